/* solve-kernel instantiations for the dynamic loss model(s); see msd_geometry.hpp */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_geometry_dynamic(int N) { return pick_geometry_t<LOSS_TABLE>(N); }
}
