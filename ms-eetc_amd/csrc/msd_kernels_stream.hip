/* solve-kernel instantiation for long horizons (stage blocks in device memory); see msd_geometry.hpp */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_stream_geometry_static(int N) { return pick_stream_geometry_t<false>(N); }
}
