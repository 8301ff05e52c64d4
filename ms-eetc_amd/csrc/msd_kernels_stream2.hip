/* solve-kernel instantiations for long horizons (stage blocks in device memory), 2048 ... 5119 intervals; see msd_geometry.hpp */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_stream_geometry_static_long(int N) { return pick_stream_geometry_long_t<LOSS_STATIC>(N); }
}
