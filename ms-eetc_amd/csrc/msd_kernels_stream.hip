/* solve-kernel instantiations for long horizons (stage blocks in device memory), up to 2047 intervals; see msd_geometry.hpp */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_stream_geometry_static(int N)
{
    const Geometry g = pick_stream_geometry_short_t<LOSS_STATIC>(N);
    return g.fn ? g : pick_stream_geometry_static_long(N);
}
}
