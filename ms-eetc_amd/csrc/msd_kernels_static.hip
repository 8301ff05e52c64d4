/* solve-kernel instantiations for the static loss model(s); see msd_geometry.hpp */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_geometry_static(int N, int full)
{
    if (full) { const Geometry g = (full == FULL_RG) ? pick_geometry_full_rg(N) : pick_geometry_full(N); if (g.fn) return g; }
    return pick_geometry_t<LOSS_STATIC>(N);
}
}
