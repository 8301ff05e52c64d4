/* solve-kernel instantiations with the collocation and adaptive shooting integrators (static loss models); see msd_geometry.hpp */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {

/* kernels that carry the collocation and the adaptive shooting integrators (msd_integ.hpp) */
Geometry pick_geometry_general(int N, bool full)
{
    const int nodes = N + 1;
    if (full) { const Geometry g = pick_geometry_general_full(N); if (g.fn) return g; }
    if (nodes <= 64) return {64, 1, solve_kernel<64, 1, 1, LOSS_STATIC, false, true, 0, 1>};
    if (nodes <= 128) return {64, 2, solve_kernel<64, 2, 1, LOSS_STATIC, false, true, 0, 1>};
    if (nodes <= 256) return {128, 2, solve_kernel<128, 2, 1, LOSS_STATIC, false, true, 0, 1>};
#ifdef MSD_MINIMAL_GEOMETRIES
    return {0, 0, nullptr};
#endif
    return pick_geometry_general_long(N);
}

}  // namespace msd
