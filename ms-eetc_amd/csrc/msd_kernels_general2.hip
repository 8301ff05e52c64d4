/* solve-kernel instantiations with the collocation and adaptive shooting integrators (static loss models), 257 ... 640 nodes; see
 * msd_kernels_general.hip and msd_geometry.hpp */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_geometry_general_long(int N)
{
    const int nodes = N + 1;
#ifdef MSD_MINIMAL_GEOMETRIES
    return {0, 0, nullptr};
#endif
    if (nodes <= 384) return {192, 2, solve_kernel<192, 2, 1, LOSS_STATIC, false, true, 0, 1>};
    if (nodes <= 512) return {256, 2, solve_kernel<256, 2, 1, LOSS_STATIC, false, true, 0, 1>};
    if (nodes <= 640) return {320, 2, solve_kernel<320, 2, 2, LOSS_STATIC, false, true, 0, 1>};
    return {0, 0, nullptr};
}
}
