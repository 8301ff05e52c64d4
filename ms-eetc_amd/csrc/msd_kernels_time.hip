/* first-pass kernels of the time-optimal problem (OptionsCasadiSolver.energyOptimal = False, ocp.py:150) with the structure of the reference's rolling stock
 * compiled in (FULL_TIME_RG); see msd_kernels_static.hip, msd_kernel.hpp: FULL */
#include <hip/hip_runtime.h>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_geometry_time_rg(int N)
{
    const int nodes = N + 1;
    if (nodes <= 64) return {64, 1, solve_kernel<64, 1, 1, LOSS_STATIC, false, false, FULL_TIME_RG, 1>};
    if (nodes <= 128) return {64, 2, solve_kernel<64, 2, 1, LOSS_STATIC, false, false, FULL_TIME_RG, 1>};
    if (nodes <= 256) return {128, 2, solve_kernel<128, 2, 1, LOSS_STATIC, false, false, FULL_TIME_RG, 1>};
    return {0, 0, nullptr};
}
}
