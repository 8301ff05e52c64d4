/* solve-kernel instantiations for combined options: collocation / adaptive shooting integrators together with the dynamic loss model
 * (ocp.py:92 with efficiency.py) or with integrateLosses (ocp.py:92 with ocp.py:231-241: the loss integrals have their own time-domain
 * integrator, train.py:367-413, whatever integrates the shooting intervals); see msd_geometry.hpp */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_geometry_general_dynamic(int N)
{
    const int nodes = N + 1;
    if (nodes <= 64) return {64, 1, solve_kernel<64, 1, 1, LOSS_TABLE, false, true, 0, 1>};
    if (nodes <= 128) return {128, 1, solve_kernel<128, 1, 1, LOSS_TABLE, false, true, 0, 1>};      /* (one node per lane: see msd_kernels_full2.hip) */
    if (nodes <= 256) return {128, 2, solve_kernel<128, 2, 1, LOSS_TABLE, false, true, 0, 1>};
    return {0, 0, nullptr};
}

Geometry pick_geometry_general_intloss(int N)
{
    const int nodes = N + 1;
    if (nodes <= 64) return {64, 1, solve_kernel<64, 1, 1, LOSS_INTEGRATED, false, true, 0, 1>};
    if (nodes <= 128) return {128, 1, solve_kernel<128, 1, 1, LOSS_INTEGRATED, false, true, 0, 1>};
    if (nodes <= 256) return {256, 1, solve_kernel<256, 1, 1, LOSS_INTEGRATED, false, true, 0, 1>};
    return {0, 0, nullptr};
}
}
