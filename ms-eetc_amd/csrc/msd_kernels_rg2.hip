/* follow-up kernels of the split solves of msd_kernels_rg.hip (solve_kernel's PART = 2): general iteration, restoration phase and second
 * attempt for the scenarios the first pass hands over -- or for the whole batch when none of its scenarios can start with the fused iteration */
#include <hip/hip_runtime.h>

#include "msd_geometry.hpp"

namespace msd {
KernelFn follow_kernel_full_rg(int NT, int SPT)
{
    /* (horizons of up to 63 intervals: the two-nodes-per-lane kernel follows up the one-node-per-lane first pass, msd_api.hip: make_plan launches it as 64 x 2) */
#ifdef MSD_FOLLOW_64X1      /* diagnostic builds: the instantiation that faulted on the device in round 4 (make_plan then launches it as 64 x 1) */
    if (NT == 64 && SPT == 1) return solve_kernel<64, 1, 1, LOSS_STATIC, false, false, FULL_RG, 2>;
#endif
    if (NT == 64 && SPT == 1) return solve_kernel<64, 2, 1, LOSS_STATIC, false, false, FULL_RG, 2>;
    if (NT == 64 && SPT == 2) return solve_kernel<64, 2, 1, LOSS_STATIC, false, false, FULL_RG, 2>;
    if (NT == 128 && SPT == 2) return solve_kernel<128, 2, 1, LOSS_STATIC, false, false, FULL_RG, 2>;
    if (NT == 192 && SPT == 2) return solve_kernel<192, 2, 1, LOSS_STATIC, false, false, FULL_RG, 2>;
    if (NT == 256 && SPT == 2) return solve_kernel<256, 2, 1, LOSS_STATIC, false, false, FULL_RG, 2>;
    return nullptr;
}
}
