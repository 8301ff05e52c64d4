/* follow-up kernels of the split solves of msd_kernels_rg.hip (solve_kernel's PART = 2): general iteration, restoration phase and second
 * attempt for the scenarios the first pass hands over -- or for the whole batch when none of its scenarios can start with the fused iteration */
#include <hip/hip_runtime.h>

#include "msd_geometry.hpp"

namespace msd {
KernelFn follow_kernel_full_rg(int NT, int SPT)
{
    /* Horizons of up to 63 intervals: the one-node-per-lane follow-up kernel again (round 6).  Rounds 4-5 launched the two-nodes-per-lane kernel behind the
     * 64 x 1 first pass because this instantiation faulted on the device -- the status words of a solve in vector registers, cold calls behind lane-masked
     * branches: msd_kernel.hpp: MSD_UNIFORM_STATUS, profiles/r06/streamed_follow_up_fault.md.  With them in scalar registers it passes the short-horizon, one-brake,
     * determinism and random-problem tests and repeats sweep seed 15 bit for bit (profiles/r06/README.md) */
    if (NT == 64 && SPT == 1) return solve_kernel<64, 1, 1, LOSS_STATIC, false, false, FULL_RG, 2>;
    if (NT == 64 && SPT == 2) return solve_kernel<64, 2, 1, LOSS_STATIC, false, false, FULL_RG, 2>;
    if (NT == 128 && SPT == 2) return solve_kernel<128, 2, 1, LOSS_STATIC, false, false, FULL_RG, 2>;
    if (NT == 192 && SPT == 2) return solve_kernel<192, 2, 1, LOSS_STATIC, false, false, FULL_RG, 2>;
    if (NT == 256 && SPT == 2) return solve_kernel<256, 2, 1, LOSS_STATIC, false, false, FULL_RG, 2>;
    return nullptr;
}
}
