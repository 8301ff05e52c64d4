/* the fused first-pass kernels with the second-order correction inside the fused iteration (msd_kernel.hpp: SOCK) -- for launches where corrections are
 * the rule: the re-solves of the shrinking-horizon loop (msd_mpc.hip).  Rolling stock of the reference's JSON files (FULL_BOTH), horizons of up to 103
 * intervals; see msd_kernels_full.hip */
#include <hip/hip_runtime.h>

#include "msd_geometry.hpp"

namespace msd {
KernelFn soc_kernel_full(int NT, int SPT, bool slds)
{
    if (NT == 64 && SPT == 1 && !slds) return solve_kernel<64, 1, 1, LOSS_STATIC, false, false, FULL_BOTH, 1, false, true>;
    if (NT == 64 && SPT == 2 && slds) return solve_kernel<64, 2, 1, LOSS_STATIC, false, false, FULL_BOTH, 1, true, true>;
    return nullptr;
}
}  // namespace msd
