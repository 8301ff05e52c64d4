/* first-pass kernels of the streamed static family with the structure of the reference's rolling stock compiled in (FULL_BOTH); see msd_kernels_stream.hip */
#include <hip/hip_runtime.h>

#include "msd_geometry.hpp"

namespace msd {
KernelFn stream_first_pass_full_both(int SPT)
{
    if (SPT == 2) return solve_kernel<512, 2, 2, LOSS_STATIC, true, false, FULL_BOTH, 1>;
    if (SPT == 4) return solve_kernel<512, 4, 2, LOSS_STATIC, true, false, FULL_BOTH, 1>;
    if (SPT == 6) return solve_kernel<512, 6, 2, LOSS_STATIC, true, false, FULL_BOTH, 1>;
    if (SPT == 10) return solve_kernel<512, 10, 2, LOSS_STATIC, true, false, FULL_BOTH, 1>;
    return nullptr;
}
}
