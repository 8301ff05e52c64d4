/* solve-kernel instantiations for the dynamic loss model with the structure of the reference's rolling stock compiled in (FULL_RG); see msd_kernels_dynamic.hip */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_geometry_dynamic_full_rg(int N)
{
    const int nodes = N + 1;
    if (nodes <= 64) return {64, 1, solve_kernel<64, 1, 1, LOSS_TABLE, false, false, FULL_RG, 1>};
    if (nodes <= 128) return tuning().two_nodes_per_lane ? Geometry{0, 0, nullptr} : Geometry{128, 1, solve_kernel<128, 1, 1, LOSS_TABLE, false, false, FULL_RG, 1>};
    if (nodes <= 256) return {128, 2, solve_kernel<128, 2, 1, LOSS_TABLE, false, false, FULL_RG, 1>};
    if (nodes <= 384) return {192, 2, solve_kernel<192, 2, 1, LOSS_TABLE, false, false, FULL_RG, 1>};
    return {0, 0, nullptr};
}
}
