/* solve-kernel instantiations for integrateLosses with a loss table (the dynamic loss model of efficiency.py or a tabulated loss function integrated over
 * the running time of the interval: msd_lossint_table.hpp, DYN = LOSS_INTEGRATED_TABLE) -- the LDS-resident first-pass kernels: one node per lane, the jets
 * of the two loss integrals are the bulk of an iteration (like msd_kernels_full2.hip).  The streamed kernel of the family follows up and takes the longer
 * horizons (msd_kernels_intloss_table2.hip); see msd_geometry.hpp */
#include <hip/hip_runtime.h>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_geometry_intloss_table(int N)
{
    const int nodes = N + 1;
    if (nodes <= 64) return {64, 1, solve_kernel<64, 1, 1, LOSS_INTEGRATED_TABLE, false, false, 0, 1>};
    if (nodes <= 128) return {128, 1, solve_kernel<128, 1, 1, LOSS_INTEGRATED_TABLE, false, false, 0, 1>};
    return {0, 0, nullptr};
}
}
