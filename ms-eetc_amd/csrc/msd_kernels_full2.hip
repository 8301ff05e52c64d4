/* solve-kernel instantiations with the structure of the reference's rolling stock compiled in (both brakes, power rows, energy
 * objective) for the other transcriptions: collocation / adaptive shooting integrators and integrateLosses; see msd_kernels_full.hip */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {

static bool no_full() { return tuning().no_full; }
static bool two_nodes_per_lane() { return tuning().two_nodes_per_lane; }      /* tuning runs (msd_tuning of mseetc_aux.h) */

/*
 * 65 ... 128 nodes: two waves per scenario with one node per lane and the whole register file of a SIMD each.  The jets through the
 * Newton solve of a collocation step, through the adaptive steps, or through the integrated loss distance are the bulk of an iteration
 * here, and a lane that carries two nodes runs them one after the other with twice the state to keep (1 400 ... 1 600 spilled registers
 * at 64 x 2 against 250 ... 400 at 128 x 1): measured on the config-1 batch 346k against 284k solves/s (integrateLosses), 186k against
 * 166k (Radau, two points), 133k against 124k (adaptive at CVODES' tolerances); profiles/r03.
 */
Geometry pick_geometry_general_full(int N)
{
    const int nodes = N + 1;
    if (no_full()) return {0, 0, nullptr};
    if (nodes <= 64) return {64, 1, solve_kernel<64, 1, 1, LOSS_STATIC, false, true, 1, 1>};
    if (nodes <= 128 && two_nodes_per_lane()) return {64, 2, solve_kernel<64, 2, 1, LOSS_STATIC, false, true, 1, 1>};
    if (nodes <= 128) return {128, 1, solve_kernel<128, 1, 1, LOSS_STATIC, false, true, 1, 1>};
    return {0, 0, nullptr};
}

Geometry pick_geometry_intloss_full(int N)
{
    const int nodes = N + 1;
    if (no_full()) return {0, 0, nullptr};
    if (nodes <= 64) return {64, 1, solve_kernel<64, 1, 1, LOSS_INTEGRATED, false, false, FULL_BOTH, 1>};
    if (nodes <= 128 && two_nodes_per_lane()) return {64, 2, solve_kernel<64, 2, 1, LOSS_INTEGRATED, false, false, FULL_BOTH, 1>};
    if (nodes <= 128) return {128, 1, solve_kernel<128, 1, 1, LOSS_INTEGRATED, false, false, FULL_BOTH, 1>};
    return {0, 0, nullptr};
}

}  // namespace msd
