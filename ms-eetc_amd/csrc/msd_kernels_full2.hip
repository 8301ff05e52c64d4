/* solve-kernel instantiations with the structure of the reference's rolling stock compiled in (both brakes, power rows, energy
 * objective) for the other transcriptions: collocation / adaptive shooting integrators and integrateLosses; see msd_kernels_full.hip */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {

static bool no_full() { const char *nf = getenv("MSD_NO_FULL"); return nf && *nf == '1'; }
static bool one_node_per_lane() { const char *g = getenv("MSD_GEOMETRY2"); return g && !strcmp(g, "128x1"); }      /* tuning runs */
static bool one_node_per_lane_w1() { const char *g = getenv("MSD_GEOMETRY2"); return g && !strcmp(g, "128x1w1"); }

Geometry pick_geometry_general_full(int N)
{
    const int nodes = N + 1;
    if (no_full()) return {0, 0, nullptr};
    if (nodes <= 64) return {64, 1, solve_kernel<64, 1, 1, LOSS_STATIC, false, true, true>};
    if (nodes <= 128 && one_node_per_lane()) return {128, 1, solve_kernel<128, 1, 2, LOSS_STATIC, false, true, true>};
    if (nodes <= 128 && one_node_per_lane_w1()) return {128, 1, solve_kernel<128, 1, 1, LOSS_STATIC, false, true, true>};
    if (nodes <= 128) return {64, 2, solve_kernel<64, 2, 1, LOSS_STATIC, false, true, true>};
    return {0, 0, nullptr};
}

Geometry pick_geometry_intloss_full(int N)
{
    const int nodes = N + 1;
    if (no_full()) return {0, 0, nullptr};
    if (nodes <= 64) return {64, 1, solve_kernel<64, 1, 1, LOSS_INTEGRATED, false, false, true>};
    if (nodes <= 128 && one_node_per_lane()) return {128, 1, solve_kernel<128, 1, 2, LOSS_INTEGRATED, false, false, true>};
    if (nodes <= 128 && one_node_per_lane_w1()) return {128, 1, solve_kernel<128, 1, 1, LOSS_INTEGRATED, false, false, true>};
    if (nodes <= 128) return {64, 2, solve_kernel<64, 2, 1, LOSS_INTEGRATED, false, false, true>};
    return {0, 0, nullptr};
}

}  // namespace msd
