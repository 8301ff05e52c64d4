/* solve-kernel instantiations with the structure of the reference's rolling stock compiled in (both brakes, power rows, energy
 * objective, constant efficiencies: BASELINE configs 1-4); see msd_geometry.hpp and Solver::rowOn in msd_kernel.hpp */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
Geometry pick_geometry_full(int N)
{
    const int nodes = N + 1;
    const char *nf = getenv("MSD_NO_FULL");      /* MSD_NO_FULL=1: the general kernels (A/B runs) */
    if (nf && *nf == '1') return {0, 0, nullptr};
    if (nodes <= 64) return {64, 1, solve_kernel<64, 1, 1, LOSS_STATIC, false, false, true>, false, XCH_FAST, 0};
    if (nodes <= 128) return {64, 2, solve_kernel<64, 2, 1, LOSS_STATIC, false, false, true>, false, XCH_FAST, 0};     /* the benchmark geometry */
    if (nodes <= 256) return {128, 2, solve_kernel<128, 2, 1, LOSS_STATIC, false, false, true>, false, XCH_FAST, RED_DOUBLES};
    return {0, 0, nullptr};
}
}
