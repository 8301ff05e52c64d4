/* solve-kernel instantiations with the structure of the reference's rolling stock compiled in (both brakes, power rows, energy
 * objective, constant efficiencies: BASELINE configs 1-4); see msd_geometry.hpp and Solver::rowOn in msd_kernel.hpp.
 * These solves are split launches (solve_kernel: PART): this unit holds the first-pass kernels -- the fused iteration alone --,
 * msd_kernels_full3.hip the follow-up kernels (general iteration, restoration phase, second attempt). */
#include <hip/hip_runtime.h>

#include <cstdlib>
#include <cstring>

#include "msd_geometry.hpp"

namespace msd {
/* the pickers below hand out XCH_FAST exchange arrays (and no reduction scratch for a single wave): the layout of a kernel whose Solver::FAST holds */
static_assert(Solver<64, 2, LOSS_STATIC, false, false, FULL_BOTH, 1>::FAST && Solver<256, 2, LOSS_STATIC, false, false, FULL_BOTH, 3>::FAST, "the tuning switches of this build (MSD_PARALLEL_RICCATI) leave no fused iteration: pick_geometry_full would size the LDS wrongly");
Geometry pick_geometry_full(int N)
{
    const int nodes = N + 1;
    if (tuning().no_full) return {0, 0, nullptr};      /* (msd_tuning("no_full", 1): the general kernels, A/B runs) */
#ifndef MSD_HOT_ONLY_64X2
    if (nodes <= 64) return {64, 1, solve_kernel<64, 1, 1, LOSS_STATIC, false, false, FULL_BOTH, 1>, false, XCH_FAST, 0, follow_kernel_full(64, 1), solve_kernel<64, 1, 1, LOSS_STATIC, false, false, FULL_BOTH, 3>, 0, soc_kernel_full(64, 1, false)};
#endif
    /* (SLDS: the node constants in LDS where they do not cost the fourth resident workgroup of a compute unit -- msd_kernel.hpp: STATIC_FIELDS) */
    if (MSD_STATIC_LDS && sizeof(double)*(size_t)(lds_doubles(N, 128, false, XCH_FAST, 0) + STATIC_FIELDS*128) <= 40*1024 && nodes > 64)
        return {64, 2, solve_kernel<64, 2, 1, LOSS_STATIC, false, false, FULL_BOTH, 1, true>, false, XCH_FAST, 0, follow_kernel_full(64, 2), solve_kernel<64, 2, 1, LOSS_STATIC, false, false, FULL_BOTH, 3, true>, STATIC_FIELDS*128, soc_kernel_full(64, 2, true)};
    if (nodes <= 128) return {64, 2, solve_kernel<64, 2, 1, LOSS_STATIC, false, false, FULL_BOTH, 1>, false, XCH_FAST, 0, follow_kernel_full(64, 2), solve_kernel<64, 2, 1, LOSS_STATIC, false, false, FULL_BOTH, 3>};     /* the benchmark geometry */
#ifdef MSD_HOT_ONLY_64X2      /* tuning builds of the benchmark geometry alone (tools/build_hot.py: a fifth of the unit's compile time) */
    return {0, 0, nullptr};
#endif
    if (nodes <= 256) return {128, 2, solve_kernel<128, 2, 1, LOSS_STATIC, false, false, FULL_BOTH, 1>, false, XCH_FAST, RED_DOUBLES, follow_kernel_full(128, 2), solve_kernel<128, 2, 1, LOSS_STATIC, false, false, FULL_BOTH, 3>};
    /* longer horizons while the five additional exchange arrays still fit the LDS of a compute unit next to the stage blocks */
    const auto fits = [&](int ns) { return sizeof(double)*(size_t)lds_doubles(N, ns, false, XCH_FAST, RED_DOUBLES) <= 160*1024; };
    if (nodes <= 384 && fits(384)) return {192, 2, solve_kernel<192, 2, 1, LOSS_STATIC, false, false, FULL_BOTH, 1>, false, XCH_FAST, RED_DOUBLES, follow_kernel_full(192, 2), solve_kernel<192, 2, 1, LOSS_STATIC, false, false, FULL_BOTH, 3>};
    if (nodes <= 512 && fits(512)) return {256, 2, solve_kernel<256, 2, 1, LOSS_STATIC, false, false, FULL_BOTH, 1>, false, XCH_FAST, RED_DOUBLES, follow_kernel_full(256, 2), solve_kernel<256, 2, 1, LOSS_STATIC, false, false, FULL_BOTH, 3>};
    return {0, 0, nullptr};
}
}
