/*
 * msd_geometry.hpp -- launch geometry of the solve kernel by horizon length.  The kernel instantiations live in four
 * translation units (static / dynamic / integrated loss model, streamed, other integrators) so that they compile in parallel.
 */
#pragma once

#include "msd_kernel.hpp"

namespace msd {

using KernelFn = void (*)(DevProb, int, const double *, const double *, double *, double *, double *, double *, int, double *);

/* NT threads per workgroup, SPT shooting nodes per thread (NT*SPT >= N + 1) */
struct Geometry {
    int NT, SPT; KernelFn fn;
    bool stream = false;                 /* stage blocks in device memory (long horizons) */
    int xch = XCH_GENERAL;               /* exchange arrays in LDS and cross-wave reduction scratch (lds_doubles) */
    int red = RED_DOUBLES;
    KernelFn fn2 = nullptr;              /* not null: `fn` is the first pass of a split solve (solve_kernel's PART = 1) and this the follow-up kernel (PART = 2) */
    KernelFn fn_lsq = nullptr;           /* first pass with the least-squares multiplier estimate in front (PART = 3): launches from the reference's starting point or a primal-only warm start */
    int extra = 0;                       /* doubles of LDS behind the layout of lds_doubles (the SLDS instantiations' node constants) */
    KernelFn fn_soc = nullptr;           /* `fn` with the second-order correction inside the fused iteration (SOCK: msd_kernels_full4.hip), same launch */
};

/* tuning switches of the pickers, set through msd_tuning() of include/mseetc_aux.h (A/B runs, one GPU test): the library reads no environment variable.
 * no_full: the kernels without the structure of the NLP compiled in; two_nodes_per_lane: the 64 x 2 geometry for 65 ... 128 nodes of the shooting-integrator
 * and integrateLosses families (default there: 128 x 1) */
struct Tuning { bool no_full = false, two_nodes_per_lane = false; };
Tuning &tuning();      /* (msd_api.hip) */

Geometry pick_geometry_static(int N, int full);     /* full: FULL_BOTH / FULL_RG / FULL_TIME_* -- the kernels with that structure compiled in (0: none) */
Geometry pick_geometry_time_rg(int N);              /* time-optimal problem, structure compiled in (FULL_TIME_RG: msd_kernels_time.hip; FULL_TIME_BOTH: msd_kernels_time2.hip) */
Geometry pick_geometry_time_both(int N);
Geometry pick_geometry_full(int N);                   /* static loss model, that structure compiled in (msd_kernels_full.hip); fn == nullptr: none for this horizon */
KernelFn follow_kernel_full(int NT, int SPT);       /* follow-up kernels of that family (msd_kernels_full3.hip) */
KernelFn soc_kernel_full(int NT, int SPT, bool slds);      /* msd_kernels_full4.hip; nullptr: none for this geometry */
Geometry pick_geometry_full_rg(int N);                /* the same with the regenerative brake alone (FULL_RG: msd_kernels_rg.hip, msd_kernels_rg2.hip) */
KernelFn follow_kernel_full_rg(int NT, int SPT);
Geometry pick_geometry_dynamic(int N, int full = 0);      /* full: FULL_BOTH / FULL_RG where the problem has that structure (msd_api.hip: make_plan) */
Geometry pick_geometry_dynamic_full_rg(int N);      /* (msd_kernels_dynamic2.hip) */
Geometry pick_geometry_dynamic_full_both(int N);    /* (msd_kernels_dynamic3.hip) */
Geometry pick_stream_geometry_static(int N, int full = 0);      /* full: FULL_BOTH / FULL_RG -- that structure compiled into the first pass (the follow-up kernel stays general) */
KernelFn stream_first_pass_full_rg(int SPT);       /* (msd_kernels_stream5.hip) */
KernelFn stream_first_pass_full_both(int SPT);     /* (msd_kernels_stream6.hip) */
Geometry pick_geometry_general_long(int N);      /* 257 ... 640 nodes of the same family (msd_kernels_general2.hip) */
Geometry pick_geometry_general(int N, bool full = false);      /* collocation / adaptive shooting integrators (static loss models, LDS-resident) */
Geometry pick_geometry_intloss(int N, bool full = false);      /* integrateLosses: loss slacks from the integrated loss power (static efficiencies, LDS-resident) */
Geometry pick_geometry_intloss_table(int N);          /* integrateLosses with a loss table (msd_lossint_table.hpp): LDS-resident first-pass kernels up to 127 intervals (msd_kernels_intloss_table.hip) */
Geometry pick_stream_geometry_intloss_table(int N);   /* ... the streamed pair of that family, up to 1023 intervals (msd_kernels_intloss_table2.hip) */
Geometry pick_geometry_general_dynamic(int N);      /* collocation / adaptive shooting integrators with the dynamic loss model (msd_kernels_compose.hip) */
Geometry pick_geometry_general_intloss(int N);      /* collocation / adaptive shooting integrators with integrateLosses (msd_kernels_compose.hip) */
/* the other transcriptions beyond the LDS-resident horizons, up to 1023 intervals (msd_kernels_stream3.hip): dynamic loss model, collocation /
 * adaptive shooting integrators, integrateLosses on the streamed kernel */
Geometry pick_stream_geometry_dynamic(int N);
Geometry pick_stream_geometry_general(int N);
Geometry pick_stream_geometry_intloss(int N);
Geometry pick_geometry_general_full(int N);  /* the same two families with the structure of the reference's rolling stock compiled in (msd_kernels_full2.hip) */
Geometry pick_geometry_intloss_full(int N);

/* LDS-resident kernels without the structure of the NLP compiled in: first-pass kernels (PART = 1: the general iteration without the cold paths); the
 * streamed kernel of the family follows up (msd_api.hip: make_plan) */
template <int DYN> inline Geometry pick_geometry_t(int N)
{
    const int nodes = N + 1;
#ifdef MSD_ONLY_192X2              /* debugging builds */
    return nodes <= 384 ? Geometry{192, 2, solve_kernel<192, 2, 1, DYN, false, false, 0, 1>} : Geometry{0, 0, nullptr};
#endif
    if (nodes <= 64) return {64, 1, solve_kernel<64, 1, 1, DYN, false, false, 0, 1>};
    /* the loss-table family on 65 ... 128 nodes: two waves with one node per lane and the whole register file of a SIMD each (round 6: 341 k against 289 k solves/s
     * on the figure-5 batch at N = 100, 307 k against 219 k at N = 120 -- the jets through the table are the bulk of its iteration, like the shooting
     * integrators' of msd_kernels_full2.hip; msd_tuning("two_nodes_per_lane", 1): the one-wave geometry) */
    if (DYN == LOSS_TABLE && nodes > 64 && nodes <= 128 && !tuning().two_nodes_per_lane) return {128, 1, solve_kernel<128, 1, 1, DYN, false, false, 0, 1>};
    if (nodes <= 128) return {64, 2, solve_kernel<64, 2, 1, DYN, false, false, 0, 1>};     /* one wave per scenario, one wave per SIMD */
    if (nodes <= 256) return {128, 2, solve_kernel<128, 2, 1, DYN, false, false, 0, 1>};
#ifdef MSD_MINIMAL_GEOMETRIES      /* tuning builds (tools/build_variant.py) */
    return {0, 0, nullptr};
#endif
    if (nodes <= 384) return {192, 2, solve_kernel<192, 2, 1, DYN, false, false, 0, 1>};
    if (nodes <= 512) return {256, 2, solve_kernel<256, 2, 1, DYN, false, false, 0, 1>};
    /* N = 512 ... 575 with static loss rows: three waves with three nodes per lane and the whole register file of a SIMD each -- the stage blocks and
     * six exchange arrays of 576 slots still fit the LDS of a compute unit.  (Round 3 ran these horizons on five waves of two nodes per lane with
     * half a register file each: 2 253 spilled registers, 25 ms per 1024 solves at N = 560 against 6.2 ms at N = 511.) */
    if (DYN == LOSS_STATIC && nodes <= 576) return {192, 3, solve_kernel<192, 3, 1, DYN, false, false, 0, 1>};
    if (nodes <= 640) return {320, 2, solve_kernel<320, 2, 2, DYN, false, false, 0, 1>};
    return {0, 0, nullptr};
}

/* horizons whose stage blocks do not fit the LDS of a compute unit: node fields, stage blocks and exchange arrays live in device memory, a
 * lane's nodes are worked off one after the other.  512 threads (two waves per SIMD, 256 registers each) with the stage-parallel KKT solve
 * and as few nodes per lane as the horizon allows (N = 1000: two; 38 -> 11 ms per solve against round 2's 1024 x 5 with serial sweeps).
 * A streamed solve is a split launch too (round 5): `fn` = the first pass (PART = 1), `fn2` = the follow-up kernel of the same geometry with the
 * restoration phase and the watchdog procedure (PART = 2), which also follows up the LDS-resident first-pass kernels of its family.
 * The instantiations are spread over translation units (msd_kernels_stream*.hip) */
template <int DYN, bool GEN, int SPT> inline Geometry stream_geometry_t()
{
    return {512, SPT, solve_kernel<512, SPT, 2, DYN, true, GEN, 0, 1>, true, XCH_GENERAL, RED_DOUBLES, solve_kernel<512, SPT, 2, DYN, true, GEN, 0, 2>};
}
template <int DYN> inline Geometry pick_stream_geometry_short_t(int N)      /* N <= 2047 */
{
    if (N + 1 <= 1024) return stream_geometry_t<DYN, false, 2>();
    if (N + 1 <= 2048) return stream_geometry_t<DYN, false, 4>();
    return {0, 0, nullptr};
}
template <int DYN> inline Geometry pick_stream_geometry_long_t(int N)       /* N <= 5119 */
{
    if (N + 1 <= 3072) return stream_geometry_t<DYN, false, 6>();
    if (N + 1 <= 5120) return stream_geometry_t<DYN, false, 10>();
    return {0, 0, nullptr};
}
Geometry pick_stream_geometry_static_long(int N);

}  // namespace msd
