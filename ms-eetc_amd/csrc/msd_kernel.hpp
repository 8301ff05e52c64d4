/*
 * msd_kernel.hpp -- device code of the MI355X (gfx950) batched multiple-shooting solver.
 *
 * One workgroup solves one OCP scenario from its starting point to convergence in a single launch (persistent over scenarios:
 * workgroups pull scenarios from a device-wide counter).  The benchmark geometry is one wavefront per scenario with two shooting
 * nodes per lane (solve_kernel<64, 2>): four single-wave workgroups per compute unit, one per SIMD, each with the SIMD's whole
 * register file.  A lane owns its nodes' states (t_i, b_i), controls (Fel_i, Fpb_i, s_i), slacks and multipliers -- in registers for
 * the whole solve; stage blocks and neighbour exchange go through LDS.  Per interior-point iteration
 *   (a) every lane integrates its intervals (RK4 + first/second sensitivities by forward-mode jets), evaluates rows, optimality
 *       error and filter quantities, and condenses its inequality rows and bounds into a stage block            [parallel over stages]
 *   (b) the block-tridiagonal KKT system is solved by a stage-parallel Riccati recursion: value functions at the lanes' chunk
 *       boundaries from an associative scan over the lanes, the ordinary recursion inside the chunks (ParallelRiccati, msd_scan.hpp);
 *       a serial sweep on one lane remains as the fallback                                                      [log depth over stages]
 *   (c) step lengths, filter line search and updates run per node again, with DPP wave reductions for the norms.
 * Kernels instantiated with FULL have the structure of the reference's rolling stock compiled in and, with constant efficiencies and
 * explicit Runge-Kutta shooting, run (a) and (c) as the fused passes of Solver::FAST (fused_pass, post_direction, merit_fast,
 * update_fast).  Nothing but the scenario record, the (shared, L2-resident) track profile and the final z* touches HBM -- apart from
 * register spills.  All arithmetic is IEEE double, like the reference's CasADi/IPOPT path.
 *
 * What it computes (reference = dkouzoup/ms-eetc):
 *   NLP        mseetc/ocp.py:134-284 (variables, bounds, rows, objective), cold start :325-339
 *   integrator mseetc/train.py:225-277 (ODE), :294-301 (RK4 = casadi.simpleRK), :324-344 (trapezoidal time); :303-322 (IRK, CVODES: msd_integ.hpp)
 *   losses     mseetc/train.py:199-216 + mseetc/utils.py:197-220 (static efficiencies -> two linear rows), mseetc/efficiency.py (table),
 *              mseetc/ocp.py:231-241 (integrateLosses: msd_lossint.hpp)
 *   NLP solver casadi.nlpsol('ipopt') (ocp.py:290,359): IPOPT's published algorithm (Waechter & Biegler, Math. Prog. 106(1), 2006)
 *              with IPOPT's default options: monotone barrier update, filter line search with second-order correction, inertia
 *              correction, gradient-based scaling.
 *
 * The state of a stage is (t, b, q) with q_i := Fel_{i-1}: the control-smoothing term 1e-3 (Fel_i - Fel_{i-1})^2 (ocp.py:245) and
 * the end-of-interval power row Fel_i sqrt(b_{i+1}) (ocp.py:189) then are stage-local, which keeps the KKT system block tridiagonal
 * with 3x3 blocks.
 */
#pragma once
#include <type_traits>

#include <hip/hip_runtime.h>

#include <float.h>
#include <math.h>

#include "../../include/mseetc_hip.h"
#include "msd_fastmath.hpp"
#include "msd_scan.hpp"

/* build-time tuning switches (defaults = what measured fastest on MI355X) */
#ifndef MSD_PHASE_FENCE
#define MSD_PHASE_FENCE 1
#endif
#ifndef MSD_FENCE_PHASES
#define MSD_FENCE_PHASES 0x155      /* bit k: fence after phase k (enum PH_*): evaluation, assembly, read-back, step lengths, update (tools/ab.sh, round 2) */
#endif
#ifndef MSD_HOT_FENCES
#define MSD_HOT_FENCES 0            /* 1: phase fences also in the first-pass kernels of the fused iteration (A/B builds) */
#endif
#ifndef MSD_FENCE_MAX_NT
#define MSD_FENCE_MAX_NT 256        /* largest workgroup the phase fences are applied to (see Solver::phase_fence) */
#endif
#ifndef MSD_FENCE_DUALS
#define MSD_FENCE_DUALS 1           /* also fence multipliers and residuals, not only the primal point */
#endif
#ifndef MSD_PROFILE_SKIP_LSQ
#define MSD_PROFILE_SKIP_LSQ 1      /* profile start begins with zero constraint multipliers: the least-squares estimate costs a KKT solve and buys no iterations there */
#endif
#ifndef MSD_RICCATI_INLINE
#define MSD_RICCATI_INLINE 1
#endif
#ifndef MSD_TELEMETRY
#define MSD_TELEMETRY 0             /* per-phase cycle counters of the logged scenario (Ctx::mark) */
#endif
#ifndef MSD_PARALLEL_RICCATI
#define MSD_PARALLEL_RICCATI 1      /* stage-parallel KKT solve (scan over the lanes, msd_scan.hpp); 0: serial sweep on one lane */
#endif

namespace msd {

struct DevProb {
    int N, withPn, hasPower, energyOpt, numSteps, numApprox, lossKind, maxIter;
    double sr0, sr1, sr2, g, rho, fmax, fmin, fminPn, pwU, pwL, accMin, accMax, ct, cr, vminSq, objDen, tol;
    const double *ds, *grad, *curv, *bmax;
    const double *loss;      /* parameter block of the dynamic loss model (lossKind == 2), see DynLoss */
    const double *lossCoef;  /* null in the host's record.  The kernels keep the head of the block (scalars and breakpoints) in LDS where it fits: their LDS copy of this
                              * record then has `loss` = that copy and `lossCoef` = the bicubic patches in device memory (solve_kernel) */
    const double *pos;       /* [N+1] node positions (cumulative ds), used by the profile start */
    /* primal warm start (set per launch by msd_solve_batch_warm, null = none) */
    const double *guess;
    long long guessStride;   /* doubles between the guesses of consecutive scenarios (nz, or the layout of the previous, longer horizon) */
    const double *guessStatus;   /* not null: stats records of the solves the guesses come from; a scenario whose record says "failed" starts cold */
    double warmMu, warmPush;
    double lossMass;         /* per-scenario override of the total mass in the dynamic loss model (0: the table's) */
    int *queue;              /* device-wide scenario counter of the launch (null: static distribution) */
    /* primal-dual warm start: multipliers per node in MSD_DUAL_STRIDE doubles (lam 2, nu 5, zL 5, zU 5, zLs 5, zUs 5) */
    double *dualOut;         /* not null: [nscen][N + 1][MSD_DUAL_STRIDE], the multipliers of every solve are recorded */
    const double *dualIn;    /* not null (with guess): multipliers to start from, dualInStride doubles between scenarios, node 0 of this
                              * problem = node dualShift of the recorded one */
    long long dualInStride;
    int dualShift;
    /* integrator of the shooting intervals other than the explicit Runge-Kutta map (kernels instantiated with GEN; msd_integ.hpp) */
    int integ;               /* 0, MSD_INTEGRATOR_ADAPTIVE or MSD_INTEGRATOR_COLLOCATION */
    int collD, newtonIters;  /* collocation points per step, Newton iterations (OptionsIRK.order, .maxIter) */
    double intAtol, intRtol; /* OptionsCVODES.absTol, .relTol */
    const double *coll;      /* C[(collD+1)^2], D[collD+1] of casadi.simpleIRK */
    int resto;               /* feasibility restoration phase where the line search breaks down (IPOPT's behaviour; msd_resto.hpp) */
    int wdTrigger;           /* shortened iterations in a row that start the watchdog procedure (IPOPT: 10; <= 0: never) */
    int oneAttempt;          /* a solve that breaks down ends there: no second attempt from the other starting point (the re-solves of msd_mpc.hip) */
    int start;               /* MSD_START_REFERENCE: cold start of ocp.py:325-339; MSD_START_PROFILE: profile start */
    /* split launches (solve_kernel's PART): the first-pass kernel appends the scenarios it does not finish to this list, the follow-up kernel drains it.
     * follow[0] entries written, [1] entries taken, [2] follow-up workgroups that found the list empty (the last one zeroes the three for the next
     * launch of the handle), [FOLLOW_HDR + 2k] scenario, [FOLLOW_HDR + 2k + 1] iterations already spent on it (>= 0: its first attempt broke down,
     * the follow-up kernel begins with the second one) or -1 (nothing decided yet: the general iteration from the same starting point) */
    int *follow;
    int *list;               /* not null: the kernel solves the scenarios of this list only (same layout as `follow`: the follow-up kernel's input is the first pass's
                              * `follow`; a first-pass kernel can be given a list too -- msd_mpc.hip's re-solves -- and hands over through `follow` as usual) */
    int *socSeen;            /* not null: a word of host memory mapped into the device; a first-pass kernel that hands a scenario over for a second-order correction sets it,
                              * and the handle's next launches take the first-pass kernel that has the correction inside its fused iteration (msd_api.hip: launch) */
};
/* behind the three counters: telemetry that is never reset -- [3] scenarios listed so far, [4 + why] by reason: 0 no fused start for the scenario
 * (a warm start whose previous solve failed), 1 wrong inertia or a scan breakdown, 2 tiny step, 3 first trial point rejected where a
 * second-order correction applies, 4 line search broke down (restoration phase), 5 a breakdown that asks for the second attempt */
constexpr int FOLLOW_HDR = 16, FOLLOW_TOTAL = 3, FOLLOW_WHY = 4;

/* IPOPT default option values */
constexpr double K_BOUND_RELAX = 1e-8;
constexpr double K_PUSH = 1e-2;      /* bound_push = bound_frac */
constexpr double K_MU_INIT = 0.1, K_EPS = 10.0, K_MU_LIN = 0.2, K_MU_SUP = 1.5, K_TAU_MIN = 0.99;
constexpr double K_SMAX = 100.0, K_SIGMA = 1e10, K_D = 1e-5;
constexpr double G_THETA = 1e-5, G_PHI = 1e-8, K_DELTA = 1.0, S_THETA = 1.1, S_PHI = 2.3, ETA_PHI = 1e-8;
constexpr double K_SOC = 0.99;
constexpr int P_MAX_SOC = 4;
constexpr double ALPHA_MIN_FRAC = 0.05;
constexpr double DW_MIN = 1e-20, DW_0 = 1e-4, DW_MAX = 1e40, KW_MINUS = 1.0/3.0, KW_PLUS = 8.0, KW_PLUS_BAR = 100.0;
constexpr double LAM_INIT_MAX = 1e3;
constexpr double ACC_TOL = 1e-6;
constexpr int ACC_ITER = 15;

constexpr int VT = 0, VB = 1, VF = 2, VP = 3, VS = 4, NV = 5;
constexpr int RPW0 = 0, RPW1 = 1, RACC = 2, RLTR = 3, RLRG = 4, NR = 5;

/* LDS layout (in doubles) */
/* stage block stride: odd -> the per-thread writes spread over the banks; the dynamic loss model needs three more entries */
/* DYN (template parameter of the kernels): loss transcription -- 0 constant efficiencies at the mid-point speed (ocp.py:221-226), 1 dynamic
 * table (efficiency.py), 2 constant efficiencies integrated over the running time (integrateLosses, ocp.py:231-241; msd_lossint.hpp).
 * 1 and 2 couple the loss slack with b and Fpb and use the wider stage block */
constexpr int LOSS_STATIC = 0, LOSS_TABLE = 1, LOSS_INTEGRATED = 2, LOSS_INTEGRATED_TABLE = 3;
/* (3, round 6: the loss table integrated over the running time -- integrateLosses with the dynamic loss model or a tabulated loss function, msd_lossint_table.hpp;
 *  2 and 3 share everything that comes from the rows' dependence on the running time t_{i+1} - t_i) */
__host__ __device__ constexpr bool loss_integrated(int dyn) { return dyn == LOSS_INTEGRATED || dyn == LOSS_INTEGRATED_TABLE; }
/* FULL (template parameter of the kernels): structure of the NLP known at compile time -- 0 nothing (row set, brakes and objective read from the
 * problem record), 1 both brakes + power rows + energy objective (the rolling stock of the reference's JSON files: BASELINE configs 1-4),
 * 2 the same with the regenerative brake alone (forceMinPn = 0: what every script of the reference sets -- figure5.py:88, figure6.py:108,
 * figure10.py:17, table3.py:18) */
constexpr int FULL_BOTH = 1, FULL_RG = 2;
/* 3, 4 (round 6): the time-optimal twins of 1 and 2 (ocp.py:150 -- minimum running time with 1e-4 (f^2 + p^2): power rows and acceleration row on and two-sided, no loss
 * rows): the problems of OptionsCasadiSolver.energyOptimal = False on the reference's rolling stock, config 4's minimum-time certificates (msd_mpc.hip) */
constexpr int FULL_TIME_BOTH = 3, FULL_TIME_RG = 4;
__host__ __device__ constexpr bool full_energy(int full) { return full == FULL_BOTH || full == FULL_RG; }
__host__ __device__ constexpr bool full_time(int full) { return full == FULL_TIME_BOTH || full == FULL_TIME_RG; }
__host__ __device__ constexpr bool full_pn(int full) { return full == FULL_BOTH || full == FULL_TIME_BOTH; }
constexpr int S_STRIDE_STATIC = 27, S_STRIDE_DYN = 31;
__host__ __device__ constexpr int stage_stride(bool dyn) { return dyn ? S_STRIDE_DYN : S_STRIDE_STATIC; }
constexpr int FILT_CAP = 64;
constexpr int RED_K = 8, RED_SLOTS = 4, MAX_WAVES = 16;
constexpr int MISC_FALLBACKS = 24;  /* misc[24]: KKT solves of this scenario that fell back from the scan to the serial sweep */
constexpr int MISC_LG = 16;        /* misc[16..22]: last interval's Fel row for the multiplier of its eliminated b row */
constexpr int HIST_COLS = 8;
constexpr int CONST_DOUBLES = 96, UNI_OFF = 48;    /* LDS copies of the problem record (DevProb) and of the scenario's uniform data (Uni), behind misc */

/* stage block slots.  Written by the owning thread in assemble(): the dynamics (0..5, never overwritten), the condensed
 * Hessian/gradient of (t, b, q | f, p) with the slack variable s already eliminated (6..22), and what the elimination needs
 * afterwards (23..).  The serial sweeps overwrite 6..22 with the feedback, the value function and the step. */
constexpr int S_TB = 0, S_TW = 1, S_BB = 2, S_BW = 3, S_RT = 4, S_RB = 5;
constexpr int S_HTT = 6, S_HBB = 7, S_HBQ = 8, S_HBF = 9, S_HBP = 10, S_HQQ = 11, S_HQF = 12, S_HFF = 13, S_HFP = 14, S_HPP = 15,
              S_HT = 18, S_HB = 19, S_HQ = 20, S_HF = 21, S_HP = 22;
/* The two forces' own curvatures, Hff - Hfp and Hpp - Hfp, accumulated on their own (round 5).  Where both brakes are free and the acceleration
 * row is active, its barrier term Sigma g g^T (1e12 at convergence) sits in Hff, Hfp and Hpp alike and the curvature of the split between the two
 * forces -- 1e-4 in a time-optimal problem -- is lost in Hff - Hfp^2/Hpp: the second pivot of the control block came out as rounding noise of
 * either sign, the inertia correction escalated (467 regularisations and the iteration limit on a re-solve of config 4 whose oracle solve takes 26
 * iterations).  With a = S_OA, b = S_OB and G = H + F^T P F:  Gff - Gfp = a + (Tw Ptq + Bw Pbq) + Pqq,  Gpp - Gfp = b - (Tw Ptq + Bw Pbq),  and the
 * pivot is  (Gff - Gfp) + (Gfp/Gpp)(Gpp - Gfp)  -- no difference of large numbers.  Written by assemble() / fused_pass; the sweeps read them before
 * they stash the value function over them (S_PN) */
constexpr int S_OA = 16, S_OB = 17;
/* s-elimination data: ds = -(GS + GBS db + GFS df + GPS dp) IS; the last interval keeps its (unreduced) s row here and the
 * backward sweep leaves the feedback of s in 23..26 */
constexpr int S_GFS = 23, S_IS = 24, S_GS = 25, S_GBS = 26 /* dynamic only */, S_GPS = 27 /* dynamic only */;
constexpr int S_KS = 23 /* 4, last interval only */;
constexpr int S_K = 6 /* 6: f and p rows */, S_KV = 12 /* 2 */, S_PN = 14 /* 6 */, S_PV = 20 /* 3 */;
constexpr int S_DT = 6, S_DB = 7, S_DF = 8, S_DP = 9, S_DS = 10, S_LT = 11, S_LB = 12;
/* dynamic loss model only: couplings of (b_i, s_i) with b_{i+1}; never overwritten by the sweeps */
constexpr int S_EB = 28, S_ES = 29;

/* cooperative evaluation of the adaptive shooting integrator (Solver::coop_adaptive): per wave a header (accepted steps, force, resistance, interval
 * length of the long interval), its step sizes and step-start values (64 each) and the 64 x 12 components of the steps' local jets.  Behind everything
 * else in the LDS of the kernels of the shooting-integrator families whose horizon is LDS-resident */
#ifndef MSD_COOP_ADAPTIVE
#define MSD_COOP_ADAPTIVE 1
#endif
constexpr int COOP_CAP = 64, COOP_HDR = 8, COOP_POOL = COOP_HDR + 2*COOP_CAP + 12*COOP_CAP;
/* ... and per lane what its own value pass left (Solver::coop_values): two step sizes and step-start values, tau and b+, a status word, and the (b, force) the
 * pass was run at -- the evaluation of the current point that follows an accepted trial point finds the value pass done */
constexpr int COOP_REC = 9;
__host__ __device__ __forceinline__ int coop_doubles(int NT, bool gen) { return (MSD_COOP_ADAPTIVE && gen && NT <= 128) ? (NT/64)*COOP_POOL + COOP_REC*NT : 0; }

/* LDS of the streamed kernel: filter, reduction scratch, misc, uniform records */
/* head of the loss table's parameter block (11 scalars, grid sizes, breakpoints of both axes) in LDS, between the uniform records and the node constants, in the
 * kernels of the loss-table families: the cell search of table_eval reads the breakpoints -- a chain of dependent loads from device memory before round 6, the
 * bulk of an evaluation (EVAL + MERIT 76 k of 175 k cycles per iteration of the figure-5 batch, profiles/r06).  A table with more breakpoints stays where it is */
constexpr int LOSS_HEAD_CAP = 128;
__host__ __device__ __forceinline__ int lds_doubles_stream() { return 2*64 + 4*16*8 + 32 + 96 + LOSS_HEAD_CAP; }

/* exchange arrays over the node slots: neighbour t, b, Fel and three outgoing contributions; the fused iteration of the kernels with the
 * problem structure compiled in (Solver::FAST) publishes sqrt(b) too and sends seven contributions in one pass */
constexpr int XCH_GENERAL = 6, XCH_FAST = 11;
constexpr int RED_DOUBLES = RED_SLOTS*MAX_WAVES*RED_K;      /* cross-wave reduction scratch; a single-wave FAST kernel has none */
/* SLDS (template parameter of the fused kernels, round 5): the five per-node constants (interval length, track resistance, scaling of the two dynamics
 * rows, upper bound of b) in LDS behind everything else; every pass loads the ones it needs instead of carrying them through the solve -- 16 spilled
 * registers and 48 B of scratch per lane less, 83.2 k -> 81.8 k cycles per iteration on config 1.  The pickers choose these instantiations where the
 * 5 NS doubles do not cost a resident workgroup (Geometry::extra; 64 x 2: up to 103 intervals, four workgroups per CU either way).  The slack steps kept
 * in the exchange arrays between post_direction and update_fast the same way bought nothing and are not */
#ifndef MSD_STATIC_LDS
#define MSD_STATIC_LDS 1
#endif
constexpr int STATIC_FIELDS = 5;
/* SOCK (template parameter of the fused first-pass kernels, round 5): the second-order correction (W&B section 2.4) inside the fused iteration, as passes of its
 * loop over the same point (Solver::run: soc_mode) instead of a handover to the follow-up kernel.  Config 4's warm loop runs at 289 k instead of 217 k successful
 * re-solves/s with it -- but the loop's extra control flow costs the hot path 4.5 % (config 1: 86.0 k instead of 81.6 k cycles per iteration, also with the
 * correction's loads and stores compiled out: the register allocator's answer to the control flow).  So it is an instantiation of its own (msd_kernels_full4.hip),
 * every line of it behind `if constexpr (SOCK)` -- the other instantiations compile what they compiled before -- and taken where corrections are the rule: the
 * re-solves of msd_mpc.hip (WarmStart::use_soc) */
__host__ __device__ __forceinline__ int lds_doubles(int N, int NS, bool dyn, int nxch = XCH_GENERAL, int red = RED_DOUBLES)
{
    return stage_stride(dyn)*(N + 1) + nxch*NS + 2*FILT_CAP + red + 32 + CONST_DOUBLES + (dyn ? LOSS_HEAD_CAP : 0);
}

/* ------------------------------------------------------------------------------------------
 * forward-mode second-order jets in (b, w)
 * ---------------------------------------------------------------------------------------- */
struct Jet { double v, g0, g1, h00, h01, h11; };

__device__ __forceinline__ Jet operator+(Jet a, Jet b) { return {a.v + b.v, a.g0 + b.g0, a.g1 + b.g1, a.h00 + b.h00, a.h01 + b.h01, a.h11 + b.h11}; }
__device__ __forceinline__ Jet operator-(Jet a, Jet b) { return {a.v - b.v, a.g0 - b.g0, a.g1 - b.g1, a.h00 - b.h00, a.h01 - b.h01, a.h11 - b.h11}; }
__device__ __forceinline__ Jet operator*(Jet a, double s) { return {a.v*s, a.g0*s, a.g1*s, a.h00*s, a.h01*s, a.h11*s}; }
__device__ __forceinline__ Jet operator*(double s, Jet a) { return a*s; }
__device__ __forceinline__ Jet operator*(Jet a, Jet b)
{
    return {a.v*b.v, a.v*b.g0 + b.v*a.g0, a.v*b.g1 + b.v*a.g1, a.v*b.h00 + 2*a.g0*b.g0 + b.v*a.h00,
            a.v*b.h01 + a.g0*b.g1 + a.g1*b.g0 + b.v*a.h01, a.v*b.h11 + 2*a.g1*b.g1 + b.v*a.h11};
}
__device__ __forceinline__ Jet operator+(Jet a, double c) { a.v += c; return a; }
__device__ __forceinline__ Jet operator+(double c, Jet a) { a.v += c; return a; }
__device__ __forceinline__ Jet operator-(Jet a, double c) { a.v -= c; return a; }
__device__ __forceinline__ Jet chain(Jet a, double F, double f1, double f2)
{
    return {F, f1*a.g0, f1*a.g1, f1*a.h00 + f2*a.g0*a.g0, f1*a.h01 + f2*a.g0*a.g1, f1*a.h11 + f2*a.g1*a.g1};
}
/* sqrt as a jet: the value is the same number the value-only evaluation of a trial point computes (xsqrt(double)), so that the residual a Newton
 * step is built on and the constraint violation the line search measures at the same point agree to the last bit -- and the two derivative
 * factors 1/(2 sqrt(v)) = r/2, -1/(4 v sqrt(v)) = -r^3/4 come from one reciprocal square root (accurate to an ulp: they scale the Jacobian the
 * dual residual is formed with) */
#ifdef MSD_HOST_EMULATION
__device__ __forceinline__ double rsqrt_(double v) { return 1.0/sqrt(v); }
#else
__device__ __forceinline__ double rsqrt_(double v) { return rsqrt(v); }
#endif
template <bool FM = false> __device__ __forceinline__ Jet xsqrt(Jet a)
{
    if constexpr (FM) { double r; const double s = fsqrt2(a.v, r), f1 = 0.5*r; return chain(a, s, f1, -0.5*f1*(r*r)); }
    else { const double r = rsqrt_(a.v), f1 = 0.5*r; return chain(a, sqrt(a.v), f1, -0.5*f1*(r*r)); }
}
template <bool FM = false> __device__ __forceinline__ Jet xrecip(Jet a) { const double r = FM ? frcp(a.v) : 1.0/a.v; return chain(a, r, -r*r, 2*r*r*r); }
template <bool FM = false> __device__ __forceinline__ double xsqrt(double a) { return FM ? fsqrt(a) : sqrt(a); }
template <bool FM = false> __device__ __forceinline__ double xrecip(double a) { return FM ? frcp(a) : 1.0/a; }
/* 1/sqrt: the integrand of the time equation (shooting integrators of msd_integ.hpp).  FM: from the square root's own refinement */
template <bool FM = false> __device__ __forceinline__ double xrsqrt(double a)
{
    if constexpr (FM) { double r; fsqrt2(a, r); return r; }
    else return 1.0/sqrt(a);
}
template <bool FM = false> __device__ __forceinline__ Jet xrsqrt(Jet a)
{
    if constexpr (FM) { const double r = xrsqrt<true>(a.v), r2 = r*r, f1 = -0.5*r*r2; return chain(a, r, f1, -1.5*f1*r2); }
    else return xrecip(xsqrt(a));
}
__device__ __forceinline__ Jet make_var(Jet, double v, int k) { return {v, k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, 0, 0, 0}; }
__device__ __forceinline__ double make_var(double, double v, int) { return v; }
__device__ __forceinline__ Jet make_zero(Jet) { return {0, 0, 0, 0, 0, 0}; }
__device__ __forceinline__ double make_zero(double) { return 0.0; }

/* d(b)/d(sigma) on the unit interval (train.py:251-259) */
template <class T, bool FM = false> __device__ __forceinline__ T ode_b(const DevProb &P, T b, T w, double G, double ds)
{
    T rr = P.sr0 + (xsqrt<FM>(b)*P.sr1 + b*P.sr2);
    return ((w - rr) - G)*(2*ds);
}

/* casadi.simpleRK(ode, numSteps, 4) on the b equation with total step H (train.py:298-301) */
template <class T> __device__ __forceinline__ T rk4_b(const DevProb &P, T b, T w, double G, double ds, double H)
{
    double h = H/P.numSteps;
    for (int s = 0; s < P.numSteps; s++) {
        T k1 = ode_b(P, b, w, G, ds);
        T k2 = ode_b(P, b + k1*(0.5*h), w, G, ds);
        T k3 = ode_b(P, b + k2*(0.5*h), w, G, ds);
        T k4 = ode_b(P, b + k3*h, w, G, ds);
        b = b + ((k1 + k2*2.0) + (k3*2.0 + k4))*(h/6);
    }
    return b;
}

/* one shooting interval: tau = t+ - t and b+ (train.py:296-301 joint RK4, :324-344 trapezoidal time) */
template <class T, bool FM = false> __device__ __forceinline__ void interval_map(const DevProb &P, double b0, double w0, double G, double ds, T &tau, T &bplus)
{
    T b = make_var(T(), b0, 0), w = make_var(T(), w0, 1);
    if (P.numApprox == 1 && P.numSteps == 1) {
        /* the integrator of simulations/config.json (one RK4 step, one trapezoid: every BASELINE config), straight-line */
        T k1 = ode_b<T, FM>(P, b, w, G, ds);
        T k2 = ode_b<T, FM>(P, b + k1*0.5, w, G, ds);
        T k3 = ode_b<T, FM>(P, b + k2*0.5, w, G, ds);
        T k4 = ode_b<T, FM>(P, b + k3*1.0, w, G, ds);
        T cur = b + ((k1 + k2*2.0) + (k3*2.0 + k4))*(1.0/6);
        tau = xrecip<FM>(xsqrt<FM>(b) + xsqrt<FM>(cur))*(2*ds*(1.0 - 0.0));
        bplus = cur;
        return;
    }
    if (P.numApprox == 0) {
        double h = 1.0/P.numSteps;
        T t = make_zero(T());
        for (int s = 0; s < P.numSteps; s++) {
            T k1b = ode_b(P, b, w, G, ds), k1t = xrecip(xsqrt(b))*ds;
            T b2 = b + k1b*(0.5*h);
            T k2b = ode_b(P, b2, w, G, ds), k2t = xrecip(xsqrt(b2))*ds;
            T b3 = b + k2b*(0.5*h);
            T k3b = ode_b(P, b3, w, G, ds), k3t = xrecip(xsqrt(b3))*ds;
            T b4 = b + k3b*h;
            T k4b = ode_b(P, b4, w, G, ds), k4t = xrecip(xsqrt(b4))*ds;
            b = b + ((k1b + k2b*2.0) + (k3b*2.0 + k4b))*(h/6);
            t = t + ((k1t + k2t*2.0) + (k3t*2.0 + k4t))*(h/6);
        }
        tau = t; bplus = b;
        return;
    }
    int ns = P.numApprox;
    T prev = b, acc = make_zero(T());
    for (int j = 1; j <= ns; j++) {
        T cur = rk4_b(P, b, w, G, ds, (double)j/ns);
        acc = acc + xrecip(xsqrt(prev) + xsqrt(cur))*(2*ds*((double)j/ns - (double)(j - 1)/ns));
        prev = cur;
    }
    tau = acc; bplus = prev;
}

#include "msd_integ.hpp"
#include "msd_lossint.hpp"

/* ------------------------------------------------------------------------------------------
 * dynamic loss model (reference: mseetc/efficiency.py:7-141 with utils.py:197-220 and train.py:214-217).
 * Parameter block: forceMax, powerMax, vTurn, vMin, vMax, auxiliaries, cgT, cgB, R, V, totalMass, nx, ny, xb[nx+1], yb[ny+1],
 * coef[nx][ny][4][4] -- bicubic patches (about the cell centres) of the not-a-knot spline of the measured motor+converter losses.
 * With vTurn = 0 the table is the total loss power itself over (signed force [N], speed) and the gear / auxiliaries / transformer terms are
 * skipped: any loss function L(F, v) the reference accepts (train.py:190-219), tabulated by the host (mseetc/efficiency.py: TabulatedLosses).
 * ---------------------------------------------------------------------------------------- */
struct DynLoss {
    double Fmax, Pmax, vTurn, vMin, vMax, aux, cgT, cgB, R, V, M;
    int nx, ny;
    const double *xb, *yb, *coef;
    /* b: the parameter block, or its head alone (LDS copy) with the patches at `patches` (DevProb::loss / lossCoef) */
    __device__ __forceinline__ DynLoss(const double *b, const double *patches, const double massOverride)
        : Fmax(b[0]), Pmax(b[1]), vTurn(b[2]), vMin(b[3]), vMax(b[4]), aux(b[5]), cgT(b[6]), cgB(b[7]), R(b[8]), V(b[9]), M(massOverride > 0 ? massOverride : b[10]),
          nx((int)b[11]), ny((int)b[12]), xb(b + 13), yb(b + 13 + (int)b[11] + 1), coef(patches ? patches : b + 13 + (int)b[11] + 1 + (int)b[12] + 1) {}
};

/* index of the cell of an axis that holds x: the last interior breakpoint at or below x (0 below the first one and for a NaN) -- what a linear search over the
 * ascending breakpoints finds, by bisection (round 6: the breakpoints in LDS, LOSS_HEAD_CAP) */
__device__ __forceinline__ int table_cell(const double *bp, int n, double x)
{
    if (n < 2 || !(x >= bp[1])) return 0;      /* (the first cell: the evaluations at f = 0 and at the linear extension's +-tol) */
    int lo = 1, hi = n;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (x >= bp[mid]) lo = mid; else hi = mid; }
    return lo;
}

/* value and derivatives up to second order of the table at (x, y): t = {p, px, py, pxx, pxy, pyy}; zero outside the x range.  iy >= 0: the cell of y, already known
 * (the three evaluations of loss_rows share their speed) */
__device__ __forceinline__ void table_eval(const DynLoss &D, double x, double y, double (&t)[6], const int iy_known = -1)
{
#pragma unroll
    for (int k = 0; k < 6; k++) t[k] = 0;
    if (x < D.xb[0] || x > D.xb[D.nx]) return;
    const int ix = table_cell(D.xb, D.nx, x), iy = iy_known >= 0 ? iy_known : table_cell(D.yb, D.ny, y);
    const double dx = x - 0.5*(D.xb[ix] + D.xb[ix + 1]), dy = y - 0.5*(D.yb[iy] + D.yb[iy + 1]);
    const double *c = D.coef + 16*(ix*D.ny + iy);
    const double X[4] = {1, dx, dx*dx, dx*dx*dx}, X1[4] = {0, 1, 2*dx, 3*dx*dx}, X2[4] = {0, 0, 2, 6*dx};
    const double Y[4] = {1, dy, dy*dy, dy*dy*dy}, Y1[4] = {0, 1, 2*dy, 3*dy*dy}, Y2[4] = {0, 0, 2, 6*dy};
#pragma unroll
    for (int p = 0; p < 4; p++) {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const double cc = c[4*p + q];
            t[0] += cc*X[p]*Y[q]; t[1] += cc*X1[p]*Y[q]; t[2] += cc*X[p]*Y1[q];
            t[3] += cc*X2[p]*Y[q]; t[4] += cc*X1[p]*Y1[q]; t[5] += cc*X[p]*Y2[q];
        }
    }
}
/* the speed the table is read at (efficiency.py:40: constant continuation outside its range) and its cell */
__device__ __forceinline__ double table_speed(const DynLoss &D, double v) { return (v >= D.vMin && v <= D.vMax) ? v : (v < D.vMin ? D.vMin : D.vMax); }
__device__ __forceinline__ int table_speed_cell(const DynLoss &D, double v) { return table_cell(D.yb, D.ny, table_speed(D, v)); }

/* specific total losses [W/kg] of the traction (f >= 0) or braking (f < 0) branch as a jet in (f, v) */
__device__ __forceinline__ Jet spec_losses(const DynLoss &D, bool traction, double f, double v, const int iy = -1)      /* iy: table_speed_cell(D, v), or -1 */
{
    const Jet F = Jet{f, 1, 0, 0, 0, 0}*D.M, vv = Jet{v, 0, 1, 0, 0, 0};
    const bool inside = (v >= D.vMin && v <= D.vMax);
    const Jet vc = inside ? vv : Jet{v < D.vMin ? D.vMin : D.vMax, 0, 0, 0, 0, 0};                       /* efficiency.py:40 */
    /* vTurn = 0 marks a direct table: total losses [W] over (signed force [N], speed) -- a user-supplied loss function L(F, v)
     * (train.py:190-219 accepts any) tabulated by the host, traction and braking side as separate splines that meet in a cell edge at F = 0 */
    const bool direct = !(D.vTurn > 0);
    const Jet absF = (traction || direct) ? F : F*(-1.0);
    const Jet load = direct ? absF : (vc.v <= D.vTurn) ? absF*(100/D.Fmax) : (absF*vc)*(100/D.Pmax);     /* efficiency.py:7-12 */
    double t[6];
    table_eval(D, load.v, vc.v, t, iy);
    if (!direct && !(t[0] > 0)) return Jet{0, 0, 0, 0, 0, 0};                                             /* efficiency.py:137 */
    Jet motor;
    motor.v = t[0];
    motor.g0 = t[1]*load.g0 + t[2]*vc.g0;
    motor.g1 = t[1]*load.g1 + t[2]*vc.g1;
    motor.h00 = t[1]*load.h00 + t[2]*vc.h00 + t[3]*load.g0*load.g0 + 2*t[4]*load.g0*vc.g0 + t[5]*vc.g0*vc.g0;
    motor.h01 = t[1]*load.h01 + t[2]*vc.h01 + t[3]*load.g0*load.g1 + t[4]*(load.g0*vc.g1 + load.g1*vc.g0) + t[5]*vc.g0*vc.g1;
    motor.h11 = t[1]*load.h11 + t[2]*vc.h11 + t[3]*load.g1*load.g1 + 2*t[4]*load.g1*vc.g1 + t[5]*vc.g1*vc.g1;
    if (direct) return motor*(1/D.M);                                                                     /* train.py:216 */
    const Jet pW = traction ? F*vv : (F*vv)*(-1.0);                                                        /* efficiency.py:108-109 */
    const Jet gear = pW*(traction ? D.cgT : D.cgB);                                                        /* efficiency.py:112-116 */
    const Jet Pm = traction ? (pW + gear) + (motor + D.aux) : (pW - gear) - (motor + D.aux);
    const Jet inner = traction ? (Pm*(-4*D.R)) + D.V*D.V : (Pm*(4*D.R)) + D.V*D.V;
    const Jet dif = (xsqrt(inner)*(-1.0)) + D.V;
    const Jet trafo = (dif*dif)*(1/(4*D.R));                                                               /* efficiency.py:127-130 */
    return ((gear + motor) + (trafo + D.aux))*(1/D.M);                                                     /* train.py:216 */
}

/*
 * The loss rows of ocp.py:225-226 as functions of (f, vbar): g = L(f, v)/v of the traction part (row 0) and of the
 * regenerative-brake part (row 1), each extended linearly through f = 0 (utils.py:197-220).  lr[row] = {g, g_f, g_v, g_ff, g_fv, g_vv}.
 * In the linear-extension branch the third derivative that g_vv would need is dropped: that branch is not active at a solution
 * and only Newton's curvature is affected, not the NLP (same convention as the oracle).
 */
/* the split loss power itself (utils.py:197-220; specific, W/kg): l = {L, L_f, L_v, L_ff, L_fv, L_vv} of the traction part (row 0) or the regenerative-brake
 * part (row 1) at (f, v); beta = spec_losses(D, true, 0, v) */
__device__ __forceinline__ void loss_split(const DynLoss &D, const int row, double f, double v, const Jet &beta, double (&l)[6], const int iy = -1)
{
    const double tol = 1e-10;
    const bool traction = (row == 0);
    const bool truth = traction ? (f >= 0) : (f < 0);
    if (truth) {
        const Jet s = spec_losses(D, traction, f, v, iy);
        l[0] = s.v; l[1] = s.g0; l[2] = s.g1; l[3] = s.h00; l[4] = s.h01; l[5] = s.h11;
    } else {
        const Jet a = spec_losses(D, traction, traction ? tol : -tol, v, iy);
        l[0] = a.g0*f + beta.v; l[1] = a.g0; l[2] = a.h01*f + beta.g1; l[3] = 0; l[4] = a.h01; l[5] = beta.h11;
    }
}

__device__ __forceinline__ void loss_rows(const DynLoss &D, double f, double v, double (&lr)[2][6])
{
    const int iy = table_speed_cell(D, v);
    const Jet beta = spec_losses(D, true, 0.0, v, iy);
#pragma unroll
    for (int row = 0; row < 2; row++) {
        double l[6];
        loss_split(D, row, f, v, beta, l, iy);
        const double L = l[0], Lf = l[1], Lv = l[2], Lff = l[3], Lfv = l[4], Lvv = l[5];
        const double iv = 1/v;
        lr[row][0] = L*iv;
        lr[row][1] = Lf*iv;
        lr[row][2] = Lv*iv - L*iv*iv;
        lr[row][3] = Lff*iv;
        lr[row][4] = Lfv*iv - Lf*iv*iv;
        lr[row][5] = Lvv*iv - 2*Lv*iv*iv + 2*L*iv*iv*iv;
    }
}

#include "msd_lossint_table.hpp"

__device__ __forceinline__ double track_resistance(const DevProb &P, double grad, double curv)
{
    double c = fabs(curv);
    double cr = (c <= 1.0/300.0) ? P.g*0.5*c/(1 - 30*c) : P.g*0.65*c/(1 - 55*c);   /* train.py:252-253 as written */
    return P.g*grad*(1/P.rho) + cr*(1/P.rho);
}

/* ------------------------------------------------------------------------------------------
 * workgroup context: LDS carve-up + reductions
 * ---------------------------------------------------------------------------------------- */
struct Ctx {
    double *S, *xt, *xb, *xf, *o1, *o2, *o3, *filt, *red, *misc;
    double *xs, *o4, *o5, *o6, *o7;      /* FAST kernels only: sqrt(b) of the published point, four more outgoing contributions */
    double *st;                          /* MSD_STATIC_LDS: the nodes' constants */
    double *pool;                        /* shooting-integrator kernels: COOP_POOL doubles per wave (coop_adaptive) */
    int tid, lane, wave, nw, nt, red_slot;
    unsigned long long tmark;
    /* telemetry (tuning builds, -DMSD_TELEMETRY=1: tools/phase_cycles.py): thread 0 accumulates the shader cycles spent since the previous
     * mark into misc[2 + phase].  A mark drains the wave's outstanding memory operations, so the product build has none */
    __device__ __forceinline__ void mark(int phase)
    {
        if (MSD_TELEMETRY && tid == 0) {
            const unsigned long long t = __builtin_readcyclecounter();
            misc[2 + phase] += (double)(t - tmark);
            tmark = t;
        }
    }
};
enum { PH_EVAL = 0, PH_KKT, PH_ASSEMBLE, PH_RICCATI, PH_READBACK, PH_GPHID, PH_STEPLEN, PH_MERIT, PH_UPDATE, PH_OTHER, PH_COUNT };
/* sub-phases of the stage-parallel KKT solve (ParallelRiccati::solve; what is left under PH_RICCATI is its roll-out): they share the slots of the
 * three phases the fused iteration does not have, and misc[12..14] */
enum { PH_R_ELEM = PH_EVAL, PH_R_SCAN_P = PH_READBACK, PH_R_RECUR = PH_STEPLEN, PH_R_SCAN_G = 10, PH_R_FEED = 11, PH_R_SCAN_X = 12, PH_SLOTS = 13 };

struct OpMax { __device__ double operator()(double a, double b) const { return fmax(a, b); } __device__ static double identity() { return -INFINITY; } };
struct OpMin { __device__ double operator()(double a, double b) const { return fmin(a, b); } __device__ static double identity() { return INFINITY; } };
struct OpSum { __device__ double operator()(double a, double b) const { return a + b; } __device__ static double identity() { return 0.0; } };

/*
 * Reduction over the 64 lanes of a wave, result in every lane.  On the GPU: DPP moves (row_shr 1/2/4/8 inside the rows of 16 lanes,
 * row_bcast 15 and 31 across them) leave the total in lane 63, v_readlane broadcasts it -- 6 dependent steps of 3 VALU instructions
 * instead of 6 round trips through the LDS crossbar (__shfl_xor = ds_bpermute), and the iteration has some forty of these.
 */
#ifndef MSD_HOST_EMULATION
template <int CTRL, int ROW_MASK, class Op> __device__ __forceinline__ double dpp_combine(double x, Op op)
{
    const double id = Op::identity();
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(id), __double2loint(x), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(id), __double2hiint(x), CTRL, ROW_MASK, 0xf, false);
    return op(x, __hiloint2double(hi, lo));
}
template <class Op> __device__ __forceinline__ double wave_reduce(double x, Op op)
{
    x = dpp_combine<0x111, 0xf>(x, op);      /* row_shr:1 */
    x = dpp_combine<0x112, 0xf>(x, op);      /* row_shr:2 */
    x = dpp_combine<0x114, 0xf>(x, op);      /* row_shr:4 */
    x = dpp_combine<0x118, 0xf>(x, op);      /* row_shr:8 */
    x = dpp_combine<0x142, 0xa>(x, op);      /* row_bcast:15 into rows 1 and 3 */
    x = dpp_combine<0x143, 0xc>(x, op);      /* row_bcast:31 into rows 2 and 3 */
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), 63), __builtin_amdgcn_readlane(__double2loint(x), 63));
}
#else
template <class Op> __device__ __forceinline__ double wave_reduce(double x, Op op)
{
    for (int off = 32; off >= 1; off >>= 1) x = op(x, __shfl_xor(x, off));
    return x;
}
#endif

template <int K, class Op> __device__ __forceinline__ void block_reduce(double (&v)[K], Op op, Ctx &c)
{
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = wave_reduce(v[k], op);
    if (c.nw == 1) return;         /* one wave per workgroup: the wave reduction already left the result in every lane */
    double *buf = c.red + (c.red_slot & (RED_SLOTS - 1))*(MAX_WAVES*RED_K);
    c.red_slot++;
    if (c.lane == 0) {
#pragma unroll
        for (int k = 0; k < K; k++) buf[c.wave*RED_K + k] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < K; k++) {
        double r = buf[k];
        for (int w = 1; w < c.nw; w++) r = op(r, buf[w*RED_K + k]);
        v[k] = r;
    }
}

/* x of lane `src` (0..63) of the caller's wave, for K doubles at once; the result is unspecified when src is outside 0..63 */
#ifndef MSD_HOST_EMULATION
template <int K> __device__ __forceinline__ void wave_fetch(const double (&x)[K], double (&y)[K], int src)
{
    const int addr = (src & 63) << 2;
#pragma unroll
    for (int k = 0; k < K; k++) {
        const int lo = __builtin_amdgcn_ds_bpermute(addr, __double2loint(x[k]));
        const int hi = __builtin_amdgcn_ds_bpermute(addr, __double2hiint(x[k]));
        y[k] = __hiloint2double(hi, lo);
    }
}
#else
template <int K> __device__ __forceinline__ void wave_fetch(const double (&x)[K], double (&y)[K], int src)
{
    emu_wave_fetch(x, y, K, src);
}
#endif

/* Sigma and barrier gradient of one bounded scalar */
__device__ __forceinline__ void bar_terms(double x, double lb, double ub, bool hasL, bool hasU, double zL, double zU, double mu, double &Sg, double &gphi)
{
    double S = 0, g = 0;
    if (hasL) { S += zL/(x - lb); g -= mu/(x - lb); }
    if (hasU) { S += zU/(ub - x); g += mu/(ub - x); }
    if (hasL && !hasU) g += K_D*mu;
    if (!hasL && hasU) g -= K_D*mu;
    Sg = S; gphi = g;
}

/* keep z within [mu/(kappa_Sigma s), kappa_Sigma mu/s] (W&B eq. (16)); one reciprocal for both ends */
__device__ __forceinline__ double sigma_clamp(double z, double mu, double s)
{
    const double r = mu/s;
    return fmax(fmin(z, K_SIGMA*r), r*(1.0/K_SIGMA));
}

/* kp: bound_push = bound_frac (1e-2) for a cold start, the caller's push for a warm one */
__device__ __forceinline__ double push_in(double x, double lb, double ub, bool hasL, bool hasU, double kp)
{
    if (hasL && hasU) {
        double pL = fmin(kp*fmax(1.0, fabs(lb)), kp*(ub - lb));
        double pU = fmin(kp*fmax(1.0, fabs(ub)), kp*(ub - lb));
        if (x < lb + pL) x = lb + pL;
        if (x > ub - pU) x = ub - pU;
    } else if (hasL) { double pL = kp*fmax(1.0, fabs(lb)); if (x < lb + pL) x = lb + pL; }
    else if (hasU) { double pU = kp*fmax(1.0, fabs(ub)); if (x > ub - pU) x = ub - pU; }
    return x;
}

/* x^y for the heuristic thresholds of the line search (switching condition and alpha_min, W&B eqs. (19), (23)): single precision
 * through the hardware log2/exp2 -- a double pow() is several hundred instructions executed by every lane, three times per iteration,
 * for quantities that only steer which acceptance test applies.  x > 0. */
__device__ __forceinline__ double hpow(double x, double y)
{
    static_assert(K_MU_SUP == 1.5, "mu^1.5 is computed as mu*sqrt(mu) in the barrier update");
#ifndef MSD_HPOW
#define MSD_HPOW 1
#endif
    return MSD_HPOW ? (double)exp2f((float)y*log2f((float)x)) : pow(x, y);
}

/* sum of log(prod_j) over a thread's nodes with one logarithm: mantissas multiplied, exponents added (the products of up to twenty
 * slacks each would underflow if multiplied directly) */
struct LogSum {
    double m = 1.0;
    int e = 0;
#ifndef MSD_LOGSUM
#define MSD_LOGSUM 1
#endif
    double acc = 0.0;
    __device__ __forceinline__ void add(double prod) { if (MSD_LOGSUM) { int ex; m *= frexp(prod, &ex); e += ex; } else acc += log(prod); }
    __device__ __forceinline__ double value() const { return MSD_LOGSUM ? log(m) + 0.6931471805599453*(double)e : acc; }
};

__device__ __forceinline__ bool cmp_le(double lhs, double rhs, double basval) { return lhs - rhs <= 10.0*DBL_EPSILON*fabs(basval); }

/* ------------------------------------------------------------------------------------------
 * per-thread (= per shooting node) state.  Kept small on purpose: it has to stay in VGPRs for the
 * whole solve next to the temporaries of the jet arithmetic (256 registers at 2 waves per SIMD).
 * Bounds are not stored per variable: only the upper bound of b is node dependent.
 * ---------------------------------------------------------------------------------------- */
constexpr unsigned F_ON_T = 1u, F_ON_B = 2u, F_ON_F = 4u, F_ON_P = 8u, F_ON_S = 16u, F_IVAL = 32u, F_NODE = 64u;

/* one field of a node's iterate: CNT doubles, in registers or in the work area (stride NS = node slots of the workgroup) */
template <int CNT, int NS, bool MEM> struct Field;
template <int CNT, int NS> struct Field<CNT, NS, false> {
    static constexpr bool in_memory = false;
    double v[CNT];
    __device__ __forceinline__ void bind(double *) {}
    __device__ __forceinline__ double &operator[](int k) { return v[k]; }
    __device__ __forceinline__ const double &operator[](int k) const { return v[k]; }
};
template <int CNT, int NS> struct Field<CNT, NS, true> {
    static constexpr bool in_memory = true;
    double *p;
    __device__ __forceinline__ void bind(double *q) { p = q; }
    __device__ __forceinline__ double &operator[](int k) const { return p[k*NS]; }
};

/* layout of the work area, in fields of NS doubles */
constexpr int W_X = 0, W_SG = 5, W_LAM = 10, W_NU = 12, W_ZL = 17, W_ZU = 22, W_ZLS = 27, W_ZUS = 32, W_DSG = 37, W_RESC = 42, W_RESD = 44,
              W_EV = 49, W_LG = 62, W_LG_FIELDS = 30, W_FIELDS_ITERATE = W_LG + W_LG_FIELDS;      /* (W_LG: 2 x 5 derivatives of the loss rows, 2 x 15 with an integrated loss table) */
/* behind the iterate: what the feasibility restoration phase keeps (msd_resto.hpp; kernels whose horizon fits the LDS only).  Row scaling of the
 * dynamics (2), the relaxation variables n, p and their multipliers per relaxed row (4 x 7), their steps (4 x 7), the reference point (x 5,
 * sigma 5), the bound multipliers of the original problem (20), D of the two dynamics rows (2), the steps of the row multipliers (5), the filter of
 * the restoration problem (2 fields >= 2 FILT_CAP doubles), one field of scalars handed between the two iterations and five fields (>= 320
 * doubles) for the dense temporaries of riccati_resto -- on the stack they would size the scratch memory of every launch of the kernel */
constexpr int W_SC = W_FIELDS_ITERATE, W_RN = W_SC + 2, W_RP = W_SC + 9, W_RZN = W_SC + 16, W_RZP = W_SC + 23, W_RDN = W_SC + 30, W_RDP = W_SC + 37, W_RDZN = W_SC + 44,
              W_RDZP = W_SC + 51, W_XR = W_SC + 58, W_SGR = W_SC + 63, W_OZL = W_SC + 68, W_OZU = W_SC + 73, W_OZLS = W_SC + 78, W_OZUS = W_SC + 83, W_RD = W_SC + 88,
              W_DNU = W_SC + 90, W_RFILT = W_SC + 95, W_SCAL = W_SC + 97, W_RTMP = W_SC + 98;
/* behind that: the reference point of the watchdog procedure -- a copy of the iterate's fields W_X ... W_ZUS in the same order (Solver::wd_restore) */
constexpr int W_WD = W_SC + 103, W_WD_FIELDS = 37;
static_assert(W_DSG == W_WD_FIELDS, "the iterate proper: the fields in front of the slack steps");
/* behind that: what assemble(MODE_RESTO) hands to riccati_resto next to the stage blocks when the loss rows depend on the running time (integrateLosses):
 * curvature and gradient of the rows' share in d = t_{i+1} - t_i -- W_dd, W_bd, W_fd, W_pd, W_sd, h_d */
constexpr int W_RX = W_WD + W_WD_FIELDS, W_RX_FIELDS = 6;
constexpr int W_FIELDS = W_RX + W_RX_FIELDS;
__host__ __device__ constexpr size_t work_doubles(int node_slots) { return (size_t)W_FIELDS*node_slots; }
/* work area of a workgroup of the streamed (long-horizon) kernel: node fields (the same fields as above), stage blocks, six exchange arrays */
__host__ __device__ constexpr size_t stream_doubles(int N, int node_slots, bool dyn)
{
    return (size_t)W_FIELDS*node_slots + (size_t)(dyn ? 31 : 27)*(N + 1) + 6*(size_t)node_slots;
}

template <int NS, bool STREAM>
struct Node {
    int i;
    unsigned flags;
    double ds, G, sct, scb, ubB;
    /* iterate */
    Field<NV, NS, STREAM> x;
    Field<NR, NS, STREAM> sg;
    Field<2, NS, STREAM> lam;
    Field<NR, NS, STREAM> nu;
    Field<NV, NS, STREAM> zL, zU;
    Field<NR, NS, STREAM> zLs, zUs;
    /* slack part of the direction; (dx, new dynamics multipliers) stay in the node's LDS stage block */
    Field<NR, NS, STREAM> dsg;
    __device__ __forceinline__ void bind(double *w)      /* w: work area of the workgroup + node slot */
    {
        x.bind(w + W_X*NS); sg.bind(w + W_SG*NS); lam.bind(w + W_LAM*NS); nu.bind(w + W_NU*NS); zL.bind(w + W_ZL*NS); zU.bind(w + W_ZU*NS);
        zLs.bind(w + W_ZLS*NS); zUs.bind(w + W_ZUS*NS); dsg.bind(w + W_DSG*NS);
    }
    __device__ __forceinline__ bool ival() const { return (flags & F_IVAL) != 0; }
    __device__ __forceinline__ bool node() const { return (flags & F_NODE) != 0; }
    __device__ __forceinline__ bool on(int k) const { return (flags >> k) & 1u; }
};

/* workgroup-uniform data (scalar registers) */
struct Uni {
    double tlo, thi, blo, flo, fhi, plo, phi, slo;     /* relaxed variable bounds */
    bool rowOn[NR], rL[NR], rU[NR];
    double dL[NR], dU[NR], rs[NR];                     /* relaxed (scaled) row bounds, row scaling */
    double irs[NR];                                    /* 1/rs */
    double sf;                                         /* objective scaling */
};

/* keeps the instruction scheduler from interleaving the work of a thread's nodes (which would double the live registers) */
#ifndef MSD_NODE_FENCE
#define MSD_NODE_FENCE 1
#endif
/* SITE: 0 KKT pass, 1 assembly, 2 read-back, 3 step lengths, 4 trial point, 5 merit, 6 trial residuals, 7 evaluation, 8 start-up,
 * 9 barrier gradient, 10 accepted step, 11 after the loop; MSD_NODE_FENCE_SITES: bit per site (tuning builds) */
#ifndef MSD_NODE_FENCE_SITES
#define MSD_NODE_FENCE_SITES 0xFFFu
#endif
template <int SITE> __device__ __forceinline__ void node_fence() { if (MSD_NODE_FENCE && ((MSD_NODE_FENCE_SITES >> SITE) & 1u)) __builtin_amdgcn_sched_barrier(0); }

/* value the optimiser must treat as redefined here (no instruction is emitted) */
__device__ __forceinline__ void opaque(double &v) { asm volatile("" : "+v"(v)); }
/* same, but pins the value in an accumulation register at the fence: state that the next phase does not touch stays out of the
 * 256 architectural VGPRs (MSD_FENCE_AGPR: 0 none, 1 bound/slack multipliers, 2 all duals, steps and residuals) */
#ifndef MSD_FENCE_AGPR
#define MSD_FENCE_AGPR 0
#endif
__device__ __forceinline__ void opaque_a(double &v) { asm volatile("" : "+a"(v)); }
__device__ __forceinline__ void opaque_z(double &v) { if (MSD_FENCE_AGPR >= 1) opaque_a(v); else opaque(v); }
__device__ __forceinline__ void opaque_d(double &v) { if (MSD_FENCE_AGPR >= 2) opaque_a(v); else opaque(v); }

/* a workgroup-uniform value into scalar registers */
__device__ __forceinline__ int wg_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
/* The status words that steer a solve between its parts -- the return value of Solver::run (solved / hand over / restoration phase due / watchdog copy due),
 * the reason the general iteration parks its iterate -- go through scalar registers (round 6).  They are the same number in every lane by construction, but
 * the compiler kept them in vector registers and so saw divergent control flow around the calls of the cold functions (resto_entry, wd_store, wd_restore) and
 * around barriers: the calls sat inside exec-masked regions with lane-wise spills and reloads around them.  In round 6 the streamed follow-up kernel
 * (solve_kernel<512, 2, ..., PART = 2>) came out of an unrelated header change with a vector register -- the one the compiler keeps the constant 0 of the LDS
 * base in -- holding a field address in lanes 31 ... 63 behind such a region: every later 64-bit address built on that "zero" pointed into nowhere
 * (HSA_STATUS_ERROR_MEMORY_APERTURE_VIOLATION in the solves that go through a restoration phase or a watchdog procedure on that kernel; found with rocgdb's
 * precise memory reporting, profiles/r06/streamed_follow_up_fault.md).  Very likely the same mechanism as round 4's fault of the 64 x 1 one-brake follow-up
 * kernel and round 5's "non-deterministic" builds, which every scheduling-only switch made go away.  With the words in scalar registers the branches are scalar
 * and no call or barrier sits under a lane mask.  0: the code of rounds 1-5 (bisection builds) */
#ifndef MSD_UNIFORM_STATUS
#define MSD_UNIFORM_STATUS 1
#endif
__device__ __forceinline__ int status_uniform(int v) { return MSD_UNIFORM_STATUS ? wg_uniform(v) : v; }
__device__ __forceinline__ double uni(double v)
{
    int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

/* derivatives of the interval map kept between the evaluation and the KKT assembly */
struct Ev {
    double sb, sb1, b1;
    double tb, tw, tbb, tbw, tww, Bb, Bw, Bbb, Bbw, Bww;
    double lg[2][15];    /* dynamic loss rows: g_f, g_v, g_ff, g_fv, g_vv of the traction / brake row at (f, vbar);
                          * integrated losses: X, X_v, X_d, X_w, X_vv | X_vd, X_vw, X_dd, X_dw, X_ww (d = running time);
                          * integrated loss table: E_k and its 4 + 10 derivatives wrt (v, d, w, f) per row (Jet4);  the kernels keep LGW entries per row */
};

/* values of the interval functions at x: c (dynamics defects) and d (inequality rows) */
template <bool DERIV, int DYN, bool GEN, int FULL>
__device__ __forceinline__ void eval_interval(const DevProb &P, const Uni &U, const double nG, const double nds, const double (&x)[NV], double t1, double b1,
                                              double (&cv)[2], double (&dv)[NR], Ev &e, const Jet *ptau = nullptr, const Jet *pbp = nullptr,
                                              const double *vtau = nullptr, const double *vbp = nullptr)
{
    const double b = x[VB], f = x[VF], p = (full_pn(FULL) || (FULL == 0 && P.withPn)) ? x[VP] : 0.0, s = x[VS];
    if (DERIV) {
        Jet tau, bp;
        if (ptau) { tau = *ptau; bp = *pbp; }      /* (the interval map evaluated by the lanes of the wave together: Solver::coop_adaptive) */
        else if (GEN) interval_map_general<Jet>(P, b, f + p, nG, nds, tau, bp); else interval_map<Jet>(P, b, f + p, nG, nds, tau, bp);
        cv[0] = t1 - (x[VT] + tau.v); cv[1] = b1 - bp.v;
        e.tb = tau.g0; e.tw = tau.g1; e.tbb = tau.h00; e.tbw = tau.h01; e.tww = tau.h11;
        e.Bb = bp.g0; e.Bw = bp.g1; e.Bbb = bp.h00; e.Bbw = bp.h01; e.Bww = bp.h11;
    } else {
        double tau, bp;
        if (vtau) { tau = *vtau; bp = *vbp; }      /* (Solver::coop_values) */
        else if (GEN) interval_map_general<double>(P, b, f + p, nG, nds, tau, bp); else interval_map<double>(P, b, f + p, nG, nds, tau, bp);
        cv[0] = t1 - (x[VT] + tau); cv[1] = b1 - bp;
    }
    const double sb = sqrt(b), sb1 = sqrt(b1);
    if (DERIV) { e.sb = sb; e.sb1 = sb1; e.b1 = b1; }
    dv[RPW0] = U.rs[RPW0]*f*sb;                                             /* ocp.py:189 */
    dv[RPW1] = U.rs[RPW1]*f*sb1;
    dv[RACC] = U.rs[RACC]*(f + p - (P.sr0 + P.sr1*sb + P.sr2*b) - nG);     /* ocp.py:199 */
    if (DYN == LOSS_INTEGRATED_TABLE) {
        /* ocp.py:233-240 with a loss table: s - E_tr, s - E_rgb, E_k = int L_k(f, v(t)) dt over t1 - t (msd_lossint_table.hpp) */
        const DynLoss D(P.loss, P.lossCoef, P.lossMass);
        if (DERIV) {
            Jet4 E[2];
            loss_energy<Jet4>(P, D, sb, t1 - x[VT], f + p, f, nG, E);
            dv[RLTR] = U.rs[RLTR]*(s - E[0].v);
            dv[RLRG] = U.rs[RLRG]*(s - E[1].v);
#pragma unroll
            for (int k = 0; k < 2; k++) {
                e.lg[k][0] = E[k].v;
#pragma unroll
                for (int m = 0; m < 4; m++) e.lg[k][1 + m] = E[k].g[m];
#pragma unroll
                for (int m = 0; m < 10; m++) e.lg[k][5 + m] = E[k].h[m];
            }
        } else {
            double E[2];
            loss_energy<double>(P, D, sb, t1 - x[VT], f + p, f, nG, E);
            dv[RLTR] = U.rs[RLTR]*(s - E[0]);
            dv[RLRG] = U.rs[RLRG]*(s - E[1]);
        }
    } else if (DYN == LOSS_INTEGRATED) {
        /* ocp.py:233-240: s - E_tr, s - E_rgb with E = int loss power dt over t1 - t = +-kappa f X(v, t1 - t, f + p) */
        if (DERIV) {
            const Jet3 X = loss_distance<Jet3>(P, sb, t1 - x[VT], f + p, nG);
            dv[RLTR] = U.rs[RLTR]*(s - P.ct*f*X.v);
            dv[RLRG] = U.rs[RLRG]*(s + P.cr*f*X.v);
            e.lg[0][0] = X.v; e.lg[0][1] = X.g[0]; e.lg[0][2] = X.g[1]; e.lg[0][3] = X.g[2]; e.lg[0][4] = X.h[0];
            e.lg[1][0] = X.h[1]; e.lg[1][1] = X.h[2]; e.lg[1][2] = X.h[3]; e.lg[1][3] = X.h[4]; e.lg[1][4] = X.h[5];
        } else {
            const double X = loss_distance<double>(P, sb, t1 - x[VT], f + p, nG);
            dv[RLTR] = U.rs[RLTR]*(s - P.ct*f*X);
            dv[RLRG] = U.rs[RLRG]*(s + P.cr*f*X);
        }
    } else if (DYN == LOSS_TABLE) {
        const DynLoss D(P.loss, P.lossCoef, P.lossMass);
        double lr[2][6];
        loss_rows(D, f, 0.5*(sb + sb1), lr);                                 /* ocp.py:221: mid-point speed of the interval */
        dv[RLTR] = U.rs[RLTR]*(s - lr[0][0]);
        dv[RLRG] = U.rs[RLRG]*(s - lr[1][0]);
        if (DERIV) {
#pragma unroll
            for (int k = 0; k < 2; k++)
#pragma unroll
                for (int m = 0; m < 5; m++) e.lg[k][m] = lr[k][1 + m];
        }
    } else {
        dv[RLTR] = U.rs[RLTR]*(s - P.ct*f);                                 /* ocp.py:225 */
        dv[RLRG] = U.rs[RLRG]*(s + P.cr*f);                                 /* ocp.py:226 */
    }
}

/* objective contribution of node i (interval terms + terminal time), scaled by sf */
template <bool LI, int FULL, class NodeT>
__device__ __forceinline__ double objective_term(const DevProb &P, const NodeT &n, const double (&x)[NV], double q, double sf)
{
    double J = 0;
    if (n.ival()) {
        const double f = x[VF], p = (full_pn(FULL) || (FULL == 0 && P.withPn)) ? x[VP] : 0.0;
        if (full_energy(FULL) || (FULL == 0 && P.energyOpt)) {
            J = LI ? n.ds*f + x[VS] : n.ds*(f + x[VS]);                       /* ocp.py:223 resp. :235 */
            if (n.i > 0) J += 1e-3*(f - q)*(f - q);                           /* ocp.py:245 */
        } else J = 1e-4*(f*f + p*p);                                          /* ocp.py:150 */
    } else if (!full_energy(FULL) && n.i == P.N && (full_time(FULL) || !P.energyOpt)) J = x[VT];
    return sf*J/P.objDen;
}


/* ------------------------------------------------------------------------------------------
 * the serial part: Riccati recursion over the stage blocks in LDS (one thread).
 * Stage i: y = (dt, db, dq | df, dp, ds), next state = F y + r with
 *   dt+ = dt + Tb db + Tw (df + dp) + rt,  db+ = Bb db + Bw (df + dp) + rb,  dq+ = df.
 * The last interval eliminates df through db_N = 0 (b_N is a parameter of the NLP).
 * Returns false when a pivot is not positive (wrong inertia of the KKT matrix).
 * ---------------------------------------------------------------------------------------- */
/*
 * The last interval on top of the terminal value function (only t_N is a free variable of the NLP; b_N is a parameter): its b row,
 * Bb db + Bw (df + dp) + rb = 0, eliminates one of the two forces -- the one with the smaller curvature -- and the other one and the loss slack
 * are ordinary pivots (s first: its reciprocal comes from the assembly, NaN when the pivot is not positive).  Round 5: rounds 1-4 always
 * eliminated df.  With Fel on a bound its barrier curvature (1e11) then sat in every reduced entry and the value function of stage N-1 came out
 * as a difference of two such numbers, P_bb = eb^2 (G_ff - G_ff^2/(G_ff + G_pp - 2 G_fp)), whose rounding error times Bw^2 decided the inertia of
 * stage N-2 on degenerate problems (a zero-cost journey, random sweep seed 176: Restoration_Failed on the device, profiles/r04).  Eliminating the
 * softer force keeps the stiff one as a pivot of its own: at most one bit is lost (the oracle does the same, ms_oracle.c: compute_direction).
 * Out: value function of stage N-1 (Pn packed tt tb tq bb bq qq, pvn), the feedback in its uniform form K = (Kft Kfb Kfq | Kpt Kpb Kpq | kf kp),
 * KS = feedback of the slack, LG = the eliminated force's row of the stage system over (dt, db, dq, df, dp, ds, 1): the multiplier of the b row
 * is that row over Bw.  Returns false when the pivot of the kept force is not positive (wrong inertia).
 */
template <int DYN>
__device__ __forceinline__ bool last_interval(const double *s, const double Ptt, const double pt, const bool pn, double (&Pn)[6], double (&pvn)[3],
                                              double (&K)[8], double (&KS)[4], double (&LG)[7])
{
    const double Tb = s[S_TB], Tw = s[S_TW], Bb = s[S_BB], Bw = s[S_BW], rt = s[S_RT], rb = s[S_RB];
    const double Htt = s[S_HTT], Hbb = s[S_HBB], Hbq = s[S_HBQ], Hbf = s[S_HBF], Hbp = s[S_HBP], Hqq = s[S_HQQ], Hqf = s[S_HQF],
                 Hff = s[S_HFF], Hfp = s[S_HFP], Hpp = s[S_HPP];
    const double ht = s[S_HT], hb = s[S_HB], hq = s[S_HQ], hf = s[S_HF], hp = s[S_HP];
    const double Prt = Ptt*rt + pt;
    const double Mbt = Tb*Ptt, Mpt = Tw*Ptt;
    const double Gtt = Htt + Ptt, Gtb = Mbt, Gtf = Mpt;
    double Gtp = Mpt;
    const double Gbb = Hbb + Tb*Mbt, Gbq = Hbq, Gbf = Hbf + Tb*Mpt;
    double Gbp = Hbp + Tb*Mpt;
    const double Gqq = Hqq, Gqf = Hqf;
    const double Gff = Hff + Tw*Mpt;
    double Gfp = Hfp + Tw*Mpt, Gpp = Hpp + Tw*Mpt;
    const double gt = ht + Prt, gb = hb + Tb*Prt, gq = hq, gf = hf + Tw*Prt;
    double gp = hp + Tw*Prt;
    if (!pn) { Gtp = 0; Gbp = 0; Gfp = 0; Gpp = 1; gp = 0; }
    const double Gfs = s[S_GFS], is = s[S_IS], gs = s[S_GS], Gbs = DYN ? s[S_GBS] : 0.0, Gps = DYN ? s[S_GPS] : 0.0;      /* (p-s coupling: integrated loss rows) */
    const double eb = -Bb/Bw, e0 = -rb/Bw;
    /* e: the force eliminated through the b row, d(e) = eb db - d(k) + e0; k: the force that is kept */
#ifndef MSD_LAST_SWAP
#define MSD_LAST_SWAP 1      /* 0: always eliminate Fel (rounds 1-4; A/B builds) */
#endif
    const double oa = pn ? s[S_OA] : 0.0, ob = pn ? s[S_OB] : 0.0;      /* Gff - Gfp, Gpp - Gfp (the value function of stage N has no q entries) */
    const bool sw = MSD_LAST_SWAP && pn && ob < oa;
    const double Gte = sw ? Gtp : Gtf, Gtk = sw ? Gtf : Gtp, Gbe = sw ? Gbp : Gbf, Gbk = sw ? Gbf : Gbp, Gqe = sw ? 0.0 : Gqf, Gqk = sw ? Gqf : 0.0;
    const double Gee = sw ? Gpp : Gff, Gkk = sw ? Gff : Gpp, Gek = Gfp, Ges = sw ? Gps : Gfs, Gks = sw ? Gfs : Gps, ge = sw ? gp : gf, gk = sw ? gf : gp;
    LG[0] = Gte; LG[1] = Gbe; LG[2] = Gqe; LG[3] = sw ? Gfp : Gff; LG[4] = sw ? Gpp : Gfp; LG[5] = Ges; LG[6] = ge;
    const double gee = ge + Gee*e0;
    /* reduced blocks over (t, b, q | k, s) */
    const double oe = sw ? ob : oa;      /* Gee - Gek */
    double Hkk2 = oa + ob, Hks2 = Gks - Ges;
    double Hkt = Gtk - Gte, Hkb = (Gbk - Gbe) - oe*eb, Hkq = Gqk - Gqe;
    const double Hsb = Ges*eb + Gbs;
    double gk2 = (gk - ge) - oe*e0;
    const double gs2 = gs + Ges*e0;
    const double Xtt = Gtt, Xtb = Gtb + Gte*eb, Xbb = Gbb + 2*eb*Gbe + eb*eb*Gee, Xbq = Gbq + eb*Gqe, Xqq = Gqq;
    const double xt = gt + Gte*e0, xb = gb + Gbe*e0 + eb*gee, xq = gq + Gqe*e0;
    if (!pn) { Hkk2 = 1; Hks2 = 0; Hkt = 0; Hkb = 0; Hkq = 0; gk2 = 0; }
    /* 2x2 pivots: s first, then k */
    const double lks = Hks2*is, dk_ = Hkk2 - Hks2*lks;
    const double ik = 1.0/dk_;
    /* columns t, b, q and the vector: rhs = -(row k, row s) */
    const double Kkt = -(Hkt)*ik, Kst = -(Hks2*Kkt)*is;
    const double Kkb = -(Hkb - lks*Hsb)*ik, Ksb = -(Hsb + Hks2*Kkb)*is;
    const double Kkq = -(Hkq)*ik, Ksq = -(Hks2*Kkq)*is;
    const double kk = -(gk2 - lks*gs2)*ik, ks = -(gs2 + Hks2*kk)*is;
    Pn[sy(0, 0)] = Xtt + Hkt*Kkt;
    Pn[sy(0, 1)] = Xtb + Hkt*Kkb;
    Pn[sy(0, 2)] = Hkt*Kkq;
    Pn[sy(1, 1)] = Xbb + Hkb*Kkb + Hsb*Ksb;
    Pn[sy(1, 2)] = Xbq + Hkb*Kkq + Hsb*Ksq;
    Pn[sy(2, 2)] = Xqq + Hkq*Kkq;
    pvn[0] = xt + Hkt*kk; pvn[1] = xb + Hkb*kk + Hsb*ks; pvn[2] = xq + Hkq*kk;
    /* uniform feedback form: d(k) = Kk x + kk, d(e) = eb db - d(k) + e0 */
    const double Ket = -Kkt, Keb = eb - Kkb, Keq = -Kkq, ke = e0 - kk;
    K[0] = sw ? Kkt : Ket; K[1] = sw ? Kkb : Keb; K[2] = sw ? Kkq : Keq; K[6] = sw ? kk : ke;
    K[3] = sw ? Ket : Kkt; K[4] = sw ? Keb : Kkb; K[5] = sw ? Keq : Kkq; K[7] = sw ? ke : kk;
    if (!pn) { K[3] = 0; K[4] = 0; K[5] = 0; K[7] = 0; }
    KS[0] = Kst; KS[1] = Ksb; KS[2] = Ksq; KS[3] = ks;
    return dk_ > 0;
}

/* lg_out: when not null only the backward sweep runs and the seven numbers the forward sweep needs for the multiplier
 * of the last interval's eliminated row are written there */
template <int DYN>
__device__ __noinline__ bool riccati_solve(const int N, const bool pn, double *S, double *lg_out)
{
    constexpr int S_STRIDE = stage_stride(DYN);
    /* terminal value function: only t_N is a free variable of the NLP */
    double Ptt = S[N*S_STRIDE + S_HTT], Ptb = 0, Ptq = 0, Pbb = 0, Pbq = 0, Pqq = 0;
    double pt = S[N*S_STRIDE + S_HT], pb = 0, pq = 0;
    /* kept from the last interval for the multiplier of its eliminated row */
    double LGtf = 0, LGbf = 0, LGqf = 0, LGff = 0, LGfp = 0, LGfs = 0, Lgf = 0;

    bool ok = true;
    /* the last interval (last_interval) is peeled off the loop so that the loop body is branch free */
    {
        double *s = S + (N - 1)*S_STRIDE;
        double Pn[6], pvn[3], K[8], KS[4], LG[7];
        if (!last_interval<DYN>(s, Ptt, pt, pn, Pn, pvn, K, KS, LG)) ok = false;
        s[S_PN + 0] = Ptt; s[S_PN + 1] = 0; s[S_PN + 2] = 0; s[S_PN + 3] = 0; s[S_PN + 4] = 0; s[S_PN + 5] = 0;      /* value function of stage N, for the multipliers */
        s[S_PV + 0] = pt; s[S_PV + 1] = 0; s[S_PV + 2] = 0;
        LGtf = LG[0]; LGbf = LG[1]; LGqf = LG[2]; LGff = LG[3]; LGfp = LG[4]; LGfs = LG[5]; Lgf = LG[6];
#pragma unroll
        for (int k = 0; k < 6; k++) s[S_K + k] = K[k];
        s[S_KV + 0] = K[6]; s[S_KV + 1] = K[7];
#pragma unroll
        for (int k = 0; k < 4; k++) s[S_KS + k] = KS[k];
        Ptt = Pn[0]; Ptb = Pn[1]; Ptq = Pn[2]; Pbb = Pn[3]; Pbq = Pn[4]; Pqq = Pn[5]; pt = pvn[0]; pb = pvn[1]; pq = pvn[2];
    }
    auto backward = [&](const int i) {
        double *s = S + i*S_STRIDE;
        const double Tb = s[S_TB], Tw = s[S_TW], Bb = s[S_BB], Bw = s[S_BW], rt = s[S_RT], rb = s[S_RB];
        const double Htt = s[S_HTT], Hbb = s[S_HBB], Hbq = s[S_HBQ], Hbf = s[S_HBF], Hbp = s[S_HBP], Hqq = s[S_HQQ], Hqf = s[S_HQF],
                     Hff = s[S_HFF], Hfp = s[S_HFP], Hpp = s[S_HPP];
        const double ht = s[S_HT], hb = s[S_HB], hq = s[S_HQ], hf = s[S_HF], hp = s[S_HP];
        const double oa = s[S_OA], ob = s[S_OB];
        /* stash the value function of stage i+1 for the multipliers */
        s[S_PN + 0] = Ptt; s[S_PN + 1] = Ptb; s[S_PN + 2] = Ptq; s[S_PN + 3] = Pbb; s[S_PN + 4] = Pbq; s[S_PN + 5] = Pqq;
        s[S_PV + 0] = pt; s[S_PV + 1] = pb; s[S_PV + 2] = pq;

        /* P r + p */
        const double Prt = Ptt*rt + Ptb*rb + pt, Prb = Ptb*rt + Pbb*rb + pb, Prq = Ptq*rt + Pbq*rb + pq;
        /* M_c = P F[:,c] for the columns b, p, f (column t is P[:,t]) */
        const double Mbt = Tb*Ptt + Bb*Ptb, Mbb = Tb*Ptb + Bb*Pbb;
        const double Mpt = Tw*Ptt + Bw*Ptb, Mpb = Tw*Ptb + Bw*Pbb, Mpq = Tw*Ptq + Bw*Pbq;
        const double Mft = Mpt + Ptq, Mfb = Mpb + Pbq, Mfq = Mpq + Pqq;
        /* G = H + F^T P F, g = h + F^T (P r + p) */
        double Gtt = Htt + Ptt, Gtb = Mbt, Gtf = Mft, Gtp = Mpt;
        double Gbb = Hbb + Tb*Mbt + Bb*Mbb, Gbq = Hbq, Gbf = Hbf + Tb*Mft + Bb*Mfb, Gbp = Hbp + Tb*Mpt + Bb*Mpb;
        double Gqq = Hqq, Gqf = Hqf;
        double Gff = Hff + Tw*Mft + Bw*Mfb + Mfq, Gfp = Hfp + Tw*Mpt + Bw*Mpb + Mpq;
        double Gpp = Hpp + Tw*Mpt + Bw*Mpb;
        double gt = ht + Prt, gb = hb + Tb*Prt + Bb*Prb, gq = hq;
        double gf = hf + Tw*Prt + Bw*Prb + Prq, gp = hp + Tw*Prt + Bw*Prb;
        if (!pn) { Gtp = 0; Gbp = 0; Gfp = 0; Gpp = 1; gp = 0; }

        double Kft, Kfb, Kfq, Kpt, Kpb, Kpq, kf, kp;
        double nPtt, nPtb, nPtq, nPbb, nPbq, nPqq, npt, npb, npq;
        {
            /* s is already eliminated (assemble()): pivots of Guu in the order p, f */
            const double ip = 1.0/Gpp, lfp = Gfp*ip;
            const double df_ = pn ? (oa + Mpq + Pqq) + lfp*(ob - Mpq) : Gff;      /* = Gff - Gfp^2/Gpp, from the own curvatures (S_OA) */
            if (!(Gpp > 0) || !(df_ > 0)) ok = false;
            const double iff = 1.0/df_;
            Kft = -(Gtf - lfp*Gtp)*iff; Kpt = -(Gtp + Gfp*Kft)*ip;
            Kfb = -(Gbf - lfp*Gbp)*iff; Kpb = -(Gbp + Gfp*Kfb)*ip;
            Kfq = -(Gqf)*iff;           Kpq = -(Gfp*Kfq)*ip;
            kf = -(gf - lfp*gp)*iff;    kp = -(gp + Gfp*kf)*ip;
            nPtt = Gtt + Gtf*Kft + Gtp*Kpt;
            nPtb = Gtb + Gtf*Kfb + Gtp*Kpb;
            nPtq = Gtf*Kfq + Gtp*Kpq;
            nPbb = Gbb + Gbf*Kfb + Gbp*Kpb;
            nPbq = Gbq + Gbf*Kfq + Gbp*Kpq;
            nPqq = Gqq + Gqf*Kfq;
            npt = gt + Gtf*kf + Gtp*kp; npb = gb + Gbf*kf + Gbp*kp; npq = gq + Gqf*kf;
        }
        if (!pn) { Kpt = 0; Kpb = 0; Kpq = 0; kp = 0; }
        s[S_K + 0] = Kft; s[S_K + 1] = Kfb; s[S_K + 2] = Kfq; s[S_K + 3] = Kpt; s[S_K + 4] = Kpb; s[S_K + 5] = Kpq;
        s[S_KV + 0] = kf; s[S_KV + 1] = kp;
        Ptt = nPtt; Ptb = nPtb; Ptq = nPtq; Pbb = nPbb; Pbq = nPbq; Pqq = nPqq; pt = npt; pb = npb; pq = npq;
    };
    for (int i = N - 2; i >= 0; i--) backward(i);     /* no early exit: a wrong inertia is rare, the branch is not */
    if (!ok) return false;
    if (lg_out) {
        lg_out[0] = LGtf; lg_out[1] = LGbf; lg_out[2] = LGqf; lg_out[3] = LGff; lg_out[4] = LGfp; lg_out[5] = LGfs; lg_out[6] = Lgf;
        return true;
    }

    /* forward sweep of (dt, db, dq) and the controls; x_0 is a parameter.  The step of s and the multipliers of the dynamics
     * follow from these per node (Solver::finish_direction), except in the last interval */
    double dt = 0, db = 0, dq = 0;
    auto forward = [&](const int i, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        double *s = S + i*S_STRIDE;
        const double Tb = s[S_TB], Tw = s[S_TW], Bb = s[S_BB], Bw = s[S_BW], rt = s[S_RT], rb = s[S_RB];
        const double df = s[S_K + 0]*dt + s[S_K + 1]*db + s[S_K + 2]*dq + s[S_KV + 0];
        const double dp = pn ? s[S_K + 3]*dt + s[S_K + 4]*db + s[S_K + 5]*dq + s[S_KV + 1] : 0.0;
        const double dw = df + dp;
        const double nt = dt + Tb*db + Tw*dw + rt;
        const double nb = LAST ? 0.0 : Bb*db + Bw*dw + rb;
        if (LAST) {
            const double dsl = s[S_KS + 0]*dt + s[S_KS + 1]*db + s[S_KS + 2]*dq + s[S_KS + 3];
            s[S_DS] = dsl;
            s[S_LB] = (LGtf*dt + LGbf*db + LGqf*dq + LGff*df + LGfp*dp + LGfs*dsl + Lgf)/Bw;
        }
        s[S_DT] = dt; s[S_DB] = db; s[S_DF] = df; s[S_DP] = dp;
        dt = nt; db = nb; dq = df;
    };
    for (int i = 0; i < N - 1; i++) forward(i, std::false_type());
    forward(N - 1, std::true_type());
    S[N*S_STRIDE + S_DT] = dt; S[N*S_STRIDE + S_DB] = 0.0; S[N*S_STRIDE + S_DF] = 0.0;
    return true;
}

/* ------------------------------------------------------------------------------------------
 * Serial sweeps of the restoration problem's Newton system (cold path, one lane; assemble(MODE_RESTO) leaves the unreduced blocks: the slack
 * variable is a control of its own here).  The two dynamics rows of an interval are relaxed there: x+ = F y + r + D lam+ on (t, b), with
 * D = n/z_n + p/z_p > 0 of the row (unscaled).  With the cross terms y_i^T E x_{i+1} of the loss rows that reach into the next node (dynamic loss
 * table: (b_i, s_i) with b_{i+1}; integrated loss rows: (t_i, b_i, f_i, p_i, s_i) with t_{i+1} through the running time) the stage system is
 *     G = H + F^T P F + E F2 + (E F2)^T - Q^T M Q,   Q = (P F)[tb,:] + E^T,   M = (P2 + D^-1)^-1 = D^1/2 (I + D^1/2 P2 D^1/2)^-1 D^1/2,
 * and the forward sweep takes lam+ = -(I + P2 D)^-1 (P a + p + E^T y) from the relaxed rows themselves (a = F y + r), so that their linearisation
 * holds to rounding -- the oracle's compute_direction, restated on the kernel's blocks.  b_N stays a parameter: in the last interval one force is
 * eliminated through  F_b y + r_b = -sqrt(D_b) v  with v = sqrt(D_b) lam_b+ of unit curvature in its slot -- the hard elimination of last_interval in
 * the limit D_b -> 0.  Leaves (dt, db, df, dp, ds) and the new multipliers (lt, lb) of every interval in its block.
 * X: the integrated loss rows' share in the running time (fields W_RX of the work area, stride NS).
 * ---------------------------------------------------------------------------------------- */
/* state of the backward sweep: value function and its gradient at the stage behind the one at work */
struct RestoSweep { double P[3][3], pv[3]; bool ok, swapLast; };

/* one stage of the backward sweep (w: stage i+1 in, stage i out); feedback, value function and gradient of stage i+1 into the stage's block */
template <int DYN, bool last>
__device__ __forceinline__ void resto_backward_stage(const int i, const bool pn, double *S, const double *Dtv, const double *Dbv, const double *X, const int NS, RestoSweep &w)
{
    constexpr int S_STRIDE = stage_stride(DYN);
    /* cross terms of stage i: E[a][m], a over (t b q f p s), m over (t+, b+) */
    auto cross = [&](const int i, const bool last_, const double *s, double (&E)[6][2], double (&H)[6][6]) {
#pragma unroll
        for (int a = 0; a < 6; a++) E[a][0] = E[a][1] = 0;
        if (DYN == LOSS_TABLE && !last_) { E[1][1] = s[S_EB]; E[5][1] = s[S_ES]; }
        if (loss_integrated(DYN)) {
            const double Wdd = X[0*NS + i], Wbd = X[1*NS + i], Wfd = X[2*NS + i], Wpd = X[3*NS + i], Wsd = X[4*NS + i];
            /* d = t+ - t: entries at t are minus, at t+ plus the running time's (H_tt already carries W_dd, the next node's too: assemble) */
            H[0][1] = H[1][0] = -Wbd; H[0][3] = H[3][0] = -Wfd; H[0][4] = H[4][0] = -Wpd; H[0][5] = H[5][0] = -Wsd;
            E[0][0] = -Wdd; E[1][0] = Wbd; E[3][0] = Wfd; E[4][0] = Wpd; E[5][0] = Wsd;
        }
    };
    constexpr int nu = 3;      /* controls to eliminate: (f, p, s), in the last interval (v, k, s) with k the force that is kept */
    double *s = S + i*S_STRIDE;
    const double Tb = s[S_TB], Tw = s[S_TW], Bb = s[S_BB], Bw = s[S_BW], rt = s[S_RT], rb = s[S_RB];
    const double Dt = Dtv[i], Db = Dbv[i];
    double H[6][6], G[6][6], h[6], g[6], PF[3][6], Pr[3], F[3][6], r[3], L[3][3], K[3][4], E[6][2], Q[2][6];
#pragma unroll
    for (int a = 0; a < 6; a++) {
        h[a] = 0;
#pragma unroll
        for (int b = 0; b < 6; b++) H[a][b] = 0;
    }
    H[0][0] = s[S_HTT]; H[1][1] = s[S_HBB]; H[1][2] = H[2][1] = s[S_HBQ]; H[1][3] = H[3][1] = s[S_HBF]; H[1][4] = H[4][1] = s[S_HBP];
    H[2][2] = s[S_HQQ]; H[2][3] = H[3][2] = s[S_HQF]; H[3][3] = s[S_HFF]; H[3][4] = H[4][3] = s[S_HFP]; H[4][4] = s[S_HPP];
    h[0] = s[S_HT]; h[1] = s[S_HB]; h[2] = s[S_HQ]; h[3] = s[S_HF]; h[4] = s[S_HP];
    H[3][5] = H[5][3] = s[S_GFS]; H[5][5] = 1.0/s[S_IS]; h[5] = s[S_GS];      /* the slack variable's row, unreduced (1/NaN when its pivot is not positive: the Cholesky below fails) */
    if (DYN) { H[1][5] = H[5][1] = s[S_GBS]; H[4][5] = H[5][4] = s[S_GPS]; }
    cross(i, last, s, E, H);
    /* value function of stage i+1 as it is, for the forward sweep */
    s[S_PN + 0] = w.P[0][0]; s[S_PN + 1] = w.P[0][1]; s[S_PN + 2] = w.P[0][2]; s[S_PN + 3] = w.P[1][1]; s[S_PN + 4] = w.P[1][2]; s[S_PN + 5] = w.P[2][2];
    s[S_PV + 0] = w.pv[0]; s[S_PV + 1] = w.pv[1]; s[S_PV + 2] = w.pv[2];
    /* through the relaxed rows */
    double M00, M01, M11;
    if (last) {
        const double den = 1 + w.P[0][0]*Dt;
        if (!(den > 0)) w.ok = false;
        M00 = Dt/den; M01 = 0; M11 = 0;
    } else {
        const double st = sqrt(Dt), sb = sqrt(Db);
        const double ma = 1 + st*w.P[0][0]*st, mb = st*w.P[0][1]*sb, mc = 1 + sb*w.P[1][1]*sb, det = ma*mc - mb*mb;
        if (!(det > 0) || !(ma > 0)) w.ok = false;
        M00 = st*(mc/det)*st; M01 = -st*(mb/det)*sb; M11 = sb*(ma/det)*sb;
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 6; b++) F[a][b] = 0;
    F[0][0] = 1; F[0][1] = Tb; F[0][3] = Tw; F[0][4] = pn ? Tw : 0.0; F[1][1] = Bb; F[1][3] = Bw; F[1][4] = pn ? Bw : 0.0; F[2][3] = 1;
    r[0] = rt; r[1] = last ? 0.0 : rb; r[2] = 0;
    if (last) {
#pragma unroll
        for (int b = 0; b < 6; b++) F[1][b] = 0;      /* db_N = 0: the b row is the equality handled below, not a transition */
    }
    /* G = H + F^T P F + E F2 + (E F2)^T, g = h + F^T (P r + p) + E r2 */
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int b = 0; b < 6; b++) PF[a][b] = w.P[a][0]*F[0][b] + w.P[a][1]*F[1][b] + w.P[a][2]*F[2][b];
        Pr[a] = w.pv[a] + w.P[a][0]*r[0] + w.P[a][1]*r[1] + w.P[a][2]*r[2];
    }
#pragma unroll
    for (int a = 0; a < 6; a++) {
#pragma unroll
        for (int b = 0; b < 6; b++)
            G[a][b] = H[a][b] + F[0][a]*PF[0][b] + F[1][a]*PF[1][b] + F[2][a]*PF[2][b]
                      + (DYN ? E[a][0]*F[0][b] + E[a][1]*F[1][b] + F[0][a]*E[b][0] + F[1][a]*E[b][1] : 0.0);
        g[a] = h[a] + F[0][a]*Pr[0] + F[1][a]*Pr[1] + F[2][a]*Pr[2] + (DYN ? E[a][0]*r[0] + E[a][1]*r[1] : 0.0);
    }
    /* - Q^T M Q on the relaxed rows */
#pragma unroll
    for (int b = 0; b < 6; b++) { Q[0][b] = PF[0][b] + (DYN ? E[b][0] : 0.0); Q[1][b] = PF[1][b] + (DYN ? E[b][1] : 0.0); }
#pragma unroll
    for (int a = 0; a < 6; a++) {
        const double q0 = M00*Q[0][a] + M01*Q[1][a], q1 = M01*Q[0][a] + M11*Q[1][a];
#pragma unroll
        for (int b = 0; b < 6; b++) G[a][b] -= q0*Q[0][b] + q1*Q[1][b];
        g[a] -= q0*Pr[0] + q1*Pr[1];
    }
    if (!pn) {
#pragma unroll
        for (int a = 0; a < 6; a++) G[4][a] = G[a][4] = 0;
        G[4][4] = 1; g[4] = 0;
    }
    if (last) {
        /* d(e) = eb db - d(k) + e0 - kap v: the force with the smaller curvature is eliminated (last_interval); when that is Fpb the two forces
         * swap their slots here, and back in the forward sweep (every index a compile-time constant) */
        w.swapLast = pn && G[4][4] < G[3][3];
        if (w.swapLast) {
#pragma unroll
            for (int a = 0; a < 6; a++) { const double v = G[3][a]; G[3][a] = G[4][a]; G[4][a] = v; }
#pragma unroll
            for (int a = 0; a < 6; a++) { const double v = G[a][3]; G[a][3] = G[a][4]; G[a][4] = v; }
            const double v = g[3]; g[3] = g[4]; g[4] = v;
        }
        const double eb = -Bb/Bw, e0 = -rb/Bw, kap = sqrt(Db)/Bw;
        double T[6][6], GT[6][6], G2[6][6], gy[6], g2[6];
#pragma unroll
        for (int a = 0; a < 6; a++)
#pragma unroll
            for (int b = 0; b < 6; b++) T[a][b] = a == b ? 1.0 : 0.0;
        T[3][3] = -kap; T[3][1] = eb; T[3][4] = pn ? -1.0 : 0.0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
            for (int b = 0; b < 6; b++) {
                double v = 0;
#pragma unroll
                for (int m = 0; m < 6; m++) v += G[a][m]*T[m][b];
                GT[a][b] = v;
            }
            gy[a] = g[a] + G[a][3]*e0;
        }
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
            for (int b = 0; b < 6; b++) {
                double v = 0;
#pragma unroll
                for (int m = 0; m < 6; m++) v += T[m][a]*GT[m][b];
                G2[a][b] = v;
            }
            double v = 0;
#pragma unroll
            for (int m = 0; m < 6; m++) v += T[m][a]*gy[m];
            g2[a] = v;
        }
        G2[3][3] += 1.0;
#pragma unroll
        for (int a = 0; a < 6; a++) {
#pragma unroll
            for (int b = 0; b < 6; b++) G[a][b] = G2[a][b];
            g[a] = g2[a];
        }
    }
    /* eliminate the controls: Cholesky of the control block */
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) L[a][b] = 0;
#pragma unroll
    for (int j = 0; j < nu; j++) {
        double d = G[3 + j][3 + j];
#pragma unroll
        for (int k = 0; k < j; k++) d -= L[j][k]*L[j][k];
        if (!(d > 0) || !isfinite(d)) { w.ok = false; d = 1.0; }
        L[j][j] = sqrt(d);
#pragma unroll
        for (int a = j + 1; a < nu; a++) {
            double v = G[3 + a][3 + j];
#pragma unroll
            for (int k = 0; k < j; k++) v -= L[a][k]*L[j][k];
            L[a][j] = v/L[j][j];
        }
    }
#pragma unroll
    for (int c = 0; c < 4; c++) {      /* K: columns t, b, q and the constant */
        double y[3] = {0, 0, 0}, x[3] = {0, 0, 0};
#pragma unroll
        for (int a = 0; a < nu; a++) {
            double v = -(c < 3 ? G[3 + a][c] : g[3 + a]);
#pragma unroll
            for (int k = 0; k < a; k++) v -= L[a][k]*y[k];
            y[a] = v/L[a][a];
        }
#pragma unroll
        for (int a = nu - 1; a >= 0; a--) {
            double v = y[a];
#pragma unroll
            for (int k = a + 1; k < nu; k++) v -= L[k][a]*x[k];
            x[a] = v/L[a][a];
        }
#pragma unroll
        for (int a = 0; a < 3; a++) K[a][c] = x[a];
    }
#pragma unroll
    for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int b = 0; b < 3; b++) {
            double v = G[a][b];
#pragma unroll
            for (int m = 0; m < nu; m++) v += G[a][3 + m]*K[m][b];
            w.P[a][b] = v;
        }
        double v = g[a];
#pragma unroll
        for (int m = 0; m < nu; m++) v += G[a][3 + m]*K[m][3];
        w.pv[a] = v;
    }
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = a + 1; b < 3; b++) { const double m = 0.5*(w.P[a][b] + w.P[b][a]); w.P[a][b] = w.P[b][a] = m; }
    if (!pn) { K[1][0] = K[1][1] = K[1][2] = K[1][3] = 0; }
    s[S_K + 0] = K[0][0]; s[S_K + 1] = K[0][1]; s[S_K + 2] = K[0][2]; s[S_K + 3] = K[1][0]; s[S_K + 4] = K[1][1]; s[S_K + 5] = K[1][2];
    s[S_KV + 0] = K[0][3]; s[S_KV + 1] = K[1][3];
    s[S_KS + 0] = K[2][0]; s[S_KS + 1] = K[2][1]; s[S_KS + 2] = K[2][2]; s[S_KS + 3] = K[2][3];      /* (over the slack row's inputs, which are used up) */
}

/* one stage of the forward sweep: (dt, db, dq) of stage i in, of stage i+1 out; steps and new multipliers into the stage's block */
template <int DYN>
__device__ __forceinline__ void resto_forward_stage(const int i, const bool last, const bool pn, double *S, const double *Dtv, const double *Dbv, const double *X, const int NS,
                                                    const bool swapLast, double &x0, double &x1, double &x2)
{
    constexpr int S_STRIDE = stage_stride(DYN);
    double *s = S + i*S_STRIDE;
    const double Tb = s[S_TB], Tw = s[S_TW], Bb = s[S_BB], Bw = s[S_BW], rt = s[S_RT], rb = s[S_RB];
    const double Dt = Dtv[i], Db = Dbv[i];
    const double u0 = s[S_K + 0]*x0 + s[S_K + 1]*x1 + s[S_K + 2]*x2 + s[S_KV + 0];
    const double u1 = pn ? s[S_K + 3]*x0 + s[S_K + 4]*x1 + s[S_K + 5]*x2 + s[S_KV + 1] : 0.0;
    const double dsl = s[S_KS + 0]*x0 + s[S_KS + 1]*x1 + s[S_KS + 2]*x2 + s[S_KS + 3];
    double df = u0, dp = u1;
    if (last) {
        /* u0: v, u1: the force that was kept */
        const double de = -Bb/Bw*x1 - u1 - rb/Bw - sqrt(Db)/Bw*u0;
        df = swapLast ? u1 : de; dp = swapLast ? de : u1;
    }
    const double dw = df + dp;
    const double at = x0 + Tb*x1 + Tw*dw + rt, ab = last ? 0.0 : Bb*x1 + Bw*dw + rb;
    const double Ptt = s[S_PN + 0], Ptb = s[S_PN + 1], Ptq = s[S_PN + 2], Pbb = s[S_PN + 3], Pbq = s[S_PN + 4];
    double g0 = Ptt*at + Ptb*ab + Ptq*df + s[S_PV + 0], g1 = Ptb*at + Pbb*ab + Pbq*df + s[S_PV + 1];
    /* + E^T y */
    if (DYN == LOSS_TABLE && !last) g1 += s[S_EB]*x1 + s[S_ES]*dsl;
    if (loss_integrated(DYN)) g0 += -X[0*NS + i]*x0 + X[1*NS + i]*x1 + X[2*NS + i]*df + X[3*NS + i]*dp + X[4*NS + i]*dsl;
    double lt, lb, nt, nb;
    if (last) { lt = -g0/(1 + Ptt*Dt); lb = u0/sqrt(Db); nt = at + Dt*lt; nb = 0; }
    else {
        const double a00 = 1 + Ptt*Dt, a01 = Ptb*Db, a10 = Ptb*Dt, a11 = 1 + Pbb*Db, det = a00*a11 - a01*a10;
        lt = -(a11*g0 - a01*g1)/det; lb = -(a00*g1 - a10*g0)/det;
        nt = at + Dt*lt; nb = ab + Db*lb;
    }
    s[S_DT] = x0; s[S_DB] = x1; s[S_DF] = df; s[S_DP] = dp; s[S_DS] = dsl; s[S_LT] = lt; s[S_LB] = lb;
    x0 = nt; x1 = nb; x2 = df;
}

template <int DYN>
__device__ __noinline__ bool riccati_resto(const int N, const bool pn, double *S, const double *Dtv, const double *Dbv, const double *X, const int NS)
{
    constexpr int S_STRIDE = stage_stride(DYN);
    /* Round 3 kept every array of the sweeps in the work area (run-time indices): the stack of this cold function sized the scratch memory of
     * every launch of the kernel it was compiled into.  Since round 4 it only lives in follow-up kernels (solve_kernel: PART = 2), so the
     * arrays are locals with compile-time indices -- registers -- and the last interval is peeled off the stage loop */
    RestoSweep w;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        w.pv[a] = 0;
#pragma unroll
        for (int b = 0; b < 3; b++) w.P[a][b] = 0;
    }
    w.P[0][0] = S[N*S_STRIDE + S_HTT]; w.pv[0] = S[N*S_STRIDE + S_HT];
    w.ok = true; w.swapLast = false;
    resto_backward_stage<DYN, true>(N - 1, pn, S, Dtv, Dbv, X, NS, w);
#pragma unroll 1
    for (int i = N - 2; i >= 0; i--) resto_backward_stage<DYN, false>(i, pn, S, Dtv, Dbv, X, NS, w);
    if (!w.ok) return false;

    double x0 = 0, x1 = 0, x2 = 0;      /* (dt, db, dq) of the stage; x_0 is a parameter */
#pragma unroll 1
    for (int i = 0; i < N; i++) resto_forward_stage<DYN>(i, i == N - 1, pn, S, Dtv, Dbv, X, NS, w.swapLast, x0, x1, x2);
    S[N*S_STRIDE + S_DT] = x0; S[N*S_STRIDE + S_DB] = 0.0; S[N*S_STRIDE + S_DF] = 0.0;
    return true;
}

/* ------------------------------------------------------------------------------------------ */
#ifndef MSD_PARALLEL_NOINLINE
#define MSD_PARALLEL_NOINLINE 0     /* the stage-parallel KKT solve as a real function: one copy of its code, its own register allocation */
#endif
#if MSD_PARALLEL_NOINLINE
#define MSD_PARALLEL_ATTR __noinline__
#else
#define MSD_PARALLEL_ATTR __forceinline__
#endif

template <int SPT, int DYN>
struct ParallelRiccati {
    /* ------------------------------------------------------------------------------------------
     * stage-parallel KKT solve (all threads; msd_scan.hpp).  Thread l owns the consecutive regular stages l*SPT .. l*SPT+SPT-1
     * (i < N-1; the last interval, which eliminates df through b_N, is the terminal piece every thread evaluates for itself).
     *   1  control-eliminated triples (A, C, J) of the own stages, composed to the chunk's triple
     *   2  suffix scan of the triples over the lanes (and waves): value-function matrix at every chunk end
     *   3  ordinary recursion over the own stages from there: pivots (inertia), feedback, value function stash; the chunk's
     *      affine map of the value-function gradient
     *   4  suffix scan of those maps: gradient at every chunk end;  5  feed-forward of the own stages, closed-loop chunk map
     *   6  prefix scan of the closed-loop maps: state at every chunk start;  7  roll-out of the own stages
     * Returns 1 (direction left in the stage blocks like riccati_solve + forward sweep), 0 (a pivot is not positive: wrong
     * inertia) or -1 (a scan step broke down numerically: the caller assembles again and takes the serial sweep).
     * ---------------------------------------------------------------------------------------- */
    static constexpr int S_STRIDE = stage_stride(DYN);
    static constexpr int RED_MAT = 0, RED_AFB = 21*8, RED_AFF = 21*8 + 12*8;
    static_assert(21*8 + 12*8 + 12*8 <= RED_SLOTS*MAX_WAVES*RED_K, "wave totals of the scans live in the reduction scratch");

    __device__ static __forceinline__ void pack_elem(const Elem &e, double (&a)[21])
    {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) a[3*i + j] = e.A[i][j];
#pragma unroll
        for (int k = 0; k < 6; k++) { a[9 + k] = e.C[k]; a[15 + k] = e.J[k]; }
    }
    __device__ static __forceinline__ void unpack_elem(const double (&a)[21], Elem &e)
    {
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) e.A[i][j] = a[3*i + j];
#pragma unroll
        for (int k = 0; k < 6; k++) { e.C[k] = a[9 + k]; e.J[k] = a[15 + k]; }
    }
    __device__ static __forceinline__ void pack_aff(const Aff &f, double (&a)[12])
    {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            a[9 + i] = f.v[i];
#pragma unroll
            for (int j = 0; j < 3; j++) a[3*i + j] = f.M[i][j];
        }
    }
    __device__ static __forceinline__ void unpack_aff(const double (&a)[12], Aff &f)
    {
#pragma unroll
        for (int i = 0; i < 3; i++) {
            f.v[i] = a[9 + i];
#pragma unroll
            for (int j = 0; j < 3; j++) f.M[i][j] = a[3*i + j];
        }
    }

    /* triple of one regular stage from its block; returns det R */
    __device__ static __forceinline__ double stage_elem(const double *s, const bool pn, Elem &e)
    {
        const double Tb = s[S_TB], Tw = s[S_TW], Bb = s[S_BB], Bw = s[S_BW];
        const double Htt = s[S_HTT], Hbb = s[S_HBB], Hbq = s[S_HBQ], Hbf = s[S_HBF], Hbp = s[S_HBP], Hqq = s[S_HQQ], Hqf = s[S_HQF],
                     Hff = s[S_HFF], Hfp = s[S_HFP], Hpp = s[S_HPP];
        double rff, rfp, rpp, det, sf, sp, sw;
        if (pn) {
            /* determinant and row sums of the inverse from the forces' own curvatures (S_OA, S_OB): no difference of large numbers */
            const double oa = s[S_OA], ob = s[S_OB];
            det = Hfp*(oa + ob) + oa*ob;
            const double id = frcp(det);
            rff = Hpp*id; rfp = -Hfp*id; rpp = Hff*id; sf = ob*id; sp = oa*id; sw = (oa + ob)*id;
        } else { det = Hff; rff = frcp(Hff); rfp = 0; rpp = 0; sf = rff; sp = 0; sw = rff; }
        e.C[sy(0, 0)] = Tw*Tw*sw; e.C[sy(0, 1)] = Tw*Bw*sw; e.C[sy(1, 1)] = Bw*Bw*sw;
        e.C[sy(0, 2)] = Tw*sf; e.C[sy(1, 2)] = Bw*sf; e.C[sy(2, 2)] = rff;
        const double ub = sf*Hbf + sp*Hbp, uq = sf*Hqf, vb = rff*Hbf + rfp*Hbp;
        e.A[0][0] = 1; e.A[0][1] = Tb - Tw*ub; e.A[0][2] = -Tw*uq;
        e.A[1][0] = 0; e.A[1][1] = Bb - Bw*ub; e.A[1][2] = -Bw*uq;
        e.A[2][0] = 0; e.A[2][1] = -vb;        e.A[2][2] = -rff*Hqf;
        e.J[sy(0, 0)] = Htt; e.J[sy(0, 1)] = 0; e.J[sy(0, 2)] = 0;
        e.J[sy(1, 1)] = Hbb - (Hbf*vb + Hbp*(rfp*Hbf + rpp*Hbp));
        e.J[sy(1, 2)] = Hbq - Hqf*vb;
        e.J[sy(2, 2)] = Hqq - Hqf*rff*Hqf;
        return det;
    }

    __device__ static __forceinline__ bool finite6(const double (&a)[6])
    {
        return isfinite(a[0] + a[1] + a[2] + a[3] + a[4] + a[5]);
    }

#if MSD_TELEMETRY      /* (tuning builds: the sub-phase marks below need the caller's cycle mark) */
    __device__ static MSD_PARALLEL_ATTR int solve(const int N, const bool pn, Ctx &c)
#else
    __device__ static MSD_PARALLEL_ATTR int solve(const int N, const bool pn, Ctx c)
#endif
    {
        double *S = c.S;
        bool ok = true, bad = false;

        /* ---- last interval on top of the terminal value function (only t_N is free): every thread, redundantly ---- */
        double Pn[6], pvn[3];          /* value function of stage N-1 */
        double LG[7];
        double lastK[8], lastKS[4];
        if (!last_interval<DYN>(S + (N - 1)*S_STRIDE, S[N*S_STRIDE + S_HTT], S[N*S_STRIDE + S_HT], pn, Pn, pvn, lastK, lastKS, LG)) ok = false;
        __syncthreads();       /* every thread has read block N-1 before thread 0 overwrites it */
        if (c.tid == 0) {
            double *s = S + (N - 1)*S_STRIDE;
            /* value function of stage N (for the multipliers of the last interval) */
            const double Ptt = S[N*S_STRIDE + S_HTT], pt = S[N*S_STRIDE + S_HT];
            s[S_PN + 0] = Ptt; s[S_PN + 1] = 0; s[S_PN + 2] = 0; s[S_PN + 3] = 0; s[S_PN + 4] = 0; s[S_PN + 5] = 0;
            s[S_PV + 0] = pt; s[S_PV + 1] = 0; s[S_PV + 2] = 0;
#pragma unroll
            for (int k = 0; k < 6; k++) s[S_K + k] = lastK[k];
            s[S_KV + 0] = lastK[6]; s[S_KV + 1] = lastK[7];
#pragma unroll
            for (int k = 0; k < 4; k++) s[S_KS + k] = lastKS[k];
#pragma unroll
            for (int k = 0; k < 7; k++) c.misc[MISC_LG + k] = LG[k];
        }

        /* ---- 1: triples of the own stages ---- */
        const int lo = c.tid*SPT;
        const int cnt = (N - 1 - lo < 0) ? 0 : (N - 1 - lo > SPT ? SPT : N - 1 - lo);
        Elem agg;
        elem_identity(agg);
        {
            bool have = false;
#pragma unroll
            for (int j = SPT - 1; j >= 0; j--) {
                if (j >= cnt) continue;
                Elem e;
                const double det = stage_elem(S + (lo + j)*S_STRIDE, pn, e);
                if (!(fabs(det) > 0) || !isfinite(det)) bad = true;
                if (have) { const double dm = combine(e, agg, agg); if (!(fabs(dm) > 0)) bad = true; }
                else agg = e;
                have = true;
            }
        }
        c.mark(PH_R_ELEM);
        /* ---- 2: suffix scan over the lanes of the wave, then over the waves ---- */
#pragma unroll 1
        for (int d = 1; d < 64; d <<= 1) {
            double mine[21], theirs[21];
            pack_elem(agg, mine);
            wave_fetch<21>(mine, theirs, c.lane + d);
            if (c.lane + d < 64 && (c.tid + d)*SPT < N - 1 && cnt > 0) {
                Elem o;
                unpack_elem(theirs, o);
                const double dm = combine(agg, o, agg);
                if (!(fabs(dm) > 0)) bad = true;
            }
        }
        double Pb[6];                  /* value function at the first stage behind this wave's range */
#pragma unroll
        for (int k = 0; k < 6; k++) Pb[k] = Pn[k];
        if (c.nw > 1) {
            if (c.lane == 0) {
                double a[21];
                pack_elem(agg, a);
#pragma unroll
                for (int k = 0; k < 21; k++) c.red[RED_MAT + 21*c.wave + k] = a[k];
            }
            __syncthreads();
            for (int w = c.nw - 1; w > c.wave; w--) {
                if (w*64*SPT >= N - 1) continue;      /* that wave has no stages */
                double a[21];
#pragma unroll
                for (int k = 0; k < 21; k++) a[k] = c.red[RED_MAT + 21*w + k];
                Elem t;
                unpack_elem(a, t);
                const double dm = combine_value(t, Pb, Pb);
                if (!(fabs(dm) > 0)) bad = true;
            }
        }
        double Ps[6];                  /* value function at the chunk start (by the scan), Pe at the chunk end */
        if (cnt > 0) { const double dm = combine_value(agg, Pb, Ps); if (!(fabs(dm) > 0)) bad = true; }
        else {
#pragma unroll
            for (int k = 0; k < 6; k++) Ps[k] = Pb[k];
        }
        double Pe[6];
        wave_fetch<6>(Ps, Pe, c.lane + 1);
        if (c.lane == 63) {
#pragma unroll
            for (int k = 0; k < 6; k++) Pe[k] = Pb[k];
        }
        if (cnt > 0 && !finite6(Pe)) bad = true;

        c.mark(PH_R_SCAN_P);
        /* ---- 3: ordinary recursion over the own stages (matrix part), chunk map of the value-function gradient ---- */
        double gam[SPT][3], gf0[SPT], gp0[SPT], lfpv[SPT], iffv[SPT], ipv[SPT], Gfpv[SPT];
        Aff bmap;
        aff_identity(bmap);
        {
            double Ptt = Pe[0], Ptb = Pe[1], Ptq = Pe[2], Pbb = Pe[3], Pbq = Pe[4], Pqq = Pe[5];
#pragma unroll
            for (int j = SPT - 1; j >= 0; j--) {
                gam[j][0] = gam[j][1] = gam[j][2] = 0; gf0[j] = gp0[j] = lfpv[j] = iffv[j] = ipv[j] = Gfpv[j] = 0;
                if (j >= cnt) continue;
                double *s = S + (lo + j)*S_STRIDE;
                const double Tb = s[S_TB], Tw = s[S_TW], Bb = s[S_BB], Bw = s[S_BW], rt = s[S_RT], rb = s[S_RB];
                const double Htt = s[S_HTT], Hbb = s[S_HBB], Hbq = s[S_HBQ], Hbf = s[S_HBF], Hbp = s[S_HBP], Hqq = s[S_HQQ], Hqf = s[S_HQF],
                             Hff = s[S_HFF], Hfp = s[S_HFP], Hpp = s[S_HPP];
                const double ht = s[S_HT], hb = s[S_HB], hq = s[S_HQ], hf = s[S_HF], hp = s[S_HP];
                const double oa = s[S_OA], ob = s[S_OB];
                s[S_PN + 0] = Ptt; s[S_PN + 1] = Ptb; s[S_PN + 2] = Ptq; s[S_PN + 3] = Pbb; s[S_PN + 4] = Pbq; s[S_PN + 5] = Pqq;
                const double Mbt = Tb*Ptt + Bb*Ptb, Mbb = Tb*Ptb + Bb*Pbb;
                const double Mpt = Tw*Ptt + Bw*Ptb, Mpb = Tw*Ptb + Bw*Pbb, Mpq = Tw*Ptq + Bw*Pbq;
                const double Mft = Mpt + Ptq, Mfb = Mpb + Pbq, Mfq = Mpq + Pqq;
                double Gtt = Htt + Ptt, Gtb = Mbt, Gtf = Mft, Gtp = Mpt;
                double Gbb = Hbb + Tb*Mbt + Bb*Mbb, Gbq = Hbq, Gbf = Hbf + Tb*Mft + Bb*Mfb, Gbp = Hbp + Tb*Mpt + Bb*Mpb;
                double Gqq = Hqq, Gqf = Hqf;
                double Gff = Hff + Tw*Mft + Bw*Mfb + Mfq, Gfp = Hfp + Tw*Mpt + Bw*Mpb + Mpq;
                double Gpp = Hpp + Tw*Mpt + Bw*Mpb;
                if (!pn) { Gtp = 0; Gbp = 0; Gfp = 0; Gpp = 1; }
                const double ip = frcp(Gpp), lfp = Gfp*ip;
                const double df_ = pn ? (oa + Mpq + Pqq) + lfp*(ob - Mpq) : Gff;      /* = Gff - Gfp^2/Gpp, from the own curvatures (S_OA) */
                if (!(Gpp > 0) || !(df_ > 0)) ok = false;
                const double iff = frcp(df_);
                double Kft = -(Gtf - lfp*Gtp)*iff, Kpt = -(Gtp + Gfp*Kft)*ip;
                double Kfb = -(Gbf - lfp*Gbp)*iff, Kpb = -(Gbp + Gfp*Kfb)*ip;
                double Kfq = -(Gqf)*iff,           Kpq = -(Gfp*Kfq)*ip;
                if (!pn) { Kpt = 0; Kpb = 0; Kpq = 0; }
                const double nPtt = Gtt + Gtf*Kft + Gtp*Kpt;
                const double nPtb = Gtb + Gtf*Kfb + Gtp*Kpb;
                const double nPtq = Gtf*Kfq + Gtp*Kpq;
                const double nPbb = Gbb + Gbf*Kfb + Gbp*Kpb;
                const double nPbq = Gbq + Gbf*Kfq + Gbp*Kpq;
                const double nPqq = Gqq + Gqf*Kfq;
                /* affine part with p+ = 0: w = P+ r */
                const double wt = Ptt*rt + Ptb*rb, wb = Ptb*rt + Pbb*rb, wq = Ptq*rt + Pbq*rb;
                const double gt = ht + wt, gb = hb + Tb*wt + Bb*wb, gq = hq;
                const double gfz = hf + Tw*wt + Bw*wb + wq, gpz = pn ? hp + Tw*wt + Bw*wb : 0.0;
                const double kfz = -(gfz - lfp*gpz)*iff, kpz = pn ? -(gpz + Gfp*kfz)*ip : 0.0;
                gam[j][0] = gt + Gtf*kfz + Gtp*kpz; gam[j][1] = gb + Gbf*kfz + Gbp*kpz; gam[j][2] = gq + Gqf*kfz;
                gf0[j] = gfz; gp0[j] = gpz; lfpv[j] = lfp; iffv[j] = iff; ipv[j] = ip; Gfpv[j] = Gfp;
                /* p_j = Phi^T p+ + gamma, Phi = Fx + Fu K (closed loop) */
                const double kt = Kft + Kpt, kb = Kfb + Kpb, kq = Kfq + Kpq;
                Aff st;
                st.M[0][0] = 1 + Tw*kt;  st.M[0][1] = Bw*kt;      st.M[0][2] = Kft;
                st.M[1][0] = Tb + Tw*kb; st.M[1][1] = Bb + Bw*kb; st.M[1][2] = Kfb;
                st.M[2][0] = Tw*kq;      st.M[2][1] = Bw*kq;      st.M[2][2] = Kfq;
                st.v[0] = gam[j][0]; st.v[1] = gam[j][1]; st.v[2] = gam[j][2];
                aff_compose(st, bmap, bmap);
                s[S_K + 0] = Kft; s[S_K + 1] = Kfb; s[S_K + 2] = Kfq; s[S_K + 3] = Kpt; s[S_K + 4] = Kpb; s[S_K + 5] = Kpq;
                Ptt = nPtt; Ptb = nPtb; Ptq = nPtq; Pbb = nPbb; Pbq = nPbq; Pqq = nPqq;
            }
        }
        /* ---- inertia and breakdown flags (uniform from here) ---- */
        {
            double v[2] = {ok ? 0.0 : 1.0, bad ? 1.0 : 0.0};
            if (c.nw > 1) __syncthreads();      /* the wave totals of the scan share the reduction scratch */
            block_reduce<2>(v, OpMax(), c);
            if (c.nw > 1) __syncthreads();
            if (uni(v[1]) != 0.0) return -1;
            if (uni(v[0]) != 0.0) return 0;
        }

        c.mark(PH_R_RECUR);
        /* ---- 4: suffix scan of the gradient maps ---- */
#pragma unroll 1
        for (int d = 1; d < 64; d <<= 1) {
            double mine[12], theirs[12];
            pack_aff(bmap, mine);
            wave_fetch<12>(mine, theirs, c.lane + d);
            if (c.lane + d < 64 && (c.tid + d)*SPT < N - 1 && cnt > 0) {
                Aff o;
                unpack_aff(theirs, o);
                aff_compose(bmap, o, bmap);
            }
        }
        double pb[3] = {pvn[0], pvn[1], pvn[2]};      /* gradient at the first stage behind this wave's range */
        if (c.nw > 1) {
            if (c.lane == 0) {
                double a[12];
                pack_aff(bmap, a);
#pragma unroll
                for (int k = 0; k < 12; k++) c.red[RED_AFB + 12*c.wave + k] = a[k];
            }
            __syncthreads();
            for (int w = c.nw - 1; w > c.wave; w--) {
                if (w*64*SPT >= N - 1) continue;
                double a[12];
#pragma unroll
                for (int k = 0; k < 12; k++) a[k] = c.red[RED_AFB + 12*w + k];
                Aff t;
                unpack_aff(a, t);
                aff_apply(t, pb, pb);
            }
        }
        double ps[3], pe[3];
        if (cnt > 0) aff_apply(bmap, pb, ps);
        else { ps[0] = pb[0]; ps[1] = pb[1]; ps[2] = pb[2]; }
        wave_fetch<3>(ps, pe, c.lane + 1);
        if (c.lane == 63) { pe[0] = pb[0]; pe[1] = pb[1]; pe[2] = pb[2]; }

        c.mark(PH_R_SCAN_G);
        /* ---- 5: feed-forward of the own stages, closed-loop chunk map ---- */
        Aff fmap;
        aff_identity(fmap);
        {
            double pt = pe[0], pbv = pe[1], pq = pe[2];
#pragma unroll
            for (int j = SPT - 1; j >= 0; j--) {
                if (j >= cnt) continue;
                double *s = S + (lo + j)*S_STRIDE;
                const double Tb = s[S_TB], Tw = s[S_TW], Bb = s[S_BB], Bw = s[S_BW], rt = s[S_RT], rb = s[S_RB];
                const double Kft = s[S_K + 0], Kfb = s[S_K + 1], Kfq = s[S_K + 2], Kpt = s[S_K + 3], Kpb = s[S_K + 4], Kpq = s[S_K + 5];
                s[S_PV + 0] = pt; s[S_PV + 1] = pbv; s[S_PV + 2] = pq;
                const double a = Tw*pt + Bw*pbv;
                const double gf = gf0[j] + a + pq, gp = pn ? gp0[j] + a : 0.0;
                const double kf = -(gf - lfpv[j]*gp)*iffv[j], kp = pn ? -(gp + Gfpv[j]*kf)*ipv[j] : 0.0;
                s[S_KV + 0] = kf; s[S_KV + 1] = kp;
                const double kt = Kft + Kpt, kb = Kfb + Kpb, kq = Kfq + Kpq, k0 = kf + kp;
                /* gradient of stage j: Phi^T p+ + gamma */
                const double npt = (1 + Tw*kt)*pt + Bw*kt*pbv + Kft*pq + gam[j][0];
                const double npb = (Tb + Tw*kb)*pt + (Bb + Bw*kb)*pbv + Kfb*pq + gam[j][1];
                const double npq = Tw*kq*pt + Bw*kq*pbv + Kfq*pq + gam[j][2];
                pt = npt; pbv = npb; pq = npq;
                /* x+ = Phi x + (Fu k + r); fmap = (later stages) o (stage j) */
                Aff st;
                st.M[0][0] = 1 + Tw*kt; st.M[0][1] = Tb + Tw*kb; st.M[0][2] = Tw*kq;
                st.M[1][0] = Bw*kt;     st.M[1][1] = Bb + Bw*kb; st.M[1][2] = Bw*kq;
                st.M[2][0] = Kft;       st.M[2][1] = Kfb;        st.M[2][2] = Kfq;
                st.v[0] = Tw*k0 + rt; st.v[1] = Bw*k0 + rb; st.v[2] = kf;
                aff_compose(fmap, st, fmap);
            }
        }
        c.mark(PH_R_FEED);
        /* ---- 6: prefix scan of the closed-loop maps ---- */
#pragma unroll 1
        for (int d = 1; d < 64; d <<= 1) {
            double mine[12], theirs[12];
            pack_aff(fmap, mine);
            wave_fetch<12>(mine, theirs, c.lane - d);
            if (c.lane - d >= 0 && cnt > 0) {
                Aff o;
                unpack_aff(theirs, o);
                aff_compose(fmap, o, fmap);
            }
        }
        double xb[3] = {0, 0, 0};      /* state at the first stage of this wave's range; x_0 is a parameter of the NLP */
        if (c.nw > 1) {
            /* the wave total sits in the wave's last lane with stages; lanes behind it hold their own (identity) map */
            const int lastLane = ((N - 2)/SPT) - 64*c.wave;      /* lane of the thread that owns stage N-2, relative to this wave */
            const int src = lastLane < 0 ? 0 : (lastLane > 63 ? 63 : lastLane);
            double mine[12], tot[12];
            pack_aff(fmap, mine);
            wave_fetch<12>(mine, tot, src);
            if (c.lane == 0) {
#pragma unroll
                for (int k = 0; k < 12; k++) c.red[RED_AFF + 12*c.wave + k] = tot[k];
            }
            __syncthreads();
            for (int w = 0; w < c.wave; w++) {
                double a[12];
#pragma unroll
                for (int k = 0; k < 12; k++) a[k] = c.red[RED_AFF + 12*w + k];
                Aff t;
                unpack_aff(a, t);
                aff_apply(t, xb, xb);
            }
        }
        double xe[3], xs[3];
        aff_apply(fmap, xb, xe);      /* state behind the own chunk */
        wave_fetch<3>(xe, xs, c.lane - 1);
        if (c.lane == 0) { xs[0] = xb[0]; xs[1] = xb[1]; xs[2] = xb[2]; }

        c.mark(PH_R_SCAN_X);
        /* ---- 7: roll-out of the own stages; the owner of stage N-2 continues through the last interval ---- */
        {
            double dt = xs[0], db = xs[1], dq = xs[2];
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                if (j >= cnt) continue;
                double *s = S + (lo + j)*S_STRIDE;
                const double Tb = s[S_TB], Tw = s[S_TW], Bb = s[S_BB], Bw = s[S_BW], rt = s[S_RT], rb = s[S_RB];
                const double df = s[S_K + 0]*dt + s[S_K + 1]*db + s[S_K + 2]*dq + s[S_KV + 0];
                const double dp = pn ? s[S_K + 3]*dt + s[S_K + 4]*db + s[S_K + 5]*dq + s[S_KV + 1] : 0.0;
                const double dw = df + dp;
                const double nt = dt + Tb*db + Tw*dw + rt, nb = Bb*db + Bw*dw + rb;
                s[S_DT] = dt; s[S_DB] = db; s[S_DF] = df; s[S_DP] = dp;
                dt = nt; db = nb; dq = df;
            }
            const bool ownsLast = (N == 1) ? (c.tid == 0) : (cnt > 0 && lo + cnt == N - 1);
            if (ownsLast) {
                double *s = S + (N - 1)*S_STRIDE;
                const double Tb = s[S_TB], Tw = s[S_TW], Bw = s[S_BW], rt = s[S_RT];
                const double df = lastK[0]*dt + lastK[1]*db + lastK[2]*dq + lastK[6];
                const double dp = pn ? lastK[3]*dt + lastK[4]*db + lastK[5]*dq + lastK[7] : 0.0;
                const double dw = df + dp;
                const double nt = dt + Tb*db + Tw*dw + rt;
                const double dsl = lastKS[0]*dt + lastKS[1]*db + lastKS[2]*dq + lastKS[3];
                s[S_DS] = dsl;
                s[S_LB] = (LG[0]*dt + LG[1]*db + LG[2]*dq + LG[3]*df + LG[4]*dp + LG[5]*dsl + LG[6])/Bw;
                s[S_DT] = dt; s[S_DB] = db; s[S_DF] = df; s[S_DP] = dp;
                double *sN = S + N*S_STRIDE;
                sN[S_DT] = nt; sN[S_DB] = 0.0; sN[S_DF] = 0.0;
            }
        }
        __syncthreads();
        return 1;
    }
};

#ifndef MSD_PARALLEL_RESTO
#define MSD_PARALLEL_RESTO 1      /* 0: the restoration problem's Newton system on riccati_resto's serial sweeps alone (A/B builds) */
#endif
#include "msd_resto_scan.hpp"

/* ------------------------------------------------------------------------------------------
 * the solver.  SPT = shooting nodes per thread: node j of thread `tid` is node tid + j*NT.
 * The benchmark geometry is one wave per scenario (NT = 64) with SPT = 2: four single-wave
 * workgroups per CU, one per SIMD, each with the whole 512-entry register file of its SIMD,
 * so the complete iterate of two nodes stays in registers and barriers are wave-local.
 * ---------------------------------------------------------------------------------------- */
enum { MODE_NEWTON = 0, MODE_LSQ = 1, MODE_RESTO = 2 };      /* (MODE_RESTO: Newton system of the restoration problem, msd_resto.hpp) */

/* PART: which part of a solve the enclosing kernel holds (solve_kernel) -- 0 everything, 1 the first pass (no restoration phase), 2 the follow-up,
 * 3 the first pass with the least-squares multiplier estimate in front (any starting point) */
template <int NT, int SPT, int DYN, bool STREAM, bool GEN, int FULL, int PART = 0, bool SLDS = false, bool SOCK = false>
struct Solver {
    static constexpr int S_STRIDE = stage_stride(DYN);
    static constexpr int NS = NT*SPT;      /* node slots of the workgroup */
    using NodeT = Node<NS, STREAM>;
    const DevProb &P;
    Ctx &c;
    double *work;                          /* the workgroup's private work area (work_doubles(NS)) */
    NodeT n[SPT];
    Uni &U;                                /* workgroup-uniform data of the scenario, in LDS (like P): loaded where needed instead of held in registers */
    /* right-hand sides of the linearised constraints: c and d - sigma (or their SOC accumulation) */
    Field<2, NS, STREAM> resc[SPT];
    Field<NR, NS, STREAM> resd[SPT];
    /* evaluation of the current point with derivatives (evaluate_current), read back by the phases that need it */
    static constexpr bool INTEG = loss_integrated(DYN), ITAB = DYN == LOSS_INTEGRATED_TABLE;
    static constexpr int LGW = ITAB ? 15 : 5;      /* derivatives kept per loss row (Ev::lg) */
    static_assert(2*LGW <= W_LG_FIELDS, "room of the loss-row derivatives in the work area");
    Field<13, NS, STREAM> evs[SPT];
    Field<2*LGW, NS, STREAM> lgs[SPT];

    __device__ __forceinline__ Solver(const DevProb &P_, Ctx &c_, double *work_, Uni &U_) : P(P_), c(c_), work(work_), U(U_) {}

    /* structure of the NLP.  FULL: traction + pneumatic brake, power rows, energy objective -- the rolling stock of the reference's JSON
     * files (BASELINE configs 1-4) -- known at compile time: no flag loads, no branches on them.  Rows then are: both power rows and the
     * acceleration row two-sided, the two loss rows bounded below (ocp.py:184-229) */
    __device__ __forceinline__ bool rowOn(int r) const { return full_energy(FULL) ? true : full_time(FULL) ? (r <= RACC) : U.rowOn[r]; }      /* (time-optimal: no loss rows) */
    __device__ __forceinline__ bool rL(int r) const { return full_energy(FULL) ? true : full_time(FULL) ? (r <= RACC) : U.rL[r]; }
    __device__ __forceinline__ bool rU(int r) const { return FULL ? (r <= RACC) : U.rU[r]; }
    __device__ __forceinline__ bool withPn() const { return FULL ? full_pn(FULL) : P.withPn != 0; }
    __device__ __forceinline__ bool energyOpt() const { return FULL ? full_energy(FULL) : P.energyOpt != 0; }
    __device__ __forceinline__ bool hasPower() const { return FULL ? true : P.hasPower != 0; }

    __device__ __forceinline__ void store_ev(int j, const Ev &e)
    {
        evs[j][0] = e.sb; evs[j][1] = e.sb1; evs[j][2] = e.b1; evs[j][3] = e.tb; evs[j][4] = e.tw; evs[j][5] = e.tbb; evs[j][6] = e.tbw; evs[j][7] = e.tww;
        evs[j][8] = e.Bb; evs[j][9] = e.Bw; evs[j][10] = e.Bbb; evs[j][11] = e.Bbw; evs[j][12] = e.Bww;
        if (DYN) {
#pragma unroll
            for (int k = 0; k < 2; k++)
#pragma unroll
                for (int m = 0; m < LGW; m++) lgs[j][LGW*k + m] = e.lg[k][m];
        }
    }
    __device__ __forceinline__ void load_ev(int j, Ev &e) const
    {
        e.sb = evs[j][0]; e.sb1 = evs[j][1]; e.b1 = evs[j][2]; e.tb = evs[j][3]; e.tw = evs[j][4]; e.tbb = evs[j][5]; e.tbw = evs[j][6]; e.tww = evs[j][7];
        e.Bb = evs[j][8]; e.Bw = evs[j][9]; e.Bbb = evs[j][10]; e.Bbw = evs[j][11]; e.Bww = evs[j][12];
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int m = 0; m < LGW; m++) e.lg[k][m] = DYN ? lgs[j][LGW*k + m] : 0.0;
    }
    /* first-order part only (sb, sb1, b1, tb, tw, Bb, Bw and the loss-row gradients): what the residual pass and the read-back need */
    __device__ __forceinline__ void load_ev1(int j, Ev &e) const
    {
        e.sb = evs[j][0]; e.sb1 = evs[j][1]; e.b1 = evs[j][2]; e.tb = evs[j][3]; e.tw = evs[j][4]; e.Bb = evs[j][8]; e.Bw = evs[j][9];
        e.tbb = e.tbw = e.tww = e.Bbb = e.Bbw = e.Bww = 0;
#pragma unroll
        for (int k = 0; k < 2; k++)
#pragma unroll
            for (int m = 0; m < LGW; m++) e.lg[k][m] = DYN ? lgs[j][LGW*k + m] : 0.0;
    }
    /* explicit home of the register-resident iterate in the work area (same layout as the memory-backed fields): stash() writes the
     * fields of node j selected by the mask, fetch() reads them back.  Between a stash and the next fetch the values are dead in
     * registers, which is the point: the phase in between gets the register file */
    static constexpr unsigned H_X = 1u, H_SG = 2u, H_LAM = 4u, H_NU = 8u, H_Z = 16u, H_ZS = 32u, H_DSG = 64u, H_RES = 128u, H_EV = 256u, H_ALL = 511u;
    template <int CNT, class F> __device__ __forceinline__ void put(const F &f, int off, int slot)
    {
        if constexpr (!F::in_memory) {
#pragma unroll
            for (int k = 0; k < CNT; k++) work[(off + k)*NS + slot] = f.v[k];
        }
    }
    template <int CNT, class F> __device__ __forceinline__ void get(F &f, int off, int slot)
    {
        if constexpr (!F::in_memory) {
#pragma unroll
            for (int k = 0; k < CNT; k++) f.v[k] = work[(off + k)*NS + slot];
        }
    }
    template <unsigned M, int BASE = 0> __device__ __forceinline__ void stash()      /* (BASE: another set of fields with the same layout -- the watchdog's copy) */
    {
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            NodeT &nd = n[j];
            const int sl = nd.i + BASE*NS;
            if (M & H_X) put<NV>(nd.x, W_X, sl);
            if (M & H_SG) put<NR>(nd.sg, W_SG, sl);
            if (M & H_LAM) put<2>(nd.lam, W_LAM, sl);
            if (M & H_NU) put<NR>(nd.nu, W_NU, sl);
            if (M & H_Z) { put<NV>(nd.zL, W_ZL, sl); put<NV>(nd.zU, W_ZU, sl); }
            if (M & H_ZS) { put<NR>(nd.zLs, W_ZLS, sl); put<NR>(nd.zUs, W_ZUS, sl); }
            if (M & H_DSG) put<NR>(nd.dsg, W_DSG, sl);
            if (M & H_RES) { put<2>(resc[j], W_RESC, sl); put<NR>(resd[j], W_RESD, sl); }
            if (M & H_EV) { put<13>(evs[j], W_EV, sl); if (DYN) put<2*LGW>(lgs[j], W_LG, sl); }
        }
        asm volatile("" ::: "memory");
    }
    template <unsigned M> __device__ __forceinline__ void fetch()
    {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            NodeT &nd = n[j];
            const int sl = nd.i;
            if (M & H_X) get<NV>(nd.x, W_X, sl);
            if (M & H_SG) get<NR>(nd.sg, W_SG, sl);
            if (M & H_LAM) get<2>(nd.lam, W_LAM, sl);
            if (M & H_NU) get<NR>(nd.nu, W_NU, sl);
            if (M & H_Z) { get<NV>(nd.zL, W_ZL, sl); get<NV>(nd.zU, W_ZU, sl); }
            if (M & H_ZS) { get<NR>(nd.zLs, W_ZLS, sl); get<NR>(nd.zUs, W_ZUS, sl); }
            if (M & H_DSG) get<NR>(nd.dsg, W_DSG, sl);
            if (M & H_RES) { get<2>(resc[j], W_RESC, sl); get<NR>(resd[j], W_RESD, sl); }
            if (M & H_EV) { get<13>(evs[j], W_EV, sl); if (DYN) get<2*LGW>(lgs[j], W_LG, sl); }
        }
    }

    template <class F> __device__ static __forceinline__ void fence_v(F &f, int k) { if constexpr (!F::in_memory) opaque(f.v[k]); }
    template <class F> __device__ static __forceinline__ void fence_z(F &f, int k) { if constexpr (!F::in_memory) opaque_z(f.v[k]); }
    template <class F> __device__ static __forceinline__ void fence_d(F &f, int k) { if constexpr (!F::in_memory) opaque_d(f.v[k]); }
    template <class F> __device__ static __forceinline__ void fence_a(F &f, int k) { if constexpr (!F::in_memory) opaque_a(f.v[k]); }

    /* every free variable has a lower bound; all but the loss slack have an upper bound (ocp.py:175-181, 263-272) */
    __device__ __forceinline__ double lbv(int k) const { return k == VT ? U.tlo : k == VB ? U.blo : k == VF ? U.flo : k == VP ? U.plo : U.slo; }
    __device__ __forceinline__ double ubv(int j, int k) const { return k == VT ? U.thi : k == VB ? n[j].ubB : k == VF ? U.fhi : k == VP ? U.phi : INFINITY; }
    __device__ static __forceinline__ bool hasU(int k) { return k != VS; }

    /* phase boundary: keeps the optimiser from carrying subexpressions of the state (slacks, reciprocals, Sigma ...) from one
     * phase to the next in registers -- recomputing them is cheaper than the spills they cause */
    __device__ __forceinline__ void phase_fence(int phase)
    {
#if MSD_PHASE_FENCE
        /* not in a first-pass kernel that holds the fused iteration alone (round 4): there the fences only cost -- with the general iteration and
         * the restoration phase out of the code object the allocator keeps the iterate where it is, and the copies into architectural
         * registers that an asm operand asks for (some 200 accumulation-register moves and 18 scratch round trips per fence) were a fifth of
         * the run time: 858 k -> 1.03 M solves/s on config 1, 1.05 -> 1.26 M at 8192 per launch (gpurun_out/abhot1, profiles/r04) */
        if (!MSD_HOT_FENCES && (PART == 1 || PART == 3) && FAST) return;
        if (!((MSD_FENCE_PHASES >> phase) & 1)) return;
        /* Up to four waves per workgroup (+30 ... 38 % at 192 x 2 and 256 x 2; the five-wave 320 x 2 kernel with its 256-register budget loses
         * 11 % to them).  Round 2 had switched them off above two waves after wrong results at 192 x 2 under the iterative-ilp scheduler it
         * used then; under the plain -O3 scheduling in use since, every geometry agrees with the oracle with the fences on (tools/
         * geometry_sweep.py: 64 scenarios per horizon, horizons that leave idle node slots in the last wave; profiles/r03) */
        if (NT > MSD_FENCE_MAX_NT || STREAM) return;
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            NodeT &nd = n[j];
#pragma unroll
            for (int k = 0; k < NV; k++) { fence_v(nd.x, k); if (MSD_FENCE_DUALS) { fence_z(nd.zL, k); fence_z(nd.zU, k); } }
#pragma unroll
            for (int r = 0; r < NR; r++) { fence_v(nd.sg, r); if (MSD_FENCE_DUALS) { fence_d(nd.nu, r); fence_z(nd.zLs, r); fence_z(nd.zUs, r); fence_d(nd.dsg, r); fence_d(resd[j], r); } }
            if (MSD_FENCE_DUALS) { fence_d(nd.lam, 0); fence_d(nd.lam, 1); fence_d(resc[j], 0); fence_d(resc[j], 1); }
        }
#endif
    }

    __device__ __forceinline__ void commit_uniforms(const Uni &u)
    {
        __syncthreads();
        if (c.tid == 0) U = u;
        __syncthreads();
    }

    /* publish (t, b, f) of a point so that neighbours can read them */
    __device__ __forceinline__ void publish(const double (&x)[SPT][NV])
    {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SPT; j++) { const int i = n[j].i; c.xt[i] = x[j][VT]; c.xb[i] = x[j][VB]; c.xf[i] = x[j][VF]; }
        __syncthreads();
    }
    __device__ __forceinline__ void publish_current()
    {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SPT; j++) { const int i = n[j].i; c.xt[i] = n[j].x[VT]; c.xb[i] = n[j].x[VB]; c.xf[i] = n[j].x[VF]; }
        __syncthreads();
    }
    __device__ __forceinline__ double nb_q(int j) const { return (n[j].i > 0 && n[j].node()) ? c.xf[n[j].i - 1] : 0.0; }

    __device__ __forceinline__ void var_terms(int j, int k, double mu_, double &Sg, double &gphi) const
    {
        bar_terms(n[j].x[k], lbv(k), ubv(j, k), true, hasU(k), n[j].zL[k], n[j].zU[k], mu_, Sg, gphi);
    }
    __device__ __forceinline__ void row_terms(int j, int r, double mu_, double &Sg, double &gphi) const
    {
        bar_terms(n[j].sg[r], U.dL[r], U.dU[r], rL(r), rU(r), n[j].zLs[r], n[j].zUs[r], mu_, Sg, gphi);
    }

    /* second derivatives of phi = f X(v(b), d, f + p) wrt (b, f, p, d) from the cached derivatives of X (integrated losses) */
    struct LossHess { double bb, bf, bp, bd, ff, fp, fd, pp, pd, dd; };
    __device__ __forceinline__ LossHess loss_hess(double f, double b, const Ev &ev) const
    {
        const double Xv = ev.lg[0][1], Xd = ev.lg[0][2], Xw = ev.lg[0][3], Xvv = ev.lg[0][4], Xvd = ev.lg[1][0], Xvw = ev.lg[1][1], Xdd = ev.lg[1][2],
                     Xdw = ev.lg[1][3], Xww = ev.lg[1][4];
        const double vb = 0.5/ev.sb, vbb = -0.25/(b*ev.sb);
        LossHess L;
        L.bb = f*(Xvv*vb*vb + Xv*vbb); L.bf = f*Xvw*vb + Xv*vb; L.bp = f*Xvw*vb; L.bd = f*Xvd*vb;
        L.ff = f*Xww + 2*Xw; L.fp = f*Xww + Xw; L.fd = f*Xdw + Xd; L.pp = f*Xww; L.pd = f*Xdw; L.dd = f*Xdd;
        return L;
    }
    /* sum over the two loss rows of nu_r hess(row_r) wrt (b, f, p, d) -- what the Hessian of the Lagrangian takes from the integrated loss rows.
     * Constant efficiencies: rows s + kappa_r f X, one function for both (loss_hess).  Loss table (ITAB): rows s - E_r(v(b), d, f + p, f), the cached jets of
     * E_tr and E_rgb */
    __device__ __forceinline__ LossHess rows_hess(int j, double f, double b, const Ev &ev) const
    {
        LossHess L;
        if constexpr (ITAB) {
            const double vb = 0.5/ev.sb, vbb = -0.25/(b*ev.sb);
            L.bb = L.bf = L.bp = L.bd = L.ff = L.fp = L.fd = L.pp = L.pd = L.dd = 0;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int r = (k == 0) ? RLTR : RLRG;
                const double wk = -n[j].nu[r]*U.rs[r];
                const double *E = ev.lg[k];      /* value, g[4] at 1, h[10] at 5 */
                const double g0 = E[1], h00 = E[5 + j4h(0, 0)], h01 = E[5 + j4h(0, 1)], h02 = E[5 + j4h(0, 2)], h03 = E[5 + j4h(0, 3)], h11 = E[5 + j4h(1, 1)],
                             h12 = E[5 + j4h(1, 2)], h13 = E[5 + j4h(1, 3)], h22 = E[5 + j4h(2, 2)], h23 = E[5 + j4h(2, 3)], h33 = E[5 + j4h(3, 3)];
                L.bb += wk*(h00*vb*vb + g0*vbb); L.bf += wk*(h02 + h03)*vb; L.bp += wk*h02*vb; L.bd += wk*h01*vb;
                L.ff += wk*(h22 + 2*h23 + h33); L.fp += wk*(h22 + h23); L.fd += wk*(h12 + h13); L.pp += wk*h22; L.pd += wk*h12; L.dd += wk*h11;
            }
        } else {
            L = loss_hess(f, b, ev);
            const double W = n[j].nu[RLTR]*U.rs[RLTR]*(-P.ct) + n[j].nu[RLRG]*U.rs[RLRG]*P.cr;
            L.bb *= W; L.bf *= W; L.bp *= W; L.bd *= W; L.ff *= W; L.fp *= W; L.fd *= W; L.pp *= W; L.pd *= W; L.dd *= W;
        }
        return L;
    }

    /* gradient entries of the rows wrt (b, f, p, s, b1) and wrt the running time t1 - t (gd; integrated losses only) */
    __device__ __forceinline__ void row_grads(int j, const Ev &ev, double (&gb)[NR], double (&gf)[NR], double (&gp)[NR], double (&gs)[NR], double (&gb1)[NR],
                                              double (&gd)[NR]) const
    {
        const double f = n[j].x[VF];
#pragma unroll
        for (int r = 0; r < NR; r++) { gb[r] = gf[r] = gp[r] = gs[r] = gb1[r] = gd[r] = 0; }
        gf[RPW0] = ev.sb; gb[RPW0] = 0.5*f/ev.sb;
        gf[RPW1] = ev.sb1; gb1[RPW1] = 0.5*f/ev.sb1;
        gf[RACC] = 1; gp[RACC] = withPn() ? 1.0 : 0.0; gb[RACC] = -(0.5*P.sr1/ev.sb + P.sr2);
        if (DYN == LOSS_INTEGRATED_TABLE) {
            /* rows s - E_k(v(b), t1 - t, f + p, f): derivatives wrt (v, d, w, f) at lg[k][1..4] */
            const double vb = 0.5/ev.sb;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int r = (k == 0) ? RLTR : RLRG;
                gs[r] = 1; gf[r] = -(ev.lg[k][3] + ev.lg[k][4]); gp[r] = withPn() ? -ev.lg[k][3] : 0.0; gb[r] = -ev.lg[k][1]*vb; gd[r] = -ev.lg[k][2];
            }
        } else if (DYN == LOSS_INTEGRATED) {
            /* rows s + kappa f X(v(b), t1 - t, f + p), kappa = -ct, +cr */
            const double X = ev.lg[0][0], Xv = ev.lg[0][1], Xd = ev.lg[0][2], Xw = ev.lg[0][3], vb = 0.5/ev.sb;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                const int r = (k == 0) ? RLTR : RLRG;
                const double kap = (k == 0) ? -P.ct : P.cr;
                gs[r] = 1; gf[r] = kap*(X + f*Xw); gp[r] = withPn() ? kap*f*Xw : 0.0; gb[r] = kap*f*Xv*vb; gd[r] = kap*f*Xd;
            }
        } else if (DYN == LOSS_TABLE) {
            /* rows s - g(f, vbar(b, b1)), vbar = (sqrt(b) + sqrt(b1))/2 */
            const double vb = 0.25/ev.sb, vb1 = 0.25/ev.sb1;
            gs[RLTR] = 1; gf[RLTR] = -ev.lg[0][0]; gb[RLTR] = -ev.lg[0][1]*vb; gb1[RLTR] = -ev.lg[0][1]*vb1;
            gs[RLRG] = 1; gf[RLRG] = -ev.lg[1][0]; gb[RLRG] = -ev.lg[1][1]*vb; gb1[RLRG] = -ev.lg[1][1]*vb1;
        } else {
            gs[RLTR] = 1; gf[RLTR] = -P.ct;
            gs[RLRG] = 1; gf[RLRG] = P.cr;
        }
#pragma unroll
        for (int r = 0; r < NR; r++) { gb[r] *= U.rs[r]; gf[r] *= U.rs[r]; gp[r] *= U.rs[r]; gs[r] *= U.rs[r]; gb1[r] *= U.rs[r]; gd[r] *= U.rs[r]; }
    }
    /*
     * In the Newton system the running time obeys the linearised time equation, d(t1 - t) = tb db + tw (df + dp) + rt: a row's
     * dependence on it folds into its (b, f, p) entries and a shift gd rt of its residual -- the stage blocks keep their pattern.
     * The multiplier of the time equation found this way is lam_t + sum_r (gd_r nu_r+ + ...), see direction().
     */
    __device__ __forceinline__ void fold_running_time(const Ev &ev, double (&gb)[NR], double (&gf)[NR], double (&gp)[NR], const double (&gd)[NR]) const
    {
        if (!INTEG) return;
#pragma unroll
        for (int r = RLTR; r <= RLRG; r++) { gb[r] += gd[r]*ev.tb; gf[r] += gd[r]*ev.tw; if (withPn()) gp[r] += gd[r]*ev.tw; }
    }

    /* objective gradient wrt (f, p, s, q) of the interval and its (constant) curvature; terminal time handled by node N */
    __device__ __forceinline__ void obj_grads(int j, double q, double &of, double &op, double &os, double &oq, double &off, double &opp) const
    {
        const double sc = U.sf/P.objDen;
        of = op = os = oq = off = opp = 0;
        if (!n[j].ival()) return;
        const double f = n[j].x[VF], p = withPn() ? n[j].x[VP] : 0.0;
        if (energyOpt()) {
            of = sc*n[j].ds; os = INTEG ? sc : sc*n[j].ds;
            if (n[j].i > 0) { of += sc*2e-3*(f - q); oq = -sc*2e-3*(f - q); off = sc*2e-3; }
        } else {
            of = sc*2e-4*f; off = sc*2e-4;
            if (withPn()) { op = sc*2e-4*p; opp = sc*2e-4; }
        }
    }

    /* primal direction and new dynamics multipliers of node j, as the Riccati sweep left them in the node's stage block */
    struct Dir { double dx[NV], lt, lb; };
    __device__ __forceinline__ void load_dir(int j, Dir &d) const
    {
        const NodeT &nd = n[j];
#pragma unroll
        for (int k = 0; k < NV; k++) d.dx[k] = 0;
        d.lt = d.lb = 0;
        if (!nd.node()) return;
        const double *s = c.S + nd.i*S_STRIDE;
        if (nd.on(VT)) d.dx[VT] = s[S_DT];
        if (nd.on(VB)) d.dx[VB] = s[S_DB];
        if (nd.ival()) {
            if (nd.on(VF)) d.dx[VF] = s[S_DF];
            if (nd.on(VP)) d.dx[VP] = s[S_DP];
            if (nd.on(VS)) d.dx[VS] = s[S_DS];
            d.lt = s[S_LT]; d.lb = s[S_LB];
        }
    }

    /*
     * One pass over the current point (evaluation `e`, residuals in resc/resd): optimality error of the scaled
     * problem (W&B eq. (5)), and the pieces of the filter's merit pair: theta, and phi(mu) = obj - mu L + kappa_d mu D.
     */
    struct Err { double dual, primal, primal_u, cmax, cmin, sd, sc, theta, L, D, obj; };

    struct RsSums { double theta, np, lognp, prox; };      /* restoration: 1-norm of the relaxed rows, sum(n + p), sum log(n p), |D_R (x - x_R)|^2 */
    __device__ __forceinline__ void kkt_pass(Err &E, const bool resto = false, const double eta = 0.0, const double rho = 0.0, RsSums *RS = nullptr)
    {
        Ev e[SPT];
#pragma unroll
        for (int j = 0; j < SPT; j++) load_ev1(j, e[j]);
        double gl[SPT][NV];
        double dual = 0, prim = 0, prim_u = 0, cmax = -INFINITY, cmin = INFINITY, sumlam = 0, sumz = 0, nlam = 0, nz = 0;
        double th = 0, logs = 0, damp = 0, obj = 0;
        double rs_th = 0, rs_np = 0, rs_log = 0, rs_prox = 0, rs_prim = 0;
        LogSum lsum;
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<0>();
            const NodeT &nd = n[j];
            const double q = nb_q(j);
            double out_q = 0, out_t1 = 0, out_b1 = 0, prod = 1.0;
#pragma unroll
            for (int k = 0; k < NV; k++) gl[j][k] = 0;
            if (nd.ival()) {
                double gb[NR], gf[NR], gp[NR], gs[NR], gb1[NR], gd[NR];
                row_grads(j, e[j], gb, gf, gp, gs, gb1, gd);
                double of, op, os, oq, off, opp;
                obj_grads(j, q, of, op, os, oq, off, opp);
                if (resto) { of = op = os = oq = 0; }
                gl[j][VF] = of; gl[j][VP] = op; gl[j][VS] = os; out_q = oq;
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    if (!rowOn(r)) continue;
                    gl[j][VB] += nd.nu[r]*gb[r]; gl[j][VF] += nd.nu[r]*gf[r]; gl[j][VP] += nd.nu[r]*gp[r]; gl[j][VS] += nd.nu[r]*gs[r]; out_b1 += nd.nu[r]*gb1[r];
                    if (INTEG) { out_t1 += nd.nu[r]*gd[r]; gl[j][VT] -= nd.nu[r]*gd[r]; }      /* running time = t1 - t */
                }
                /* dynamics rows: c_t = t1 - t - tau, c_b = b1 - b+ */
                out_t1 += nd.lam[0]; gl[j][VT] -= nd.lam[0];
                gl[j][VB] -= nd.lam[0]*e[j].tb + nd.lam[1]*e[j].Bb;
                gl[j][VF] -= nd.lam[0]*e[j].tw + nd.lam[1]*e[j].Bw;
                if (withPn()) gl[j][VP] -= nd.lam[0]*e[j].tw + nd.lam[1]*e[j].Bw;
                out_b1 += nd.lam[1];
                prim = fmax(prim, fmax(nd.sct*fabs(resc[j][0]), nd.scb*fabs(resc[j][1])));
                prim_u = fmax(prim_u, fmax(fabs(resc[j][0]), fabs(resc[j][1])));
                th += nd.sct*fabs(resc[j][0]) + nd.scb*fabs(resc[j][1]);
                sumlam += fabs(nd.lam[0])/nd.sct + fabs(nd.lam[1])/nd.scb; nlam += 2;
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    if (!rowOn(r)) continue;
                    const double viol = fabs(resd[j][r]);
                    prim = fmax(prim, viol); prim_u = fmax(prim_u, viol/U.rs[r]); th += viol;
                    sumlam += fabs(nd.nu[r]); nlam += 1;
                    double gsl = -nd.nu[r];
                    if (rL(r)) { const double s = nd.sg[r] - U.dL[r]; gsl -= nd.zLs[r]; const double cp = s*nd.zLs[r]; cmax = fmax(cmax, cp); cmin = fmin(cmin, cp); sumz += nd.zLs[r]; nz += 1; prod *= s; }
                    if (rU(r)) { const double s = U.dU[r] - nd.sg[r]; gsl += nd.zUs[r]; const double cp = s*nd.zUs[r]; cmax = fmax(cmax, cp); cmin = fmin(cmin, cp); sumz += nd.zUs[r]; nz += 1; prod *= s; }
                    if (rL(r) && !rU(r)) damp += nd.sg[r] - U.dL[r];
                    if (!rL(r) && rU(r)) damp += U.dU[r] - nd.sg[r];
                    dual = fmax(dual, fabs(gsl));
                }
                if (resto) {
                    /* optimality error of the restoration problem: the relaxed rows row + n - p, stationarity in (n, p), their complementarity */
#pragma unroll
                    for (int jr = 0; jr < 2 + NR; jr++) {
                        if (!rs_on(jr)) continue;
                        const double row = jr == 0 ? nd.sct*resc[j][0] : jr == 1 ? nd.scb*resc[j][1] : resd[j][jr >= 2 ? jr - 2 : 0];
                        const double y = jr == 0 ? nd.lam[0]/nd.sct : jr == 1 ? nd.lam[1]/nd.scb : nd.nu[jr >= 2 ? jr - 2 : 0];
                        const double rn = wf(W_RN + jr, nd.i), rp = wf(W_RP + jr, nd.i), zn = wf(W_RZN + jr, nd.i), zp = wf(W_RZP + jr, nd.i);
                        const double v = fabs(row + rn - rp);
                        rs_prim = fmax(rs_prim, v); rs_th += v;
                        dual = fmax(dual, fmax(fabs(rho + y - zn), fabs(rho - y - zp)));
                        cmax = fmax(cmax, fmax(rn*zn, rp*zp)); cmin = fmin(cmin, fmin(rn*zn, rp*zp));
                        sumz += zn + zp; nz += 2;
                        rs_np += rn + rp; rs_log += log(rn) + log(rp);
                    }
                }
            } else if (nd.i == P.N && !energyOpt() && !resto) gl[j][VT] = U.sf/P.objDen;
            if (nd.node()) {
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    if (!nd.on(k)) continue;
                    prod *= nd.x[k] - lbv(k);
                    if (hasU(k)) prod *= ubv(j, k) - nd.x[k];
                    else damp += nd.x[k] - lbv(k);
                }
                { double xl[NV];
#pragma unroll
                    for (int k = 0; k < NV; k++) xl[k] = nd.x[k];
                    obj += objective_term<loss_integrated(DYN), FULL>(P, nd, xl, q, U.sf); }
            }
            lsum.add(prod);
            /* the contributions that belong to the neighbours' variables */
            c.o1[nd.i] = out_q; c.o2[nd.i] = out_t1; c.o3[nd.i] = out_b1;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<0>();
            const NodeT &nd = n[j];
            if (!nd.node()) continue;
            if (nd.i > 0) { gl[j][VT] += c.o2[nd.i - 1]; gl[j][VB] += c.o3[nd.i - 1]; }
            if (nd.i + 1 < P.N) gl[j][VF] += c.o1[nd.i + 1];
#pragma unroll
            for (int k = 0; k < NV; k++) {
                if (!nd.on(k)) continue;
                double g = gl[j][k];
                if (resto) { const double xr = wf(W_XR + k, nd.i), dr = 1.0/fmax(1.0, fabs(xr)), qv = dr*(nd.x[k] - xr); g += eta*dr*qv; rs_prox += qv*qv; }
                { const double s = nd.x[k] - lbv(k); g -= nd.zL[k]; const double cp = s*nd.zL[k]; cmax = fmax(cmax, cp); cmin = fmin(cmin, cp); sumz += nd.zL[k]; nz += 1; }
                if (hasU(k)) { const double s = ubv(j, k) - nd.x[k]; g += nd.zU[k]; const double cp = s*nd.zU[k]; cmax = fmax(cmax, cp); cmin = fmin(cmin, cp); sumz += nd.zU[k]; nz += 1; }
                dual = fmax(dual, fabs(g));
            }
        }
        logs = lsum.value();
        if (resto) prim = rs_prim;
        double vm[5] = {dual, prim, prim_u, cmax, -cmin};
        block_reduce<5>(vm, OpMax(), c);
        double vs[8] = {sumlam, sumz, nlam, nz, th, logs, damp, obj};
        block_reduce<8>(vs, OpSum(), c);
        E.dual = uni(vm[0]); E.primal = uni(vm[1]); E.primal_u = uni(vm[2]); E.cmax = uni(vm[3]); E.cmin = -uni(vm[4]);
        E.sd = uni(fmax(K_SMAX, (vs[0] + vs[1])/fmax(1.0, vs[2] + vs[3]))/K_SMAX);
        E.sc = uni(fmax(K_SMAX, vs[1]/fmax(1.0, vs[3]))/K_SMAX);
        E.theta = uni(vs[4]); E.L = uni(vs[5]); E.D = uni(vs[6]); E.obj = uni(vs[7]);
        if (resto) {
            double vr[4] = {rs_th, rs_np, rs_log, rs_prox};
            block_reduce<4>(vr, OpSum(), c);
            RS->theta = uni(vr[0]); RS->np = uni(vr[1]); RS->lognp = uni(vr[2]); RS->prox = uni(vr[3]);
        }
    }
    __device__ static __forceinline__ double compl_err(const Err &E, double mu_) { return (E.cmax >= E.cmin) ? fmax(fabs(E.cmax - mu_), fabs(E.cmin - mu_)) : 0.0; }
    __device__ static __forceinline__ double total_err(const Err &E, double mu_) { return fmax(E.dual/E.sd, fmax(E.primal, compl_err(E, mu_)/E.sc)); }

    /*
     * Condensed stage block of every node into LDS (W&B eq. (13) with slacks and bound multipliers eliminated).
     * MODE_LSQ: least-squares multiplier system (W = 0, Sigma = I, gradient = grad f - zL + zU).
     */
    __device__ __forceinline__ void assemble(const int mode, double mu_, double dw, const double eta = 0.0)
    {
        constexpr bool UNFOLD = loss_integrated(DYN);      /* (see below: MODE_RESTO keeps the running time of the integrated loss rows unfolded) */
        Ev e[SPT];
#pragma unroll
        for (int j = 0; j < SPT; j++) load_ev(j, e[j]);
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<1>();
            const NodeT &nd = n[j];
            const double q = nb_q(j);
            double Htt = 0, Hbb = 0, Hbq = 0, Hbf = 0, Hbp = 0, Hqq = 0, Hqf = 0, Hff = 0, Hfp = 0, Hfs = 0, Hpp = 0, Hss = 0;
            double ht = 0, hb = 0, hq = 0, hf = 0, hp = 0, hs = 0;
            double nHbb = 0, nHbq = 0, nhb = 0;
            double Hbs = 0, Hps = 0, Eb = 0, Es = 0;     /* only the dynamic / integrated loss rows fill these */
            double oa = 0, ob = 0;                       /* Hff - Hfp, Hpp - Hfp accumulated on their own (S_OA, S_OB) */
            /* MODE_RESTO with integrated loss rows: the rows' share in the running time d = t_{i+1} - t_i stays unfolded (the time row is relaxed there, so
             * d no longer obeys its linearisation): curvature W_xd over x = (b, f, p, s), W_dd, gradient h_d -- riccati_resto turns them into entries of
             * (t_i, t_{i+1}) */
            const bool unfold = UNFOLD && mode == MODE_RESTO;
            double Wdd = 0, Wbd = 0, Wfd = 0, Wpd = 0, Wsd = 0, hd = 0;
            if (nd.ival()) {
                const double f = nd.x[VF];
                double gb[NR], gf[NR], gp[NR], gs[NR], gb1[NR], gd[NR];
                row_grads(j, e[j], gb, gf, gp, gs, gb1, gd);
                if (!unfold) fold_running_time(e[j], gb, gf, gp, gd);
                const double rt = (mode == MODE_NEWTON) ? -resc[j][0] : 0.0;
                double of, op, os, oq, off, opp;
                obj_grads(j, q, of, op, os, oq, off, opp);
                if (mode == MODE_RESTO) { of = op = os = oq = off = opp = 0; }      /* the restoration problem has no objective but the proximity term */
                hf = of; hp = op; hs = os; hq = oq;
                if (mode != MODE_LSQ) {
                    Hff = off; Hpp = opp; oa = off; ob = opp;
                    if (energyOpt() && nd.i > 0) { Hqq = off; Hqf = -off; }
                    /* - lam_t hess(tau) - lam_b hess(b+) */
                    const double hbb = -(nd.lam[0]*e[j].tbb + nd.lam[1]*e[j].Bbb), hbw = -(nd.lam[0]*e[j].tbw + nd.lam[1]*e[j].Bbw),
                                 hww = -(nd.lam[0]*e[j].tww + nd.lam[1]*e[j].Bww);
                    Hbb += hbb; Hbf += hbw; Hff += hww;
                    if (withPn()) { Hbp += hbw; Hfp += hww; Hpp += hww; } else oa += hww;
                    /* nu * hess(row) */
                    const double b = nd.x[VB];
                    if (rowOn(RPW0)) { Hbf += nd.nu[RPW0]*U.rs[RPW0]*0.5/e[j].sb; Hbb += nd.nu[RPW0]*U.rs[RPW0]*(-0.25*f/(b*e[j].sb)); }
                    if (rowOn(RPW1)) { nHbq += nd.nu[RPW1]*U.rs[RPW1]*0.5/e[j].sb1; nHbb += nd.nu[RPW1]*U.rs[RPW1]*(-0.25*f/(e[j].b1*e[j].sb1)); }
                    if (rowOn(RACC)) Hbb += nd.nu[RACC]*U.rs[RACC]*0.25*P.sr1/(b*e[j].sb);
                    if (INTEG && rowOn(RLTR)) {
                        /* rows s + kappa f X(v(b), d, f + p) resp. s - E_k(v(b), d, f + p, f), d = t1 - t: sum_r nu_r hess(row_r) in (b, f, p, d) (rows_hess), then
                         * d folded away with dd = tb db + tw (df + dp) + rt (quadratic form; the part linear in rt goes to the gradient) */
                        const LossHess L = rows_hess(j, f, b, e[j]);
                        constexpr double W = 1.0;
                        const double tb = unfold ? 0.0 : e[j].tb, tw = unfold ? 0.0 : e[j].tw;
                        if (unfold) { Wbd += W*L.bd; Wfd += W*L.fd; Wdd += W*L.dd; if (withPn()) Wpd += W*L.pd; }
                        Hbb += W*(L.bb + 2*tb*L.bd + tb*tb*L.dd);
                        Hbf += W*(L.bf + tb*L.fd + tw*L.bd + tb*tw*L.dd);
                        Hff += W*(L.ff + 2*tw*L.fd + tw*tw*L.dd);
                        if (withPn()) { oa += W*((L.ff - L.fp) + tw*(L.fd - L.pd)); ob += W*((L.pp - L.fp) + tw*(L.pd - L.fd)); }
                        else oa += W*(L.ff + 2*tw*L.fd + tw*tw*L.dd);
                        hb += W*rt*(L.bd + L.dd*tb); hf += W*rt*(L.fd + L.dd*tw);
                        if (withPn()) {
                            Hbp += W*(L.bp + tb*L.pd + tw*L.bd + tb*tw*L.dd);
                            Hfp += W*(L.fp + tw*L.pd + tw*L.fd + tw*tw*L.dd);
                            Hpp += W*(L.pp + 2*tw*L.pd + tw*tw*L.dd);
                            hp += W*rt*(L.pd + L.dd*tw);
                        }
                    }
                    if (DYN == LOSS_TABLE && rowOn(RLTR)) {
                        /* rows s - g(f, vbar(b, b1)): nu * hess = -nu * hess(g) */
                        const double vb = 0.25/e[j].sb, vb1 = 0.25/e[j].sb1, vbb = -0.125/(b*e[j].sb), vb1b1 = -0.125/(e[j].b1*e[j].sb1);
#pragma unroll
                        for (int k = 0; k < 2; k++) {
                            const int r = (k == 0) ? RLTR : RLRG;
                            const double w = -nd.nu[r]*U.rs[r];
                            const double gv = e[j].lg[k][1], gff = e[j].lg[k][2], gfv = e[j].lg[k][3], gvv = e[j].lg[k][4];
                            Hff += w*gff; oa += w*gff; Hbf += w*gfv*vb; nHbq += w*gfv*vb1;
                            Hbb += w*(gvv*vb*vb + gv*vbb); nHbb += w*(gvv*vb1*vb1 + gv*vb1b1); Eb += w*gvv*vb*vb1;
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    if (!rowOn(r)) continue;
                    double Sg, coef;
                    if (mode == MODE_NEWTON) { double gphi; row_terms(j, r, mu_, Sg, gphi); Sg += dw; coef = Sg*(resd[j][r] + gd[r]*rt) + gphi; }
                    else if (mode == MODE_RESTO) {
                        /* relaxed row: Sigma -> 1/(D + 1/Sigma); resd carries rhat */
                        double gphi; row_terms(j, r, mu_, Sg, gphi); Sg += dw;
                        const double St = 1.0/(rs_D(nd.i, 2 + r) + 1.0/Sg);
                        coef = St*resd[j][r] + gphi*St/Sg; Sg = St;
                    }
                    else { Sg = 1.0; coef = -nd.zLs[r] + nd.zUs[r]; }
                    if (unfold) { hd += coef*gd[r]; Wbd += Sg*gb[r]*gd[r]; Wfd += Sg*gf[r]*gd[r]; Wpd += Sg*gp[r]*gd[r]; Wsd += Sg*gs[r]*gd[r]; Wdd += Sg*gd[r]*gd[r]; }
                    hb += coef*gb[r]; hf += coef*gf[r]; hp += coef*gp[r]; hs += coef*gs[r]; nhb += coef*gb1[r];
                    Hbb += Sg*gb[r]*gb[r]; Hbf += Sg*gb[r]*gf[r]; Hbp += Sg*gb[r]*gp[r];
                    Hff += Sg*gf[r]*gf[r]; Hfp += Sg*gf[r]*gp[r]; Hfs += Sg*gf[r]*gs[r];
                    Hpp += Sg*gp[r]*gp[r]; Hss += Sg*gs[r]*gs[r];
                    oa += Sg*gf[r]*(gf[r] - gp[r]); ob += Sg*gp[r]*(gp[r] - gf[r]);      /* (the acceleration row: gf = gp, nothing of it in either) */
                    nHbq += Sg*gf[r]*gb1[r]; nHbb += Sg*gb1[r]*gb1[r];
                    if (DYN) { Hbs += Sg*gb[r]*gs[r]; Hps += Sg*gp[r]*gs[r]; Eb += Sg*gb[r]*gb1[r]; Es += Sg*gs[r]*gb1[r]; }
                }
                if (unfold) { Htt += Wdd; ht -= hd; }      /* (d = t_{i+1} - t_i: the entries at t_i; those at t_{i+1} go to the next node below) */
            } else if (nd.i == P.N && !energyOpt() && mode != MODE_RESTO) ht = U.sf/P.objDen;
            /* bounds of the node's own variables + regularisation */
            if (nd.node()) {
                double Sv[NV], gv[NV];
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    Sv[k] = 0; gv[k] = 0;
                    if (!nd.on(k)) continue;
                    if (mode == MODE_NEWTON) { var_terms(j, k, mu_, Sv[k], gv[k]); Sv[k] += dw; }
                    else if (mode == MODE_RESTO) {
                        var_terms(j, k, mu_, Sv[k], gv[k]); Sv[k] += dw;
                        const double xr = wf(W_XR + k, nd.i), dr = 1.0/fmax(1.0, fabs(xr)), w = eta*dr*dr;      /* eta/2 |D_R (x - x_R)|^2 */
                        Sv[k] += w; gv[k] += w*(nd.x[k] - xr);
                    }
                    else { Sv[k] = 1.0; gv[k] = -nd.zL[k] + nd.zU[k]; }
                }
                Htt += Sv[VT]; ht += gv[VT]; Hbb += Sv[VB]; hb += gv[VB]; Hff += Sv[VF]; hf += gv[VF]; Hpp += Sv[VP]; hp += gv[VP]; Hss += Sv[VS]; hs += gv[VS];
                oa += Sv[VF]; ob += Sv[VP];
                double *s = c.S + nd.i*S_STRIDE;
                if (nd.ival()) {
                    s[S_TB] = e[j].tb; s[S_TW] = e[j].tw; s[S_BB] = e[j].Bb; s[S_BW] = e[j].Bw;
                    s[S_RT] = (mode == MODE_NEWTON) ? -resc[j][0] : (mode == MODE_RESTO) ? -resc[j][0]/nd.sct : 0.0;
                    s[S_RB] = (mode == MODE_NEWTON) ? -resc[j][1] : (mode == MODE_RESTO) ? -resc[j][1]/nd.scb : 0.0;
                }
                if (nd.ival()) {
                    /* everything that does not depend on the value function is folded in here, off the serial path: the couplings
                     * with b_{i+1} (dynamic loss rows; db+ = Bb db + Bw (df + dp) + rb) and the elimination of the slack variable s,
                     * whose pivot is Hss.  A pivot that is not positive poisons the block with NaN: the sweep then reports the
                     * wrong inertia.  The last interval keeps its s row (b_N is a parameter; riccati_solve) */
                    const double Bb = e[j].Bb, Bw = e[j].Bw, rb = (mode == MODE_NEWTON) ? -resc[j][1] : 0.0;
                    const bool last = nd.i == P.N - 1;
                    double Gbs = Hbs, Gfs = Hfs, Gps = Hps, gsv = hs;
                    /* (MODE_RESTO: the dynamics rows are relaxed, nothing folds through them, and the slack variable stays a control of its own: riccati_resto
                     *  works on the unreduced blocks, with the couplings Eb, Es of (b_i, s_i) with b_{i+1} as cross terms) */
                    if (DYN && !last && mode != MODE_RESTO) {
                        Hbb += 2*Eb*Bb; Hbf += Eb*Bw; hb += Eb*rb;
                        if (withPn()) Hbp += Eb*Bw;
                        Gbs += Es*Bb; Gfs += Es*Bw; gsv += Es*rb;
                        if (withPn()) Gps += Es*Bw;
                    }
                    const double is = (Hss > 0) ? 1.0/Hss : NAN;
                    if (!last && mode != MODE_RESTO) {
                        const double wf = Gfs*is;
                        Hff -= Gfs*wf; hf -= wf*gsv;
                        oa -= (Gfs - Gps)*wf; ob -= Gps*(Gps - Gfs)*is;
                        if (DYN) {
                            const double wb = Gbs*is, wp = Gps*is;
                            Hbb -= Gbs*wb; Hbf -= Gfs*wb; Hbp -= Gps*wb; Hfp -= Gps*wf; Hpp -= Gps*wp;
                            hb -= wb*gsv; hp -= wp*gsv;
                        }
                    }
                    s[S_GFS] = Gfs; s[S_IS] = is; s[S_GS] = gsv;
                    if (DYN) { s[S_GBS] = Gbs; s[S_GPS] = Gps; s[S_EB] = last ? 0.0 : Eb; s[S_ES] = last ? 0.0 : Es; }
                }
                s[S_HTT] = Htt; s[S_HBB] = Hbb; s[S_HBQ] = Hbq; s[S_HBF] = Hbf; s[S_HBP] = Hbp; s[S_HQQ] = Hqq; s[S_HQF] = Hqf;
                s[S_HFF] = Hff; s[S_HFP] = Hfp; s[S_HPP] = Hpp; s[S_OA] = oa; s[S_OB] = ob;
                s[S_HT] = ht; s[S_HB] = hb; s[S_HQ] = hq; s[S_HF] = hf; s[S_HP] = hp;
            }
            /* the end-of-interval power row lives in the next stage's (b, q) block */
            c.o1[nd.i] = nHbb; c.o2[nd.i] = nHbq; c.o3[nd.i] = nhb;
            if constexpr (UNFOLD && STREAM) {
                if (unfold) {
                    wf(W_RX + 0, nd.i) = Wdd; wf(W_RX + 1, nd.i) = Wbd; wf(W_RX + 2, nd.i) = Wfd; wf(W_RX + 3, nd.i) = Wpd; wf(W_RX + 4, nd.i) = Wsd; wf(W_RX + 5, nd.i) = hd;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<1>();
            const NodeT &nd = n[j];
            if (nd.node() && nd.i > 0) {
                double *s = c.S + nd.i*S_STRIDE;
                s[S_HBB] += c.o1[nd.i - 1]; s[S_HBQ] += c.o2[nd.i - 1]; s[S_HB] += c.o3[nd.i - 1];
                if constexpr (UNFOLD && STREAM) {
                    if (mode == MODE_RESTO) { s[S_HTT] += wf(W_RX + 0, nd.i - 1); s[S_HT] += wf(W_RX + 5, nd.i - 1); }      /* the entries at t_{i+1} of the interval before */
                }
            }
        }
        __syncthreads();
    }

    /*
     * What the serial sweeps leave to the nodes: the step of the slack variable s (eliminated in assemble()) and the new
     * multipliers of the dynamics, lam+ = -(P+ x+ + p+ + E^T y), from the value function stashed in the node's own block and
     * the step of the next node.  The last interval's ds and lam_b come from the sweep itself.
     */
    __device__ __forceinline__ void finish_direction()
    {
        const int N = P.N;
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            const NodeT &nd = n[j];
            if (!nd.ival()) continue;
            double *s = c.S + nd.i*S_STRIDE;
            const double *s1 = s + S_STRIDE;
            const double nt = s1[S_DT], nb = s1[S_DB], df = s[S_DF];
            const double lt = -(s[S_PN + 0]*nt + s[S_PN + 1]*nb + s[S_PN + 2]*df + s[S_PV + 0]);
            if (nd.i < N - 1) {
                const double db = s[S_DB], dp = s[S_DP];
                double acc = s[S_GS] + s[S_GFS]*df;
                if (DYN) acc += s[S_GBS]*db + s[S_GPS]*dp;
                const double dsl = -acc*s[S_IS];
                double lb = -(s[S_PN + 1]*nt + s[S_PN + 3]*nb + s[S_PN + 4]*df + s[S_PV + 1]);
                if (DYN) lb -= s[S_EB]*db + s[S_ES]*dsl;      /* coupling terms of the dynamic loss rows */
                s[S_DS] = dsl; s[S_LB] = lb;
            }
            s[S_LT] = lt;
        }
        __syncthreads();
    }

    /* KKT solve: assemble, serial Riccati, read the direction back.  Returns the inertia flag (uniform). */
    __device__ __forceinline__ bool direction(const int mode, double mu_, double dw)
    {
        c.mark(PH_OTHER); phase_fence(PH_OTHER);
        assemble(mode, mu_, dw);
        c.mark(PH_ASSEMBLE); phase_fence(PH_ASSEMBLE);
        int par = -1;
#if MSD_PARALLEL_RICCATI
        /* (the scan keeps its wave totals for up to eight waves: the streamed kernels of 512 threads take it too, on their stage blocks in
         * device memory -- a serial sweep there pays a memory round trip per stage) */
        if (!STREAM || NT <= 512) {
        par = ParallelRiccati<SPT, DYN>::solve(P.N, withPn(), c);
        c.red_slot++;        /* one block reduction inside */
        if (par < 0) {       /* the scan broke down (cold path): the sweeps overwrite the blocks, so assemble again */
            assemble(mode, mu_, dw);
            if (c.tid == 0) c.misc[MISC_FALLBACKS] += 1.0;
        }
        }      /* (the 1024-thread streamed kernels take the serial sweeps) */
#endif
        if (par < 0) {       /* serial sweeps on one lane: the fallback of the scan (cold) or, with MSD_PARALLEL_RICCATI = 0, the only path */
            if (c.tid == 0) {
                const unsigned long long t0 = __builtin_readcyclecounter();
                c.misc[0] = riccati_solve<DYN>(P.N, withPn(), c.S, nullptr) ? 1.0 : 0.0;
                c.misc[1] += (double)(__builtin_readcyclecounter() - t0);
            }
            __syncthreads();
        }
        c.mark(PH_RICCATI); phase_fence(PH_RICCATI);
        const bool ok = (par >= 0) ? (par == 1) : (uni(c.misc[0]) != 0.0);
        if (ok) finish_direction();
        if (ok) {
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                node_fence<2>();
                NodeT &nd = n[j];
#pragma unroll
                for (int r = 0; r < NR; r++) nd.dsg[r] = 0;
                if (!nd.ival()) continue;
                Dir d; load_dir(j, d);
                const double db1 = c.S[(nd.i + 1)*S_STRIDE + S_DB];
                const double dd = INTEG ? c.S[(nd.i + 1)*S_STRIDE + S_DT] - d.dx[VT] : 0.0;      /* step of the running time */
                double gb[NR], gf[NR], gp[NR], gs[NR], gb1[NR], gd[NR];
                Ev ej; load_ev1(j, ej);
                row_grads(j, ej, gb, gf, gp, gs, gb1, gd);
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    if (!rowOn(r)) continue;
                    const double lin = gb[r]*d.dx[VB] + gf[r]*d.dx[VF] + gp[r]*d.dx[VP] + gs[r]*d.dx[VS] + gb1[r]*db1 + gd[r]*dd;
                    /* Newton: slack step; least squares: nu = Sigma dsigma + (-zL + zU) with Sigma = 1, parked in dsg */
                    nd.dsg[r] = (mode == MODE_NEWTON) ? resd[j][r] + lin : lin + (-nd.zLs[r] + nd.zUs[r]);
                }
                if (INTEG && rowOn(RLTR)) {
                    /* the sweeps solved the system with the running time folded into (b, f, p): their multiplier of the time equation is
                     * lam_t + sum_r (gd_r nu_r+ + nu_r d(grad_d row_r)); take the rows' share out again (fold_running_time) */
                    double corr = 0;
#pragma unroll
                    for (int r = RLTR; r <= RLRG; r++) {
                        double nup = nd.dsg[r];
                        if (mode == MODE_NEWTON) { double Sg, gphi; row_terms(j, r, mu_, Sg, gphi); nup = (Sg + dw)*nd.dsg[r] + gphi; }
                        corr += gd[r]*nup;
                    }
                    if (mode == MODE_NEWTON) {
                        const LossHess L = rows_hess(j, nd.x[VF], nd.x[VB], ej);
                        constexpr double W = 1.0;
                        corr += W*(L.bd*d.dx[VB] + L.fd*d.dx[VF] + L.pd*d.dx[VP] + L.dd*dd);
                    }
                    c.S[nd.i*S_STRIDE + S_LT] -= corr;
                }
            }
        }
        __syncthreads();
        c.mark(PH_READBACK); phase_fence(PH_READBACK);
        return ok;
    }

    __device__ __forceinline__ double dzL_var(int j, int k, double mu_, double dxk) const { const double s = n[j].x[k] - lbv(k); return mu_/s - n[j].zL[k] - n[j].zL[k]/s*dxk; }
    __device__ __forceinline__ double dzU_var(int j, int k, double mu_, double dxk) const { const double s = ubv(j, k) - n[j].x[k]; return mu_/s - n[j].zU[k] + n[j].zU[k]/s*dxk; }
    __device__ __forceinline__ double dzL_row(int j, int r, double mu_) const { const double s = n[j].sg[r] - U.dL[r]; return mu_/s - n[j].zLs[r] - n[j].zLs[r]/s*n[j].dsg[r]; }
    __device__ __forceinline__ double dzU_row(int j, int r, double mu_) const { const double s = U.dU[r] - n[j].sg[r]; return mu_/s - n[j].zUs[r] + n[j].zUs[r]/s*n[j].dsg[r]; }

    /* fraction-to-the-boundary step lengths of the current direction: primal, dual */
    __device__ __forceinline__ void step_lengths(double mu_, double tau_, double &apr, double &adu)
    {
        double ap = 1.0, ad = 1.0;
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<3>();
            const NodeT &nd = n[j];
            if (nd.node()) {
                Dir dd; load_dir(j, dd);
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    if (!nd.on(k)) continue;
                    const double d = dd.dx[k];
                    { if (d < 0) ap = fmin(ap, -tau_*(nd.x[k] - lbv(k))/d); const double dz = dzL_var(j, k, mu_, d); if (dz < 0) ad = fmin(ad, -tau_*nd.zL[k]/dz); }
                    if (hasU(k)) { if (d > 0) ap = fmin(ap, tau_*(ubv(j, k) - nd.x[k])/d); const double dz = dzU_var(j, k, mu_, d); if (dz < 0) ad = fmin(ad, -tau_*nd.zU[k]/dz); }
                }
            }
            if (nd.ival()) {
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    if (!rowOn(r)) continue;
                    const double d = nd.dsg[r];
                    if (rL(r)) { if (d < 0) ap = fmin(ap, -tau_*(nd.sg[r] - U.dL[r])/d); const double dz = dzL_row(j, r, mu_); if (dz < 0) ad = fmin(ad, -tau_*nd.zLs[r]/dz); }
                    if (rU(r)) { if (d > 0) ap = fmin(ap, tau_*(U.dU[r] - nd.sg[r])/d); const double dz = dzU_row(j, r, mu_); if (dz < 0) ad = fmin(ad, -tau_*nd.zUs[r]/dz); }
                }
            }
        }
        double v[2] = {ap, ad};
        block_reduce<2>(v, OpMin(), c);
        apr = uni(v[0]); adu = uni(v[1]);
    }

    /* the point x + alpha d */
    __device__ __forceinline__ void trial_point(double alpha, double (&xt)[SPT][NV], double (&st)[SPT][NR]) const
    {
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<4>();
            Dir dd; load_dir(j, dd);
#pragma unroll
            for (int k = 0; k < NV; k++) xt[j][k] = n[j].x[k] + alpha*dd.dx[k];
#pragma unroll
            for (int r = 0; r < NR; r++) st[j][r] = n[j].sg[r] + (rowOn(r) ? alpha*n[j].dsg[r] : 0.0);
        }
    }

    /* theta (1-norm of the scaled constraint rows), barrier objective and validity of the point x + alpha d */
    __device__ __forceinline__ void merit(double alpha, double mu_, double &theta, double &phi, bool &ok, const bool resto = false, const double eta = 0.0,
                                          const double rho = 0.0)
    {
        double xt[SPT][NV], st[SPT][NR];
        trial_point(alpha, xt, st);
        publish(xt);
        double ctau = 0, cbp = 0;
        bool pre = false;
        if constexpr (COOP) {
            if (P.integ == MSD_INTEGRATOR_ADAPTIVE) {
                coop_values(n[0].ival(), xt[0][VB], xt[0][VF] + (withPn() ? xt[0][VP] : 0.0), n[0].G, n[0].ds, ctau, cbp);
                coop_valid = true; pre = true;
            }
        }
        double th = 0, logs = 0, damp = 0, bad = 0, obj = 0;
        LogSum lsum;
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<5>();
            const NodeT &nd = n[j];
            double prod = 1.0;
            if (nd.ival()) {
                double cv[2], dv[NR]; Ev dummy;
                eval_interval<false, DYN, GEN, FULL>(P, U, nd.G, nd.ds, xt[j], c.xt[nd.i + 1], c.xb[nd.i + 1], cv, dv, dummy, nullptr, nullptr, pre ? &ctau : nullptr, pre ? &cbp : nullptr);
                if (!resto) th += nd.sct*fabs(cv[0]) + nd.scb*fabs(cv[1]);
                else {
                    /* restoration problem: relaxed rows with the trial (n, p); barrier and penalty terms of (n, p) */
#pragma unroll
                    for (int jr = 0; jr < 2 + NR; jr++) {
                        if (!rs_on(jr)) continue;
                        const double row = jr == 0 ? nd.sct*cv[0] : jr == 1 ? nd.scb*cv[1] : dv[jr >= 2 ? jr - 2 : 0] - st[j][jr >= 2 ? jr - 2 : 0];
                        const double rn = wf(W_RN + jr, nd.i) + alpha*wf(W_RDN + jr, nd.i), rp = wf(W_RP + jr, nd.i) + alpha*wf(W_RDP + jr, nd.i);
                        th += fabs(row + rn - rp);
                        if (rn <= 0 || rp <= 0) bad = 1; else lsum.add(rn*rp);
                        damp += rn + rp; obj += rho*(rn + rp);
                    }
                }
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    if (!rowOn(r)) continue;
                    if (!resto) th += fabs(dv[r] - st[j][r]);
                    if (rL(r)) { const double s = st[j][r] - U.dL[r]; if (s <= 0) bad = 1; else prod *= s; }
                    if (rU(r)) { const double s = U.dU[r] - st[j][r]; if (s <= 0) bad = 1; else prod *= s; }
                    if (rL(r) && !rU(r)) damp += st[j][r] - U.dL[r];
                    if (!rL(r) && rU(r)) damp += U.dU[r] - st[j][r];
                }
            }
            if (nd.node()) {
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    if (!nd.on(k)) continue;
                    { const double s = xt[j][k] - lbv(k); if (s <= 0) bad = 1; else prod *= s; }
                    if (hasU(k)) { const double s = ubv(j, k) - xt[j][k]; if (s <= 0) bad = 1; else prod *= s; }
                    else damp += xt[j][k] - lbv(k);
                    if (resto) { const double xr = wf(W_XR + k, nd.i), qv = (xt[j][k] - xr)/fmax(1.0, fabs(xr)); obj += 0.5*eta*qv*qv; }
                }
                if (!resto) obj += objective_term<loss_integrated(DYN), FULL>(P, nd, xt[j], (nd.i > 0) ? c.xf[nd.i - 1] : 0.0, U.sf);
            }
            lsum.add(prod);
        }
        logs = lsum.value();
        double v[5] = {th, logs, damp, obj, bad};
        block_reduce<5>(v, OpSum(), c);
        theta = uni(v[0]); phi = uni(v[3] - mu_*v[1] + K_D*mu_*v[2]);
        ok = (uni(v[4]) == 0.0) && isfinite(theta) && isfinite(phi);
    }

    /* c and d - sigma at the trial point x + alpha d (needed by the second-order correction only) */
    __device__ __forceinline__ void trial_residuals(double alpha, double (&tc)[SPT][2], double (&td)[SPT][NR])
    {
        double xt[SPT][NV], st[SPT][NR];
        trial_point(alpha, xt, st);
        publish(xt);
        double ctau = 0, cbp = 0;
        bool pre = false;
        if constexpr (COOP) {
            if (P.integ == MSD_INTEGRATOR_ADAPTIVE) {
                coop_values(n[0].ival(), xt[0][VB], xt[0][VF] + (withPn() ? xt[0][VP] : 0.0), n[0].G, n[0].ds, ctau, cbp);
                coop_valid = true; pre = true;
            }
        }
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<6>();
            tc[j][0] = tc[j][1] = 0;
#pragma unroll
            for (int r = 0; r < NR; r++) td[j][r] = 0;
            if (n[j].ival()) {
                double dv[NR]; Ev dummy;
                eval_interval<false, DYN, GEN, FULL>(P, U, n[j].G, n[j].ds, xt[j], c.xt[n[j].i + 1], c.xb[n[j].i + 1], tc[j], dv, dummy, nullptr, nullptr, pre ? &ctau : nullptr, pre ? &cbp : nullptr);
#pragma unroll
                for (int r = 0; r < NR; r++) td[j][r] = rowOn(r) ? dv[r] - st[j][r] : 0.0;
            }
        }
    }

    /* evaluate the current point with derivatives; residuals of the Newton system into resc/resd */
    /*
     * The interval maps of the adaptive shooting integrator with their jets, evaluated by the lanes of a wave together (msd_integ.hpp: DopriValues,
     * dopri_step_jet, jet_after).  Every lane runs the step-size controller on values; an interval that is through after two accepted steps (all but
     * the first and the last of a journey from and to standstill) replays them in jets itself.  The lowest lane with a longer interval -- the owner --
     * finishes its value pass, leaves step sizes and step-start values in the wave's pool, lane k evaluates the local jet of step k, and the owner
     * composes them by the chain rule: 26 steps cost one set of jet stages plus 26 small compositions instead of 26 sets.  A second long interval in
     * the same wave, or one with more than COOP_CAP steps, is evaluated the sequential way (dopri_tb_jet).  Values (tau, b+) are the value pass's: the
     * numbers the value-only evaluation of a trial point computes.  Uniform control flow: every thread of the workgroup calls it.
     */
    static constexpr bool COOP = GEN && !STREAM && SPT == 1 && MSD_COOP_ADAPTIVE && NT <= 128;      /* (one node per lane: the benchmark geometries of these families) */
    __device__ __forceinline__ double &coop_rec(int f) const { return c.pool[c.nw*COOP_POOL + f*NT + c.tid]; }
    /* the value pass of every lane's interval at (b0, force wv): step-size controller on values, records for the jets (above), tau and b+ */
    __device__ __forceinline__ void coop_values(const bool act, const double b0, const double wv, const double G, const double ds, double &tauv, double &bpv)
    {
        double *pool = c.pool + c.wave*COOP_POOL;
        double *rec_h = pool + COOP_HDR, *rec_y = rec_h + COOP_CAP;
        DopriValues s;
        double rh[2] = {0, 0}, ry[2] = {0, 0};
        int na = 0, state = act ? 0 : 1, tries = 0;      /* state: 0 under way, 1 at the end of the interval, -1 the step control collapsed */
        if (act) s.start(P, b0, wv, G, ds);
        while (state == 0 && na < 2) {
            double h, y;
            const int rc = s.attempt(P, wv, G, ds, h, y);
            if (rc == 1) { rh[na] = h; ry[na] = y; na++; if (s.finished()) state = 1; }
            else if (rc < 0 || ++tries >= 100000) state = -1;
        }
        const bool lng = act && state == 0;
        const int owner = (int)wave_reduce(lng ? (double)c.lane : 64.0, OpMin());
        int k = 2;
        if (lng) {
            const bool own = c.lane == owner;
            if (own) { pool[1] = wv; pool[2] = G; pool[3] = ds; rec_h[0] = rh[0]; rec_y[0] = ry[0]; rec_h[1] = rh[1]; rec_y[1] = ry[1]; }
            while (state == 0) {
                double h, y;
                const int rc = s.attempt(P, wv, G, ds, h, y);
                if (rc == 1) { if (own && k < COOP_CAP) { rec_h[k] = h; rec_y[k] = y; } k++; if (s.finished()) state = 1; }
                else if (rc < 0 || ++tries >= 100000) state = -1;
            }
            if (own) pool[0] = (state == 1 && k <= COOP_CAP) ? (double)k : -1.0;
        }
        if (c.lane == 0) pool[4] = (double)owner;
        tauv = (act && state == 1) ? s.yt : NAN; bpv = (act && state == 1) ? s.yb : NAN;
        coop_rec(0) = rh[0]; coop_rec(1) = rh[1]; coop_rec(2) = ry[0]; coop_rec(3) = ry[1]; coop_rec(4) = tauv; coop_rec(5) = bpv;
        coop_rec(6) = (double)(na + 4*(state + 1) + 16*(lng ? 1 : 0));
        coop_rec(7) = b0; coop_rec(8) = wv;
        __syncthreads();
    }

    /*
     * The interval maps of the adaptive shooting integrator with their jets, evaluated by the lanes of a wave together (msd_integ.hpp: DopriValues,
     * dopri_step_jet, jet_after).  Every lane runs the step-size controller on values (coop_values -- or finds that pass done: the current point is the
     * trial point the line search accepted); an interval that is through after two accepted steps (all but the first and the last of a journey from and
     * to standstill) replays them in jets itself.  The lowest lane with a longer interval -- the owner -- has left step sizes and step-start values in
     * the wave's pool, lane k evaluates the local jet of step k, and the owner composes them by the chain rule: 26 steps cost one set of jet stages plus
     * 26 small compositions instead of 26 sets.  A second long interval in the same wave, or one with more than COOP_CAP steps, is evaluated the
     * sequential way (dopri_tb_jet).  Values (tau, b+) are the value pass's: the numbers the value-only evaluation of a trial point computes.  Uniform
     * control flow: every thread of the workgroup calls it.
     */
    __device__ __forceinline__ void coop_adaptive(Jet (&ptau)[SPT], Jet (&pbp)[SPT])
    {
        const NodeT &nd = n[0];
        const bool act = nd.ival();
        const double b0 = nd.x[VB], wv = nd.x[VF] + (withPn() ? nd.x[VP] : 0.0), G = nd.G, ds = nd.ds;
        {
            /* is the value pass on record the one of this point?  (bit-equal b and force in every lane: x + alpha d is formed the same way twice) */
            double v[1] = {(!act || (coop_valid && coop_rec(7) == b0 && coop_rec(8) == wv)) ? 1.0 : 0.0};
            block_reduce<1>(v, OpMin(), c);
            double tv, bv;
            if (uni(v[0]) == 0.0) coop_values(act, b0, wv, G, ds, tv, bv);
            coop_valid = true;
        }
        double *pool = c.pool + c.wave*COOP_POOL;
        const double *rec_h = pool + COOP_HDR, *rec_y = rec_h + COOP_CAP;
        double *out = pool + COOP_HDR + 2*COOP_CAP;
        const int code = (int)coop_rec(6), na = code & 3, state = ((code >> 2) & 3) - 1;
        const bool lng = (code & 16) != 0;
        const int owner = (int)pool[4];
        const int nL = owner < 64 ? (int)pool[0] : 0;      /* steps of the owner's interval to be shared out (0: none) */
        Jet tau = {NAN, NAN, NAN, NAN, NAN, NAN}, bp = tau;
        if (act && !lng) {
            if (state == 1) {
                Jet yb = make_var(Jet(), b0, 0), T = make_zero(Jet());
                const Jet w = make_var(Jet(), wv, 1);
                for (int k = 0; k < na; k++) { Jet d; dopri_step_jet(P, yb, d, w, G, ds, coop_rec(k)); T = T + d; }
                tau = T; bp = yb; tau.v = coop_rec(4); bp.v = coop_rec(5);
            }
        } else if (lng && !(c.lane == owner && nL > 0)) {
            dopri_tb_jet(P, make_var(Jet(), b0, 0), make_var(Jet(), wv, 1), G, ds, tau, bp);
        }
        if (nL > 0 && c.lane < nL) {
            Jet yb = make_var(Jet(), rec_y[c.lane], 0), d;
            const Jet w = make_var(Jet(), pool[1], 1);
            dopri_step_jet(P, yb, d, w, pool[2], pool[3], rec_h[c.lane]);
            double *o = out + 12*c.lane;
            o[0] = yb.v; o[1] = yb.g0; o[2] = yb.g1; o[3] = yb.h00; o[4] = yb.h01; o[5] = yb.h11;
            o[6] = d.v; o[7] = d.g0; o[8] = d.g1; o[9] = d.h00; o[10] = d.h01; o[11] = d.h11;
        }
        __syncthreads();
        if (nL > 0 && c.lane == owner) {
            Jet B = make_var(Jet(), b0, 0), T = make_zero(Jet());
            for (int k = 0; k < nL; k++) {
                const double *o = out + 12*k;
                const Jet phi = {o[0], o[1], o[2], o[3], o[4], o[5]}, psi = {o[6], o[7], o[8], o[9], o[10], o[11]};
                T = T + jet_after(psi, B);
                B = jet_after(phi, B);
            }
            tau = T; bp = B; tau.v = coop_rec(4); bp.v = coop_rec(5);
        }
        ptau[0] = tau; pbp[0] = bp;
    }
    bool coop_valid = false;      /* the records of coop_values are this solve's (the LDS is the previous scenario's at the start) */

    __device__ __forceinline__ void evaluate_current()
    {
        publish_current();
        Jet ptau[SPT], pbp[SPT];
        bool pre = false;
        if constexpr (COOP) {
            if (P.integ == MSD_INTEGRATOR_ADAPTIVE) { coop_adaptive(ptau, pbp); pre = true; }
        }
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<7>();
            resc[j][0] = resc[j][1] = 0;
#pragma unroll
            for (int r = 0; r < NR; r++) resd[j][r] = 0;
            if (n[j].ival()) {
                double dv[NR], cv[2], xl[NV];
                Ev ej;
#pragma unroll
                for (int k = 0; k < NV; k++) xl[k] = n[j].x[k];
                eval_interval<true, DYN, GEN, FULL>(P, U, n[j].G, n[j].ds, xl, c.xt[n[j].i + 1], c.xb[n[j].i + 1], cv, dv, ej, pre ? &ptau[j] : nullptr, pre ? &pbp[j] : nullptr);
                resc[j][0] = cv[0]; resc[j][1] = cv[1];
#pragma unroll
                for (int r = 0; r < NR; r++) resd[j][r] = rowOn(r) ? dv[r] - n[j].sg[r] : 0.0;
                store_ev(j, ej);
            }
        }
    }

    __device__ __forceinline__ bool filter_ok(int nfilt, double theta, double phi) const
    {
        for (int j = 0; j < nfilt; j++)
            if (theta >= c.filt[2*j] && phi >= c.filt[2*j + 1]) return false;
        return true;
    }

    /* acceptance of a trial point by the filter line search (W&B section 2.3); th_ref/phi: current point */
    __device__ __forceinline__ bool acceptable(bool okt, double th_t, double ph_t, double th_ref, double phi, double alpha_test, double gphid, bool ftype,
                                              double theta_max, double theta_min, int nfilt) const
    {
        if (!okt || th_t > theta_max) return false;
        bool acc;
        if (ftype && th_ref <= theta_min) acc = cmp_le(ph_t - phi, ETA_PHI*alpha_test*gphid, phi);
        else acc = cmp_le(th_t, (1 - G_THETA)*th_ref, th_ref) || cmp_le(ph_t - phi, -G_PHI*th_ref, phi);
        return acc && filter_ok(nfilt, th_t, ph_t);
    }

    /* ---------------------------------------------------------------------------------------- */
    /*
     * Profile start (not in the reference, which starts every solve from ocp.py:325-339): a speed profile that respects the
     * limits, accelerates from v0 and brakes to vN with 0.3 m/s^2 and cruises at the speed that uses up the running time; times
     * from the trapezoidal rule; forces from the acceleration the profile needs (profile floored at 5 m/s there, so that the
     * integrator stays away from b = 0), cut to the force and power limits; slacks just above the loss rows.  Same optimum,
     * about half the iterations.  Node-parallel except for the running sum of the times (thread 0, N additions).
     */
    __device__ __forceinline__ void profile_start(double t0, double tEnd, double v0sq, double vNsq)
    {
        const int N = P.N;
        constexpr double A = 0.3, MARG = 0.97*0.97, BFL = 25.0, S0 = 0.02, TFR = 0.995;
        const double L = P.pos[N], span = tEnd - t0;
        double cs = L/span;
        double cap[SPT], up[SPT], dn[SPT];
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            const int i = n[j].i;
            const double ps = (i <= N) ? P.pos[i] : L;
            cap[j] = (i >= 1 && i < N) ? MARG*P.bmax[i] : INFINITY;
            up[j] = v0sq + 2*A*ps; dn[j] = vNsq + 2*A*(L - ps);
        }
#pragma unroll 1
        for (int it = 0; it < 3; it++) {
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                const int i = n[j].i;
                double bi = fmin(fmin(cap[j], cs*cs), fmin(up[j], dn[j]));
                if (i == 0) bi = v0sq;
                if (i == N) bi = vNsq;
                n[j].x[VB] = bi;
                if (i <= N) { c.xb[i] = sqrt(bi); c.o2[i] = bi; }
            }
            __syncthreads();
            double tt = 0;
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                if (!n[j].ival()) continue;
                const int i = n[j].i;
                const double dti = 2*n[j].ds/(c.xb[i] + c.xb[i + 1]);
                c.o1[i] = dti; tt += dti;
            }
            double v[1] = {tt};
            block_reduce<1>(v, OpSum(), c);
            if (it < 2) cs *= uni(v[0])/(span*TFR);
            __syncthreads();
        }
        if (c.tid == 0) {
            double acc = t0;
            for (int i = 0; i <= N; i++) { c.xt[i] = fmin(acc, tEnd); if (i < N) acc += c.o1[i]; }
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            NodeT &nd = n[j];
            const int i = nd.i;
            if (!nd.node()) continue;
            nd.x[VT] = c.xt[i];
            if (!nd.ival()) continue;
            const double b0 = c.o2[i], b1 = c.o2[i + 1], v0 = c.xb[i], v1 = c.xb[i + 1];
            const double bs0 = fmax(b0, BFL), bs1 = fmax(b1, BFL);
            const double vm = 0.5*(sqrt(bs0) + sqrt(bs1));
            const double f = (bs1 - bs0)/(2*nd.ds) + P.sr0 + P.sr1*vm + P.sr2*vm*vm + nd.G;
            double fel = fmin(fmax(f, P.fmin), P.fmax);
            if (hasPower()) { const double vmx = fmax(v0, v1); fel = fmin(fmax(fel, -fabs(P.pwL)/vmx), fabs(P.pwU)/vmx); }
            double fpb = 0.0;
            if (withPn()) {
                /* the interior push will move Fpb at least this far below its upper bound 0: start there and let Fel make up for
                 * it, so that the pushed point still has the acceleration the profile needs (a start from standstill must not stall) */
                const double pb = K_PUSH*fmin(1.0, fabs(P.fminPn));
                fpb = fmin(fmin(fmax(f - fel, P.fminPn), 0.0), -pb);
                fel = fmin(fmax(f - fpb, P.fmin), P.fmax);
                if (hasPower()) { const double vmx = fmax(v0, v1); fel = fmin(fmax(fel, -fabs(P.pwL)/vmx), fabs(P.pwU)/vmx); }
            }
            double sl;
            if (DYN == LOSS_TABLE || ITAB) { const DynLoss D(P.loss, P.lossCoef, P.lossMass); double lr[2][6]; loss_rows(D, fel, 0.5*(v0 + v1), lr); sl = fmax(lr[0][0], lr[1][0])*(ITAB ? nd.ds : 1.0) + S0; }
            else sl = fmax(P.ct*fel, -P.cr*fel)*(INTEG ? nd.ds : 1.0) + S0;      /* integrated losses: the slack is an energy per interval */
            nd.x[VF] = fel; nd.x[VP] = fpb; nd.x[VS] = sl;
        }
        __syncthreads();
    }

    /* ------------------------------------------------------------------------------------------
     * FAST: the iteration of the kernels with the problem structure compiled in (FULL, constant efficiencies, explicit
     * Runge-Kutta shooting, LDS-resident).  Same algorithm, same formulas, arranged so that a point is looked at once:
     *   fused_pass       evaluation with derivatives + optimality error + condensed stage blocks in one pass over the nodes
     *                    (kkt_pass + assemble of the general path; one reciprocal per bound; the jets never leave the pass);
     *                    the barrier parameter is chosen after the pass, so the gradient side of the blocks is kept as h0 + mu h1
     *   post_direction   slack steps + directional derivative + fraction-to-the-boundary ratios (one reciprocal per bound)
     *   merit_fast       the trial point: its theta, barrier sums and objective become those of the next current point
     *   update_fast      accepted step
     * Anything rare -- wrong inertia, scan breakdown, a rejected first trial point (backtracking, second-order correction) --
     * repeats the iteration on the general path (run()).
     * ---------------------------------------------------------------------------------------- */
    static constexpr bool FAST = full_energy(FULL) && DYN == LOSS_STATIC && !STREAM && !GEN && MSD_PARALLEL_RICCATI;
    static constexpr int HV = 6;      /* gradient side of a stage block: t, b, q, f, p and the slack row */
    double cnt_lam = 0, cnt_z = 0;    /* numbers of constraint and of bound multipliers (fused_pass) */

    __device__ static __forceinline__ double step_to(double x, double alpha, double d) { return fma(alpha, d, x); }

    static constexpr bool STATIC_LDS = SLDS;      /* (the instantiations the pickers choose where the LDS has the room: STATIC_FIELDS) */
    __device__ __forceinline__ void store_static()
    {
        if constexpr (STATIC_LDS) {
#pragma unroll
            for (int j = 0; j < SPT; j++) { const NodeT &nd = n[j]; const int i = nd.i; c.st[i] = nd.ds; c.st[NS + i] = nd.G; c.st[2*NS + i] = nd.sct; c.st[3*NS + i] = nd.scb; c.st[4*NS + i] = nd.ubB; }
            __syncthreads();
        }
    }
    /* WHAT: bit 0 ds, G; 1 sct, scb; 2 ubB */
    template <int WHAT> __device__ __forceinline__ void load_static(int j)
    {
        if constexpr (STATIC_LDS) {
            NodeT &nd = n[j]; const int i = nd.i;
            if (WHAT & 1) { nd.ds = c.st[i]; nd.G = c.st[NS + i]; }
            if (WHAT & 2) { nd.sct = c.st[2*NS + i]; nd.scb = c.st[3*NS + i]; }
            if (WHAT & 4) nd.ubB = c.st[4*NS + i];
        }
    }

    /* (t, b, sqrt(b), Fel) of a point into the exchange arrays */
    __device__ __forceinline__ void publish_fast(const double (&x)[SPT][NV])
    {
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SPT; j++) { const int i = n[j].i; c.xt[i] = x[j][VT]; c.xb[i] = x[j][VB]; c.xs[i] = fsqrt(x[j][VB]); c.xf[i] = x[j][VF]; }
        __syncthreads();
    }

    /* the rows' non-zero gradient entries at (f, sb, sb1), scaled (ocp.py:189-229 with constant efficiencies) */
    struct RowG { double g0f, g0b, g1f, g1b1, g2f, g2b, g3f, g3s, g4f, g4s; };
    __device__ __forceinline__ RowG row_grads_fast(double f, double sb, double sb1, double isb, double isb1) const
    {
        RowG g;
        const double r0 = U.rs[RPW0], r1 = U.rs[RPW1], r2 = U.rs[RACC], r3 = U.rs[RLTR], r4 = U.rs[RLRG];
        g.g0f = r0*sb; g.g0b = r0*(0.5*f*isb);
        g.g1f = r1*sb1; g.g1b1 = r1*(0.5*f*isb1);
        g.g2f = r2; g.g2b = -r2*(0.5*P.sr1*isb + P.sr2);
        g.g3f = -r3*P.ct; g.g3s = r3; g.g4f = r4*P.cr; g.g4s = r4;
        return g;
    }

    /*
     * One pass over the current point (published in the exchange arrays): constraint values and derivatives, optimality error
     * (W&B eq. (5)), condensed stage blocks (W&B eq. (13)) with the gradient side split as h0 + mu h1.  MERIT: also theta, the
     * barrier sums and the objective (first iteration, or after an iteration on the general path); otherwise E keeps the values
     * the accepted trial point left there.
     */
    __device__ __forceinline__ void fused_pass(const bool MERIT, Err &E, double (&h0)[SPT][HV], double (&h1)[SPT][HV], const double dw = 0.0, const bool soc = false)
    {
        const int N = P.N;
        double dual = 0, prim = 0, prim_u = 0, cmax = -INFINITY, cmin = INFINITY, sumlam = 0, sumz = 0, nlam = 0, nz = 0;      /* (nlam, nz: MERIT passes only) */
        double th = 0, damp = 0, obj = 0;
        LogSum lsum;
        double gl[SPT][NV], Hbb_[SPT], Hbq_[SPT], resd[SPT][NR];      /* (resd: local, not the general path's field) */
        const double sc = U.sf/P.objDen;
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<0>();
            load_static<7>(j);
            NodeT &nd = n[j];
            const int i = nd.i;
            double Htt = 0, Hbb = 0, Hbq = 0, Hbf = 0, Hbp = 0, Hqq = 0, Hqf = 0, Hff = 0, Hfp = 0, Hfs = 0, Hpp = 0, Hss = 0;
            double a0[HV] = {0, 0, 0, 0, 0, 0}, a1[HV] = {0, 0, 0, 0, 0, 0};      /* t b q f p s */
            double nHbb = 0, nHbq = 0, nhb0 = 0, nhb1 = 0, out_q = 0, out_t1 = 0, out_b1 = 0, prod = 1.0;
            double tb = 0, tw = 0, Bb = 0, Bw = 0, rt = 0, rb = 0, oa = 0;
#pragma unroll
            for (int k = 0; k < NV; k++) gl[j][k] = 0;
#pragma unroll
            for (int r = 0; r < NR; r++) resd[j][r] = 0;
            if (nd.ival()) {
                const double t = nd.x[VT], b = nd.x[VB], f = nd.x[VF], p = withPn() ? nd.x[VP] : 0.0, s = nd.x[VS];
                const double t1 = c.xt[i + 1], b1 = c.xb[i + 1], sb = c.xs[i], sb1 = c.xs[i + 1];
                const double q = (i > 0) ? c.xf[i - 1] : 0.0;
                Jet tau, bp;
                interval_map<Jet, true>(P, b, f + p, nd.G, nd.ds, tau, bp);
                double cv0_ = t1 - (t + tau.v), cv1_ = b1 - bp.v;
                if constexpr (SOCK) { if (soc) { cv0_ = wf(W_RESC, i); cv1_ = wf(W_RESC + 1, i); } }      /* (second-order correction: the accumulated residuals, merit_fast) */
                const double cv0 = cv0_, cv1 = cv1_;
                tb = tau.g0; tw = tau.g1; Bb = bp.g0; Bw = bp.g1; rt = -cv0; rb = -cv1;
                const double isb = frcp(sb), isb1 = frcp(sb1);
                const RowG g = row_grads_fast(f, sb, sb1, isb, isb1);
                /* rows d(x), scaled (eval_interval) */
                double dv[NR];
                dv[RPW0] = U.rs[RPW0]*f*sb; dv[RPW1] = U.rs[RPW1]*f*sb1;
                dv[RACC] = U.rs[RACC]*(f + p - (P.sr0 + P.sr1*sb + P.sr2*b) - nd.G);
                dv[RLTR] = U.rs[RLTR]*(s - P.ct*f); dv[RLRG] = U.rs[RLRG]*(s + P.cr*f);
                /* objective (obj_grads) */
                double of = sc*nd.ds, oq = 0, off = 0;
                const double os = sc*nd.ds;
                if (i > 0) { of += sc*2e-3*(f - q); oq = -sc*2e-3*(f - q); off = sc*2e-3; }
                /* gradient of the Lagrangian wrt the interval's own variables, and what belongs to the neighbours */
                const double l0 = nd.lam[0], l1 = nd.lam[1];
                const double dyn_w = l0*tw + l1*Bw;
                gl[j][VT] = -l0;
                gl[j][VB] = nd.nu[RPW0]*g.g0b + nd.nu[RACC]*g.g2b - (l0*tb + l1*Bb);
                gl[j][VF] = of + nd.nu[RPW0]*g.g0f + nd.nu[RPW1]*g.g1f + nd.nu[RACC]*g.g2f + nd.nu[RLTR]*g.g3f + nd.nu[RLRG]*g.g4f - dyn_w;
                if (withPn()) gl[j][VP] = nd.nu[RACC]*g.g2f - dyn_w;
                gl[j][VS] = os + nd.nu[RLTR]*g.g3s + nd.nu[RLRG]*g.g4s;
                out_q = oq; out_t1 = l0; out_b1 = nd.nu[RPW1]*g.g1b1 + l1;
                prim = fmax(prim, fmax(nd.sct*fabs(cv0), nd.scb*fabs(cv1)));
                prim_u = fmax(prim_u, fmax(fabs(cv0), fabs(cv1)));
                if (MERIT) th += nd.sct*fabs(cv0) + nd.scb*fabs(cv1);
                sumlam += fabs(l0)*frcp(nd.sct) + fabs(l1)*frcp(nd.scb); if (MERIT) nlam += 2;
                /* Hessian of the Lagrangian: objective, dynamics, power and acceleration rows (assemble) */
                a0[2] = oq; a0[3] = of; a0[5] = os;
                Hff = off;
                if (i > 0) { Hqq = off; Hqf = -off; }
                {
                    const double hbb = -(l0*tau.h00 + l1*bp.h00), hbw = -(l0*tau.h01 + l1*bp.h01), hww = -(l0*tau.h11 + l1*bp.h11);
                    Hbb += hbb; Hbf += hbw; Hff += hww;
                    if (withPn()) { Hbp += hbw; Hfp += hww; Hpp += hww; }
                }
                {
                    const double ib = isb*isb, ib1 = isb1*isb1;
                    Hbf += nd.nu[RPW0]*U.rs[RPW0]*0.5*isb; Hbb += nd.nu[RPW0]*U.rs[RPW0]*(-0.25*f*(ib*isb));
                    nHbq += nd.nu[RPW1]*U.rs[RPW1]*0.5*isb1; nHbb += nd.nu[RPW1]*U.rs[RPW1]*(-0.25*f*(ib1*isb1));
                    Hbb += nd.nu[RACC]*U.rs[RACC]*0.25*P.sr1*(ib*isb);
                }
                /* rows: residuals, slack bounds (one reciprocal each), condensation */
                double Sg[NR], c0[NR], c1[NR];
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    double rd0_ = dv[r] - nd.sg[r];
                    if constexpr (SOCK) { if (soc) rd0_ = wf(W_RESD + r, i); }
                    const double rd_ = rd0_;
                    resd[j][r] = rd_;
                    const double viol = fabs(rd_);
                    prim = fmax(prim, viol); prim_u = fmax(prim_u, viol*U.irs[r]);
                    if (MERIT) th += viol;
                    sumlam += fabs(nd.nu[r]); if (MERIT) nlam += 1;
                    double gsl = -nd.nu[r], S_, g1_;
                    {
                        const double sl = nd.sg[r] - U.dL[r], z = nd.zLs[r], ri = frcp(sl), cp = sl*z;
                        gsl -= z; cmax = fmax(cmax, cp); cmin = fmin(cmin, cp); sumz += z; nz += 1; prod *= sl;
                        S_ = z*ri; g1_ = -ri;
                    }
                    if (r <= RACC) {
                        const double su = U.dU[r] - nd.sg[r], z = nd.zUs[r], ri = frcp(su), cp = su*z;
                        gsl += z; cmax = fmax(cmax, cp); cmin = fmin(cmin, cp); sumz += z; nz += 1; prod *= su;
                        S_ += z*ri; g1_ += ri;
                    } else { g1_ += K_D; if (MERIT) damp += nd.sg[r] - U.dL[r]; }
                    dual = fmax(dual, fabs(gsl));
                    S_ += dw;      /* (inertia correction: delta_w on the slack's diagonal entry, like assemble()) */
                    Sg[r] = S_; c0[r] = S_*rd_; c1[r] = g1_;
                }
                /* h += coef grad(row), H += Sigma grad grad^T over the rows' non-zeros */
                a0[1] += c0[RPW0]*g.g0b + c0[RACC]*g.g2b;                       a1[1] += c1[RPW0]*g.g0b + c1[RACC]*g.g2b;
                a0[3] += c0[RPW0]*g.g0f + c0[RPW1]*g.g1f + c0[RACC]*g.g2f + c0[RLTR]*g.g3f + c0[RLRG]*g.g4f;
                a1[3] += c1[RPW0]*g.g0f + c1[RPW1]*g.g1f + c1[RACC]*g.g2f + c1[RLTR]*g.g3f + c1[RLRG]*g.g4f;
                if (withPn()) { a0[4] += c0[RACC]*g.g2f; a1[4] += c1[RACC]*g.g2f; }
                a0[5] += c0[RLTR]*g.g3s + c0[RLRG]*g.g4s;                        a1[5] += c1[RLTR]*g.g3s + c1[RLRG]*g.g4s;
                nhb0 = c0[RPW1]*g.g1b1; nhb1 = c1[RPW1]*g.g1b1;
                Hbb += Sg[RPW0]*g.g0b*g.g0b + Sg[RACC]*g.g2b*g.g2b;
                Hbf += Sg[RPW0]*g.g0b*g.g0f + Sg[RACC]*g.g2b*g.g2f;
                if (withPn()) Hbp += Sg[RACC]*g.g2b*g.g2f;
                /* own curvature of Fel (S_OA): everything in Hff that is not in Hfp as well -- with both brakes: all but the dynamics' and the acceleration row's */
                oa = off + (Sg[RPW0]*g.g0f*g.g0f + Sg[RPW1]*g.g1f*g.g1f + Sg[RLTR]*g.g3f*g.g3f + Sg[RLRG]*g.g4f*g.g4f);
                Hff += Sg[RPW0]*g.g0f*g.g0f + Sg[RPW1]*g.g1f*g.g1f + Sg[RACC]*g.g2f*g.g2f + Sg[RLTR]*g.g3f*g.g3f + Sg[RLRG]*g.g4f*g.g4f;
                if (withPn()) { Hfp += Sg[RACC]*g.g2f*g.g2f; Hpp += Sg[RACC]*g.g2f*g.g2f; }
                Hfs += Sg[RLTR]*g.g3f*g.g3s + Sg[RLRG]*g.g4f*g.g4s;
                Hss += Sg[RLTR]*g.g3s*g.g3s + Sg[RLRG]*g.g4s*g.g4s;
                nHbq += Sg[RPW1]*g.g1f*g.g1b1; nHbb += Sg[RPW1]*g.g1b1*g.g1b1;
            }
            /* bounds of the node's own variables */
            if (nd.node()) {
                double Sv[NV], g1v[NV];
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    Sv[k] = 0; g1v[k] = 0;
                    if ((k == VP && !withPn()) || !nd.on(k)) continue;
                    {
                        const double sl = nd.x[k] - lbv(k), z = nd.zL[k], ri = frcp(sl), cp = sl*z;
                        gl[j][k] -= z; cmax = fmax(cmax, cp); cmin = fmin(cmin, cp); sumz += z; nz += 1; prod *= sl;
                        Sv[k] = z*ri; g1v[k] = -ri;
                    }
                    if (hasU(k)) {
                        const double su = ubv(j, k) - nd.x[k], z = nd.zU[k], ri = frcp(su), cp = su*z;
                        gl[j][k] += z; cmax = fmax(cmax, cp); cmin = fmin(cmin, cp); sumz += z; nz += 1; prod *= su;
                        Sv[k] += z*ri; g1v[k] += ri;
                    } else { g1v[k] += K_D; if (MERIT) damp += nd.x[k] - lbv(k); }
                    Sv[k] += dw;
                }
                Htt += Sv[VT]; a1[0] += g1v[VT]; Hbb += Sv[VB]; a1[1] += g1v[VB]; Hff += Sv[VF]; a1[3] += g1v[VF]; Hpp += Sv[VP]; a1[4] += g1v[VP];
                Hss += Sv[VS]; a1[5] += g1v[VS];
                if (MERIT) {
                    double xl[NV];
#pragma unroll
                    for (int k = 0; k < NV; k++) xl[k] = nd.x[k];
                    obj += objective_term<false, FULL>(P, nd, xl, (i > 0) ? c.xf[i - 1] : 0.0, U.sf);
                }
                double *sB = c.S + i*S_STRIDE;
                if (nd.ival()) {
                    sB[S_TB] = tb; sB[S_TW] = tw; sB[S_BB] = Bb; sB[S_BW] = Bw; sB[S_RT] = rt; sB[S_RB] = rb;
                    /* the slack variable is eliminated here (pivot Hss), except in the last interval (assemble) */
                    const bool last = i == N - 1;
                    const double is = (Hss > 0) ? frcp(Hss) : NAN;
                    oa += Sv[VF];
                    if (!last) {
                        const double wf = Hfs*is;
                        Hff -= Hfs*wf; a0[3] -= wf*a0[5]; a1[3] -= wf*a1[5]; oa -= Hfs*wf;
                    }
                    sB[S_GFS] = Hfs; sB[S_IS] = is;
                    if (withPn()) { sB[S_OA] = oa; sB[S_OB] = Sv[VP]; }      /* (own curvature of Fpb: its bounds) */
                }
                sB[S_HTT] = Htt; sB[S_HBF] = Hbf; sB[S_HBP] = Hbp; sB[S_HQQ] = Hqq; sB[S_HQF] = Hqf; sB[S_HFF] = Hff; sB[S_HFP] = Hfp; sB[S_HPP] = Hpp;
            }
            Hbb_[j] = Hbb; Hbq_[j] = Hbq;
#pragma unroll
            for (int k = 0; k < HV; k++) { h0[j][k] = a0[k]; h1[j][k] = a1[k]; }
            if (MERIT) lsum.add(prod);
            c.o1[i] = nHbb; c.o2[i] = nHbq; c.o3[i] = nhb0; c.o4[i] = nhb1; c.o5[i] = out_q; c.o6[i] = out_t1; c.o7[i] = out_b1;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<0>();
            const NodeT &nd = n[j];
            if (!nd.node()) continue;
            const int i = nd.i;
            double Hbb = Hbb_[j], Hbq = Hbq_[j];
            if (i > 0) {
                Hbb += c.o1[i - 1]; Hbq += c.o2[i - 1]; h0[j][1] += c.o3[i - 1]; h1[j][1] += c.o4[i - 1];
                gl[j][VT] += c.o6[i - 1]; gl[j][VB] += c.o7[i - 1];
            }
            if (i + 1 < N) gl[j][VF] += c.o5[i + 1];
            double *sB = c.S + i*S_STRIDE;
            sB[S_HBB] = Hbb; sB[S_HBQ] = Hbq;
#pragma unroll
            for (int k = 0; k < NV; k++) if (!(k == VP && !withPn()) && nd.on(k)) dual = fmax(dual, fabs(gl[j][k]));
        }
        /* the row residuals wait for post_direction in five exchange arrays that are free until the next pass: not in registers
         * across the KKT solve */
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            const int i = n[j].i;
            c.o1[i] = resd[j][RPW0]; c.o2[i] = resd[j][RPW1]; c.o3[i] = resd[j][RACC]; c.o4[i] = resd[j][RLTR]; c.o6[i] = resd[j][RLRG];
        }
        double vm[5] = {dual, prim, prim_u, cmax, -cmin};
        block_reduce<5>(vm, OpMax(), c);
        E.dual = uni(vm[0]); E.primal = uni(vm[1]); E.primal_u = uni(vm[2]); E.cmax = uni(vm[3]); E.cmin = -uni(vm[4]);
        if (MERIT) {
            double vs[8] = {sumlam, sumz, nlam, nz, th, lsum.value(), damp, obj};
            block_reduce<8>(vs, OpSum(), c);
            E.sd = uni(fmax(K_SMAX, (vs[0] + vs[1])/fmax(1.0, vs[2] + vs[3]))/K_SMAX);
            E.sc = uni(fmax(K_SMAX, vs[1]/fmax(1.0, vs[3]))/K_SMAX);
            E.theta = uni(vs[4]); E.L = uni(vs[5]); E.D = uni(vs[6]); E.obj = uni(vs[7]);
            cnt_lam = uni(vs[2]); cnt_z = uni(vs[3]);      /* constants of the scenario */
        } else {
            double vs[2] = {sumlam, sumz};
            block_reduce<2>(vs, OpSum(), c);
            E.sd = uni(fmax(K_SMAX, (vs[0] + vs[1])/fmax(1.0, cnt_lam + cnt_z))/K_SMAX);
            E.sc = uni(fmax(K_SMAX, vs[1]/fmax(1.0, cnt_z))/K_SMAX);
        }
    }

    /* gradient side of the stage blocks for the barrier parameter of this iteration */
    __device__ __forceinline__ void finish_blocks(const double (&h0)[SPT][HV], const double (&h1)[SPT][HV], double mu_)
    {
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            const NodeT &nd = n[j];
            if (!nd.node()) continue;
            double *sB = c.S + nd.i*S_STRIDE;
            sB[S_HT] = h0[j][0] + mu_*h1[j][0]; sB[S_HB] = h0[j][1] + mu_*h1[j][1]; sB[S_HQ] = h0[j][2] + mu_*h1[j][2];
            sB[S_HF] = h0[j][3] + mu_*h1[j][3]; sB[S_HP] = h0[j][4] + mu_*h1[j][4];
            if (nd.ival()) sB[S_GS] = h0[j][5] + mu_*h1[j][5];
        }
        __syncthreads();
    }

    /*
     * After the KKT solve: slack steps, directional derivative of the barrier function, step norms, fraction-to-the-boundary
     * ratios.  Per bound one reciprocal, w = 1/(slack z): 1/slack = z w, 1/z = slack w.
     */
    __device__ __forceinline__ void post_direction(double mu_, double tau_, double &gphid, double &dnorm, bool &tiny_step, double &amax, double &adu)
    {
        const int N = P.N;
        finish_direction();
        const double sc = U.sf/P.objDen;
        double gd = 0, dn = 0, rel = -1.0, rp = 0, rd = 0;
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<9>();
            load_static<5>(j);
            NodeT &nd = n[j];
#pragma unroll
            for (int r = 0; r < NR; r++) nd.dsg[r] = 0;
            if (!nd.node()) continue;
            const int i = nd.i;
            Dir d; load_dir(j, d);
            double og[NV] = {0, 0, 0, 0, 0};
            if (nd.ival()) {
                const double f = nd.x[VF], sb = c.xs[i], sb1 = c.xs[i + 1];
                const double q = (i > 0) ? c.xf[i - 1] : 0.0;
                og[VF] = sc*nd.ds; og[VS] = sc*nd.ds;
                if (i > 0) og[VF] += sc*2e-3*(f - q);
                if (i + 1 < N) og[VF] += c.o5[i + 1];       /* d(obj)/dq of the next interval (fused_pass left it there) */
                const double isb = frcp(sb), isb1 = frcp(sb1);
                const RowG g = row_grads_fast(f, sb, sb1, isb, isb1);
                const double db1 = c.S[(i + 1)*S_STRIDE + S_DB];
                nd.dsg[RPW0] = c.o1[i] + (g.g0b*d.dx[VB] + g.g0f*d.dx[VF]);      /* (residuals: fused_pass left them in the exchange arrays) */
                nd.dsg[RPW1] = c.o2[i] + (g.g1f*d.dx[VF] + g.g1b1*db1);
                nd.dsg[RACC] = c.o3[i] + (g.g2b*d.dx[VB] + g.g2f*d.dx[VF] + g.g2f*d.dx[VP]);
                nd.dsg[RLTR] = c.o4[i] + (g.g3f*d.dx[VF] + g.g3s*d.dx[VS]);
                nd.dsg[RLRG] = c.o6[i] + (g.g4f*d.dx[VF] + g.g4s*d.dx[VS]);
            }
#pragma unroll
            for (int k = 0; k < NV; k++) {
                if ((k == VP && !withPn()) || !nd.on(k)) continue;
                const double dk = d.dx[k];
                double gp;
                {
                    const double s = nd.x[k] - lbv(k), z = nd.zL[k], w = frcp(s*z), r = z*w, iz = s*w;
                    gp = -mu_*r; rp = fmax(rp, -dk*r);
                    rd = fmax(rd, -(r*(mu_ - z*dk) - z)*iz);
                }
                if (hasU(k)) {
                    const double s = ubv(j, k) - nd.x[k], z = nd.zU[k], w = frcp(s*z), r = z*w, iz = s*w;
                    gp += mu_*r; rp = fmax(rp, dk*r);
                    rd = fmax(rd, -(r*(mu_ + z*dk) - z)*iz);
                } else gp += K_D*mu_;
                gd += (og[k] + gp)*dk;
                dn = fmax(dn, fabs(dk)); rel = fmax(rel, fabs(dk) - 10*DBL_EPSILON*(1 + fabs(nd.x[k])));
            }
            if (nd.ival()) {
#pragma unroll
                for (int r_ = 0; r_ < NR; r_++) {
                    const double dk = nd.dsg[r_];
                    double gp;
                    {
                        const double s = nd.sg[r_] - U.dL[r_], z = nd.zLs[r_], w = frcp(s*z), r = z*w, iz = s*w;
                        gp = -mu_*r; rp = fmax(rp, -dk*r);
                        rd = fmax(rd, -(r*(mu_ - z*dk) - z)*iz);
                    }
                    if (r_ <= RACC) {
                        const double s = U.dU[r_] - nd.sg[r_], z = nd.zUs[r_], w = frcp(s*z), r = z*w, iz = s*w;
                        gp += mu_*r; rp = fmax(rp, dk*r);
                        rd = fmax(rd, -(r*(mu_ + z*dk) - z)*iz);
                    } else gp += K_D*mu_;
                    gd += gp*dk;
                    dn = fmax(dn, fabs(dk)); rel = fmax(rel, fabs(dk) - 10*DBL_EPSILON*(1 + fabs(nd.sg[r_])));
                }
            }
        }
        double v1[1] = {gd}; block_reduce<1>(v1, OpSum(), c);
        double v2[4] = {dn, rel, rp, rd}; block_reduce<4>(v2, OpMax(), c);
        gphid = uni(v1[0]); dnorm = uni(v2[0]); tiny_step = uni(v2[1]) < 0;
        const double rpm = uni(v2[2]), rdm = uni(v2[3]);
        amax = (rpm > tau_) ? tau_/rpm : 1.0;
        adu = (rdm > tau_) ? tau_/rdm : 1.0;
    }

    /* the point x + alpha d: published, theta / barrier sums / objective into T (which become the next current point's when it is accepted) */
    /* soc_store (SOCK kernels; second-order correction, W&B section 2.4): 1 / 2 -- also c_soc = alpha c_soc + c(x + alpha d) per lane into the work area (fields
     * W_RESC, W_RESD: the general iteration's residual fields, unused by these kernels otherwise), from where the next pass over the current point takes its
     * right-hand sides (fused_pass(..., soc)); 1: c_soc starts as c(x), the residuals the last pass left in the stage blocks (dynamics) and in the exchange
     * arrays (rows) */
    __device__ __forceinline__ void merit_fast(double alpha, double mu_, Err &T, double &phi, bool &ok, const int soc_store = 0)
    {
        double xt[SPT][NV], st[SPT][NR];
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<4>();
            Dir dd; load_dir(j, dd);
#pragma unroll
            for (int k = 0; k < NV; k++) xt[j][k] = step_to(n[j].x[k], alpha, dd.dx[k]);
#pragma unroll
            for (int r = 0; r < NR; r++) st[j][r] = step_to(n[j].sg[r], alpha, n[j].dsg[r]);
        }
        publish_fast(xt);
        double th = 0, damp = 0, bad = 0, obj = 0;
        LogSum lsum;
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<5>();
            load_static<7>(j);
            const NodeT &nd = n[j];
            const int i = nd.i;
            double prod = 1.0;
            if (nd.ival()) {
                const double t = xt[j][VT], b = xt[j][VB], f = xt[j][VF], p = withPn() ? xt[j][VP] : 0.0, s = xt[j][VS];
                const double t1 = c.xt[i + 1], b1 = c.xb[i + 1], sb = c.xs[i], sb1 = c.xs[i + 1];
                double tau, bp;
                interval_map<double, true>(P, b, f + p, nd.G, nd.ds, tau, bp);
                th += nd.sct*fabs(t1 - (t + tau)) + nd.scb*fabs(b1 - bp);
                double dv[NR];
                dv[RPW0] = U.rs[RPW0]*f*sb; dv[RPW1] = U.rs[RPW1]*f*sb1;
                dv[RACC] = U.rs[RACC]*(f + p - (P.sr0 + P.sr1*sb + P.sr2*b) - nd.G);
                dv[RLTR] = U.rs[RLTR]*(s - P.ct*f); dv[RLRG] = U.rs[RLRG]*(s + P.cr*f);
                if constexpr (SOCK) {
                    if (soc_store != 0) {
                        const bool first = soc_store == 1;
                        const double *sB = c.S + i*S_STRIDE;
                        const double o0 = first ? -sB[S_RT] : wf(W_RESC, i), o1 = first ? -sB[S_RB] : wf(W_RESC + 1, i);
                        wf(W_RESC, i) = alpha*o0 + (t1 - (t + tau)); wf(W_RESC + 1, i) = alpha*o1 + (b1 - bp);
                        const double oldr[NR] = {first ? c.o1[i] : wf(W_RESD + RPW0, i), first ? c.o2[i] : wf(W_RESD + RPW1, i), first ? c.o3[i] : wf(W_RESD + RACC, i),
                                                 first ? c.o4[i] : wf(W_RESD + RLTR, i), first ? c.o6[i] : wf(W_RESD + RLRG, i)};
#pragma unroll
                        for (int r = 0; r < NR; r++) wf(W_RESD + r, i) = alpha*oldr[r] + (dv[r] - st[j][r]);
                    }
                }
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    th += fabs(dv[r] - st[j][r]);
                    { const double sl = st[j][r] - U.dL[r]; if (sl <= 0) bad = 1; else prod *= sl; }
                    if (r <= RACC) { const double su = U.dU[r] - st[j][r]; if (su <= 0) bad = 1; else prod *= su; }
                    else damp += st[j][r] - U.dL[r];
                }
            }
            if (nd.node()) {
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    if ((k == VP && !withPn()) || !nd.on(k)) continue;
                    { const double sl = xt[j][k] - lbv(k); if (sl <= 0) bad = 1; else prod *= sl; }
                    if (hasU(k)) { const double su = ubv(j, k) - xt[j][k]; if (su <= 0) bad = 1; else prod *= su; }
                    else damp += xt[j][k] - lbv(k);
                }
                obj += objective_term<false, FULL>(P, nd, xt[j], (i > 0) ? c.xf[i - 1] : 0.0, U.sf);
            }
            lsum.add(prod);
        }
        double v[5] = {th, lsum.value(), damp, obj, bad};
        block_reduce<5>(v, OpSum(), c);
        T.theta = uni(v[0]); T.L = uni(v[1]); T.D = uni(v[2]); T.obj = uni(v[3]);
        phi = T.obj - mu_*T.L + K_D*mu_*T.D;
        ok = (uni(v[4]) == 0.0) && isfinite(T.theta) && isfinite(phi);
    }

    /* accept x + alpha d (the point merit_fast published); multipliers like the general path's update.  The safeguard that keeps
     * Sigma = z/slack within [mu/(kappa_Sigma slack), kappa_Sigma mu/slack] (W&B eq. (16)) is tested on the product z slack; the division it
     * needs when it acts (kappa_Sigma = 1e10: next to never) is left to a second pass that runs only then */
    __device__ __forceinline__ void update_fast(double apr, double adu, double mu_, const double dw = 0.0)
    {
        const double hi = K_SIGMA*mu_, lo = mu_*(1.0/K_SIGMA);
        bool clamp = false;
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<10>();
            load_static<4>(j);
            NodeT &nd = n[j];
            if (!nd.node()) continue;
            Dir dd; load_dir(j, dd);
#pragma unroll
            for (int k = 0; k < NV; k++) {
                if ((k == VP && !withPn()) || !nd.on(k)) continue;
                const double dk = dd.dx[k], xo = nd.x[k], xn = step_to(xo, apr, dk);
                {
                    const double s = xo - lbv(k), z = nd.zL[k], r = frcp(s);
                    const double zn = z + adu*(r*(mu_ - z*dk) - z), cn = zn*(xn - lbv(k));
                    clamp |= !(cn <= hi && cn >= lo);
                    nd.zL[k] = zn;
                }
                if (hasU(k)) {
                    const double s = ubv(j, k) - xo, z = nd.zU[k], r = frcp(s);
                    const double zn = z + adu*(r*(mu_ + z*dk) - z), cn = zn*(ubv(j, k) - xn);
                    clamp |= !(cn <= hi && cn >= lo);
                    nd.zU[k] = zn;
                }
                nd.x[k] = xn;
            }
            if (nd.ival()) {
#pragma unroll
                for (int r_ = 0; r_ < NR; r_++) {
                    const double dk = nd.dsg[r_], so = nd.sg[r_], sn_ = step_to(so, apr, dk);
                    double Sg, gphi;
                    {
                        const double s = so - U.dL[r_], z = nd.zLs[r_], r = frcp(s);
                        Sg = z*r; gphi = -mu_*r;
                        const double zn = z + adu*(r*(mu_ - z*dk) - z), cn = zn*(sn_ - U.dL[r_]);
                        clamp |= !(cn <= hi && cn >= lo);
                        nd.zLs[r_] = zn;
                    }
                    if (r_ <= RACC) {
                        const double s = U.dU[r_] - so, z = nd.zUs[r_], r = frcp(s);
                        Sg += z*r; gphi += mu_*r;
                        const double zn = z + adu*(r*(mu_ + z*dk) - z), cn = zn*(U.dU[r_] - sn_);
                        clamp |= !(cn <= hi && cn >= lo);
                        nd.zUs[r_] = zn;
                    } else gphi += K_D*mu_;
                    /* new inequality multiplier nu+ = Sigma dsigma + grad phi_sigma, at the old point */
                    nd.nu[r_] += apr*((Sg + dw)*dk + gphi - nd.nu[r_]);
                    nd.sg[r_] = sn_;
                }
                nd.lam[0] += apr*(dd.lt - nd.lam[0]); nd.lam[1] += apr*(dd.lb - nd.lam[1]);
            }
        }
        double v[1] = {clamp ? 1.0 : 0.0};
        block_reduce<1>(v, OpMax(), c);
        if (uni(v[0]) != 0.0) {
            /* (every index a compile-time constant: a run-time index into the nodes' fields would move the whole iterate into scratch memory) */
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                NodeT &nd = n[j];
                if (!nd.node()) continue;
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    if ((k == VP && !withPn()) || !nd.on(k)) continue;
                    nd.zL[k] = sigma_clamp(nd.zL[k], mu_, nd.x[k] - lbv(k));
                    if (hasU(k)) nd.zU[k] = sigma_clamp(nd.zU[k], mu_, ubv(j, k) - nd.x[k]);
                }
                if (!nd.ival()) continue;
#pragma unroll
                for (int r_ = 0; r_ < NR; r_++) {
                    nd.zLs[r_] = sigma_clamp(nd.zLs[r_], mu_, nd.sg[r_] - U.dL[r_]);
                    if (r_ <= RACC) nd.zUs[r_] = sigma_clamp(nd.zUs[r_], mu_, U.dU[r_] - nd.sg[r_]);
                }
            }
        }
    }

    /* one solve; startKind: MSD_START_* (ignored with an external guess); returns the status, iters_out = iterations spent.
     * FL: the fused iteration of the FAST kernels (no least-squares multiplier estimate: profile start or primal-dual warm start only);
     * it returns STATUS_GENERAL when something rare asks for the general iteration, and the caller solves the scenario again with FL = false */
    static constexpr int STATUS_GENERAL = -100;
    int why_general = 0;      /* what sent the fused iteration to the general one (DevProb::follow telemetry) */
#include "msd_resto.hpp"

    /* the general iteration hands a scenario whose line search broke down to the restoration phase (STATUS_RESTO, the iterate parked in the
     * work area) and is entered again with `resume` afterwards: the kernels with static loss rows and explicit Runge-Kutta shooting of up to four
     * waves (N <= 511: every BASELINE configuration); the others keep the restart from the other starting point */
#ifndef MSD_RESTO
#define MSD_RESTO 1      /* 0: kernels without the restoration phase (A/B builds) */
#endif
    static constexpr bool FIRST = PART == 1 || PART == 3;
    /* Every family has the phase (round 5: also the dynamic loss table and integrateLosses -- assemble(MODE_RESTO) leaves their couplings with b_{i+1} resp.
     * the running time unfolded and riccati_resto carries them as cross terms, like the oracle's compute_direction).  Where it lives: in the follow-up
     * kernels.  The kernels with the structure of the reference's rolling stock compiled in have LDS-resident follow-up kernels of their own; every other
     * LDS-resident kernel is a first-pass kernel (PART = 1) followed up by the streamed kernel of its family, and a streamed solve is a first-pass kernel
     * plus a follow-up kernel of the same geometry too (msd_api.hip: make_plan) -- the cold paths cost the streamed kernels a quarter of their speed when
     * they sat in the same code object (round 4) */
    static constexpr bool FAMILY_HAS_RESTO = MSD_RESTO && (STREAM || FIRST || PART == 2 || (DYN == LOSS_STATIC && !GEN && NT <= 256));
    static constexpr bool HAS_RESTO = FAMILY_HAS_RESTO && !FIRST;      /* (a first-pass kernel leaves the phase to its follow-up kernel) */
    /* assemble(MODE_RESTO) hands the integrated loss rows' share in the running time to riccati_resto through the work area (fields W_RX), which only the
     * streamed kernels write: an LDS-resident kernel of these families with the phase inside would read fields nobody wrote */
    static_assert(!(loss_integrated(DYN) && HAS_RESTO && !STREAM), "integrated loss rows: the restoration phase lives in the streamed follow-up kernels");
    static constexpr int STATUS_RESTO = -101;

    /* ---- watchdog procedure (IPOPT: IpBacktrackingLineSearch::StartWatchDog / StopWatchDog, FilterLSAcceptor::StartWatchDog / StopWatchDog; the options
     * the reference leaves at their defaults, ocp.py:290: watchdog_shortened_iter_trigger = 10, watchdog_trial_iter_max = 3).  Restated step for step in
     * oracle/ms_oracle.c (solve_core; the header there says what it was restated from).  The general iteration runs it; the fused iteration only counts
     * the shortened iterations and hands the scenario over when the procedure would start.  Reference point: fields W_WD of the work area ---- */
#ifndef MSD_WATCHDOG
#define MSD_WATCHDOG 1
#endif
    static constexpr int WD_TRIGGER_DEFAULT = 10, WD_TRIAL_MAX = 3;
    /* Who runs the procedure: the follow-up kernels and the streamed kernels -- the kernels that hold the cold paths.  With its blocks compiled into the
     * general iteration of the LDS-resident kernels that are some family's hot kernel (dynamic loss table, integrateLosses, the other shooting
     * integrators) the register allocation of their loop fell apart: 591 -> 3 172 spilled registers on the 64 x 2 kernel of the dynamic loss model, 295 k
     * -> 163 k solves/s (profiles/r04).  A first-pass kernel counts the shortened iterations and hands the scenario over when the procedure is due; a
     * complete LDS-resident kernel without a follow-up kernel (PART = 0: those families) only counts -- no watchdog procedure there */
    static constexpr bool WD_FULL = MSD_WATCHDOG && (PART == 2 || (STREAM && PART == 0));
    static constexpr bool WD_HANDOVER = MSD_WATCHDOG && !WD_FULL && (PART == 1 || PART == 3);
    static constexpr bool RESUMABLE = (FAMILY_HAS_RESTO && !(PART == 1 || PART == 3)) || WD_FULL;      /* the general iteration can be entered again (`resume`) */
    /* StopWatchDog, the part outside the iteration (solve_kernel calls it between two entries of run): the watchdog's copy back to the iterate's fields in
     * the work area -- their home (fetch) in the register-resident kernels, the fields themselves in a streamed one */
    __device__ static __noinline__ void wd_restore(double *work, const int tid, const int nt)
    {
        for (int j = 0; j < SPT; j++) {
            const int i = tid + j*nt;
            for (int f = 0; f < W_WD_FIELDS; f++) work[(size_t)f*NS + i] = work[(size_t)(W_WD + f)*NS + i];
        }
    }
    __device__ static __noinline__ void wd_store(double *work, const int tid, const int nt)      /* StartWatchDog: the parked iterate to the watchdog's copy */
    {
        for (int j = 0; j < SPT; j++) {
            const int i = tid + j*nt;
            for (int f = 0; f < W_WD_FIELDS; f++) work[(size_t)(W_WD + f)*NS + i] = work[(size_t)f*NS + i];
        }
    }
    static constexpr int STATUS_WDSTART = -103;     /* the watchdog procedure is due: the iterate is parked, the caller copies it (wd_store) and enters again */
    static constexpr int STATUS_WDSTOP = -102;      /* the watchdog procedure has put its reference point back (in the work area): enter again with `resume` */

    template <bool FL>
    __device__ __forceinline__ int run(const double *scen, const double *guess, const double *dual_in, int startKind, int iter_offset, int &iters_out,
                                       double *z_out, double *lam_out, double *dual_out, double *stats, double *hist, int hist_cap, const bool resume = false)
    {
        const int N = P.N;
        const bool ext = guess != nullptr;
        const bool warm = ext || startKind == MSD_START_PROFILE;
        const double kp = ext ? P.warmPush : K_PUSH;
        const double mu_start = ext ? P.warmMu : K_MU_INIT;
        const unsigned long long cyc0 = __builtin_readcyclecounter();
        if (c.tid == 0) { c.misc[1] = 0.0; c.misc[MISC_FALLBACKS] = 0.0; for (int k = 0; k < PH_SLOTS; k++) c.misc[2 + k] = 0.0; }
        c.tmark = cyc0;
        const double t0 = scen[MSD_SC_T0], tEnd = scen[MSD_SC_TEND], v0sq = scen[MSD_SC_V0SQ], vNsq = scen[MSD_SC_VNSQ];

        /* ---- static data of the nodes: profile (coalesced reads), bounds (ocp.py:175-181, 247-272) ---- */
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<8>();
            NodeT &nd = n[j];
            nd.i = c.tid + j*c.nt;
            nd.bind(work + nd.i);
            resc[j].bind(work + W_RESC*NS + nd.i); resd[j].bind(work + W_RESD*NS + nd.i);
            evs[j].bind(work + W_EV*NS + nd.i); lgs[j].bind(work + W_LG*NS + nd.i);
            if constexpr (!STREAM) {
                /* defined values in the interval derivatives of every node slot.  They are written for intervals only (store_ev) and loaded for every slot
                 * (load_ev, stash): a register read before it was written -- harmless as long as the value went unused, which held for every library up to
                 * the end of round 5, when a header change that compiled to the same code elsewhere made the one-brake follow-up kernels non-deterministic
                 * from the reference's starting point (status -3 at iteration 0; profiles/r05/README.md).  Very likely round 4's unexplained fault of the
                 * 64 x 1 follow-up kernel as well (profiles/r05/follow_64x1_fault.md) */
#pragma unroll
                for (int k = 0; k < 13; k++) evs[j][k] = 0.0;
            }
            const bool ival = nd.i < N, node = nd.i <= N;
            nd.ds = ival ? P.ds[nd.i] : 0.0;
            nd.G = ival ? track_resistance(P, P.grad[nd.i], P.curv[nd.i]) : 0.0;
            const double bm = (nd.i >= 1 && nd.i < N) ? P.bmax[nd.i] : INFINITY;
            unsigned fl = (ival ? F_IVAL : 0u) | (node ? F_NODE : 0u);
            /* fixed variables (lb == ub) are parameters of the NLP: x_0, b_N (fixed_variable_treatment = make_parameter) */
            if (nd.i >= 1 && node && t0 != tEnd) fl |= F_ON_T;
            if (nd.i >= 1 && nd.i < N && P.vminSq != bm) fl |= F_ON_B;
            if (ival && P.fmin != P.fmax) fl |= F_ON_F;
            if (ival && withPn() && P.fminPn != 0.0) fl |= F_ON_P;
            if (ival) fl |= F_ON_S;
            nd.flags = fl;
            nd.ubB = bm + K_BOUND_RELAX*fmax(1.0, fabs(bm));
            if (STREAM && RESUMABLE && !FL && resume) continue;      /* (the iterate is the one the restoration phase / the watchdog procedure left: in the work area, which is where a streamed kernel's fields live) */
            /* cold start (ocp.py:325-339) */
            const double dt = (tEnd - t0)/N, vel0 = (60/3.6)*(60/3.6);
            nd.x[VT] = t0 + dt*nd.i; nd.x[VB] = vel0; nd.x[VF] = 0.5; nd.x[VP] = withPn() ? -0.1 : 0.0; nd.x[VS] = 1;
            if (ext && node) {
                /* layout of z_out (ocp.py:166-272): [Fel,(Fpb),s,t,b] per interval, then t_N, b_N */
                const int nu = 1 + withPn();
                const double *q = guess + (nu + 3)*nd.i;
                if (ival) { nd.x[VF] = q[0]; nd.x[VP] = withPn() ? q[1] : 0.0; nd.x[VS] = q[nu]; nd.x[VT] = q[nu + 1]; nd.x[VB] = q[nu + 2]; }
                else { nd.x[VT] = q[0]; nd.x[VB] = q[1]; }
            }
            if (nd.i == 0) { nd.x[VT] = t0; nd.x[VB] = v0sq; }
            if (nd.i == N) nd.x[VB] = vNsq;
#pragma unroll
            for (int k = 0; k < NV; k++) nd.zL[k] = nd.zU[k] = 0;
#pragma unroll
            for (int r = 0; r < NR; r++) { nd.sg[r] = nd.nu[r] = nd.zLs[r] = nd.zUs[r] = 0; nd.dsg[r] = 0; }
            nd.lam[0] = nd.lam[1] = 0;
            nd.sct = nd.scb = 1;
        }
        double mu = mu_start, tau = fmax(K_TAU_MIN, 1 - mu);
        if (!(RESUMABLE && !FL && resume)) {
        if (!ext && startKind == MSD_START_PROFILE) profile_start(t0, tEnd, v0sq, vNsq);
        Uni u;     /* built in registers (uniform), published to the LDS copy every phase reads */
        u.tlo = t0 - K_BOUND_RELAX*fmax(1.0, fabs(t0)); u.thi = tEnd + K_BOUND_RELAX*fmax(1.0, fabs(tEnd));
        u.blo = P.vminSq - K_BOUND_RELAX*fmax(1.0, fabs(P.vminSq));
        u.flo = P.fmin - K_BOUND_RELAX*fmax(1.0, fabs(P.fmin)); u.fhi = P.fmax + K_BOUND_RELAX*fmax(1.0, fabs(P.fmax));
        u.plo = P.fminPn - K_BOUND_RELAX*fmax(1.0, fabs(P.fminPn)); u.phi = K_BOUND_RELAX;
        u.slo = -K_BOUND_RELAX;
#pragma unroll
        for (int r = 0; r < NR; r++) { u.rowOn[r] = false; u.dL[r] = -INFINITY; u.dU[r] = INFINITY; u.rs[r] = 1.0; u.irs[r] = 1.0; u.rL[r] = u.rU[r] = false; }
        if (hasPower()) { u.rowOn[RPW0] = u.rowOn[RPW1] = true; u.dL[RPW0] = u.dL[RPW1] = -fabs(P.pwL); u.dU[RPW0] = u.dU[RPW1] = fabs(P.pwU); }
        u.rowOn[RACC] = true; u.dL[RACC] = P.accMin; u.dU[RACC] = P.accMax;
        if (energyOpt()) { u.rowOn[RLTR] = u.rowOn[RLRG] = true; u.dL[RLTR] = u.dL[RLRG] = 0; }
        u.sf = 1;

        commit_uniforms(u);

        /* ---- gradient-based scaling at the starting point (nlp_scaling_max_gradient = 100) ---- */
        evaluate_current();
        Ev e[SPT];
#pragma unroll
        for (int j = 0; j < SPT; j++) load_ev1(j, e[j]);
        {
            double gmax = 0, rmax[NR] = {0, 0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                node_fence<8>();
                NodeT &nd = n[j];
                if (nd.ival()) {
                    double of, op, os, oq, off, opp;
                    obj_grads(j, nb_q(j), of, op, os, oq, off, opp);
                    gmax = fmax(gmax, fmax(fabs(of), fmax(fabs(op), fabs(os))));
                    double mb = (nd.i == N - 1) ? 0.0 : 1.0;
                    if (nd.i > 0) mb = fmax(mb, fabs(e[j].Bb));
                    mb = fmax(mb, fabs(e[j].Bw));
                    nd.scb = mb > 100 ? 100/mb : 1;
                    double mt = 1.0;
                    if (nd.i > 0) mt = fmax(mt, fabs(e[j].tb));
                    mt = fmax(mt, fabs(e[j].tw));
                    nd.sct = mt > 100 ? 100/mt : 1;
                    double gb[NR], gf[NR], gp[NR], gs[NR], gb1[NR], gd[NR];
                    row_grads(j, e[j], gb, gf, gp, gs, gb1, gd);
#pragma unroll
                    for (int r = 0; r < NR; r++) {
                        double m = fmax(fmax(fabs(gf[r]), fabs(gd[r])), fmax(fabs(gp[r]), fabs(gs[r])));      /* (gd: the entries at t and t1) */
                        if (nd.i > 0) m = fmax(m, fabs(gb[r]));
                        if (nd.i < N - 1) m = fmax(m, fabs(gb1[r]));
                        rmax[r] = fmax(rmax[r], m);
                    }
                } else if (nd.i == N && !energyOpt()) gmax = fmax(gmax, 1.0/P.objDen);
            }
            double v[6] = {gmax, rmax[0], rmax[1], rmax[2], rmax[3], rmax[4]};
            block_reduce<6>(v, OpMax(), c);
            if (uni(v[0]) > 100) u.sf = uni(100/v[0]);
#pragma unroll
            for (int r = 0; r < NR; r++) if (u.rowOn[r] && uni(v[1 + r]) > 100) u.rs[r] = uni(100/v[1 + r]);
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
            if (!u.rowOn[r]) continue;
            u.dL[r] *= u.rs[r]; u.dU[r] *= u.rs[r]; u.irs[r] = 1.0/u.rs[r];
            u.rL[r] = isfinite(u.dL[r]); u.rU[r] = isfinite(u.dU[r]);
            if (u.rL[r]) u.dL[r] -= K_BOUND_RELAX*fmax(1.0, fabs(u.dL[r]));
            if (u.rU[r]) u.dU[r] += K_BOUND_RELAX*fmax(1.0, fabs(u.dU[r]));
        }
        commit_uniforms(u);

        /* ---- push into the interior, slacks, bound multipliers.  Primal-dual warm start: constraint multipliers as recorded, bound and
         *      slack multipliers too but not below 1e-3 of their central-path value mu/slack at the pushed point (a multiplier that was
         *      zero must be able to grow) ---- */
        const bool dualStart = ext && dual_in != nullptr;
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<8>();
            const double *q = dualStart ? dual_in + (size_t)MSD_DUAL_STRIDE*n[j].i : nullptr;
#pragma unroll
            for (int k = 0; k < NV; k++) {
                if (!n[j].on(k)) continue;
                n[j].x[k] = push_in(n[j].x[k], lbv(k), ubv(j, k), true, hasU(k), kp);
                n[j].zL[k] = warm ? mu_start/(n[j].x[k] - lbv(k)) : 1.0;
                n[j].zU[k] = hasU(k) ? (warm ? mu_start/(ubv(j, k) - n[j].x[k]) : 1.0) : 0.0;
                if (dualStart) { n[j].zL[k] = fmax(q[7 + k], 1e-3*n[j].zL[k]); if (hasU(k)) n[j].zU[k] = fmax(q[12 + k], 1e-3*n[j].zU[k]); }
            }
        }
        evaluate_current();     /* resd = d(x) since the slacks are still zero */
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<8>();
            if (!n[j].ival()) continue;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (!rowOn(r)) continue;
                n[j].sg[r] = push_in(resd[j][r], U.dL[r], U.dU[r], rL(r), rU(r), kp);
                n[j].zLs[r] = rL(r) ? (warm ? mu_start/(n[j].sg[r] - U.dL[r]) : 1.0) : 0.0;
                n[j].zUs[r] = rU(r) ? (warm ? mu_start/(U.dU[r] - n[j].sg[r]) : 1.0) : 0.0;
                resd[j][r] -= n[j].sg[r];
            }
            if (dualStart) {
                const double *q = dual_in + (size_t)MSD_DUAL_STRIDE*n[j].i;
                n[j].lam[0] = q[0]; n[j].lam[1] = q[1];
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    if (!rowOn(r)) continue;
                    n[j].nu[r] = q[2 + r];
                    if (rL(r)) n[j].zLs[r] = fmax(q[17 + r], 1e-3*n[j].zLs[r]);
                    if (rU(r)) n[j].zUs[r] = fmax(q[22 + r], 1e-3*n[j].zUs[r]);
                }
            }
        }


        /* ---- least-squares multiplier estimate (W&B section 3.6) ----
         * (the fused iteration of the first-pass kernels with PART = 3 takes it too -- one KKT solve of the general kind in front of the loop, not
         * inside it -- and so serves the reference's starting point and primal-only warm starts; the kernels for the profile start and the
         * primal-dual warm start, PART = 1, do not carry the code: 1.5 % on config 1) */
        if constexpr (!FL || PART == 3)
        if (!dualStart)
#if MSD_PROFILE_SKIP_LSQ
        if (ext || startKind != MSD_START_PROFILE)
#endif
        {
            const bool ok = direction(MODE_LSQ, 0.0, 0.0);
            double lmax = 0;
            Dir dd[SPT];
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                load_dir(j, dd[j]);
                if (ok && n[j].ival()) {
                    lmax = fmax(lmax, fmax(fabs(dd[j].lt)/n[j].sct, fabs(dd[j].lb)/n[j].scb));
#pragma unroll
                    for (int r = 0; r < NR; r++) if (rowOn(r)) lmax = fmax(lmax, fabs(n[j].dsg[r]));
                }
            }
            double v[1] = {lmax};
            block_reduce<1>(v, OpMax(), c);
            const double lm = uni(v[0]);
            const bool use = ok && lm <= LAM_INIT_MAX && isfinite(lm);
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                NodeT &nd = n[j];
                const bool tk = use && nd.ival();
                nd.lam[0] = tk ? dd[j].lt : 0.0; nd.lam[1] = tk ? dd[j].lb : 0.0;
#pragma unroll
                for (int r = 0; r < NR; r++) { nd.nu[r] = (tk && rowOn(r)) ? nd.dsg[r] : 0.0; nd.dsg[r] = 0; }
            }
        }
        }      /* (start-up; skipped when the iteration is resumed after a restoration phase) */

        int nfilt = 0;
        double delta_last = 0, theta_max = 0, theta_min = 0;
        int status = MSD_STATUS_MAXITER, iter = 0, acc_count = 0, tiny_count = 0;
        /* inertia correction of the fused iteration (W&B Algorithm IC): the pass over the current point runs again with delta_w on the diagonal */
        double ic_dw = 0;
        bool ic_retry = false;
        /* SOCK: second-order correction of the fused iteration: 1 passes over the current point with the accumulated residuals, 2 the Newton step once more
         * after corrections that did not help, backtracking goes on from half the step.  (What a correction keeps between its passes lives in LDS,
         * misc[27..31]: uniform values held in registers across the loop are what the hot path spills first) */
        int soc_mode = 0;
        constexpr int MISC_SOC = 27;
        int n_reg = 0, n_soc = 0, n_back = 0, n_resto = 0, iter_first = 0, forced = 0;
        int wd_short = 0, wd_trial = 0, n_wd = 0, wd_reg_inc = 0;      /* watchdog: successive shortened iterations, trial iterations of a running procedure, procedures started */
        bool in_wd = false, wd_arm = false;      /* (wd_arm: the iterate of this iteration has been copied for the procedure, which starts at its line search) */
        double wd_theta = 0, wd_phi = 0, wd_gphid = 0, wd_delta_last = 0;
        Err E;
        double alpha_pr = 0, alpha_du = 0, dnorm = 0, objv = 0;
        const double mu_floor = fmin(P.tol, 1e-4)/(K_EPS + 1.0);
        bool skip_first = false;      /* watchdog: the line search of this iteration starts from half the maximal step (the procedure has been stopped) */
        if constexpr (RESUMABLE && !FL) {
            if (resume) {
                /* the iterate from the work area (the restoration phase left the new point there, or the watchdog procedure its reference point), the
                 * scalars of the interrupted iteration */
                fetch<H_ALL>();
#pragma unroll
                for (int j = 0; j < SPT; j++) { n[j].sct = wf(W_SC, n[j].i); n[j].scb = wf(W_SC + 1, n[j].i); }
                mu = uni(wf(W_SCAL, SC_MU)); tau = fmax(K_TAU_MIN, 1 - mu);
                iter_first = (int)uni(wf(W_SCAL, SC_ITER)); nfilt = (int)uni(wf(W_SCAL, SC_NFILT));
                theta_max = uni(wf(W_SCAL, SC_THETA_MAX)); theta_min = uni(wf(W_SCAL, SC_THETA_MIN)); delta_last = uni(wf(W_SCAL, SC_DELTA_LAST));
                n_reg = (int)uni(wf(W_SCAL, SC_N_REG)); n_soc = (int)uni(wf(W_SCAL, SC_N_SOC)); n_back = (int)uni(wf(W_SCAL, SC_N_BACK));
                n_resto = (int)uni(wf(W_SCAL, SC_N_RESTO)); forced = (int)uni(wf(W_SCAL, SC_FORCED));
                wd_short = (int)uni(wf(W_SCAL, SC_WD_SHORT)); n_wd = (int)uni(wf(W_SCAL, SC_N_WD)); skip_first = uni(wf(W_SCAL, SC_SKIP_FIRST)) != 0.0; wd_arm = uni(wf(W_SCAL, SC_WD_ARM)) != 0.0;
            }
        }

        if constexpr (FL) store_static();
        int park = 0;                      /* why the iterate is parked at the top of the next pass (STATUS_RESTO) */
        double park_theta = 0, park_phi = 0;
        for (iter = iter_first;; iter++) {
            if constexpr (RESUMABLE && !FL) {
                /* The one place where the general iteration leaves with its iterate parked in the work area, to be entered again (`resume`): for the
                 * restoration phase (the line search of the pass before broke down) and for the watchdog procedure, which is due when ten shortened
                 * iterations have gone by -- solve_kernel then copies the parked iterate (Solver::wd_store), so that no store of the procedure sits inside
                 * this loop: with its blocks in here the follow-up kernel went from 656 to 2 867 spilled registers (profiles/r04) */
                int reason = park;
                if (WD_FULL && reason == 0 && P.wdTrigger > 0 && !in_wd && !wd_arm && !skip_first && wd_short >= P.wdTrigger) reason = STATUS_WDSTART;
                reason = status_uniform(reason);
                if (reason != 0) {
                    __syncthreads();
                    stash<H_ALL>();
#pragma unroll
                    for (int j = 0; j < SPT; j++) { wf(W_SC, n[j].i) = n[j].sct; wf(W_SC + 1, n[j].i) = n[j].scb; }
                    if (c.tid == 0) {
                        wf(W_SCAL, SC_MU) = mu; wf(W_SCAL, SC_THETA) = park_theta; wf(W_SCAL, SC_PHI) = park_phi; wf(W_SCAL, SC_ITER) = iter; wf(W_SCAL, SC_NFILT) = nfilt;
                        wf(W_SCAL, SC_THETA_MAX) = theta_max; wf(W_SCAL, SC_THETA_MIN) = theta_min; wf(W_SCAL, SC_DELTA_LAST) = delta_last;
                        wf(W_SCAL, SC_N_REG) = n_reg; wf(W_SCAL, SC_N_SOC) = n_soc; wf(W_SCAL, SC_N_BACK) = n_back; wf(W_SCAL, SC_N_RESTO) = n_resto + (reason == STATUS_RESTO ? 1 : 0);
                        wf(W_SCAL, SC_FORCED) = 0; wf(W_SCAL, SC_WD_SHORT) = wd_short; wf(W_SCAL, SC_N_WD) = n_wd; wf(W_SCAL, SC_SKIP_FIRST) = 0;
                        wf(W_SCAL, SC_WD_ARM) = reason == STATUS_WDSTART ? 1.0 : 0.0;
                    }
                    __syncthreads();
                    status = reason; break;
                }
            }
            c.mark(PH_OTHER); phase_fence(PH_OTHER);
            double h0[SPT][HV], h1[SPT][HV];
            if constexpr (FL) {
                if (iter == 0 || (SOCK && soc_mode != 0)) {      /* (soc_mode: the exchange arrays hold the rejected trial point) */
                    double xc[SPT][NV];
#pragma unroll
                    for (int j = 0; j < SPT; j++)
#pragma unroll
                        for (int k = 0; k < NV; k++) xc[j][k] = n[j].x[k];
                    publish_fast(xc);
                }
                fused_pass(iter == 0, E, h0, h1, ic_dw, SOCK && soc_mode == 1);      /* later iterations: theta, barrier sums and objective are the accepted trial point's */
                c.mark(PH_KKT); phase_fence(PH_KKT);
            } else {
                if (iter > 0 || (RESUMABLE && !FL && resume)) evaluate_current();
                c.mark(PH_EVAL); phase_fence(PH_EVAL);
                kkt_pass(E);
                c.mark(PH_KKT); phase_fence(PH_KKT);
            }
            objv = E.obj/U.sf;
            if (iter == 0) { theta_max = 1e4*fmax(1.0, E.theta); theta_min = 1e-4*fmax(1.0, E.theta); }
            if (HAS_RESTO && !FL && forced != 0) { status = forced; break; }      /* the restoration phase ended the solve: outputs at the point it left */
            if (hist && c.tid == 0 && iter < hist_cap) {
                double *hh = hist + HIST_COLS*iter;
                hh[0] = iter; hh[1] = objv; hh[2] = E.primal; hh[3] = E.dual; hh[4] = log10(mu); hh[5] = dnorm; hh[6] = alpha_du; hh[7] = alpha_pr;
            }
            const bool soc_pass = SOCK && FL && soc_mode != 0;      /* (the same point once more, with other residuals: no tests, no barrier update) */
            const double E0 = total_err(E, 0.0);
            const double dual_u = E.dual/U.sf, compl_u = compl_err(E, 0.0)/U.sf;
            if (!soc_pass && E0 <= P.tol && dual_u <= 1.0 && E.primal_u <= 1e-4 && compl_u <= 1e-4) { status = MSD_STATUS_SOLVED; break; }
            /* (acc_now: the current point meets the acceptable tolerances -- where the line search then finds no step the solve ends with
             *  Solved_To_Acceptable_Level, IPOPT's "Restoration phase called at acceptable point", instead of breaking down: below) */
            const bool acc_now = E0 <= ACC_TOL && dual_u <= 1e10 && E.primal_u <= 1e-2 && compl_u <= 1e-2;
            if (soc_pass) { }
            else if (acc_now) { if (!(FL && ic_retry) && ++acc_count >= ACC_ITER) { status = MSD_STATUS_ACCEPTABLE; break; } }
            else acc_count = 0;
            if (!soc_pass && iter >= P.maxIter) { status = MSD_STATUS_MAXITER; break; }
            if (!soc_pass && !isfinite(E0)) { status = MSD_STATUS_NUMERIC; break; }

            /* barrier parameter (monotone, W&B eq. (7)); E_mu differs from E_0 only in the complementarity part */
            {
                bool changed = false;
                while (!(FL && ic_retry) && !soc_pass && total_err(E, mu) <= K_EPS*mu && mu > mu_floor) {      /* (ic_retry, soc_pass: the same point once more, below) */
                    const double nm = fmax(mu_floor, fmin(K_MU_LIN*mu, mu*sqrt(mu)));      /* mu^theta_mu with theta_mu = K_MU_SUP = 1.5 */
                    if (nm >= mu) break;
                    mu = nm; tau = fmax(K_TAU_MIN, 1 - mu); changed = true;
                }
                if (changed) { nfilt = 0; in_wd = false; wd_short = 0; }      /* (a new barrier problem: filter and watchdog start afresh) */
            }
            double theta = E.theta, phi = E.obj - mu*E.L + K_D*mu*E.D;

            if constexpr (FL) {
                c.mark(PH_OTHER); phase_fence(PH_OTHER);
                finish_blocks(h0, h1, mu);
                c.mark(PH_ASSEMBLE); phase_fence(PH_ASSEMBLE);
                const int par = ParallelRiccati<SPT, DYN>::solve(P.N, withPn(), c);
                c.red_slot++;
                c.mark(PH_RICCATI); phase_fence(PH_RICCATI);
                if (par == 0) {
                    /* wrong inertia (a pivot of the stage recursion is not positive): delta_w on the diagonal and the pass over this point again -- W&B
                     * Algorithm IC, the schedule of the general iteration below.  Round 5: warm-started re-solves of config 4 meet it in 3 % of the
                     * solves; they used to be solved again from scratch by the follow-up kernel (40 % of the loop's device time) */
                    if (!ic_retry) { n_reg++; ic_dw = (delta_last == 0) ? DW_0 : fmax(DW_MIN, KW_MINUS*delta_last); }
                    else ic_dw *= (delta_last == 0) ? KW_PLUS_BAR : KW_PLUS;
                    if (ic_dw <= DW_MAX) { ic_retry = true; iter--; continue; }
                }
                if (par != 1) { status = STATUS_GENERAL; why_general = 1; break; }      /* scan breakdown, or no delta_w up to DW_MAX gives the inertia */
                if (ic_dw > 0) delta_last = ic_dw;
                double gphid, amax;
                bool tiny_step;
                post_direction(mu, tau, gphid, dnorm, tiny_step, amax, alpha_du);
                c.mark(PH_GPHID); phase_fence(PH_GPHID);
                if (tiny_step) { status = STATUS_GENERAL; why_general = 2; break; }
                if (MSD_WATCHDOG && P.wdTrigger > 0 && wd_short >= P.wdTrigger) { status = STATUS_GENERAL; why_general = 6; break; }      /* the watchdog procedure would start here */
                tiny_count = 0;
                double amin = G_THETA;
                if (gphid < 0) {
                    amin = fmin(amin, G_PHI*theta/(-gphid));
                    if (theta <= theta_min) amin = fmin(amin, K_DELTA*hpow(theta, S_THETA)/hpow(-gphid, S_PHI));
                }
                amin *= ALPHA_MIN_FRAC;
                if constexpr (SOCK) {
                double alpha = amax;
                bool accepted = false, ftype_armijo = false, soc_go = false;
                Err T = E;
                int ls = 0, store = 0;
                const bool soc_trial = soc_mode == 1;      /* this pass's step is a corrected one: tested like the step it corrects (its alpha, its directional derivative) */
                bool resume_half = false;
                if (soc_mode == 2) { soc_mode = 0; alpha = 0.5*amax; ls = 1; n_back++; resume_half = alpha < amin; }
                if (!resume_half)
                for (;;) {
                    double ph_t; bool okt;
                    merit_fast(alpha, mu, T, ph_t, okt, store);
                    if (store != 0) { soc_go = true; break; }      /* (the trial point once more, for its residuals: the corrected step comes with the next pass) */
                    if (soc_trial) {
                        n_soc++;
                        const double soc_alpha0 = uni(c.misc[MISC_SOC]), soc_gphid = uni(c.misc[MISC_SOC + 1]), soc_thprev = uni(c.misc[MISC_SOC + 2]);
                        const int soc_count = (int)uni(c.misc[MISC_SOC + 3]) + 1;
                        const bool soc_ftype = uni(c.misc[MISC_SOC + 4]) != 0.0;
                        if (acceptable(okt, T.theta, ph_t, theta, phi, soc_alpha0, soc_gphid, soc_ftype, theta_max, theta_min, nfilt)) {
                            accepted = true; ftype_armijo = soc_ftype && cmp_le(ph_t - phi, ETA_PHI*soc_alpha0*soc_gphid, phi);
                            soc_mode = 0;
                            break;
                        }
                        if (okt && T.theta <= K_SOC*soc_thprev && soc_count < P_MAX_SOC) {
                            __syncthreads();
                            if (c.tid == 0) { c.misc[MISC_SOC + 2] = T.theta; c.misc[MISC_SOC + 3] = (double)soc_count; }
                            __syncthreads();
                            store = 2; continue;
                        }
                        soc_mode = 2; soc_go = true;      /* no help: the Newton step once more, backtracking goes on */
                        break;
                    }
                    bool ftype = false;
                    if (gphid < 0) {
                        double lhs = alpha*hpow(-gphid, S_PHI), rhs = K_DELTA*hpow(theta, S_THETA);
                        if (fabs(lhs - rhs) <= 1e-4*fmax(lhs, rhs)) { lhs = alpha*pow(-gphid, S_PHI); rhs = K_DELTA*pow(theta, S_THETA); }
                        ftype = lhs > rhs;
                    }
                    if (acceptable(okt, T.theta, ph_t, theta, phi, alpha, gphid, ftype, theta_max, theta_min, nfilt)) {
                        accepted = true; ftype_armijo = ftype && cmp_le(ph_t - phi, ETA_PHI*alpha*gphid, phi);
                        break;
                    }
                    /* second-order correction: the trial point evaluated once more to leave its residuals (store), then fused_pass with the accumulated
                     * residuals as right-hand sides, the KKT solve, and the corrected step through this same line search -- one call site each */
                    if (ls == 0 && okt && T.theta >= theta) {
                        __syncthreads();
                        if (c.tid == 0) { c.misc[MISC_SOC] = alpha; c.misc[MISC_SOC + 1] = gphid; c.misc[MISC_SOC + 2] = T.theta; c.misc[MISC_SOC + 3] = 0.0; c.misc[MISC_SOC + 4] = ftype ? 1.0 : 0.0; }
                        __syncthreads();
                        soc_mode = 1; store = 1; continue;
                    }
                    alpha *= 0.5; n_back++; ls++;
                    if (alpha < amin) break;
                }
                if (soc_go) iter--;      /* (the same point once more: the loop's own back-edge) */
                else {
                if (!accepted && acc_now) { status = MSD_STATUS_ACCEPTABLE; break; }
                if (!accepted) { status = (HAS_RESTO && P.resto) ? STATUS_GENERAL : MSD_STATUS_LINESEARCH; why_general = 4; break; }
                alpha_pr = alpha;
                if (ls == 0) wd_short = 0; else if (ls > 1) wd_short++;
                c.mark(PH_MERIT); phase_fence(PH_MERIT);
                if (!ftype_armijo && nfilt < FILT_CAP) {
                    __syncthreads();
                    if (c.tid == 0) { c.filt[2*nfilt] = (1 - G_THETA)*theta; c.filt[2*nfilt + 1] = phi - G_PHI*theta; }
                    nfilt++;
                    __syncthreads();
                }
                update_fast(alpha_pr, alpha_du, mu, ic_dw);
                ic_dw = 0; ic_retry = false;
                E.theta = T.theta; E.L = T.L; E.D = T.D; E.obj = T.obj;
                }
                c.mark(PH_UPDATE); phase_fence(PH_UPDATE);
                } else {
                double alpha = amax;
                bool accepted = false, ftype_armijo = false, general = false;
                Err T = E;
                int ls = 0;
                for (;; ls++) {
                    double ph_t; bool okt;
                    merit_fast(alpha, mu, T, ph_t, okt);
                    bool ftype = false;
                    if (gphid < 0) {
                        double lhs = alpha*hpow(-gphid, S_PHI), rhs = K_DELTA*hpow(theta, S_THETA);
                        if (fabs(lhs - rhs) <= 1e-4*fmax(lhs, rhs)) { lhs = alpha*pow(-gphid, S_PHI); rhs = K_DELTA*pow(theta, S_THETA); }
                        ftype = lhs > rhs;
                    }
                    if (acceptable(okt, T.theta, ph_t, theta, phi, alpha, gphid, ftype, theta_max, theta_min, nfilt)) {
                        accepted = true; ftype_armijo = ftype && cmp_le(ph_t - phi, ETA_PHI*alpha*gphid, phi);
                        break;
                    }
                    /* second-order correction (W&B section 2.4): the follow-up kernel's general iteration.  (Round 4 tried it in here, as a cold
                     * block made of the general iteration's pieces: same iterates as the oracle, but the block's register demand reaches into the
                     * loop's allocation -- 594 instead of 256 spilled registers, 779 k instead of 1 044 k solves/s on config 1; profiles/r04) */
                    if (ls == 0 && okt && T.theta >= theta) { general = true; break; }
                    alpha *= 0.5; n_back++;
                    if (alpha < amin) break;
                }
                if (general) { status = STATUS_GENERAL; why_general = 3; break; }
                if (!accepted && acc_now) { status = MSD_STATUS_ACCEPTABLE; break; }      /* (no step from an acceptable point: IPOPT's ACCEPTABLE_POINT_REACHED) */
                if (!accepted) { status = (HAS_RESTO && P.resto) ? STATUS_GENERAL : MSD_STATUS_LINESEARCH; why_general = 4; break; }      /* (the general iteration has the restoration phase) */
                alpha_pr = alpha;
                if (ls == 0) wd_short = 0; else if (ls > 1) wd_short++;      /* (shortened iterations: what starts the watchdog) */
                c.mark(PH_MERIT); phase_fence(PH_MERIT);
                if (!ftype_armijo && nfilt < FILT_CAP) {      /* filter augmentation (W&B eq. (22)) */
                    __syncthreads();
                    if (c.tid == 0) { c.filt[2*nfilt] = (1 - G_THETA)*theta; c.filt[2*nfilt + 1] = phi - G_PHI*theta; }
                    nfilt++;
                    __syncthreads();
                }
                update_fast(alpha_pr, alpha_du, mu, ic_dw);
                ic_dw = 0; ic_retry = false;
                E.theta = T.theta; E.L = T.L; E.D = T.D; E.obj = T.obj;
                c.mark(PH_UPDATE); phase_fence(PH_UPDATE);
                }
            } else {

            /* search direction with inertia correction (W&B Algorithm IC); one call site */
            const double delta_last_in = delta_last;      /* (inertia history as this iteration found it: what a repeat of the iteration has to start from) */
            const int n_reg_in = n_reg;
            double dw = 0;
            bool ok;
            for (bool first = true;; first = false) {
                ok = direction(MODE_NEWTON, mu, dw);
                if (ok) break;
                if (first) { n_reg++; dw = (delta_last == 0) ? DW_0 : fmax(DW_MIN, KW_MINUS*delta_last); }
                else dw *= (delta_last == 0) ? KW_PLUS_BAR : KW_PLUS;
                if (dw > DW_MAX) break;
            }
            if (!ok) { status = MSD_STATUS_REGULARIZATION; break; }
            if (dw > 0) delta_last = dw;

            /* directional derivative of the barrier function, step norms */
            double gphid = 0, amax = 1.0, alpha = 0, th_ref = theta, ph_ref = phi, gd_ref = 0;
            bool tiny = false, accepted = false, ftype_armijo = false, wd_forced = false, wd_stop = false;
            int ls = 0;
            bool tiny_step;
            {
                double gd = 0, dn = 0, rel = -1.0, rp = 0, rd = 0;
                double og[SPT][NV];
#pragma unroll
                for (int j = 0; j < SPT; j++) {
                    node_fence<9>();
                    double of = 0, op = 0, os = 0, oq = 0, off = 0, opp = 0;
                    if (n[j].node()) obj_grads(j, nb_q(j), of, op, os, oq, off, opp);
                    og[j][VT] = (n[j].i == N && !energyOpt()) ? U.sf/P.objDen : 0.0; og[j][VB] = 0; og[j][VF] = of; og[j][VP] = op; og[j][VS] = os;
                    /* d(obj)/dq of the next interval belongs to this node's f */
                    c.o1[n[j].i] = oq;
                }
                __syncthreads();
#pragma unroll
                for (int j = 0; j < SPT; j++) {
                    node_fence<9>();
                    const NodeT &nd = n[j];
                    if (!nd.node()) continue;
                    if (nd.i + 1 < N) og[j][VF] += c.o1[nd.i + 1];
                    Dir dd; load_dir(j, dd);
                    /* per bound: one reciprocal of the slack serves the barrier gradient, the primal ratio -d/s and the dual step
                     * dz = (mu -+ z d)/s - z; fraction to the boundary: alpha = min(1, tau/max ratio) */
#pragma unroll
                    for (int k = 0; k < NV; k++) {
                        if (!nd.on(k)) continue;
                        const double d = dd.dx[k];
                        double gp;
                        {
                            const double r = 1.0/(nd.x[k] - lbv(k)), z = nd.zL[k];
                            gp = -mu*r; rp = fmax(rp, -d*r);
                            rd = fmax(rd, -(r*(mu - z*d) - z)/z);
                        }
                        if (hasU(k)) {
                            const double r = 1.0/(ubv(j, k) - nd.x[k]), z = nd.zU[k];
                            gp += mu*r; rp = fmax(rp, d*r);
                            rd = fmax(rd, -(r*(mu + z*d) - z)/z);
                        } else gp += K_D*mu;
                        gd += (og[j][k] + gp)*d;
                        dn = fmax(dn, fabs(d)); rel = fmax(rel, fabs(d) - 10*DBL_EPSILON*(1 + fabs(nd.x[k])));
                    }
                    if (nd.ival()) {
#pragma unroll
                        for (int r_ = 0; r_ < NR; r_++) {
                            if (!rowOn(r_)) continue;
                            const double d = nd.dsg[r_];
                            double gp = 0;
                            if (rL(r_)) {
                                const double r = 1.0/(nd.sg[r_] - U.dL[r_]), z = nd.zLs[r_];
                                gp -= mu*r; rp = fmax(rp, -d*r);
                                rd = fmax(rd, -(r*(mu - z*d) - z)/z);
                            }
                            if (rU(r_)) {
                                const double r = 1.0/(U.dU[r_] - nd.sg[r_]), z = nd.zUs[r_];
                                gp += mu*r; rp = fmax(rp, d*r);
                                rd = fmax(rd, -(r*(mu + z*d) - z)/z);
                            }
                            if (rL(r_) && !rU(r_)) gp += K_D*mu;
                            if (!rL(r_) && rU(r_)) gp -= K_D*mu;
                            gd += gp*d;
                            dn = fmax(dn, fabs(d)); rel = fmax(rel, fabs(d) - 10*DBL_EPSILON*(1 + fabs(nd.sg[r_])));
                        }
                    }
                }
                double v1[1] = {gd}; block_reduce<1>(v1, OpSum(), c);
                double v2[4] = {dn, rel, rp, rd}; block_reduce<4>(v2, OpMax(), c);
                gphid = uni(v1[0]); dnorm = uni(v2[0]); tiny_step = uni(v2[1]) < 0;
                const double rpm = uni(v2[2]), rdm = uni(v2[3]);
                amax = (rpm > tau) ? tau/rpm : 1.0;
                alpha_du = (rdm > tau) ? tau/rdm : 1.0;
            }
            c.mark(PH_GPHID); phase_fence(PH_GPHID);

            tiny = tiny_step;
            bool wd_skip = false;      /* how the iteration is entered again after the watchdog procedure has been stopped */
            if (WD_FULL && in_wd && tiny) wd_stop = true;      /* a tiny step ends a running procedure: everything resumes from the stored point */
            if (WD_HANDOVER && P.wdTrigger > 0 && !tiny && wd_short >= P.wdTrigger) { status = STATUS_GENERAL; why_general = 6; break; }      /* the procedure is due: the follow-up kernel's */
            if (WD_FULL && wd_arm) {
                /* StartWatchDog: the iterate of this iteration is in the watchdog's copy already (parked at the top of the loop, copied by the caller).
                 * The search direction is not kept: it is the Newton direction of that iterate, the same numbers when the iteration is entered again */
                wd_arm = false;
                if (P.wdTrigger > 0 && !in_wd && !tiny && wd_short >= P.wdTrigger) {
                    wd_theta = theta; wd_phi = phi; wd_gphid = gphid; wd_delta_last = delta_last_in; wd_reg_inc = n_reg - n_reg_in; wd_trial = 0; in_wd = true; n_wd++;
                }
            }
            alpha = amax; accepted = false;
            if (tiny && !wd_stop) {
                accepted = true;
                if (++tiny_count >= 2 && mu <= mu_floor*(1 + 1e-12)) { status = MSD_STATUS_TINY_STEP; break; }
            } else tiny_count = 0;

            /* reference point of the acceptance tests: the current one, or the watchdog's */
            th_ref = in_wd ? wd_theta : theta; ph_ref = in_wd ? wd_phi : phi; gd_ref = in_wd ? wd_gphid : gphid;
            double amin = G_THETA;
            if (gd_ref < 0) {
                amin = fmin(amin, G_PHI*th_ref/(-gd_ref));
                if (th_ref <= theta_min) amin = fmin(amin, K_DELTA*hpow(th_ref, S_THETA)/hpow(-gd_ref, S_PHI));
            }
            amin *= ALPHA_MIN_FRAC;

            if (skip_first) alpha = 0.5*amax;
            ls = 0;
            bool okt_last = true;
            while (!accepted && !wd_stop) {
                double th_t, ph_t; bool okt;
                merit(alpha, mu, th_t, ph_t, okt);
                okt_last = okt;
                /* switching condition (W&B eq. (19)): single-precision powers decide unless the two sides are within 1e-4 of each other */
                bool ftype = false;
                if (gd_ref < 0) {
                    double lhs = alpha*hpow(-gd_ref, S_PHI), rhs = K_DELTA*hpow(th_ref, S_THETA);
                    if (fabs(lhs - rhs) <= 1e-4*fmax(lhs, rhs)) { lhs = alpha*pow(-gd_ref, S_PHI); rhs = K_DELTA*pow(th_ref, S_THETA); }
                    ftype = lhs > rhs;
                }
                if (acceptable(okt, th_t, ph_t, th_ref, ph_ref, alpha, gd_ref, ftype, theta_max, theta_min, nfilt)) {
                    accepted = true; ftype_armijo = ftype && cmp_le(ph_t - ph_ref, ETA_PHI*alpha*gd_ref, ph_ref);
                    break;
                }
                if (WD_FULL && in_wd) break;      /* only the full step is tried while the watchdog procedure runs */
                /* second-order correction (W&B section 2.4): rare, kept out of the hot path */
                if (ls == 0 && !skip_first && okt && th_t >= th_ref) {
                    double th_prev = th_t, alpha_soc = alpha; int nsoc = 0;
                    while (nsoc < P_MAX_SOC) {
                        /* c_soc = alpha_soc c_soc + c(trial) */
                        double tc[SPT][2], td[SPT][NR];
                        trial_residuals(alpha_soc, tc, td);
#pragma unroll
                        for (int j = 0; j < SPT; j++) {
                            resc[j][0] = alpha_soc*resc[j][0] + tc[j][0]; resc[j][1] = alpha_soc*resc[j][1] + tc[j][1];
#pragma unroll
                            for (int r = 0; r < NR; r++) resd[j][r] = alpha_soc*resd[j][r] + td[j][r];
                        }
                        publish_current();     /* the neighbours' Fel in LDS are those of the trial point */
                        if (!direction(MODE_NEWTON, mu, dw)) break;
                        double adu_soc;
                        step_lengths(mu, tau, alpha_soc, adu_soc);
                        double th_s, ph_s; bool oks;
                        merit(alpha_soc, mu, th_s, ph_s, oks);
                        nsoc++; n_soc++;
                        if (acceptable(oks, th_s, ph_s, th_ref, ph_ref, alpha, gd_ref, ftype, theta_max, theta_min, nfilt)) {
                            accepted = true; ftype_armijo = ftype && cmp_le(ph_s - ph_ref, ETA_PHI*alpha*gd_ref, ph_ref);
                            alpha = alpha_soc; alpha_du = adu_soc;
                            break;
                        }
                        if (!oks || th_s > K_SOC*th_prev) break;
                        th_prev = th_s;
                    }
                    if (accepted) break;
                    /* back to the Newton step: recompute it */
                    evaluate_current();
                    direction(MODE_NEWTON, mu, dw);
                    double apr_dummy;
                    step_lengths(mu, tau, apr_dummy, alpha_du);
                }
                alpha *= 0.5; ls++; n_back++;
                if (alpha < amin) break;
            }
            if (WD_FULL && in_wd && !wd_stop) {
                if (accepted) in_wd = false;      /* the procedure has succeeded: the filter gets the reference point below */
                else {
                    wd_trial++;
                    if (okt_last && wd_trial <= WD_TRIAL_MAX) { accepted = true; wd_forced = true; }      /* taken although the filter does not accept it */
                    else { wd_stop = true; wd_skip = true; }      /* no success: back to the stored point, ordinary line search from half the maximal step */
                }
            }
            if (WD_FULL && wd_stop) {
                /* StopWatchDog: the scalars of this iteration to the work area; the caller copies the stored iterate back (wd_restore) and enters the
                 * iteration again (`resume`), which recomputes what belongs to that point -- the search direction included */
                __syncthreads();
#pragma unroll
                for (int j = 0; j < SPT; j++) { wf(W_SC, n[j].i) = n[j].sct; wf(W_SC + 1, n[j].i) = n[j].scb; }
                if (c.tid == 0) {
                    wf(W_SCAL, SC_MU) = mu; wf(W_SCAL, SC_ITER) = iter; wf(W_SCAL, SC_NFILT) = nfilt;
                    wf(W_SCAL, SC_THETA_MAX) = theta_max; wf(W_SCAL, SC_THETA_MIN) = theta_min; wf(W_SCAL, SC_DELTA_LAST) = wd_delta_last;
                    wf(W_SCAL, SC_N_REG) = n_reg - wd_reg_inc; wf(W_SCAL, SC_N_SOC) = n_soc; wf(W_SCAL, SC_N_BACK) = n_back; wf(W_SCAL, SC_N_RESTO) = n_resto;
                    wf(W_SCAL, SC_FORCED) = 0; wf(W_SCAL, SC_WD_SHORT) = 0; wf(W_SCAL, SC_N_WD) = n_wd; wf(W_SCAL, SC_SKIP_FIRST) = wd_skip ? 1.0 : 0.0; wf(W_SCAL, SC_WD_ARM) = 0;
                }
                __syncthreads();
                status = STATUS_WDSTOP; break;
            }
            if (!accepted && acc_now) { status = MSD_STATUS_ACCEPTABLE; break; }      /* (no step from an acceptable point: IPOPT's ACCEPTABLE_POINT_REACHED; the oracle has the comment) */
            if (!accepted) {
                /* the step became too small: feasibility restoration (IpBacktrackingLineSearch), unless the point is almost feasible
                 * (resto_failure_feasibility_threshold = 100 tol).  The current point enters the filter; the iterate and the scalars
                 * of this iteration go to the work area */
                if (HAS_RESTO && P.resto && E.primal > 1e2*P.tol) {
                    __syncthreads();
                    if (nfilt < FILT_CAP) { if (c.tid == 0) { c.filt[2*nfilt] = (1 - G_THETA)*theta; c.filt[2*nfilt + 1] = phi - G_PHI*theta; } nfilt++; }
                    __syncthreads();
                    park = STATUS_RESTO; park_theta = theta; park_phi = phi;
                    iter--; continue;      /* (to the park site at the top of the loop, in the same iteration) */
                }
                status = MSD_STATUS_LINESEARCH; break;
            }
            alpha_pr = alpha; skip_first = false;
            if (ls == 0) wd_short = 0; else if (ls > 1) wd_short++;      /* (n_steps == 0 / n_steps > 1 of IpBacktrackingLineSearch: shortened iterations) */
            c.mark(PH_MERIT); phase_fence(PH_MERIT);

            /* filter augmentation (W&B eq. (22)), with the reference point of the tests */
            if (!tiny && !wd_forced && !ftype_armijo && nfilt < FILT_CAP) {
                __syncthreads();
                if (c.tid == 0) { c.filt[2*nfilt] = (1 - G_THETA)*th_ref; c.filt[2*nfilt + 1] = ph_ref - G_PHI*th_ref; }
                nfilt++;
                __syncthreads();
            }

            /* accept x + alpha d; multipliers: equalities with the primal step, bounds with alpha_du (of the accepted direction) */
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                node_fence<10>();
                NodeT &nd = n[j];
                if (!nd.node()) continue;
                Dir dd; load_dir(j, dd);
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    if (!nd.on(k)) continue;
                    const double dzl = dzL_var(j, k, mu, dd.dx[k]), dzu = hasU(k) ? dzU_var(j, k, mu, dd.dx[k]) : 0.0;
                    nd.x[k] += alpha_pr*dd.dx[k];
                    nd.zL[k] += alpha_du*dzl;
                    if (hasU(k)) nd.zU[k] += alpha_du*dzu;
                    /* keep Sigma within [mu/(kappa_Sigma s), kappa_Sigma mu/s] (W&B eq. (16)) */
                    { const double s = nd.x[k] - lbv(k); nd.zL[k] = sigma_clamp(nd.zL[k], mu, s); }
                    if (hasU(k)) { const double s = ubv(j, k) - nd.x[k]; nd.zU[k] = sigma_clamp(nd.zU[k], mu, s); }
                }
                if (nd.ival()) {
#pragma unroll
                    for (int r = 0; r < NR; r++) {
                        if (!rowOn(r)) continue;
                        const double dzl = rL(r) ? dzL_row(j, r, mu) : 0.0, dzu = rU(r) ? dzU_row(j, r, mu) : 0.0;
                        /* new inequality multiplier nu+ = (Sigma + delta_w) dsigma + grad phi_sigma, at the old point */
                        double Sg, gphi; row_terms(j, r, mu, Sg, gphi);
                        const double dnu = (Sg + dw)*nd.dsg[r] + gphi - nd.nu[r];
                        nd.sg[r] += alpha_pr*nd.dsg[r];
                        nd.nu[r] += alpha_pr*dnu;
                        if (rL(r)) { nd.zLs[r] += alpha_du*dzl; const double s = nd.sg[r] - U.dL[r]; nd.zLs[r] = sigma_clamp(nd.zLs[r], mu, s); }
                        if (rU(r)) { nd.zUs[r] += alpha_du*dzu; const double s = U.dU[r] - nd.sg[r]; nd.zUs[r] = sigma_clamp(nd.zUs[r], mu, s); }
                    }
                    nd.lam[0] += alpha_pr*(dd.lt - nd.lam[0]); nd.lam[1] += alpha_pr*(dd.lb - nd.lam[1]);
                }
            }
            c.mark(PH_UPDATE); phase_fence(PH_UPDATE);
            }      /* (general iteration) */
        }
        if ((FL || WD_HANDOVER) && status == STATUS_GENERAL) { iters_out = iter; return status; }      /* nothing is written: the general path / the follow-up kernel solves the scenario */
        if (status == STATUS_RESTO || status == STATUS_WDSTOP || status == STATUS_WDSTART) { iters_out = iter; return status; }

        /* ---- the multipliers for a later primal-dual warm start ---- */
        if (dual_out) {
#pragma unroll
            for (int j = 0; j < SPT; j++) {
                node_fence<11>();
                const NodeT &nd = n[j];
                if (!nd.node()) continue;
                double *q = dual_out + (size_t)MSD_DUAL_STRIDE*nd.i;
                q[0] = nd.ival() ? nd.lam[0] : 0.0; q[1] = nd.ival() ? nd.lam[1] : 0.0;
#pragma unroll
                for (int r = 0; r < NR; r++) { q[2 + r] = nd.ival() ? nd.nu[r] : 0.0; q[17 + r] = nd.ival() ? nd.zLs[r] : 0.0; q[22 + r] = nd.ival() ? nd.zUs[r] : 0.0; }
#pragma unroll
                for (int k = 0; k < NV; k++) { q[7 + k] = nd.zL[k]; q[12 + k] = nd.zU[k]; }
            }
        }

        /* ---- outputs: z in the reference's layout (ocp.py:166-272), multipliers in the reference's row order ---- */
        const int stp = 4 + withPn();
#pragma unroll
        for (int j = 0; j < SPT; j++) {
            node_fence<11>();
            const NodeT &nd = n[j];
            if (nd.ival()) {
                double *zi = z_out + stp*nd.i; int k = 0;
                zi[k++] = nd.x[VF]; if (withPn()) zi[k++] = nd.x[VP];
                zi[k++] = nd.x[VS]; zi[k++] = nd.x[VT]; zi[k++] = nd.x[VB];
                if (lam_out) {
                    const int rpi = (hasPower() ? 2 : 0) + 3 + (energyOpt() ? 2 : 0);
                    double *l = lam_out + rpi*nd.i; int m = 0;
                    if (hasPower()) { l[m++] = nd.nu[RPW0]*U.rs[RPW0]/U.sf; l[m++] = nd.nu[RPW1]*U.rs[RPW1]/U.sf; }
                    l[m++] = nd.nu[RACC]*U.rs[RACC]/U.sf;
                    l[m++] = nd.lam[0]/U.sf; l[m++] = nd.lam[1]/U.sf;
                    if (energyOpt()) { l[m++] = nd.nu[RLTR]*U.rs[RLTR]/U.sf; l[m++] = nd.nu[RLRG]*U.rs[RLRG]/U.sf; }
                }
            } else if (nd.i == N) { z_out[stp*N] = nd.x[VT]; z_out[stp*N + 1] = nd.x[VB]; }
        }
        if (c.tid == 0) {
            stats[MSD_ST_STATUS] = status; stats[MSD_ST_ITERS] = iter + iter_offset; stats[MSD_ST_OBJ] = objv;
            stats[MSD_ST_KKT] = total_err(E, 0.0); stats[MSD_ST_MU] = mu; stats[MSD_ST_DUAL_INF] = E.dual/U.sf;
            stats[MSD_ST_CONSTR_VIOL] = E.primal_u; stats[MSD_ST_COMPL] = compl_err(E, 0.0)/U.sf;
            stats[MSD_ST_N_REG] = n_reg; stats[MSD_ST_N_SOC] = n_soc; stats[MSD_ST_N_BACKTRACK] = n_back; stats[MSD_ST_N_RESTO] = n_resto; stats[MSD_ST_N_WATCHDOG] = n_wd;
            stats[MSD_ST_CYC_TOTAL] = (double)(__builtin_readcyclecounter() - cyc0); stats[MSD_ST_CYC_KKT] = c.misc[1]; stats[MSD_ST_N_FALLBACK] = c.misc[MISC_FALLBACKS];
            /* phase telemetry of the logged scenario: the last two rows of the history buffer */
            if (hist && hist_cap >= 4) for (int k = 0; k < PH_SLOTS; k++) hist[HIST_COLS*(hist_cap - 2) + k] = c.misc[2 + k];
        }
        __syncthreads();
        iters_out = iter;
        return status;
    }
};

/* the restoration phase as a function of its own: its code and its registers stay out of the iteration's */
template <int NT, int SPT, int DYN, bool GEN, int FULL>
__device__ __noinline__ int resto_entry(const DevProb *P, Ctx c, double *work, Uni *U, const double *scen, double *hist, int hist_cap)
{
    Solver<NT, SPT, DYN, true, GEN, FULL> r(*P, c, work, *U);
    int nit = 0;
    const int rr = r.restoration(scen, hist, hist_cap, nit);
    /* hand-over to the general iteration: where it continues, and with which status if the solve ends here */
    __syncthreads();
    if (c.tid == 0) {
        double *sc = work + (size_t)W_SCAL*NT*SPT;
        sc[Solver<NT, SPT, DYN, true, GEN, FULL>::SC_ITER] += nit;
        sc[Solver<NT, SPT, DYN, true, GEN, FULL>::SC_FORCED] = rr == 1 ? 0 : rr == -1 ? MSD_STATUS_INFEASIBLE : rr == -2 ? MSD_STATUS_MAXITER : MSD_STATUS_LINESEARCH;
    }
    __syncthreads();
    return rr;
}

/*
 * grid = min(nscen, resident workgroups); block = NT threads (multiple of 64), NT*SPT >= N + 1.
 * Dynamic LDS: lds_doubles(N, NT*SPT) * 8 bytes.  work: gridDim.x * work_doubles(NT*SPT) doubles of device memory, private to
 * the workgroups (the part of the iterate that does not stay in registers between the phases).  WPS = minimum waves per SIMD the register budget is planned for.
 *
 * PART -- a solve as one launch or as two (round 4):
 *   0  everything in one kernel: fused iteration (FAST kernels), general iteration, restoration phase, second attempt
 *   1  the first pass: the fused iteration alone (FAST kernels; otherwise the general iteration without the restoration phase).  A scenario
 *      that needs anything else -- no fused start for it, a rare event that asks for the general iteration, a breakdown -- is appended to the
 *      list P.follow and left to the follow-up kernel; the code and the registers of the cold paths stay out of this kernel's code object
 *   2  the follow-up: general iteration + restoration phase + second attempt, for the scenarios of the list (P.follow) or, without a
 *      list, for the whole batch (launches none of whose scenarios can start fused: the reference's starting point, a primal-only warm start)
 */
template <int NT, int SPT, int WPS, int DYN, bool STREAM = false, bool GEN = false, int FULL = 0, int PART = 0, bool SLDS = false, bool SOCK = false>
__global__ void __launch_bounds__(NT, WPS) solve_kernel(DevProb P, int nscen, const double *scen, const double *overrides, double *z_out, double *lam_out,
                                                       double *stats, double *hist, int hist_cap, double *work)
{
    HIP_DYNAMIC_SHARED(double, lds)
    constexpr int NS = NT*SPT;     /* node slots */
    if (P.list && P.list[0] == 0) return;      /* nothing listed (the usual case): the list header is clear already */
    Ctx c;
    c.tid = threadIdx.x; c.lane = threadIdx.x & 63; c.wave = threadIdx.x >> 6; c.nw = NT/64; c.nt = NT; c.red_slot = 0;
    double *wg_work = work + (STREAM ? stream_doubles(P.N, NS, DYN) : work_doubles(NS))*blockIdx.x;
    if (STREAM) {
        /* long horizons (N > 560): stage blocks and exchange arrays behind the node fields in the workgroup's work area (device
         * memory, L2-resident); LDS keeps the filter, the reduction scratch and the uniform records */
        c.S = wg_work + (size_t)W_FIELDS*NS;
        c.xt = c.S + stage_stride(DYN)*(P.N + 1);
        c.filt = lds;
    } else {
        c.S = lds;
        c.xt = c.S + stage_stride(DYN)*(P.N + 1);
    }
    c.xb = c.xt + NS; c.xf = c.xb + NS;
    c.o1 = c.xf + NS; c.o2 = c.o1 + NS; c.o3 = c.o2 + NS;
    using SolverT = Solver<NT, SPT, DYN, STREAM, GEN, FULL, PART, SLDS, SOCK>;
    constexpr bool FASTK = SolverT::FAST;
    c.xs = c.o4 = c.o5 = c.o6 = c.o7 = nullptr;
    if (FASTK) { c.xs = c.o3 + NS; c.o4 = c.xs + NS; c.o5 = c.o4 + NS; c.o6 = c.o5 + NS; c.o7 = c.o6 + NS; }
    if (!STREAM) c.filt = c.o3 + NS + (FASTK ? (XCH_FAST - XCH_GENERAL)*NS : 0);
    c.red = c.filt + 2*FILT_CAP; c.misc = c.red + ((FASTK && NT == 64) ? 0 : RED_DOUBLES);      /* (a single wave reduces in registers) */
    /* the problem record and the scenario's uniform data live in LDS: phases read what they need (broadcast reads) instead of
     * carrying some eighty uniform values through the whole solve in registers */
    double *loss_lds = c.misc + 32 + CONST_DOUBLES;
    c.st = loss_lds + (DYN != LOSS_STATIC ? LOSS_HEAD_CAP : 0);      /* (lds_doubles: every family with the wide stage blocks has the room) */
    c.pool = c.st;      /* (kernels of the shooting-integrator families have no node constants there) */
    const double *loss_head = P.loss, *loss_coef = nullptr;
    if ((DYN == LOSS_TABLE || DYN == LOSS_INTEGRATED_TABLE) && P.loss) {
        const int len = 13 + ((int)P.loss[11] + 1) + ((int)P.loss[12] + 1);
        loss_coef = P.loss + len;
        if (len <= LOSS_HEAD_CAP) {
            for (int k = threadIdx.x; k < len; k += NT) loss_lds[k] = P.loss[k];      /* (visible behind the barrier in front of the first scenario's record) */
            loss_head = loss_lds;
        }
    }
    DevProb *Pl = reinterpret_cast<DevProb *>(c.misc + 32);
    Uni *Ul = reinterpret_cast<Uni *>(c.misc + 32 + UNI_OFF);
    static_assert(sizeof(DevProb) <= 8*UNI_OFF && sizeof(Uni) <= 8*(CONST_DOUBLES - UNI_OFF), "LDS room for the uniform records");
    const int nz = (4 + P.withPn)*P.N + 2;
    const int rpi = (P.hasPower ? 2 : 0) + 3 + (P.energyOpt ? 2 : 0);
    /* scenarios are pulled from a device-wide counter (zeroed by the launch code): solves differ in their iteration counts, and a
     * workgroup that finishes early takes the next scenario instead of idling behind a static stride (matters once the batch is
     * several times the resident workgroups: configs 2-4).  queue == null: static stride. */
    constexpr int MISC_NEXT = 25, MISC_SPENT = 26;
    const bool listed = P.list != nullptr;      /* the follow-up kernel works off the list of the first pass (any complete kernel can be launched on a list: msd_mpc.hip) */
    for (int turn = 0;; turn++) {
        int sidx, spent0 = -1;
        if (listed) {
            __syncthreads();
            if (c.tid == 0) {
                const int k = atomicAdd(P.list + 1, 1);
                const bool have = k < P.list[0];      /* (complete: the kernel that wrote the list has ended) */
                c.misc[MISC_NEXT] = have ? (double)P.list[FOLLOW_HDR + 2*k] : (double)nscen;
                c.misc[MISC_SPENT] = have ? (double)P.list[FOLLOW_HDR + 2*k + 1] : -1.0;
            }
            __syncthreads();
            sidx = wg_uniform((int)c.misc[MISC_NEXT]); spent0 = wg_uniform((int)c.misc[MISC_SPENT]);
        } else if (P.queue) {
            __syncthreads();
            if (c.tid == 0) c.misc[MISC_NEXT] = (double)atomicAdd(P.queue, 1);
            __syncthreads();
            sidx = wg_uniform((int)c.misc[MISC_NEXT]);     /* scalar: every pointer derived from it stays out of the vector registers */
        } else sidx = blockIdx.x + turn*gridDim.x;
        if (sidx >= nscen) break;
        /* per-scenario rolling stock (uniform over the workgroup) */
        DevProb Ps = P;
        Ps.loss = loss_head; Ps.lossCoef = loss_coef;
        if (overrides) {
            const double *o = overrides + (size_t)MSD_OV_COUNT*sidx;
            Ps.sr0 = o[MSD_OV_SR0]; Ps.sr1 = o[MSD_OV_SR1]; Ps.sr2 = o[MSD_OV_SR2];
            Ps.fmax = o[MSD_OV_F_MAX]; Ps.fmin = o[MSD_OV_F_MIN]; Ps.fminPn = o[MSD_OV_F_MIN_PN];
            Ps.pwU = o[MSD_OV_PW_UPPER]; Ps.pwL = o[MSD_OV_PW_LOWER]; Ps.objDen = o[MSD_OV_OBJ_DEN]; Ps.lossMass = o[MSD_OV_TOTAL_MASS];
        }
        __syncthreads();
        if (c.tid == 0) *Pl = Ps;
        __syncthreads();
        SolverT s(*Pl, c, wg_work, *Ul);
        const double *guess = P.guess ? P.guess + (size_t)P.guessStride*sidx : nullptr;
        if (guess && P.guessStatus && P.guessStatus[(size_t)MSD_ST_COUNT*sidx + MSD_ST_STATUS] < 0) guess = nullptr;
        const double *dual_in = (guess && P.dualIn) ? P.dualIn + (size_t)P.dualInStride*sidx + (size_t)MSD_DUAL_STRIDE*P.dualShift : nullptr;
        double *dual_out = P.dualOut ? P.dualOut + (size_t)MSD_DUAL_STRIDE*(P.N + 1)*sidx : nullptr;
        int startKind = P.start, spent = 0, attempt0 = 0;
        if ((PART == 2 || PART == 0) && spent0 >= 0) {
            /* the first pass broke down on this scenario: its second attempt */
            attempt0 = 1; spent = spent0;
            if (guess) guess = nullptr; else startKind = (startKind == MSD_START_PROFILE) ? MSD_START_REFERENCE : MSD_START_PROFILE;
        }
        /* a solve that breaks down (not: runs out of iterations) is repeated from the other starting point */
#ifndef MSD_MAX_ATTEMPTS
#define MSD_MAX_ATTEMPTS 2          /* (1: diagnostic builds that look at the first attempt alone) */
#endif
#pragma unroll 1
        for (int attempt = attempt0; attempt < MSD_MAX_ATTEMPTS; attempt++) {
            int iters = 0;
            int st = SolverT::STATUS_GENERAL;
            if constexpr (FASTK && PART != 2) {
                /* the fused iteration needs no least-squares multiplier estimate: profile start or primal-dual warm start */
                if (PART == 3 || (guess && dual_in) || (!guess && startKind == MSD_START_PROFILE))
                    st = status_uniform(s.template run<true>(scen + (size_t)MSD_SC_COUNT*sidx, guess, guess ? dual_in : nullptr, startKind, spent, iters, z_out + (size_t)nz*sidx,
                                              lam_out ? lam_out + (size_t)rpi*P.N*sidx : nullptr, dual_out, stats + (size_t)MSD_ST_COUNT*sidx,
                                              (hist && sidx == 0) ? hist : nullptr, hist_cap));
                __syncthreads();
            }
            if constexpr (SolverT::FIRST && !FASTK) {
                /* first pass of a family without a fused iteration: the general one without the restoration phase */
                bool resume = false;
#pragma unroll 1
                for (;;) {      /* (entered again when the watchdog procedure has put its reference point back) */
                    st = status_uniform(s.template run<false>(scen + (size_t)MSD_SC_COUNT*sidx, guess, guess ? dual_in : nullptr, startKind, spent, iters, z_out + (size_t)nz*sidx,
                                               lam_out ? lam_out + (size_t)rpi*P.N*sidx : nullptr, dual_out, stats + (size_t)MSD_ST_COUNT*sidx,
                                               (hist && sidx == 0) ? hist : nullptr, hist_cap, resume));
                    __syncthreads();
                    if constexpr (SolverT::WD_FULL) {
                        if (st == SolverT::STATUS_WDSTART) SolverT::wd_store(wg_work, c.tid, c.nt);
                        else if (st == SolverT::STATUS_WDSTOP) SolverT::wd_restore(wg_work, c.tid, c.nt);
                        else break;
                        __syncthreads();
                        resume = true;
                    } else break;
                }
            }
            if constexpr (SolverT::FIRST) {
                /* done (solved, or out of iterations: no second attempt for that) -- or the follow-up kernel's: the general iteration from the same
                 * starting point (also a line search that broke down where the follow-up kernel has the restoration phase).  A breakdown is
                 * repeated here, from the other starting point, like in the kernels that hold everything */
                if (st >= 0 || st == MSD_STATUS_INFEASIBLE || st == MSD_STATUS_MAXITER) break;
                const bool again = st == SolverT::STATUS_GENERAL || (SolverT::FAMILY_HAS_RESTO && st == MSD_STATUS_LINESEARCH && P.resto);
                if (again) {
                    if (c.tid == 0) {
                        const int k = atomicAdd(P.follow, 1);
                        P.follow[FOLLOW_HDR + 2*k] = sidx; P.follow[FOLLOW_HDR + 2*k + 1] = attempt == 0 ? -1 : spent;
                        atomicAdd(P.follow + FOLLOW_TOTAL, 1);
                        atomicAdd(P.follow + FOLLOW_WHY + (st == SolverT::STATUS_GENERAL ? s.why_general : 4), 1);
                        if (P.socSeen && st == SolverT::STATUS_GENERAL && s.why_general == 3) *P.socSeen = 1;
                    }
                    break;
                }
                if (P.oneAttempt) break;
                if (attempt == 0 && c.tid == 0) atomicAdd(P.follow + FOLLOW_WHY + 5, 1);      /* (telemetry: second attempts made here) */
                spent = iters;
                if (guess) guess = nullptr;
                else startKind = (startKind == MSD_START_PROFILE) ? MSD_START_REFERENCE : MSD_START_PROFILE;
            } else {
            if (st == SolverT::STATUS_GENERAL) {
                /* one call site: a scenario whose line search broke down comes back with STATUS_RESTO, goes through the restoration phase and is
                 * resumed -- with the status the phase ended the solve with, if it did */
                bool resume = false;
#pragma unroll 1
                for (;;) {
                    st = status_uniform(s.template run<false>(scen + (size_t)MSD_SC_COUNT*sidx, guess, guess ? dual_in : nullptr, startKind, spent, iters, z_out + (size_t)nz*sidx,
                                               lam_out ? lam_out + (size_t)rpi*P.N*sidx : nullptr, dual_out, stats + (size_t)MSD_ST_COUNT*sidx,
                                               (hist && sidx == 0) ? hist : nullptr, hist_cap, resume));
                    __syncthreads();
                    if constexpr (SolverT::WD_FULL) {
                        if (st == SolverT::STATUS_WDSTOP) { SolverT::wd_restore(wg_work, c.tid, c.nt); __syncthreads(); resume = true; continue; }      /* the watchdog procedure puts its reference point back */
                        if (st == SolverT::STATUS_WDSTART) { SolverT::wd_store(wg_work, c.tid, c.nt); __syncthreads(); resume = true; continue; }      /* ... takes its copy of the iterate */
                    }
                    if constexpr (SolverT::HAS_RESTO) {
                        if (st != SolverT::STATUS_RESTO) break;
                        resto_entry<NT, SPT, DYN, GEN, FULL>(Pl, c, wg_work, Ul, scen + (size_t)MSD_SC_COUNT*sidx, (hist && sidx == 0) ? hist : nullptr, hist_cap);
                        resume = true;
                    } else break;
                }
            }
            __syncthreads();
            /* no second attempt after a success, a verdict of infeasibility or the iteration limit -- unless the solve ran into that limit after a
             * restoration phase (it can leave the iterate where the original iteration only crawls; the other starting point is the way out) */
            if (st >= 0 || st == MSD_STATUS_INFEASIBLE) break;
            if (st == MSD_STATUS_MAXITER && !(SolverT::HAS_RESTO && stats[(size_t)MSD_ST_COUNT*sidx + MSD_ST_N_RESTO] > 0)) break;
            if (P.oneAttempt) break;
            spent = iters;
            if (guess) guess = nullptr;      /* a warm start that breaks down: once more from the problem's own starting point */
            else startKind = (startKind == MSD_START_PROFILE) ? MSD_START_REFERENCE : MSD_START_PROFILE;
            }
        }
    }
    if (listed) {
        /* the last workgroup to find the list empty clears it for the next launch of the handle (launches of a handle are ordered on its stream) */
        __syncthreads();
        if (c.tid == 0) {
            __threadfence();
            if (atomicAdd(P.list + 2, 1) == (int)gridDim.x - 1) { P.list[0] = 0; P.list[1] = 0; P.list[2] = 0; __threadfence(); }
        }
    }
}

/* one thread per interval: TrainIntegrator.solve (train.py:347-364) with sensitivities */
template <int UNUSED>
__global__ void stage_eval_kernel(DevProb P, int n, const double *b, const double *w, const double *ds, const double *grad, const double *curv, double *out)
{
    int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    Jet tau, bp;
    if (P.integ) interval_map_general<Jet>(P, b[k], w[k], track_resistance(P, grad[k], curv[k]), ds[k], tau, bp);
    else interval_map<Jet>(P, b[k], w[k], track_resistance(P, grad[k], curv[k]), ds[k], tau, bp);
    double *o = out + 12*(size_t)k;
    o[0] = tau.v; o[1] = bp.v; o[2] = tau.g0; o[3] = tau.g1; o[4] = bp.g0; o[5] = bp.g1;
    o[6] = tau.h00; o[7] = tau.h01; o[8] = tau.h11; o[9] = bp.h00; o[10] = bp.h01; o[11] = bp.h11;
}

}  // namespace msd
