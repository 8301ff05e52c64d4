/*
 * msd_kernel.hpp -- device code of the MI355X (gfx950) batched multiple-shooting solver.
 *
 * One workgroup solves one OCP scenario from cold start to convergence in a single launch
 * (persistent over scenarios: workgroups stride through the batch).  Thread i of the workgroup
 * owns shooting node i: its state (t_i, b_i), its controls (Fel_i, Fpb_i, s_i), the slacks and
 * multipliers of interval i -- all in registers for the whole solve.  Per interior-point iteration
 *   (a) every thread integrates its interval (RK4 + first/second sensitivities by forward-mode
 *       jets) and condenses its inequality rows and bounds into a stage block     [parallel over stages]
 *   (b) the stage blocks meet in LDS where the block-tridiagonal KKT system is solved by a
 *       Riccati sweep that exploits the sparsity of the 3-state/3-control stage     [serial over stages]
 *   (c) step lengths, filter line search and updates run again one thread per stage, with
 *       wave-shuffle + LDS reductions for the norms.
 * Nothing but the scenario record, the (shared, L2-resident) track profile and the final z*
 * touches HBM.  All arithmetic is IEEE double, like the reference's CasADi/IPOPT path.
 *
 * What it computes (reference = dkouzoup/ms-eetc):
 *   NLP        mseetc/ocp.py:134-284 (variables, bounds, rows, objective), cold start :325-339
 *   integrator mseetc/train.py:225-277 (ODE), :294-301 (RK4 = casadi.simpleRK), :324-344 (trapezoidal time)
 *   losses     mseetc/train.py:199-216 + mseetc/utils.py:197-220 (static efficiencies -> two linear rows)
 *   NLP solver casadi.nlpsol('ipopt') (ocp.py:290,359): IPOPT's published algorithm (Waechter & Biegler,
 *              Math. Prog. 106(1), 2006) with IPOPT's default options: monotone barrier update, filter line
 *              search with second-order correction, inertia correction, gradient-based scaling.
 *
 * The state of a stage is (t, b, q) with q_i := Fel_{i-1}: the control-smoothing term
 * 1e-3 (Fel_i - Fel_{i-1})^2 (ocp.py:245) and the end-of-interval power row Fel_i sqrt(b_{i+1})
 * (ocp.py:189) then are stage-local, which keeps the KKT system block tridiagonal with 3x3 blocks.
 */
#pragma once

#include <hip/hip_runtime.h>

#include <float.h>
#include <math.h>

#include "../../include/mseetc_hip.h"

namespace msd {

struct DevProb {
    int N, withPn, hasPower, energyOpt, numSteps, numApprox, lossKind, maxIter;
    double sr0, sr1, sr2, g, rho, fmax, fmin, fminPn, pwU, pwL, accMin, accMax, ct, cr, vminSq, objDen, tol;
    const double *ds, *grad, *curv, *bmax;
};

/* IPOPT default option values */
constexpr double K_BOUND_RELAX = 1e-8;
constexpr double K_PUSH = 1e-2, K_FRAC = 1e-2;
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
constexpr int S_STRIDE = 27;    /* stage block: odd stride -> the per-thread writes spread over the banks */
constexpr int FILT_CAP = 64;
constexpr int RED_K = 8, RED_SLOTS = 4, MAX_WAVES = 16;
constexpr int HIST_COLS = 8;

/* stage block slots */
constexpr int S_TB = 0, S_TW = 1, S_BB = 2, S_BW = 3, S_RT = 4, S_RB = 5;
constexpr int S_HTT = 6, S_HBB = 7, S_HBQ = 8, S_HBF = 9, S_HBP = 10, S_HQQ = 11, S_HQF = 12, S_HFF = 13, S_HFP = 14, S_HFS = 15,
              S_HPP = 16, S_HSS = 17, S_HT = 18, S_HB = 19, S_HQ = 20, S_HF = 21, S_HP = 22, S_HS = 23;
constexpr int S_K = 6 /* 9 */, S_KV = 15 /* 3 */, S_PN = 18 /* 6 */, S_PV = 24 /* 3 */;
constexpr int S_DT = 6, S_DB = 7, S_DF = 8, S_DP = 9, S_DS = 10, S_LT = 11, S_LB = 12;

__host__ __device__ __forceinline__ int lds_doubles(int N, int NT)
{
    return S_STRIDE*(N + 1) + 6*NT + 2*FILT_CAP + RED_SLOTS*MAX_WAVES*RED_K + 16;
}

/* ------------------------------------------------------------------------------------------
 * forward-mode second-order jets in (b, w)
 * ---------------------------------------------------------------------------------------- */
struct Jet { double v, g0, g1, h00, h01, h11; };

__device__ __forceinline__ Jet operator+(Jet a, Jet b) { return {a.v + b.v, a.g0 + b.g0, a.g1 + b.g1, a.h00 + b.h00, a.h01 + b.h01, a.h11 + b.h11}; }
__device__ __forceinline__ Jet operator-(Jet a, Jet b) { return {a.v - b.v, a.g0 - b.g0, a.g1 - b.g1, a.h00 - b.h00, a.h01 - b.h01, a.h11 - b.h11}; }
__device__ __forceinline__ Jet operator*(Jet a, double s) { return {a.v*s, a.g0*s, a.g1*s, a.h00*s, a.h01*s, a.h11*s}; }
__device__ __forceinline__ Jet operator*(double s, Jet a) { return a*s; }
__device__ __forceinline__ Jet operator+(Jet a, double c) { a.v += c; return a; }
__device__ __forceinline__ Jet operator+(double c, Jet a) { a.v += c; return a; }
__device__ __forceinline__ Jet operator-(Jet a, double c) { a.v -= c; return a; }
__device__ __forceinline__ Jet chain(Jet a, double F, double f1, double f2)
{
    return {F, f1*a.g0, f1*a.g1, f1*a.h00 + f2*a.g0*a.g0, f1*a.h01 + f2*a.g0*a.g1, f1*a.h11 + f2*a.g1*a.g1};
}
__device__ __forceinline__ Jet xsqrt(Jet a) { double s = sqrt(a.v); return chain(a, s, 0.5/s, -0.25/(a.v*s)); }
__device__ __forceinline__ Jet xrecip(Jet a) { double r = 1.0/a.v; return chain(a, r, -r*r, 2*r*r*r); }
__device__ __forceinline__ double xsqrt(double a) { return sqrt(a); }
__device__ __forceinline__ double xrecip(double a) { return 1.0/a; }
__device__ __forceinline__ Jet make_var(Jet, double v, int k) { return {v, k == 0 ? 1.0 : 0.0, k == 1 ? 1.0 : 0.0, 0, 0, 0}; }
__device__ __forceinline__ double make_var(double, double v, int) { return v; }
__device__ __forceinline__ Jet make_zero(Jet) { return {0, 0, 0, 0, 0, 0}; }
__device__ __forceinline__ double make_zero(double) { return 0.0; }

/* d(b)/d(sigma) on the unit interval (train.py:251-259) */
template <class T> __device__ __forceinline__ T ode_b(const DevProb &P, T b, T w, double G, double ds)
{
    T rr = P.sr0 + (xsqrt(b)*P.sr1 + b*P.sr2);
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
template <class T> __device__ __forceinline__ void interval_map(const DevProb &P, double b0, double w0, double G, double ds, T &tau, T &bplus)
{
    T b = make_var(T(), b0, 0), w = make_var(T(), w0, 1);
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
    int tid, lane, wave, nw, red_slot;
};

struct OpMax { __device__ double operator()(double a, double b) const { return fmax(a, b); } };
struct OpMin { __device__ double operator()(double a, double b) const { return fmin(a, b); } };
struct OpSum { __device__ double operator()(double a, double b) const { return a + b; } };

template <int K, class Op> __device__ __forceinline__ void block_reduce(double (&v)[K], Op op, Ctx &c)
{
#pragma unroll
    for (int k = 0; k < K; k++) {
        double x = v[k];
        for (int off = 32; off >= 1; off >>= 1) x = op(x, __shfl_xor(x, off));
        v[k] = x;
    }
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

__device__ __forceinline__ double push_in(double x, double lb, double ub, bool hasL, bool hasU)
{
    if (hasL && hasU) {
        double pL = fmin(K_PUSH*fmax(1.0, fabs(lb)), K_FRAC*(ub - lb));
        double pU = fmin(K_PUSH*fmax(1.0, fabs(ub)), K_FRAC*(ub - lb));
        if (x < lb + pL) x = lb + pL;
        if (x > ub - pU) x = ub - pU;
    } else if (hasL) { double pL = K_PUSH*fmax(1.0, fabs(lb)); if (x < lb + pL) x = lb + pL; }
    else if (hasU) { double pU = K_PUSH*fmax(1.0, fabs(ub)); if (x > ub - pU) x = ub - pU; }
    return x;
}

__device__ __forceinline__ bool cmp_le(double lhs, double rhs, double basval) { return lhs - rhs <= 10.0*DBL_EPSILON*fabs(basval); }

/* ------------------------------------------------------------------------------------------
 * per-thread (= per shooting node) state
 * ---------------------------------------------------------------------------------------- */
struct Node {
    /* static */
    int i;
    bool ival;                 /* has an interval (i < N)                */
    bool on[NV], hasL[NV], hasU[NV];
    double lb[NV], ub[NV];
    double ds, G, sct, scb;
    /* iterate */
    double x[NV], sg[NR], lam[2], nu[NR], zL[NV], zU[NV], zLs[NR], zUs[NR];
    /* direction */
    double dx[NV], dsg[NR], dlam[2], dnu[NR];
};

struct Rows {                  /* workgroup-uniform row data */
    bool on[NR], hasL[NR], hasU[NR];
    double dL[NR], dU[NR], rs[NR];
};

/* values of the interval functions */
struct Ev {
    double c[2], d[NR];
    double sb, sb1, b1;
    double tb, tw, tbb, tbw, tww, Bb, Bw, Bbb, Bbw, Bww;
};

template <bool DERIV>
__device__ __forceinline__ void eval_interval(const DevProb &P, const Rows &R, const Node &n, const double *x, double t1, double b1, Ev &e)
{
    double b = x[VB], f = x[VF], p = P.withPn ? x[VP] : 0.0, s = x[VS];
    if (DERIV) {
        Jet tau, bp;
        interval_map<Jet>(P, b, f + p, n.G, n.ds, tau, bp);
        e.c[0] = t1 - (x[VT] + tau.v); e.c[1] = b1 - bp.v;
        e.tb = tau.g0; e.tw = tau.g1; e.tbb = tau.h00; e.tbw = tau.h01; e.tww = tau.h11;
        e.Bb = bp.g0; e.Bw = bp.g1; e.Bbb = bp.h00; e.Bbw = bp.h01; e.Bww = bp.h11;
    } else {
        double tau, bp;
        interval_map<double>(P, b, f + p, n.G, n.ds, tau, bp);
        e.c[0] = t1 - (x[VT] + tau); e.c[1] = b1 - bp;
    }
    double sb = sqrt(b), sb1 = sqrt(b1);
    e.sb = sb; e.sb1 = sb1; e.b1 = b1;
    e.d[RPW0] = R.rs[RPW0]*f*sb;                                             /* ocp.py:189 */
    e.d[RPW1] = R.rs[RPW1]*f*sb1;
    e.d[RACC] = R.rs[RACC]*(f + p - (P.sr0 + P.sr1*sb + P.sr2*b) - n.G);     /* ocp.py:199 */
    e.d[RLTR] = R.rs[RLTR]*(s - P.ct*f);                                     /* ocp.py:225 */
    e.d[RLRG] = R.rs[RLRG]*(s + P.cr*f);                                     /* ocp.py:226 */
}

/* objective contribution of node i (interval terms + terminal time), scaled by sf */
__device__ __forceinline__ double objective_term(const DevProb &P, const Node &n, const double *x, double q, double sf)
{
    double J = 0;
    if (n.ival) {
        double f = x[VF], p = P.withPn ? x[VP] : 0.0;
        if (P.energyOpt) {
            J = n.ds*(f + x[VS]);                                             /* ocp.py:223 */
            if (n.i > 0) J += 1e-3*(f - q)*(f - q);                           /* ocp.py:245 */
        } else J = 1e-4*(f*f + p*p);                                          /* ocp.py:150 */
    } else if (n.i == P.N && !P.energyOpt) J = x[VT];
    return sf*J/P.objDen;
}

/* ------------------------------------------------------------------------------------------
 * the serial part: Riccati recursion over the stage blocks in LDS (one thread).
 * Stage i: y = (dt, db, dq | df, dp, ds), next state = F y + r with
 *   dt+ = dt + Tb db + Tw (df + dp) + rt,  db+ = Bb db + Bw (df + dp) + rb,  dq+ = df.
 * The last interval eliminates df through db_N = 0 (b_N is a parameter of the NLP).
 * Returns false when a pivot is not positive (wrong inertia of the KKT matrix).
 * ---------------------------------------------------------------------------------------- */
__device__ __noinline__ bool riccati_solve(const DevProb &P, double *S)
{
    const int N = P.N;
    const bool pn = P.withPn != 0;
    /* terminal value function: only t_N is a free variable of the NLP */
    double Ptt = S[N*S_STRIDE + S_HTT], Ptb = 0, Ptq = 0, Pbb = 0, Pbq = 0, Pqq = 0;
    double pt = S[N*S_STRIDE + S_HT], pb = 0, pq = 0;
    /* kept from the last interval for the multiplier of its eliminated row */
    double LGtf = 0, LGbf = 0, LGqf = 0, LGff = 0, LGfp = 0, LGfs = 0, Lgf = 0;

    for (int i = N - 1; i >= 0; i--) {
        double *s = S + i*S_STRIDE;
        const double Tb = s[S_TB], Tw = s[S_TW], Bb = s[S_BB], Bw = s[S_BW], rt = s[S_RT], rb = s[S_RB];
        const double Htt = s[S_HTT], Hbb = s[S_HBB], Hbq = s[S_HBQ], Hbf = s[S_HBF], Hbp = s[S_HBP], Hqq = s[S_HQQ], Hqf = s[S_HQF],
                     Hff = s[S_HFF], Hfp = s[S_HFP], Hfs = s[S_HFS], Hpp = s[S_HPP], Hss = s[S_HSS];
        const double ht = s[S_HT], hb = s[S_HB], hq = s[S_HQ], hf = s[S_HF], hp = s[S_HP], hs = s[S_HS];
        /* stash the value function of stage i+1 for the forward sweep */
        s[S_PN + 0] = Ptt; s[S_PN + 1] = Ptb; s[S_PN + 2] = Ptq; s[S_PN + 3] = Pbb; s[S_PN + 4] = Pbq; s[S_PN + 5] = Pqq;
        s[S_PV + 0] = pt; s[S_PV + 1] = pb; s[S_PV + 2] = pq;

        /* P r + p */
        const double Prt = Ptt*rt + Ptb*rb + pt, Prb = Ptb*rt + Pbb*rb + pb, Prq = Ptq*rt + Pbq*rb + pq;
        /* M_c = P F[:,c] for the columns b, p, f (column t is P[:,t]) */
        const double Mbt = Tb*Ptt + Bb*Ptb, Mbb = Tb*Ptb + Bb*Pbb, Mbq = Tb*Ptq + Bb*Pbq;
        const double Mpt = Tw*Ptt + Bw*Ptb, Mpb = Tw*Ptb + Bw*Pbb, Mpq = Tw*Ptq + Bw*Pbq;
        const double Mft = Mpt + Ptq, Mfb = Mpb + Pbq, Mfq = Mpq + Pqq;
        /* G = H + F^T P F, g = h + F^T (P r + p) */
        double Gtt = Htt + Ptt, Gtb = Mbt, Gtf = Mft, Gtp = Mpt;
        double Gbb = Hbb + Tb*Mbt + Bb*Mbb, Gbq = Hbq, Gbf = Hbf + Tb*Mft + Bb*Mfb, Gbp = Hbp + Tb*Mpt + Bb*Mpb;
        double Gqq = Hqq, Gqf = Hqf;
        double Gff = Hff + Tw*Mft + Bw*Mfb + Mfq, Gfp = Hfp + Tw*Mpt + Bw*Mpb + Mpq, Gfs = Hfs;
        double Gpp = Hpp + Tw*Mpt + Bw*Mpb, Gss = Hss;
        double gt = ht + Prt, gb = hb + Tb*Prt + Bb*Prb, gq = hq;
        double gf = hf + Tw*Prt + Bw*Prb + Prq, gp = hp + Tw*Prt + Bw*Prb, gs = hs;
        if (!pn) { Gtp = 0; Gbp = 0; Gfp = 0; Gpp = 1; gp = 0; }

        double Kft, Kfb, Kfq, Kpt, Kpb, Kpq, Kst, Ksb, Ksq, kf, kp, ks;
        double nPtt, nPtb, nPtq, nPbb, nPbq, nPqq, npt, npb, npq;

        if (i == N - 1) {
            /* df = eb db - dp + e0 from the b row */
            const double eb = -Bb/Bw, e0 = -rb/Bw;
            LGtf = Gtf; LGbf = Gbf; LGqf = Gqf; LGff = Gff; LGfp = Gfp; LGfs = Gfs; Lgf = gf;
            const double gfe = gf + Gff*e0;
            /* reduced blocks over (t, b, q | p, s) */
            double Hpp2 = Gpp - 2*Gfp + Gff, Hps2 = -Gfs, Hss2 = Gss;
            double Hpt = Gtp - Gtf, Hpb = Gbp + Gfp*eb - Gbf - Gff*eb, Hpq = -Gqf;
            double Hsb = Gfs*eb;
            double gp2 = gp + Gfp*e0 - gfe, gs2 = gs + Gfs*e0;
            double Xtt = Gtt, Xtb = Gtb + Gtf*eb, Xbb = Gbb + 2*eb*Gbf + eb*eb*Gff, Xbq = Gbq + eb*Gqf, Xqq = Gqq;
            double xt = gt + Gtf*e0, xb = gb + Gbf*e0 + eb*gfe, xq = gq + Gqf*e0;
            if (!pn) { Hpp2 = 1; Hps2 = 0; Hpt = 0; Hpb = 0; Hpq = 0; gp2 = 0; }
            /* 2x2 pivots: s first, then p */
            if (!(Hss2 > 0)) return false;
            const double is = 1.0/Hss2, lps = Hps2*is, dp_ = Hpp2 - Hps2*lps;
            if (!(dp_ > 0)) return false;
            const double ip = 1.0/dp_;
            /* columns t, b, q and the vector: rhs = -(row p, row s) */
            double Kp2t = -(Hpt)*ip, Ks2t = -(Hps2*Kp2t)*is;
            double Kp2b = -(Hpb - lps*Hsb)*ip, Ks2b = -(Hsb + Hps2*Kp2b)*is;
            double Kp2q = -(Hpq)*ip, Ks2q = -(Hps2*Kp2q)*is;
            double kp2 = -(gp2 - lps*gs2)*ip, ks2 = -(gs2 + Hps2*kp2)*is;
            /* value function of stage N-1 */
            nPtt = Xtt + Hpt*Kp2t;
            nPtb = Xtb + Hpt*Kp2b;
            nPtq = Hpt*Kp2q;
            nPbb = Xbb + Hpb*Kp2b + Hsb*Ks2b;
            nPbq = Xbq + Hpb*Kp2q + Hsb*Ks2q;
            nPqq = Xqq + Hpq*Kp2q;
            npt = xt + Hpt*kp2; npb = xb + Hpb*kp2 + Hsb*ks2; npq = xq + Hpq*kp2;
            /* uniform feedback form: df = eb db - dp + e0 */
            Kpt = Kp2t; Kpb = Kp2b; Kpq = Kp2q; kp = kp2;
            Kst = Ks2t; Ksb = Ks2b; Ksq = Ks2q; ks = ks2;
            Kft = -Kp2t; Kfb = eb - Kp2b; Kfq = -Kp2q; kf = e0 - kp2;
        } else {
            /* pivots of Guu in the order s, p, f (s and p couple only with f) */
            if (!(Gss > 0) || !(Gpp > 0)) return false;
            const double is = 1.0/Gss, ip = 1.0/Gpp;
            const double lfs = Gfs*is, lfp = Gfp*ip;
            const double df_ = Gff - Gfs*lfs - Gfp*lfp;
            if (!(df_ > 0)) return false;
            const double iff = 1.0/df_;
            /* rhs columns: -(Gfx, Gpx, Gsx); Gsx = 0 */
            Kft = -(Gtf - lfp*Gtp)*iff; Kpt = -(Gtp + Gfp*Kft)*ip; Kst = -(Gfs*Kft)*is;
            Kfb = -(Gbf - lfp*Gbp)*iff; Kpb = -(Gbp + Gfp*Kfb)*ip; Ksb = -(Gfs*Kfb)*is;
            Kfq = -(Gqf)*iff;           Kpq = -(Gfp*Kfq)*ip;       Ksq = -(Gfs*Kfq)*is;
            kf = -(gf - lfs*gs - lfp*gp)*iff; kp = -(gp + Gfp*kf)*ip; ks = -(gs + Gfs*kf)*is;
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
        s[S_K + 6] = Kst; s[S_K + 7] = Ksb; s[S_K + 8] = Ksq; s[S_KV + 0] = kf; s[S_KV + 1] = kp; s[S_KV + 2] = ks;
        Ptt = nPtt; Ptb = nPtb; Ptq = nPtq; Pbb = nPbb; Pbq = nPbq; Pqq = nPqq; pt = npt; pb = npb; pq = npq;
    }

    /* forward sweep; x_0 is a parameter */
    double dt = 0, db = 0, dq = 0;
    for (int i = 0; i < N; i++) {
        double *s = S + i*S_STRIDE;
        const double Tb = s[S_TB], Tw = s[S_TW], Bb = s[S_BB], Bw = s[S_BW], rt = s[S_RT], rb = s[S_RB];
        const double df = s[S_K + 0]*dt + s[S_K + 1]*db + s[S_K + 2]*dq + s[S_KV + 0];
        const double dp = pn ? s[S_K + 3]*dt + s[S_K + 4]*db + s[S_K + 5]*dq + s[S_KV + 1] : 0.0;
        const double dsl = s[S_K + 6]*dt + s[S_K + 7]*db + s[S_K + 8]*dq + s[S_KV + 2];
        const double dw = df + dp;
        const double nt = dt + Tb*db + Tw*dw + rt;
        const double nb = (i == N - 1) ? 0.0 : Bb*db + Bw*dw + rb;
        const double nq = df;
        const double *Pn = s + S_PN, *pv = s + S_PV;
        double lt = -(Pn[0]*nt + Pn[1]*nb + Pn[2]*nq + pv[0]);
        double lb = -(Pn[1]*nt + Pn[3]*nb + Pn[4]*nq + pv[1]);
        if (i == N - 1) lb = (LGtf*dt + LGbf*db + LGqf*dq + LGff*df + LGfp*dp + LGfs*dsl + Lgf)/Bw;
        s[S_DT] = dt; s[S_DB] = db; s[S_DF] = df; s[S_DP] = dp; s[S_DS] = dsl; s[S_LT] = lt; s[S_LB] = lb;
        dt = nt; db = nb; dq = nq;
    }
    S[N*S_STRIDE + S_DT] = dt; S[N*S_STRIDE + S_DB] = 0.0; S[N*S_STRIDE + S_DF] = 0.0;
    return true;
}

/* ------------------------------------------------------------------------------------------
 * the solver
 * ---------------------------------------------------------------------------------------- */
enum { MODE_NEWTON = 0, MODE_LSQ = 1 };

struct Solver {
    const DevProb &P;
    Ctx &c;
    Node n;
    Rows R;
    Ev e;
    double sf, mu, tau;
    double resc[2], resd[NR];     /* right-hand sides of the linearised constraints (c, d - sigma or their SOC accumulation) */

    __device__ Solver(const DevProb &P_, Ctx &c_) : P(P_), c(c_) {}

    /* publish (t, b, f) of a point so that neighbours can read them */
    __device__ __forceinline__ void publish(const double *x)
    {
        __syncthreads();
        c.xt[c.tid] = x[VT]; c.xb[c.tid] = x[VB]; c.xf[c.tid] = x[VF];
        __syncthreads();
    }
    __device__ __forceinline__ double nb_q() const { return (n.i > 0 && n.i <= P.N) ? c.xf[c.tid - 1] : 0.0; }

    __device__ __forceinline__ void row_slack_terms(int r, double mu_, double dw, double &Sg, double &gphi) const
    {
        bar_terms(n.sg[r], R.dL[r], R.dU[r], R.hasL[r], R.hasU[r], n.zLs[r], n.zUs[r], mu_, Sg, gphi);
        Sg += dw;
    }

    /* gradient entries of the rows wrt (b, f, p, s, b1) */
    __device__ __forceinline__ void row_grads(double f, double (&gb)[NR], double (&gf)[NR], double (&gp)[NR], double (&gs)[NR], double (&gb1)[NR]) const
    {
#pragma unroll
        for (int r = 0; r < NR; r++) { gb[r] = gf[r] = gp[r] = gs[r] = gb1[r] = 0; }
        gf[RPW0] = e.sb; gb[RPW0] = 0.5*f/e.sb;
        gf[RPW1] = e.sb1; gb1[RPW1] = 0.5*f/e.sb1;
        gf[RACC] = 1; gp[RACC] = P.withPn ? 1.0 : 0.0; gb[RACC] = -(0.5*P.sr1/e.sb + P.sr2);
        gs[RLTR] = 1; gf[RLTR] = -P.ct;
        gs[RLRG] = 1; gf[RLRG] = P.cr;
#pragma unroll
        for (int r = 0; r < NR; r++) { gb[r] *= R.rs[r]; gf[r] *= R.rs[r]; gp[r] *= R.rs[r]; gs[r] *= R.rs[r]; gb1[r] *= R.rs[r]; }
    }

    /* objective gradient wrt (f, p, s, q) of the interval; terminal time handled by node N */
    __device__ __forceinline__ void obj_grads(double q, double &of, double &op, double &os, double &oq, double &off, double &opp) const
    {
        double sc = sf/P.objDen;
        of = op = os = oq = off = opp = 0;
        if (!n.ival) return;
        double f = n.x[VF], p = P.withPn ? n.x[VP] : 0.0;
        if (P.energyOpt) {
            of = sc*n.ds; os = sc*n.ds;
            if (n.i > 0) { of += sc*2e-3*(f - q); oq = -sc*2e-3*(f - q); off = sc*2e-3; }
        } else {
            of = sc*2e-4*f; off = sc*2e-4;
            if (P.withPn) { op = sc*2e-4*p; opp = sc*2e-4; }
        }
    }

    /*
     * Optimality error of the scaled problem (W&B eq. (5)) with the current evaluation `e`.
     * out: dual, primal (scaled), cmax/cmin of the complementarity products, sum|lam|, sum z, counts, unscaled primal
     */
    struct Err { double dual, primal, primal_u, cmax, cmin, sd, sc; };

    __device__ __noinline__ void kkt_error(Err &E)
    {
        const double q = nb_q();
        double gl[NV] = {0, 0, 0, 0, 0};
        double out_q = 0, out_t1 = 0, out_b1 = 0;
        double dual = 0, prim = 0, prim_u = 0, cmax = -INFINITY, cmin = INFINITY, sumlam = 0, sumz = 0, nlam = 0, nz = 0;
        if (n.ival) {
            double gb[NR], gf[NR], gp[NR], gs[NR], gb1[NR];
            row_grads(n.x[VF], gb, gf, gp, gs, gb1);
            double of, op, os, oq, off, opp;
            obj_grads(q, of, op, os, oq, off, opp);
            gl[VF] = of; gl[VP] = op; gl[VS] = os; out_q = oq;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (!R.on[r]) continue;
                gl[VB] += n.nu[r]*gb[r]; gl[VF] += n.nu[r]*gf[r]; gl[VP] += n.nu[r]*gp[r]; gl[VS] += n.nu[r]*gs[r]; out_b1 += n.nu[r]*gb1[r];
            }
            /* dynamics rows: c_t = t1 - t - tau, c_b = b1 - b+ */
            out_t1 += n.lam[0]; gl[VT] -= n.lam[0];
            gl[VB] -= n.lam[0]*e.tb + n.lam[1]*e.Bb;
            gl[VF] -= n.lam[0]*e.tw + n.lam[1]*e.Bw;
            if (P.withPn) gl[VP] -= n.lam[0]*e.tw + n.lam[1]*e.Bw;
            out_b1 += n.lam[1];
            prim = fmax(n.sct*fabs(e.c[0]), n.scb*fabs(e.c[1]));
            prim_u = fmax(fabs(e.c[0]), fabs(e.c[1]));
            sumlam = fabs(n.lam[0])/n.sct + fabs(n.lam[1])/n.scb; nlam = 2;
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (!R.on[r]) continue;
                double viol = fabs(e.d[r] - n.sg[r]);
                prim = fmax(prim, viol); prim_u = fmax(prim_u, viol/R.rs[r]);
                sumlam += fabs(n.nu[r]); nlam += 1;
                double gsl = -n.nu[r];
                if (R.hasL[r]) { gsl -= n.zLs[r]; double cp = (n.sg[r] - R.dL[r])*n.zLs[r]; cmax = fmax(cmax, cp); cmin = fmin(cmin, cp); sumz += n.zLs[r]; nz += 1; }
                if (R.hasU[r]) { gsl += n.zUs[r]; double cp = (R.dU[r] - n.sg[r])*n.zUs[r]; cmax = fmax(cmax, cp); cmin = fmin(cmin, cp); sumz += n.zUs[r]; nz += 1; }
                dual = fmax(dual, fabs(gsl));
            }
        } else if (n.i == P.N && !P.energyOpt) gl[VT] = sf/P.objDen;
        /* exchange the contributions that belong to the neighbours' variables */
        c.o1[c.tid] = out_q; c.o2[c.tid] = out_t1; c.o3[c.tid] = out_b1;
        __syncthreads();
        if (n.i <= P.N) {
            if (n.i > 0) { gl[VT] += c.o2[c.tid - 1]; gl[VB] += c.o3[c.tid - 1]; }
            if (n.i + 1 < P.N) gl[VF] += c.o1[c.tid + 1];
#pragma unroll
            for (int k = 0; k < NV; k++) {
                if (!n.on[k]) continue;
                double g = gl[k];
                if (n.hasL[k]) { g -= n.zL[k]; double cp = (n.x[k] - n.lb[k])*n.zL[k]; cmax = fmax(cmax, cp); cmin = fmin(cmin, cp); sumz += n.zL[k]; nz += 1; }
                if (n.hasU[k]) { g += n.zU[k]; double cp = (n.ub[k] - n.x[k])*n.zU[k]; cmax = fmax(cmax, cp); cmin = fmin(cmin, cp); sumz += n.zU[k]; nz += 1; }
                dual = fmax(dual, fabs(g));
            }
        }
        double vm[5] = {dual, prim, prim_u, cmax, -cmin};
        block_reduce<5>(vm, OpMax(), c);
        double vs[4] = {sumlam, sumz, nlam, nz};
        block_reduce<4>(vs, OpSum(), c);
        E.dual = vm[0]; E.primal = vm[1]; E.primal_u = vm[2]; E.cmax = vm[3]; E.cmin = -vm[4];
        E.sd = fmax(K_SMAX, (vs[0] + vs[1])/fmax(1.0, vs[2] + vs[3]))/K_SMAX;
        E.sc = fmax(K_SMAX, vs[1]/fmax(1.0, vs[3]))/K_SMAX;
    }
    __device__ static __forceinline__ double compl_err(const Err &E, double mu_) { return (E.cmax >= E.cmin) ? fmax(fabs(E.cmax - mu_), fabs(E.cmin - mu_)) : 0.0; }
    __device__ static __forceinline__ double total_err(const Err &E, double mu_) { return fmax(E.dual/E.sd, fmax(E.primal, compl_err(E, mu_)/E.sc)); }

    /*
     * Condensed stage block of node i into LDS (W&B eq. (13) with slacks and bound multipliers eliminated).
     * MODE_LSQ: least-squares multiplier system (W = 0, Sigma = I, gradient = grad f - zL + zU).
     */
    template <int MODE> __device__ __forceinline__ void assemble(double mu_, double dw)
    {
        const double q = nb_q();
        double Htt = 0, Hbb = 0, Hbq = 0, Hbf = 0, Hbp = 0, Hqq = 0, Hqf = 0, Hff = 0, Hfp = 0, Hfs = 0, Hpp = 0, Hss = 0;
        double ht = 0, hb = 0, hq = 0, hf = 0, hp = 0, hs = 0;
        double nHbb = 0, nHbq = 0, nhb = 0;
        if (n.ival) {
            const double f = n.x[VF];
            double gb[NR], gf[NR], gp[NR], gs[NR], gb1[NR];
            row_grads(f, gb, gf, gp, gs, gb1);
            double of, op, os, oq, off, opp;
            obj_grads(q, of, op, os, oq, off, opp);
            hf = of; hp = op; hs = os; hq = oq;
            if (MODE == MODE_NEWTON) {
                Hff = off; Hpp = opp;
                if (P.energyOpt && n.i > 0) { Hqq = off; Hqf = -off; }
                /* - lam_t hess(tau) - lam_b hess(b+) */
                const double hbb = -(n.lam[0]*e.tbb + n.lam[1]*e.Bbb), hbw = -(n.lam[0]*e.tbw + n.lam[1]*e.Bbw), hww = -(n.lam[0]*e.tww + n.lam[1]*e.Bww);
                Hbb += hbb; Hbf += hbw; Hff += hww;
                if (P.withPn) { Hbp += hbw; Hfp += hww; Hpp += hww; }
                /* nu * hess(row) */
                const double b = n.x[VB];
                if (R.on[RPW0]) { Hbf += n.nu[RPW0]*R.rs[RPW0]*0.5/e.sb; Hbb += n.nu[RPW0]*R.rs[RPW0]*(-0.25*f/(b*e.sb)); }
                if (R.on[RPW1]) { nHbq += n.nu[RPW1]*R.rs[RPW1]*0.5/e.sb1; nHbb += n.nu[RPW1]*R.rs[RPW1]*(-0.25*f/(e.b1*e.sb1)); }
                if (R.on[RACC]) Hbb += n.nu[RACC]*R.rs[RACC]*0.25*P.sr1/(b*e.sb);
            }
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (!R.on[r]) continue;
                double Sg, coef;
                if (MODE == MODE_NEWTON) { double gphi; row_slack_terms(r, mu_, dw, Sg, gphi); coef = Sg*resd[r] + gphi; }
                else { Sg = 1.0; coef = -(R.hasL[r] ? 1.0 : 0.0) + (R.hasU[r] ? 1.0 : 0.0); }
                hb += coef*gb[r]; hf += coef*gf[r]; hp += coef*gp[r]; hs += coef*gs[r]; nhb += coef*gb1[r];
                Hbb += Sg*gb[r]*gb[r]; Hbf += Sg*gb[r]*gf[r]; Hbp += Sg*gb[r]*gp[r];
                Hff += Sg*gf[r]*gf[r]; Hfp += Sg*gf[r]*gp[r]; Hfs += Sg*gf[r]*gs[r];
                Hpp += Sg*gp[r]*gp[r]; Hss += Sg*gs[r]*gs[r];
                nHbq += Sg*gf[r]*gb1[r]; nHbb += Sg*gb1[r]*gb1[r];
            }
        } else if (n.i == P.N && !P.energyOpt) ht = sf/P.objDen;
        /* bounds of the node's own variables + regularisation */
        if (n.i <= P.N) {
            double Sv[NV], gv[NV];
#pragma unroll
            for (int k = 0; k < NV; k++) {
                Sv[k] = 0; gv[k] = 0;
                if (!n.on[k]) continue;
                if (MODE == MODE_NEWTON) { bar_terms(n.x[k], n.lb[k], n.ub[k], n.hasL[k], n.hasU[k], n.zL[k], n.zU[k], mu_, Sv[k], gv[k]); Sv[k] += dw; }
                else { Sv[k] = 1.0; gv[k] = -(n.hasL[k] ? 1.0 : 0.0) + (n.hasU[k] ? 1.0 : 0.0); }
            }
            Htt += Sv[VT]; ht += gv[VT]; Hbb += Sv[VB]; hb += gv[VB]; Hff += Sv[VF]; hf += gv[VF]; Hpp += Sv[VP]; hp += gv[VP]; Hss += Sv[VS]; hs += gv[VS];
        }
        /* the end-of-interval power row lives in the next stage's (b, q) block */
        c.o1[c.tid] = nHbb; c.o2[c.tid] = nHbq; c.o3[c.tid] = nhb;
        __syncthreads();
        if (n.i <= P.N) {
            if (n.i > 0) { Hbb += c.o1[c.tid - 1]; Hbq += c.o2[c.tid - 1]; hb += c.o3[c.tid - 1]; }
            double *s = c.S + n.i*S_STRIDE;
            if (n.ival) {
                s[S_TB] = e.tb; s[S_TW] = e.tw; s[S_BB] = e.Bb; s[S_BW] = e.Bw;
                s[S_RT] = (MODE == MODE_NEWTON) ? -resc[0] : 0.0; s[S_RB] = (MODE == MODE_NEWTON) ? -resc[1] : 0.0;
            }
            s[S_HTT] = Htt; s[S_HBB] = Hbb; s[S_HBQ] = Hbq; s[S_HBF] = Hbf; s[S_HBP] = Hbp; s[S_HQQ] = Hqq; s[S_HQF] = Hqf;
            s[S_HFF] = Hff; s[S_HFP] = Hfp; s[S_HFS] = Hfs; s[S_HPP] = Hpp; s[S_HSS] = Hss;
            s[S_HT] = ht; s[S_HB] = hb; s[S_HQ] = hq; s[S_HF] = hf; s[S_HP] = hp; s[S_HS] = hs;
        }
        __syncthreads();
    }

    /* KKT solve: assemble, serial Riccati, read the direction back.  Returns the inertia flag (uniform). */
    template <int MODE> __device__ __noinline__ bool direction(double mu_, double dw)
    {
        assemble<MODE>(mu_, dw);
        if (c.tid == 0) c.misc[0] = riccati_solve(P, c.S) ? 1.0 : 0.0;
        __syncthreads();
        const bool ok = c.misc[0] != 0.0;
        if (ok && n.i <= P.N) {
            const double *s = c.S + n.i*S_STRIDE;
            n.dx[VT] = s[S_DT]; n.dx[VB] = s[S_DB];
            if (n.ival) {
                n.dx[VF] = s[S_DF]; n.dx[VP] = s[S_DP]; n.dx[VS] = s[S_DS];
                const double lt = s[S_LT], lb = s[S_LB];
                const double *s1 = s + S_STRIDE;
                const double db1 = s1[S_DB];
                double gb[NR], gf[NR], gp[NR], gs[NR], gb1[NR];
                row_grads(n.x[VF], gb, gf, gp, gs, gb1);
                if (MODE == MODE_NEWTON) { n.dlam[0] = lt - n.lam[0]; n.dlam[1] = lb - n.lam[1]; }
                else { n.dlam[0] = lt; n.dlam[1] = lb; }
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    n.dsg[r] = 0; n.dnu[r] = 0;
                    if (!R.on[r]) continue;
                    double lin = gb[r]*n.dx[VB] + gf[r]*n.dx[VF] + gp[r]*n.dx[VP] + gs[r]*n.dx[VS] + gb1[r]*db1;
                    if (MODE == MODE_NEWTON) {
                        double Sg, gphi; row_slack_terms(r, mu_, dw, Sg, gphi);
                        n.dsg[r] = resd[r] + lin;
                        n.dnu[r] = Sg*n.dsg[r] + gphi - n.nu[r];
                    } else {
                        /* nu = Sigma dsigma + (-zL + zU) with Sigma = 1 */
                        n.dnu[r] = lin + (-(R.hasL[r] ? 1.0 : 0.0) + (R.hasU[r] ? 1.0 : 0.0));
                    }
                }
            } else { n.dx[VF] = n.dx[VP] = n.dx[VS] = 0; }
#pragma unroll
            for (int k = 0; k < NV; k++) if (!n.on[k]) n.dx[k] = 0;
        }
        __syncthreads();
        return ok;
    }

    __device__ __forceinline__ double dzL_var(int k, double mu_) const { double s = n.x[k] - n.lb[k]; return mu_/s - n.zL[k] - n.zL[k]/s*n.dx[k]; }
    __device__ __forceinline__ double dzU_var(int k, double mu_) const { double s = n.ub[k] - n.x[k]; return mu_/s - n.zU[k] + n.zU[k]/s*n.dx[k]; }
    __device__ __forceinline__ double dzL_row(int r, double mu_) const { double s = n.sg[r] - R.dL[r]; return mu_/s - n.zLs[r] - n.zLs[r]/s*n.dsg[r]; }
    __device__ __forceinline__ double dzU_row(int r, double mu_) const { double s = R.dU[r] - n.sg[r]; return mu_/s - n.zUs[r] + n.zUs[r]/s*n.dsg[r]; }

    /* fraction-to-the-boundary step lengths of the current direction: primal, dual */
    __device__ __noinline__ void step_lengths(double mu_, double tau_, double &apr, double &adu)
    {
        double ap = 1.0, ad = 1.0;
        if (n.i <= P.N) {
#pragma unroll
            for (int k = 0; k < NV; k++) {
                if (!n.on[k]) continue;
                double d = n.dx[k];
                if (n.hasL[k]) { if (d < 0) ap = fmin(ap, -tau_*(n.x[k] - n.lb[k])/d); double dz = dzL_var(k, mu_); if (dz < 0) ad = fmin(ad, -tau_*n.zL[k]/dz); }
                if (n.hasU[k]) { if (d > 0) ap = fmin(ap, tau_*(n.ub[k] - n.x[k])/d); double dz = dzU_var(k, mu_); if (dz < 0) ad = fmin(ad, -tau_*n.zU[k]/dz); }
            }
        }
        if (n.ival) {
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (!R.on[r]) continue;
                double d = n.dsg[r];
                if (R.hasL[r]) { if (d < 0) ap = fmin(ap, -tau_*(n.sg[r] - R.dL[r])/d); double dz = dzL_row(r, mu_); if (dz < 0) ad = fmin(ad, -tau_*n.zLs[r]/dz); }
                if (R.hasU[r]) { if (d > 0) ap = fmin(ap, tau_*(R.dU[r] - n.sg[r])/d); double dz = dzU_row(r, mu_); if (dz < 0) ad = fmin(ad, -tau_*n.zUs[r]/dz); }
            }
        }
        double v[2] = {ap, ad};
        block_reduce<2>(v, OpMin(), c);
        apr = v[0]; adu = v[1];
    }

    /* theta (1-norm of the scaled constraint rows), barrier objective and validity of the point x + alpha d */
    __device__ __noinline__ void merit(double alpha, double mu_, double &theta, double &phi, bool &ok, Ev *evout, double (*xout)[NV], double (*sgout)[NR])
    {
        double xt[NV], st[NR];
#pragma unroll
        for (int k = 0; k < NV; k++) xt[k] = n.x[k] + (n.on[k] ? alpha*n.dx[k] : 0.0);
#pragma unroll
        for (int r = 0; r < NR; r++) st[r] = n.sg[r] + (R.on[r] ? alpha*n.dsg[r] : 0.0);
        publish(xt);
        double th = 0, bar = 0, bad = 0, obj = 0;
        Ev et;
        if (n.ival) {
            eval_interval<false>(P, R, n, xt, c.xt[c.tid + 1], c.xb[c.tid + 1], et);
            th = n.sct*fabs(et.c[0]) + n.scb*fabs(et.c[1]);
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (!R.on[r]) continue;
                th += fabs(et.d[r] - st[r]);
                if (R.hasL[r]) { double s = st[r] - R.dL[r]; if (s <= 0) bad = 1; else bar -= mu_*log(s); }
                if (R.hasU[r]) { double s = R.dU[r] - st[r]; if (s <= 0) bad = 1; else bar -= mu_*log(s); }
                if (R.hasL[r] && !R.hasU[r]) bar += K_D*mu_*(st[r] - R.dL[r]);
                if (!R.hasL[r] && R.hasU[r]) bar += K_D*mu_*(R.dU[r] - st[r]);
            }
        }
        if (n.i <= P.N) {
#pragma unroll
            for (int k = 0; k < NV; k++) {
                if (!n.on[k]) continue;
                if (n.hasL[k]) { double s = xt[k] - n.lb[k]; if (s <= 0) bad = 1; else bar -= mu_*log(s); }
                if (n.hasU[k]) { double s = n.ub[k] - xt[k]; if (s <= 0) bad = 1; else bar -= mu_*log(s); }
                if (n.hasL[k] && !n.hasU[k]) bar += K_D*mu_*(xt[k] - n.lb[k]);
                if (!n.hasL[k] && n.hasU[k]) bar += K_D*mu_*(n.ub[k] - xt[k]);
            }
            obj = objective_term(P, n, xt, (n.i > 0) ? c.xf[c.tid - 1] : 0.0, sf);
        }
        double v[4] = {th, bar, obj, bad};
        block_reduce<4>(v, OpSum(), c);
        theta = v[0]; phi = v[2] + v[1];
        ok = (v[3] == 0.0) && isfinite(theta) && isfinite(phi);
        if (evout) *evout = et;
        if (xout) {
#pragma unroll
            for (int k = 0; k < NV; k++) (*xout)[k] = xt[k];
#pragma unroll
            for (int r = 0; r < NR; r++) (*sgout)[r] = st[r];
        }
    }

    __device__ __forceinline__ bool filter_ok(int nfilt, double theta, double phi) const
    {
        for (int j = 0; j < nfilt; j++)
            if (theta >= c.filt[2*j] && phi >= c.filt[2*j + 1]) return false;
        return true;
    }

    /* ---------------------------------------------------------------------------------------- */
    __device__ __forceinline__ void run(const double *scen, double *z_out, double *lam_out, double *stats, double *hist, int hist_cap)
    {
        const int N = P.N;
        n.i = c.tid;
        n.ival = n.i < N;
        const double t0 = scen[MSD_SC_T0], tEnd = scen[MSD_SC_TEND], v0sq = scen[MSD_SC_V0SQ], vNsq = scen[MSD_SC_VNSQ];

        /* ---- static data of the node: profile (coalesced reads), bounds (ocp.py:175-181, 247-272) ---- */
        n.ds = n.ival ? P.ds[n.i] : 0.0;
        n.G = n.ival ? track_resistance(P, P.grad[n.i], P.curv[n.i]) : 0.0;
#pragma unroll
        for (int k = 0; k < NV; k++) { n.lb[k] = -INFINITY; n.ub[k] = INFINITY; n.on[k] = false; n.hasL[k] = n.hasU[k] = false; }
        if (n.i <= N) {
            n.on[VT] = n.on[VB] = true;
            n.on[VF] = n.on[VS] = n.ival; n.on[VP] = n.ival && P.withPn;
            n.lb[VF] = P.fmin; n.ub[VF] = P.fmax; n.lb[VP] = P.fminPn; n.ub[VP] = 0; n.lb[VS] = 0;
            if (n.i == 0) { n.lb[VT] = n.ub[VT] = t0; n.lb[VB] = n.ub[VB] = v0sq; }
            else if (n.i == N) { n.lb[VT] = t0; n.ub[VT] = tEnd; n.lb[VB] = n.ub[VB] = vNsq; }
            else { n.lb[VT] = t0; n.ub[VT] = tEnd; n.lb[VB] = P.vminSq; n.ub[VB] = P.bmax[n.i]; }
#pragma unroll
            for (int k = 0; k < NV; k++) {
                if (!n.on[k]) continue;
                if (n.lb[k] == n.ub[k]) { n.on[k] = false; continue; }      /* fixed variables are parameters */
                n.hasL[k] = isfinite(n.lb[k]); n.hasU[k] = isfinite(n.ub[k]);
                if (n.hasL[k]) n.lb[k] -= K_BOUND_RELAX*fmax(1.0, fabs(n.lb[k]));
                if (n.hasU[k]) n.ub[k] += K_BOUND_RELAX*fmax(1.0, fabs(n.ub[k]));
            }
        }
#pragma unroll
        for (int r = 0; r < NR; r++) { R.on[r] = false; R.dL[r] = -INFINITY; R.dU[r] = INFINITY; R.rs[r] = 1.0; R.hasL[r] = R.hasU[r] = false; }
        if (P.hasPower) { R.on[RPW0] = R.on[RPW1] = true; R.dL[RPW0] = R.dL[RPW1] = -fabs(P.pwL); R.dU[RPW0] = R.dU[RPW1] = fabs(P.pwU); }
        R.on[RACC] = true; R.dL[RACC] = P.accMin; R.dU[RACC] = P.accMax;
        if (P.energyOpt) { R.on[RLTR] = R.on[RLRG] = true; R.dL[RLTR] = R.dL[RLRG] = 0; }

        /* ---- cold start (ocp.py:325-339) ---- */
        {
            const double dt = (tEnd - t0)/N, vel0 = (60/3.6)*(60/3.6);
            n.x[VT] = t0 + dt*n.i; n.x[VB] = vel0; n.x[VF] = 0.5; n.x[VP] = P.withPn ? -0.1 : 0.0; n.x[VS] = 1;
            if (n.i == 0) { n.x[VT] = t0; n.x[VB] = v0sq; }
            if (n.i == N) n.x[VB] = vNsq;
        }
#pragma unroll
        for (int k = 0; k < NV; k++) { n.zL[k] = n.zU[k] = 0; n.dx[k] = 0; }
#pragma unroll
        for (int r = 0; r < NR; r++) { n.sg[r] = n.nu[r] = n.zLs[r] = n.zUs[r] = 0; n.dsg[r] = n.dnu[r] = 0; resd[r] = 0; }
        n.lam[0] = n.lam[1] = 0; n.dlam[0] = n.dlam[1] = 0; resc[0] = resc[1] = 0;
        n.sct = n.scb = 1; sf = 1;

        /* ---- gradient-based scaling at the starting point (max gradient 100) ---- */
        publish(n.x);
        {
            double gmax = 0, rmax[NR] = {0, 0, 0, 0, 0};
            if (n.ival) {
                eval_interval<true>(P, R, n, n.x, c.xt[c.tid + 1], c.xb[c.tid + 1], e);
                double of, op, os, oq, off, opp;
                obj_grads(nb_q(), of, op, os, oq, off, opp);
                gmax = fmax(fabs(of), fmax(fabs(op), fabs(os)));
                double mb = (n.i == N - 1) ? 0.0 : 1.0;
                if (n.i > 0) mb = fmax(mb, fabs(e.Bb));
                mb = fmax(mb, fabs(e.Bw));
                n.scb = mb > 100 ? 100/mb : 1;
                double mt = 1.0;
                if (n.i > 0) mt = fmax(mt, fabs(e.tb));
                mt = fmax(mt, fabs(e.tw));
                n.sct = mt > 100 ? 100/mt : 1;
                double gb[NR], gf[NR], gp[NR], gs[NR], gb1[NR];
                row_grads(n.x[VF], gb, gf, gp, gs, gb1);
#pragma unroll
                for (int r = 0; r < NR; r++) {
                    double m = fmax(fabs(gf[r]), fmax(fabs(gp[r]), fabs(gs[r])));
                    if (n.i > 0) m = fmax(m, fabs(gb[r]));
                    if (n.i < N - 1) m = fmax(m, fabs(gb1[r]));
                    rmax[r] = m;
                }
            } else if (n.i == N && !P.energyOpt) gmax = 1.0/P.objDen;
            double v[6] = {gmax, rmax[0], rmax[1], rmax[2], rmax[3], rmax[4]};
            block_reduce<6>(v, OpMax(), c);
            if (v[0] > 100) sf = 100/v[0];
#pragma unroll
            for (int r = 0; r < NR; r++) if (R.on[r] && v[1 + r] > 100) R.rs[r] = 100/v[1 + r];
        }
#pragma unroll
        for (int r = 0; r < NR; r++) {
            if (!R.on[r]) continue;
            R.dL[r] *= R.rs[r]; R.dU[r] *= R.rs[r];
            R.hasL[r] = isfinite(R.dL[r]); R.hasU[r] = isfinite(R.dU[r]);
            if (R.hasL[r]) R.dL[r] -= K_BOUND_RELAX*fmax(1.0, fabs(R.dL[r]));
            if (R.hasU[r]) R.dU[r] += K_BOUND_RELAX*fmax(1.0, fabs(R.dU[r]));
        }

        /* ---- push into the interior, slacks, bound multipliers ---- */
#pragma unroll
        for (int k = 0; k < NV; k++) {
            if (!n.on[k]) continue;
            n.x[k] = push_in(n.x[k], n.lb[k], n.ub[k], n.hasL[k], n.hasU[k]);
            n.zL[k] = n.hasL[k] ? 1.0 : 0.0; n.zU[k] = n.hasU[k] ? 1.0 : 0.0;
        }
        publish(n.x);
        if (n.ival) {
            eval_interval<true>(P, R, n, n.x, c.xt[c.tid + 1], c.xb[c.tid + 1], e);
#pragma unroll
            for (int r = 0; r < NR; r++) {
                if (!R.on[r]) continue;
                n.sg[r] = push_in(e.d[r], R.dL[r], R.dU[r], R.hasL[r], R.hasU[r]);
                n.zLs[r] = R.hasL[r] ? 1.0 : 0.0; n.zUs[r] = R.hasU[r] ? 1.0 : 0.0;
            }
        }

        mu = K_MU_INIT; tau = fmax(K_TAU_MIN, 1 - mu);

        /* ---- least-squares multiplier estimate (W&B section 3.6) ---- */
        {
            bool ok = direction<MODE_LSQ>(0.0, 0.0);
            double lmax = 0;
            if (ok && n.ival) {
                lmax = fmax(fabs(n.dlam[0])/n.sct, fabs(n.dlam[1])/n.scb);
#pragma unroll
                for (int r = 0; r < NR; r++) if (R.on[r]) lmax = fmax(lmax, fabs(n.dnu[r]));
            }
            double v[1] = {lmax};
            block_reduce<1>(v, OpMax(), c);
            const bool use = ok && v[0] <= LAM_INIT_MAX && isfinite(v[0]);
            n.lam[0] = use ? n.dlam[0] : 0.0; n.lam[1] = use ? n.dlam[1] : 0.0;
#pragma unroll
            for (int r = 0; r < NR; r++) n.nu[r] = (use && R.on[r] && n.ival) ? n.dnu[r] : 0.0;
#pragma unroll
            for (int k = 0; k < NV; k++) n.dx[k] = 0;
#pragma unroll
            for (int r = 0; r < NR; r++) { n.dsg[r] = 0; n.dnu[r] = 0; }
            n.dlam[0] = n.dlam[1] = 0;
        }

        /* ---- filter ---- */
        double theta, phi; bool okp;
        merit(0.0, mu, theta, phi, okp, nullptr, nullptr, nullptr);
        const double theta_max = 1e4*fmax(1.0, theta), theta_min = 1e-4*fmax(1.0, theta);
        int nfilt = 0;
        double delta_last = 0;

        int status = MSD_STATUS_MAXITER, iter = 0, acc_count = 0, tiny_count = 0;
        int n_reg = 0, n_soc = 0, n_back = 0;
        Err E; E.dual = E.primal = E.primal_u = 0; E.cmax = E.cmin = 0; E.sd = E.sc = 1;
        double alpha_pr = 0, alpha_du = 0, dnorm = 0, objv = 0;
        const double mu_floor = fmin(P.tol, 1e-4)/(K_EPS + 1.0);

        for (iter = 0;; iter++) {
            publish(n.x);
            if (n.ival) eval_interval<true>(P, R, n, n.x, c.xt[c.tid + 1], c.xb[c.tid + 1], e);
            kkt_error(E);
            {
                double v[1] = {(n.i <= N) ? objective_term(P, n, n.x, nb_q(), sf) : 0.0};
                block_reduce<1>(v, OpSum(), c);
                objv = v[0]/sf;
            }
            if (hist && c.tid == 0 && iter < hist_cap) {
                double *hh = hist + HIST_COLS*iter;
                hh[0] = iter; hh[1] = objv; hh[2] = E.primal; hh[3] = E.dual; hh[4] = log10(mu); hh[5] = dnorm; hh[6] = alpha_du; hh[7] = alpha_pr;
            }
            const double E0 = total_err(E, 0.0);
            const double dual_u = E.dual/sf, compl_u = compl_err(E, 0.0)/sf;
            if (E0 <= P.tol && dual_u <= 1.0 && E.primal_u <= 1e-4 && compl_u <= 1e-4) { status = MSD_STATUS_SOLVED; break; }
            if (E0 <= ACC_TOL && dual_u <= 1e10 && E.primal_u <= 1e-2 && compl_u <= 1e-2) { if (++acc_count >= ACC_ITER) { status = MSD_STATUS_ACCEPTABLE; break; } }
            else acc_count = 0;
            if (iter >= P.maxIter) { status = MSD_STATUS_MAXITER; break; }
            if (!isfinite(E0)) { status = MSD_STATUS_NUMERIC; break; }

            /* barrier parameter (monotone, W&B eq. (7)); E_mu only differs from E_0 in the complementarity part */
            {
                bool changed = false;
                while (total_err(E, mu) <= K_EPS*mu && mu > mu_floor) {
                    double nm = fmax(mu_floor, fmin(K_MU_LIN*mu, pow(mu, K_MU_SUP)));
                    if (nm >= mu) break;
                    mu = nm; tau = fmax(K_TAU_MIN, 1 - mu); changed = true;
                }
                if (changed) nfilt = 0;
            }
            merit(0.0, mu, theta, phi, okp, nullptr, nullptr, nullptr);

            /* search direction with inertia correction (W&B Algorithm IC) */
            resc[0] = n.ival ? e.c[0] : 0.0; resc[1] = n.ival ? e.c[1] : 0.0;
#pragma unroll
            for (int r = 0; r < NR; r++) resd[r] = (n.ival && R.on[r]) ? e.d[r] - n.sg[r] : 0.0;
            double dw = 0;
            bool ok = direction<MODE_NEWTON>(mu, 0.0);
            if (!ok) {
                n_reg++;
                dw = (delta_last == 0) ? DW_0 : fmax(DW_MIN, KW_MINUS*delta_last);
                for (;;) {
                    ok = direction<MODE_NEWTON>(mu, dw);
                    if (ok) break;
                    dw *= (delta_last == 0) ? KW_PLUS_BAR : KW_PLUS;
                    if (dw > DW_MAX) break;
                }
                if (!ok) { status = MSD_STATUS_REGULARIZATION; break; }
                delta_last = dw;
            }

            /* directional derivative of the barrier function, step norms */
            double gphid, rel_step;
            {
                double gd = 0, dn = 0, rel = 0;
                double of = 0, op = 0, os = 0, oq = 0, off = 0, opp = 0;
                if (n.i <= N) obj_grads(nb_q(), of, op, os, oq, off, opp);
                /* d(obj)/dq of the next interval belongs to this node's f (barrier in uniform control flow) */
                c.o1[c.tid] = oq;
                __syncthreads();
                if (n.i <= N) {
                    double og[NV] = {(n.i == N && !P.energyOpt) ? sf/P.objDen : 0.0, 0.0, of, op, os};
                    if (n.i + 1 < N) og[VF] += c.o1[c.tid + 1];
#pragma unroll
                    for (int k = 0; k < NV; k++) {
                        if (!n.on[k]) continue;
                        double Sg, gp; bar_terms(n.x[k], n.lb[k], n.ub[k], n.hasL[k], n.hasU[k], n.zL[k], n.zU[k], mu, Sg, gp);
                        gd += (og[k] + gp)*n.dx[k];
                        dn = fmax(dn, fabs(n.dx[k])); rel = fmax(rel, fabs(n.dx[k])/(1 + fabs(n.x[k])));
                    }
                    if (n.ival) {
#pragma unroll
                        for (int r = 0; r < NR; r++) {
                            if (!R.on[r]) continue;
                            double Sg, gp; bar_terms(n.sg[r], R.dL[r], R.dU[r], R.hasL[r], R.hasU[r], n.zLs[r], n.zUs[r], mu, Sg, gp);
                            gd += gp*n.dsg[r];
                            dn = fmax(dn, fabs(n.dsg[r])); rel = fmax(rel, fabs(n.dsg[r])/(1 + fabs(n.sg[r])));
                        }
                    }
                }
                double v1[1] = {gd}; block_reduce<1>(v1, OpSum(), c);
                double v2[2] = {dn, rel}; block_reduce<2>(v2, OpMax(), c);
                gphid = v1[0]; dnorm = v2[0]; rel_step = v2[1];
            }

            double amax;
            step_lengths(mu, tau, amax, alpha_du);

            const bool tiny = rel_step < 10*DBL_EPSILON;
            double alpha = amax;
            bool accepted = false, ftype_armijo = false, took_soc = false;
            /* point to be accepted */
            double xacc[NV], sacc[NR];
            if (tiny) {
                accepted = true;
                double th_t, ph_t; bool okt;
                merit(alpha, mu, th_t, ph_t, okt, nullptr, &xacc, &sacc);
                if (++tiny_count >= 2 && mu <= mu_floor*(1 + 1e-12)) { status = MSD_STATUS_TINY_STEP; break; }
            } else tiny_count = 0;

            double amin = G_THETA;
            if (gphid < 0) {
                amin = fmin(amin, G_PHI*theta/(-gphid));
                if (theta <= theta_min) amin = fmin(amin, K_DELTA*pow(theta, S_THETA)/pow(-gphid, S_PHI));
            }
            amin *= ALPHA_MIN_FRAC;

            /* saved Newton direction for the case a second-order correction replaces it */
            int ls = 0;
            while (!accepted) {
                double th_t, ph_t; bool okt; Ev et;
                merit(alpha, mu, th_t, ph_t, okt, &et, &xacc, &sacc);
                const bool ftype = (gphid < 0) && (alpha*pow(-gphid, S_PHI) > K_DELTA*pow(theta, S_THETA));
                bool acc = false;
                if (okt && th_t <= theta_max) {
                    if (ftype && theta <= theta_min) acc = cmp_le(ph_t - phi, ETA_PHI*alpha*gphid, phi);
                    else acc = cmp_le(th_t, (1 - G_THETA)*theta, theta) || cmp_le(ph_t - phi, -G_PHI*theta, phi);
                    if (acc) acc = filter_ok(nfilt, th_t, ph_t);
                }
                if (acc) { accepted = true; ftype_armijo = ftype && cmp_le(ph_t - phi, ETA_PHI*alpha*gphid, phi); break; }

                /* second-order correction (W&B section 2.4) */
                if (ls == 0 && okt && th_t >= theta) {
                    const double th_old = theta; double th_prev = th_t; int nsoc = 0;
                    /* keep the Newton step */
                    double sdx[NV], sdsg[NR], sdlam[2], sdnu[NR], src[2], srd[NR];
#pragma unroll
                    for (int k = 0; k < NV; k++) sdx[k] = n.dx[k];
#pragma unroll
                    for (int r = 0; r < NR; r++) { sdsg[r] = n.dsg[r]; sdnu[r] = n.dnu[r]; srd[r] = resd[r]; }
                    sdlam[0] = n.dlam[0]; sdlam[1] = n.dlam[1]; src[0] = resc[0]; src[1] = resc[1];
                    double alpha_soc = alpha;
                    while (nsoc < P_MAX_SOC) {
                        /* c_soc = alpha_soc c_soc + c(trial) */
                        resc[0] = alpha_soc*resc[0] + (n.ival ? et.c[0] : 0.0); resc[1] = alpha_soc*resc[1] + (n.ival ? et.c[1] : 0.0);
#pragma unroll
                        for (int r = 0; r < NR; r++) if (n.ival && R.on[r]) resd[r] = alpha_soc*resd[r] + (et.d[r] - sacc[r]);
                        publish(n.x);          /* the neighbours' Fel in LDS are those of the trial point */
                        if (!direction<MODE_NEWTON>(mu, dw)) break;
                        double adu_soc;
                        step_lengths(mu, tau, alpha_soc, adu_soc);
                        double th_s, ph_s; bool oks;
                        merit(alpha_soc, mu, th_s, ph_s, oks, &et, &xacc, &sacc);
                        nsoc++; n_soc++;
                        bool accs = false;
                        if (oks && th_s <= theta_max) {
                            if (ftype && th_old <= theta_min) accs = cmp_le(ph_s - phi, ETA_PHI*alpha*gphid, phi);
                            else accs = cmp_le(th_s, (1 - G_THETA)*th_old, th_old) || cmp_le(ph_s - phi, -G_PHI*th_old, phi);
                            if (accs) accs = filter_ok(nfilt, th_s, ph_s);
                        }
                        if (accs) {
                            accepted = true; took_soc = true; ftype_armijo = ftype && cmp_le(ph_s - phi, ETA_PHI*alpha*gphid, phi);
                            alpha = alpha_soc; alpha_du = adu_soc;
                            break;
                        }
                        if (!oks || th_s > K_SOC*th_prev) break;
                        th_prev = th_s;
                    }
                    if (accepted) break;
                    /* back to the Newton step */
#pragma unroll
                    for (int k = 0; k < NV; k++) n.dx[k] = sdx[k];
#pragma unroll
                    for (int r = 0; r < NR; r++) { n.dsg[r] = sdsg[r]; n.dnu[r] = sdnu[r]; resd[r] = srd[r]; }
                    n.dlam[0] = sdlam[0]; n.dlam[1] = sdlam[1]; resc[0] = src[0]; resc[1] = src[1];
                }
                alpha *= 0.5; ls++; n_back++;
                if (alpha < amin) break;
            }
            if (!accepted) { status = MSD_STATUS_LINESEARCH; break; }
            alpha_pr = alpha;
            (void)took_soc;

            /* filter augmentation (W&B eq. (22)) */
            if (!tiny && !ftype_armijo && nfilt < FILT_CAP) {
                __syncthreads();
                if (c.tid == 0) { c.filt[2*nfilt] = (1 - G_THETA)*theta; c.filt[2*nfilt + 1] = phi - G_PHI*theta; }
                nfilt++;
                __syncthreads();
            }

            /* accept the trial point; multipliers: equality with the primal step, bounds with alpha_du (of the accepted direction) */
            if (n.i <= N) {
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    if (!n.on[k]) continue;
                    const double dzl = n.hasL[k] ? dzL_var(k, mu) : 0.0, dzu = n.hasU[k] ? dzU_var(k, mu) : 0.0;
                    n.x[k] = xacc[k];
                    if (n.hasL[k]) n.zL[k] += alpha_du*dzl;
                    if (n.hasU[k]) n.zU[k] += alpha_du*dzu;
                }
                if (n.ival) {
#pragma unroll
                    for (int r = 0; r < NR; r++) {
                        if (!R.on[r]) continue;
                        const double dzl = R.hasL[r] ? dzL_row(r, mu) : 0.0, dzu = R.hasU[r] ? dzU_row(r, mu) : 0.0;
                        n.sg[r] = sacc[r];
                        n.nu[r] += alpha_pr*n.dnu[r];
                        if (R.hasL[r]) n.zLs[r] += alpha_du*dzl;
                        if (R.hasU[r]) n.zUs[r] += alpha_du*dzu;
                    }
                    n.lam[0] += alpha_pr*n.dlam[0]; n.lam[1] += alpha_pr*n.dlam[1];
                }
                /* keep Sigma within [mu/(kappa_Sigma s), kappa_Sigma mu/s] (W&B eq. (16)) */
#pragma unroll
                for (int k = 0; k < NV; k++) {
                    if (!n.on[k]) continue;
                    if (n.hasL[k]) { double s = n.x[k] - n.lb[k]; n.zL[k] = fmax(fmin(n.zL[k], K_SIGMA*mu/s), mu/(K_SIGMA*s)); }
                    if (n.hasU[k]) { double s = n.ub[k] - n.x[k]; n.zU[k] = fmax(fmin(n.zU[k], K_SIGMA*mu/s), mu/(K_SIGMA*s)); }
                }
                if (n.ival) {
#pragma unroll
                    for (int r = 0; r < NR; r++) {
                        if (!R.on[r]) continue;
                        if (R.hasL[r]) { double s = n.sg[r] - R.dL[r]; n.zLs[r] = fmax(fmin(n.zLs[r], K_SIGMA*mu/s), mu/(K_SIGMA*s)); }
                        if (R.hasU[r]) { double s = R.dU[r] - n.sg[r]; n.zUs[r] = fmax(fmin(n.zUs[r], K_SIGMA*mu/s), mu/(K_SIGMA*s)); }
                    }
                }
            }
        }

        /* ---- outputs: z in the reference's layout (ocp.py:166-272), multipliers in the reference's row order ---- */
        const int stp = 4 + P.withPn;
        if (n.ival) {
            double *zi = z_out + stp*n.i; int k = 0;
            zi[k++] = n.x[VF]; if (P.withPn) zi[k++] = n.x[VP];
            zi[k++] = n.x[VS]; zi[k++] = n.x[VT]; zi[k++] = n.x[VB];
            if (lam_out) {
                const int rpi = (P.hasPower ? 2 : 0) + 3 + (P.energyOpt ? 2 : 0);
                double *l = lam_out + rpi*n.i; int m = 0;
                if (P.hasPower) { l[m++] = n.nu[RPW0]*R.rs[RPW0]/sf; l[m++] = n.nu[RPW1]*R.rs[RPW1]/sf; }
                l[m++] = n.nu[RACC]*R.rs[RACC]/sf;
                l[m++] = n.lam[0]/sf; l[m++] = n.lam[1]/sf;
                if (P.energyOpt) { l[m++] = n.nu[RLTR]*R.rs[RLTR]/sf; l[m++] = n.nu[RLRG]*R.rs[RLRG]/sf; }
            }
        } else if (n.i == N) { z_out[stp*N] = n.x[VT]; z_out[stp*N + 1] = n.x[VB]; }
        if (c.tid == 0) {
            stats[MSD_ST_STATUS] = status; stats[MSD_ST_ITERS] = iter; stats[MSD_ST_OBJ] = objv;
            stats[MSD_ST_KKT] = total_err(E, 0.0); stats[MSD_ST_MU] = mu; stats[MSD_ST_DUAL_INF] = E.dual/sf;
            stats[MSD_ST_CONSTR_VIOL] = E.primal_u; stats[MSD_ST_COMPL] = compl_err(E, 0.0)/sf;
            stats[MSD_ST_N_REG] = n_reg; stats[MSD_ST_N_SOC] = n_soc; stats[MSD_ST_N_BACKTRACK] = n_back;
        }
        __syncthreads();
    }
};

/*
 * grid = min(nscen, resident workgroups); block = NT threads (multiple of 64, >= N + 1).
 * Dynamic LDS: lds_doubles(N, NT) * 8 bytes.
 */
template <int NT>
__global__ void __launch_bounds__(NT, 2) solve_kernel(DevProb P, int nscen, const double *scen, double *z_out, double *lam_out, double *stats,
                                                   double *hist, int hist_cap)
{
    HIP_DYNAMIC_SHARED(double, lds)
    Ctx c;
    c.tid = threadIdx.x; c.lane = threadIdx.x & 63; c.wave = threadIdx.x >> 6; c.nw = NT/64; c.red_slot = 0;
    c.S = lds;
    c.xt = c.S + S_STRIDE*(P.N + 1); c.xb = c.xt + NT; c.xf = c.xb + NT;
    c.o1 = c.xf + NT; c.o2 = c.o1 + NT; c.o3 = c.o2 + NT;
    c.filt = c.o3 + NT; c.red = c.filt + 2*FILT_CAP; c.misc = c.red + RED_SLOTS*MAX_WAVES*RED_K;
    const int nz = (4 + P.withPn)*P.N + 2;
    const int rpi = (P.hasPower ? 2 : 0) + 3 + (P.energyOpt ? 2 : 0);
    for (int sidx = blockIdx.x; sidx < nscen; sidx += gridDim.x) {
        Solver s(P, c);
        s.run(scen + (size_t)MSD_SC_COUNT*sidx, z_out + (size_t)nz*sidx, lam_out ? lam_out + (size_t)rpi*P.N*sidx : nullptr,
              stats + (size_t)MSD_ST_COUNT*sidx, (hist && sidx == 0) ? hist : nullptr, hist_cap);
        __syncthreads();
    }
}

/* one thread per interval: TrainIntegrator.solve (train.py:347-364) with sensitivities */
__global__ void stage_eval_kernel(DevProb P, int n, const double *b, const double *w, const double *ds, const double *grad, const double *curv, double *out)
{
    int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    Jet tau, bp;
    interval_map<Jet>(P, b[k], w[k], track_resistance(P, grad[k], curv[k]), ds[k], tau, bp);
    double *o = out + 12*(size_t)k;
    o[0] = tau.v; o[1] = bp.v; o[2] = tau.g0; o[3] = tau.g1; o[4] = bp.g0; o[5] = bp.g1;
    o[6] = tau.h00; o[7] = tau.h01; o[8] = tau.h11; o[9] = bp.h00; o[10] = bp.h01; o[11] = bp.h11;
}

}  // namespace msd
