/*
 * msd_resto_scan.hpp -- the Newton system of the restoration problem on the stage-parallel scan (included by msd_kernel.hpp behind ParallelRiccati).
 *
 * riccati_resto (msd_kernel.hpp) sweeps the relaxed stage system on one lane: N dependent stages of some 1 500 flops each while the other lanes of the
 * workgroup wait -- two thirds of the time of a restoration phase (alt.loose_schedules_reference_start: 45 ms with the phases against 17 ms without,
 * profiles/r05).  Round 6: the same recursion on the scan of msd_scan.hpp.  With the dynamics rows relaxed, x+ = F y + r + D lam+ (D = n/z_n + p/z_p
 * on t and b), the value function seen through the rows is  P' = (P^-1 + D)^-1 = W P,  W = (I + P D)^-1,  and the backward map of a stage
 *     P_i = J + A^T P+ (I + (C + D) P+)^-1 A
 * is the ordinary one with D added to C = Fu R^-1 Fu^T: the triples (A, C + D, J) compose like ParallelRiccati's.  Every lane then runs the ordinary
 * recursion over its own stages from the scanned boundary value -- pivots (inertia) and feedback are the serial sweep's --, and the two affine
 * recursions are scans over 3 x 3 maps again:
 *     p_i = gamma + (Phi^T W) p+,   Phi = Fx + Fu K          (gradient, backward)
 *     x+  = W^T (Phi x + Fu k + r) - Y p+,   Y = E M E^T     (state, forward; M = (P2 + D^-1)^-1 on the relaxed rows)
 * with k = k0 - Guu^-1 Fu^T W p+.  What the sweeps leave in a stage's block (feedback, P+, p+) is what riccati_resto leaves, and the roll-out is its
 * forward stage (resto_forward_stage), so steps and the new multipliers of the relaxed rows come out of the rows themselves as there.  The last
 * interval (b_N is a parameter: one force eliminated through the relaxed b row) is riccati_resto's stage, run by one thread in front of the scan.
 * Static loss rows (no cross terms between a stage and the next node); the other loss models keep the serial sweeps.  A breakdown of the scan
 * (singular control block, non-finite boundary value) returns -1 and the caller sweeps serially: the parallel path never decides alone that a
 * system cannot be solved.  Cold path: a function of its own.
 */
#pragma once

template <int SPT>
struct ParallelResto {
    static constexpr int DYN = LOSS_STATIC;
    static constexpr int S_STRIDE = stage_stride(DYN);
    using PR = ParallelRiccati<SPT, LOSS_STATIC>;
    /* the last stage's results for every thread: free slots of the terminal node's block (it holds H_tt, h_t and the terminal step only) */
    static constexpr int T_PN = 9, T_PV = 15, T_OK = 19, T_SWAP = 20;

    /* blocks of a regular stage over x = (t, b, q), u = (f, p, s) */
    struct Stage {
        double Fx[3][3], Fu[3][3], r[3], Dt, Db;
        double Hxx[3][3], Hxu[3][3], Huu[3][3], hx[3], hu[3];
    };
    __device__ static __forceinline__ void load(const double *s, const double Dt, const double Db, const bool pn, const bool with_h, Stage &g)
    {
        const double Tb = s[S_TB], Tw = s[S_TW], Bb = s[S_BB], Bw = s[S_BW];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) { g.Fx[a][b] = 0; g.Fu[a][b] = 0; g.Hxx[a][b] = 0; g.Hxu[a][b] = 0; g.Huu[a][b] = 0; }
        g.Fx[0][0] = 1; g.Fx[0][1] = Tb; g.Fx[1][1] = Bb;
        g.Fu[0][0] = Tw; g.Fu[1][0] = Bw; g.Fu[2][0] = 1;
        if (pn) { g.Fu[0][1] = Tw; g.Fu[1][1] = Bw; }
        g.r[0] = s[S_RT]; g.r[1] = s[S_RB]; g.r[2] = 0;
        g.Dt = Dt; g.Db = Db;
        if (!with_h) return;
        g.Hxx[0][0] = s[S_HTT]; g.Hxx[1][1] = s[S_HBB]; g.Hxx[1][2] = g.Hxx[2][1] = s[S_HBQ]; g.Hxx[2][2] = s[S_HQQ];
        g.Hxu[1][0] = s[S_HBF]; g.Hxu[2][0] = s[S_HQF];
        g.Huu[0][0] = s[S_HFF]; g.Huu[0][2] = g.Huu[2][0] = s[S_GFS]; g.Huu[2][2] = 1.0/s[S_IS];
        g.hx[0] = s[S_HT]; g.hx[1] = s[S_HB]; g.hx[2] = s[S_HQ];
        g.hu[0] = s[S_HF]; g.hu[1] = 0; g.hu[2] = s[S_GS];
        if (pn) { g.Hxu[1][1] = s[S_HBP]; g.Huu[0][1] = g.Huu[1][0] = s[S_HFP]; g.Huu[1][1] = s[S_HPP]; g.hu[1] = s[S_HP]; }
        else g.Huu[1][1] = 1;      /* (no pneumatic brake: the slot is an identity row, like riccati_resto's) */
    }

    /* through the relaxed rows: M = (P2 + D^-1)^-1 on (t, b), W = (I + P D)^-1 */
    __device__ static __forceinline__ bool relax(const double (&P)[3][3], const double Dt, const double Db, double (&M)[3], double (&W)[3][3])
    {
        const double st = sqrt(Dt), sb = sqrt(Db);
        const double ma = 1 + st*P[0][0]*st, mb = st*P[0][1]*sb, mc = 1 + sb*P[1][1]*sb, det = ma*mc - mb*mb;
        M[0] = st*(mc/det)*st; M[1] = -st*(mb/det)*sb; M[2] = sb*(ma/det)*sb;
#pragma unroll
        for (int a = 0; a < 3; a++) {
            W[a][0] = (a == 0 ? 1.0 : 0.0) - (P[a][0]*M[0] + P[a][1]*M[1]);
            W[a][1] = (a == 1 ? 1.0 : 0.0) - (P[a][0]*M[1] + P[a][1]*M[2]);
            W[a][2] = (a == 2 ? 1.0 : 0.0);
        }
        return det > 0 && ma > 0;
    }
    __device__ static __forceinline__ void unpack(const double (&p)[6], double (&P)[3][3])
    {
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) P[a][b] = p[sy(a, b)];
    }
    __device__ static __forceinline__ void mul(const double (&A)[3][3], const double (&B)[3][3], double (&C)[3][3])      /* C = A B */
    {
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) C[a][b] = A[a][0]*B[0][b] + A[a][1]*B[1][b] + A[a][2]*B[2][b];
    }
    __device__ static __forceinline__ void mulT(const double (&A)[3][3], const double (&B)[3][3], double (&C)[3][3])     /* C = A^T B */
    {
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) C[a][b] = A[0][a]*B[0][b] + A[1][a]*B[1][b] + A[2][a]*B[2][b];
    }

    /* triple of a regular stage; returns det R */
    __device__ static __forceinline__ double stage_elem(const Stage &g, Elem &e)
    {
        double Ri[3][3], RiS[3][3], RiFt[3][3], T[3][3];
        const double det = inv3(g.Huu, Ri);
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) {
                RiS[a][b] = Ri[a][0]*g.Hxu[b][0] + Ri[a][1]*g.Hxu[b][1] + Ri[a][2]*g.Hxu[b][2];      /* R^-1 Hux */
                RiFt[a][b] = Ri[a][0]*g.Fu[b][0] + Ri[a][1]*g.Fu[b][1] + Ri[a][2]*g.Fu[b][2];        /* R^-1 Fu^T */
            }
        mul(g.Fu, RiS, T);
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) e.A[a][b] = g.Fx[a][b] - T[a][b];
        mul(g.Fu, RiFt, T);
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = a; b < 3; b++) e.C[sy(a, b)] = 0.5*(T[a][b] + T[b][a]);
        e.C[sy(0, 0)] += g.Dt; e.C[sy(1, 1)] += g.Db;
        mul(g.Hxu, RiS, T);
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = a; b < 3; b++) e.J[sy(a, b)] = g.Hxx[a][b] - 0.5*(T[a][b] + T[b][a]);
        return det;
    }

    /* matrix part of a stage's backward step: P (stage i+1) in, P (stage i) out; feedback K, Guu^-1, and with the stage's vectors k0 and gamma (p+ = 0) */
    __device__ static __forceinline__ bool stage_matrix(const Stage &g, double (&P)[3][3], double (&K)[3][3], double (&Gi)[6], double (&k0)[3], double (&gam)[3])
    {
        double M[3], W[3][3], Pp[3][3], PFx[3][3], PFu[3][3], Gxx[3][3], Gxu[3][3], Guu[3][3], L[3][3], Li[3][3];
        bool ok = relax(P, g.Dt, g.Db, M, W);
        mul(W, P, Pp);
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = a + 1; b < 3; b++) { const double m = 0.5*(Pp[a][b] + Pp[b][a]); Pp[a][b] = Pp[b][a] = m; }
        mul(Pp, g.Fx, PFx); mul(Pp, g.Fu, PFu);
        mulT(g.Fx, PFx, Gxx); mulT(g.Fx, PFu, Gxu); mulT(g.Fu, PFu, Guu);
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) { Gxx[a][b] += g.Hxx[a][b]; Gxu[a][b] += g.Hxu[a][b]; Guu[a][b] += g.Huu[a][b]; L[a][b] = 0; Li[a][b] = 0; }
        /* Cholesky of the control block (the serial sweep's pivots) */
#pragma unroll
        for (int j = 0; j < 3; j++) {
            double d = Guu[j][j];
#pragma unroll
            for (int k = 0; k < j; k++) d -= L[j][k]*L[j][k];
            if (!(d > 0) || !isfinite(d)) { ok = false; d = 1.0; }
            L[j][j] = sqrt(d);
#pragma unroll
            for (int a = j + 1; a < 3; a++) {
                double v = Guu[a][j];
#pragma unroll
                for (int k = 0; k < j; k++) v -= L[a][k]*L[j][k];
                L[a][j] = v/L[j][j];
            }
        }
        /* L^-1 (lower), Guu^-1 = L^-T L^-1 */
#pragma unroll
        for (int j = 0; j < 3; j++) {
            Li[j][j] = 1.0/L[j][j];
#pragma unroll
            for (int a = j + 1; a < 3; a++) {
                double v = 0;
#pragma unroll
                for (int k = j; k < a; k++) v -= L[a][k]*Li[k][j];
                Li[a][j] = v/L[a][a];
            }
        }
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = a; b < 3; b++) Gi[sy(a, b)] = Li[0][a]*Li[0][b] + Li[1][a]*Li[1][b] + Li[2][a]*Li[2][b];
        /* K = -Guu^-1 Gux */
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) K[a][b] = -(Gi[sy(a, 0)]*Gxu[b][0] + Gi[sy(a, 1)]*Gxu[b][1] + Gi[sy(a, 2)]*Gxu[b][2]);
        /* vectors with p+ = 0: P' r through the rows */
        double pr[3], gx[3], gu[3];
#pragma unroll
        for (int a = 0; a < 3; a++) pr[a] = Pp[a][0]*g.r[0] + Pp[a][1]*g.r[1] + Pp[a][2]*g.r[2];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            gx[a] = g.hx[a] + g.Fx[0][a]*pr[0] + g.Fx[1][a]*pr[1] + g.Fx[2][a]*pr[2];
            gu[a] = g.hu[a] + g.Fu[0][a]*pr[0] + g.Fu[1][a]*pr[1] + g.Fu[2][a]*pr[2];
        }
#pragma unroll
        for (int a = 0; a < 3; a++) k0[a] = -(Gi[sy(a, 0)]*gu[0] + Gi[sy(a, 1)]*gu[1] + Gi[sy(a, 2)]*gu[2]);
#pragma unroll
        for (int a = 0; a < 3; a++) gam[a] = gx[a] + Gxu[a][0]*k0[0] + Gxu[a][1]*k0[1] + Gxu[a][2]*k0[2];
        /* P_i = Gxx + Gxu K */
        double Pn[3][3];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) Pn[a][b] = Gxx[a][b] + Gxu[a][0]*K[0][b] + Gxu[a][1]*K[1][b] + Gxu[a][2]*K[2][b];
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) P[a][b] = 0.5*(Pn[a][b] + Pn[b][a]);
        return ok;
    }

    /* what both affine recursions need of a stage whose block holds the feedback and P+: W, M, Phi = Fx + Fu K */
    __device__ static __forceinline__ void stage_maps(const double *s, const Stage &g, double (&M)[3], double (&W)[3][3], double (&Phi)[3][3], double (&K)[3][3])
    {
        double Pn[3][3];
        const double pk[6] = {s[S_PN + 0], s[S_PN + 1], s[S_PN + 2], s[S_PN + 3], s[S_PN + 4], s[S_PN + 5]};
        unpack(pk, Pn);
        relax(Pn, g.Dt, g.Db, M, W);
        K[0][0] = s[S_K + 0]; K[0][1] = s[S_K + 1]; K[0][2] = s[S_K + 2];
        K[1][0] = s[S_K + 3]; K[1][1] = s[S_K + 4]; K[1][2] = s[S_K + 5];
        K[2][0] = s[S_KS + 0]; K[2][1] = s[S_KS + 1]; K[2][2] = s[S_KS + 2];
        mul(g.Fu, K, Phi);
#pragma unroll
        for (int a = 0; a < 3; a++)
#pragma unroll
            for (int b = 0; b < 3; b++) Phi[a][b] += g.Fx[a][b];
    }

    /* returns 1 solved, 0 wrong inertia (a pivot of the recursion is not positive: delta_w), -1 breakdown of the scan (the caller sweeps serially) */
    __device__ static __noinline__ int solve(const int N, const bool pn, Ctx c, const double *Dtv, const double *Dbv)
    {
        double *S = c.S;
        if (N < 3) return -1;
        bool ok = true, bad = false;
        double *sT = S + N*S_STRIDE;

        /* ---- the last interval on top of the terminal value function: riccati_resto's stage, one thread ---- */
        if (c.tid == 0) {
            RestoSweep w;
#pragma unroll
            for (int a = 0; a < 3; a++) {
                w.pv[a] = 0;
#pragma unroll
                for (int b = 0; b < 3; b++) w.P[a][b] = 0;
            }
            w.P[0][0] = sT[S_HTT]; w.pv[0] = sT[S_HT];
            w.ok = true; w.swapLast = false;
            resto_backward_stage<DYN, true>(N - 1, pn, S, Dtv, Dbv, nullptr, 0, w);
#pragma unroll
            for (int a = 0; a < 3; a++) {
                sT[T_PV + a] = w.pv[a];
#pragma unroll
                for (int b = a; b < 3; b++) sT[T_PN + sy(a, b)] = w.P[a][b];
            }
            sT[T_OK] = w.ok ? 1.0 : 0.0; sT[T_SWAP] = w.swapLast ? 1.0 : 0.0;
        }
        __syncthreads();
        double Pn[6], pvn[3];
#pragma unroll
        for (int k = 0; k < 6; k++) Pn[k] = sT[T_PN + k];
#pragma unroll
        for (int k = 0; k < 3; k++) pvn[k] = sT[T_PV + k];
        if (!(sT[T_OK] != 0.0)) ok = false;
        const bool swapLast = sT[T_SWAP] != 0.0;

        /* ---- 1: triples of the own stages ---- */
        const int lo = c.tid*SPT;
        const int cnt = (N - 1 - lo < 0) ? 0 : (N - 1 - lo > SPT ? SPT : N - 1 - lo);
        Elem agg;
        elem_identity(agg);
        {
            bool have = false;
#pragma unroll 1
            for (int j = SPT - 1; j >= 0; j--) {
                if (j >= cnt) continue;
                Stage g; Elem e;
                load(S + (lo + j)*S_STRIDE, Dtv[lo + j], Dbv[lo + j], pn, true, g);
                const double det = stage_elem(g, e);
                if (!(fabs(det) > 0) || !isfinite(det)) bad = true;
                if (have) { const double dm = combine(e, agg, agg); if (!(fabs(dm) > 0)) bad = true; }
                else agg = e;
                have = true;
            }
        }
        /* ---- 2: suffix scan over the lanes of the wave, then over the waves ---- */
#pragma unroll 1
        for (int d = 1; d < 64; d <<= 1) {
            double mine[21], theirs[21];
            PR::pack_elem(agg, mine);
            wave_fetch<21>(mine, theirs, c.lane + d);
            if (c.lane + d < 64 && (c.tid + d)*SPT < N - 1 && cnt > 0) {
                Elem o;
                PR::unpack_elem(theirs, o);
                const double dm = combine(agg, o, agg);
                if (!(fabs(dm) > 0)) bad = true;
            }
        }
        double Pb[6];
#pragma unroll
        for (int k = 0; k < 6; k++) Pb[k] = Pn[k];
        if (c.nw > 1) {
            if (c.lane == 0) {
                double a[21];
                PR::pack_elem(agg, a);
#pragma unroll
                for (int k = 0; k < 21; k++) c.red[PR::RED_MAT + 21*c.wave + k] = a[k];
            }
            __syncthreads();
            for (int w = c.nw - 1; w > c.wave; w--) {
                if (w*64*SPT >= N - 1) continue;
                double a[21];
#pragma unroll
                for (int k = 0; k < 21; k++) a[k] = c.red[PR::RED_MAT + 21*w + k];
                Elem t;
                PR::unpack_elem(a, t);
                const double dm = combine_value(t, Pb, Pb);
                if (!(fabs(dm) > 0)) bad = true;
            }
        }
        double Ps[6], Pe[6];
        if (cnt > 0) { const double dm = combine_value(agg, Pb, Ps); if (!(fabs(dm) > 0)) bad = true; }
        else {
#pragma unroll
            for (int k = 0; k < 6; k++) Ps[k] = Pb[k];
        }
        wave_fetch<6>(Ps, Pe, c.lane + 1);
        if (c.lane == 63) {
#pragma unroll
            for (int k = 0; k < 6; k++) Pe[k] = Pb[k];
        }
        if (cnt > 0 && !PR::finite6(Pe)) bad = true;

        /* ---- 3: the recursion over the own stages (matrix part), chunk map of the gradient ---- */
        double Giv[SPT][6], k0v[SPT][3], gamv[SPT][3];
        Aff bmap;
        aff_identity(bmap);
        {
            double P[3][3];
            unpack(Pe, P);
#pragma unroll 1
            for (int j = SPT - 1; j >= 0; j--) {
#pragma unroll
                for (int k = 0; k < 6; k++) Giv[j][k] = 0;
#pragma unroll
                for (int k = 0; k < 3; k++) { k0v[j][k] = 0; gamv[j][k] = 0; }
                if (j >= cnt) continue;
                double *s = S + (lo + j)*S_STRIDE;
                Stage g;
                load(s, Dtv[lo + j], Dbv[lo + j], pn, true, g);
                double K[3][3], M[3], W[3][3], Phi[3][3];
                const double pk[6] = {P[0][0], P[0][1], P[0][2], P[1][1], P[1][2], P[2][2]};
                relax(P, g.Dt, g.Db, M, W);      /* (of P+, before the step overwrites it) */
                if (!stage_matrix(g, P, K, Giv[j], k0v[j], gamv[j])) ok = false;
                /* the block from here on: P+, feedback (the H entries are used up) */
#pragma unroll
                for (int k = 0; k < 6; k++) s[S_PN + k] = pk[k];
                s[S_K + 0] = K[0][0]; s[S_K + 1] = K[0][1]; s[S_K + 2] = K[0][2]; s[S_K + 3] = K[1][0]; s[S_K + 4] = K[1][1]; s[S_K + 5] = K[1][2];
                s[S_KS + 0] = K[2][0]; s[S_KS + 1] = K[2][1]; s[S_KS + 2] = K[2][2];
                /* p_i = gamma + (Phi^T W) p+ */
                mul(g.Fu, K, Phi);
#pragma unroll
                for (int a = 0; a < 3; a++)
#pragma unroll
                    for (int b = 0; b < 3; b++) Phi[a][b] += g.Fx[a][b];
                Aff st;
                mulT(Phi, W, st.M);
                st.v[0] = gamv[j][0]; st.v[1] = gamv[j][1]; st.v[2] = gamv[j][2];
                aff_compose(st, bmap, bmap);
            }
        }
        {
            double v[2] = {ok ? 0.0 : 1.0, bad ? 1.0 : 0.0};
            if (c.nw > 1) __syncthreads();      /* the wave totals of the scan share the reduction scratch */
            block_reduce<2>(v, OpMax(), c);
            if (c.nw > 1) __syncthreads();
            if (uni(v[1]) != 0.0) return -1;
            if (uni(v[0]) != 0.0) return 0;
        }

        /* ---- 4: suffix scan of the gradient maps ---- */
#pragma unroll 1
        for (int d = 1; d < 64; d <<= 1) {
            double mine[12], theirs[12];
            PR::pack_aff(bmap, mine);
            wave_fetch<12>(mine, theirs, c.lane + d);
            if (c.lane + d < 64 && (c.tid + d)*SPT < N - 1 && cnt > 0) {
                Aff o;
                PR::unpack_aff(theirs, o);
                aff_compose(bmap, o, bmap);
            }
        }
        double pb[3] = {pvn[0], pvn[1], pvn[2]};
        if (c.nw > 1) {
            if (c.lane == 0) {
                double a[12];
                PR::pack_aff(bmap, a);
#pragma unroll
                for (int k = 0; k < 12; k++) c.red[PR::RED_AFB + 12*c.wave + k] = a[k];
            }
            __syncthreads();
            for (int w = c.nw - 1; w > c.wave; w--) {
                if (w*64*SPT >= N - 1) continue;
                double a[12];
#pragma unroll
                for (int k = 0; k < 12; k++) a[k] = c.red[PR::RED_AFB + 12*w + k];
                Aff t;
                PR::unpack_aff(a, t);
                aff_apply(t, pb, pb);
            }
        }
        double ps[3], pe[3];
        if (cnt > 0) aff_apply(bmap, pb, ps);
        else { ps[0] = pb[0]; ps[1] = pb[1]; ps[2] = pb[2]; }
        wave_fetch<3>(ps, pe, c.lane + 1);
        if (c.lane == 63) { pe[0] = pb[0]; pe[1] = pb[1]; pe[2] = pb[2]; }

        /* ---- 5: feed-forward of the own stages, closed-loop chunk map ---- */
        Aff fmap;
        aff_identity(fmap);
        {
            double p[3] = {pe[0], pe[1], pe[2]};
#pragma unroll 1
            for (int j = SPT - 1; j >= 0; j--) {
                if (j >= cnt) continue;
                double *s = S + (lo + j)*S_STRIDE;
                Stage g;
                load(s, Dtv[lo + j], Dbv[lo + j], pn, false, g);
                double M[3], W[3][3], Phi[3][3], K[3][3];
                stage_maps(s, g, M, W, Phi, K);
                s[S_PV + 0] = p[0]; s[S_PV + 1] = p[1]; s[S_PV + 2] = p[2];
                /* k = k0 - Guu^-1 Fu^T W p+ */
                double wp[3], fw[3], k[3];
#pragma unroll
                for (int a = 0; a < 3; a++) wp[a] = W[a][0]*p[0] + W[a][1]*p[1] + W[a][2]*p[2];
#pragma unroll
                for (int a = 0; a < 3; a++) fw[a] = g.Fu[0][a]*wp[0] + g.Fu[1][a]*wp[1] + g.Fu[2][a]*wp[2];
#pragma unroll
                for (int a = 0; a < 3; a++) k[a] = k0v[j][a] - (Giv[j][sy(a, 0)]*fw[0] + Giv[j][sy(a, 1)]*fw[1] + Giv[j][sy(a, 2)]*fw[2]);
                if (!pn) k[1] = 0;
                s[S_KV + 0] = k[0]; s[S_KV + 1] = k[1]; s[S_KS + 3] = k[2];
                /* x+ = W^T (Phi x + Fu k + r) - Y p+,  Y = E M E^T */
                double a0[3];
#pragma unroll
                for (int a = 0; a < 3; a++) a0[a] = g.Fu[a][0]*k[0] + g.Fu[a][1]*k[1] + g.Fu[a][2]*k[2] + g.r[a];
                Aff st;
                mulT(W, Phi, st.M);
#pragma unroll
                for (int a = 0; a < 3; a++) st.v[a] = W[0][a]*a0[0] + W[1][a]*a0[1] + W[2][a]*a0[2];
                st.v[0] -= M[0]*p[0] + M[1]*p[1]; st.v[1] -= M[1]*p[0] + M[2]*p[1];
                aff_compose(fmap, st, fmap);
                /* gradient of stage j */
                double q[3], pn_[3];
#pragma unroll
                for (int a = 0; a < 3; a++) q[a] = wp[a];
#pragma unroll
                for (int a = 0; a < 3; a++) pn_[a] = Phi[0][a]*q[0] + Phi[1][a]*q[1] + Phi[2][a]*q[2] + gamv[j][a];
                p[0] = pn_[0]; p[1] = pn_[1]; p[2] = pn_[2];
            }
        }
        /* ---- 6: prefix scan of the closed-loop maps ---- */
#pragma unroll 1
        for (int d = 1; d < 64; d <<= 1) {
            double mine[12], theirs[12];
            PR::pack_aff(fmap, mine);
            wave_fetch<12>(mine, theirs, c.lane - d);
            if (c.lane - d >= 0 && cnt > 0) {
                Aff o;
                PR::unpack_aff(theirs, o);
                aff_compose(fmap, o, fmap);
            }
        }
        double xb[3] = {0, 0, 0};
        if (c.nw > 1) {
            const int lastLane = ((N - 2)/SPT) - 64*c.wave;
            const int src = lastLane < 0 ? 0 : (lastLane > 63 ? 63 : lastLane);
            double mine[12], tot[12];
            PR::pack_aff(fmap, mine);
            wave_fetch<12>(mine, tot, src);
            if (c.lane == 0) {
#pragma unroll
                for (int k = 0; k < 12; k++) c.red[PR::RED_AFF + 12*c.wave + k] = tot[k];
            }
            __syncthreads();
            for (int w = 0; w < c.wave; w++) {
                double a[12];
#pragma unroll
                for (int k = 0; k < 12; k++) a[k] = c.red[PR::RED_AFF + 12*w + k];
                Aff t;
                PR::unpack_aff(a, t);
                aff_apply(t, xb, xb);
            }
        }
        double xe[3], xs[3];
        aff_apply(fmap, xb, xe);
        wave_fetch<3>(xe, xs, c.lane - 1);
        if (c.lane == 0) { xs[0] = xb[0]; xs[1] = xb[1]; xs[2] = xb[2]; }

        /* ---- 7: roll-out of the own stages (riccati_resto's forward stage); the owner of stage N-2 continues through the last interval ---- */
        {
            double x0 = xs[0], x1 = xs[1], x2 = xs[2];
#pragma unroll 1
            for (int j = 0; j < SPT; j++) {
                if (j >= cnt) continue;
                resto_forward_stage<DYN>(lo + j, false, pn, S, Dtv, Dbv, nullptr, 0, swapLast, x0, x1, x2);
            }
            if (cnt > 0 && lo + cnt == N - 1) {
                resto_forward_stage<DYN>(N - 1, true, pn, S, Dtv, Dbv, nullptr, 0, swapLast, x0, x1, x2);
                sT[S_DT] = x0; sT[S_DB] = 0.0; sT[S_DF] = 0.0;
            }
        }
        __syncthreads();
        return 1;
    }
};
