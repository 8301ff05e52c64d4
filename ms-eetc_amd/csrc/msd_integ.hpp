/*
 * msd_integ.hpp -- the other two integrators TrainIntegrator offers for the shooting intervals of the NLP (reference:
 * mseetc/train.py:303-322, selected by OptionsCasadiSolver.integrationMethod, ocp.py:26-27,92): collocation (casadi.simpleIRK) and
 * integration to tolerances (casadi.integrator('cvodes'); SUNDIALS is third-party code that is not part of the reference repository,
 * its role is taken by an adaptive Dormand-Prince pair at the same tolerances).  Both are written once for values (T = double: line
 * search) and for second-order jets in (b, w) (T = Jet: the evaluation with derivatives).  They are compiled into their own
 * instantiations of the solve kernel (template parameter GEN), the explicit Runge-Kutta kernels do not carry them.
 * Included by msd_kernel.hpp inside namespace msd, after the jet arithmetic and ode_b.
 */
#pragma once

constexpr int COLL_MAX = 9;      /* OptionsIRK.order <= 9 (train.py:502) */

__device__ __forceinline__ double jval(const Jet &a) { return a.v; }
__device__ __forceinline__ double jval(double a) { return a; }
__device__ __forceinline__ Jet jconst(Jet, double c) { return {c, 0, 0, 0, 0, 0}; }
__device__ __forceinline__ double jconst(double, double c) { return c; }
/* derivative component k (1..5) of a jet, component 0 = value; a double has the value only */
__device__ __forceinline__ int jcomps(const Jet &) { return 6; }
__device__ __forceinline__ int jcomps(double) { return 1; }
__device__ __forceinline__ double &jcomp(Jet &a, int k) { return (&a.v)[k]; }
__device__ __forceinline__ double &jcomp(double &a, int) { return a; }
static_assert(sizeof(Jet) == 6*sizeof(double), "jet components are addressed as an array");

/* LU factorisation with partial pivoting (in place; false = singular) and the solve with it */
__device__ inline bool lu_factor(int n, double (&A)[COLL_MAX][COLL_MAX], int (&piv)[COLL_MAX])
{
    for (int c = 0; c < n; c++) {
        int p = c; double big = fabs(A[c][c]);
        for (int r = c + 1; r < n; r++) if (fabs(A[r][c]) > big) { big = fabs(A[r][c]); p = r; }
        if (!(big > 0)) return false;
        piv[c] = p;
        if (p != c) for (int m = 0; m < n; m++) { const double x = A[c][m]; A[c][m] = A[p][m]; A[p][m] = x; }
        for (int r = c + 1; r < n; r++) {
            A[r][c] /= A[c][c];
            for (int m = c + 1; m < n; m++) A[r][m] -= A[r][c]*A[c][m];
        }
    }
    return true;
}
__device__ inline void lu_solve(int n, const double (&A)[COLL_MAX][COLL_MAX], const int (&piv)[COLL_MAX], double (&x)[COLL_MAX])
{
    for (int c = 0; c < n; c++) if (piv[c] != c) { const double y = x[c]; x[c] = x[piv[c]]; x[piv[c]] = y; }
    for (int c = 0; c < n; c++) for (int r = c + 1; r < n; r++) x[r] -= A[r][c]*x[c];
    for (int c = n - 1; c >= 0; c--) {
        for (int m = c + 1; m < n; m++) x[c] -= A[c][m]*x[m];
        x[c] /= A[c][c];
    }
}
/* the same solve for the components first..last of an array of jets (or of doubles: component 0 only) */
template <class T>
__device__ inline void lu_solve_comps(int n, const double (&A)[COLL_MAX][COLL_MAX], const int (&piv)[COLL_MAX], T (&R)[COLL_MAX], int first)
{
    double x[COLL_MAX];
    const int last = jcomps(R[0]);
    for (int k = first; k < last; k++) {
        for (int j = 0; j < n; j++) x[j] = jcomp(R[j], k);
        lu_solve(n, A, piv, x);
        for (int j = 0; j < n; j++) jcomp(R[j], k) = x[j];
    }
}

/*
 * The same factorisation and solves for a compile-time dimension D <= 4 (OptionsIRK.order 1 ... 4; the default is 2, train.py:485): every
 * index is a constant after unrolling, so the Newton system of a collocation step lives in registers -- with the run-time dimension
 * above it lives in scratch memory.  The pivot search bubbles the largest entry of the column up by compare-and-swap of neighbouring
 * candidates (same pivot as the search for the maximum; the order of the rows below it may differ, which changes nothing but rounding).
 */
template <int D> struct LuSmall { double A[D][D]; bool sw[D][D]; };

template <int D> __device__ __forceinline__ bool lu_factor_small(LuSmall<D> &L)
{
    bool ok = true;
#pragma unroll
    for (int c = 0; c < D; c++) {
#pragma unroll
        for (int r = c + 1; r < D; r++) {
            const bool s = fabs(L.A[r][c]) > fabs(L.A[c][c]);
            L.sw[c][r] = s;
#pragma unroll
            for (int m = 0; m < D; m++) { const double x = L.A[c][m], y = L.A[r][m]; L.A[c][m] = s ? y : x; L.A[r][m] = s ? x : y; }
        }
        if (!(fabs(L.A[c][c]) > 0)) ok = false;
        const double ip = 1.0/L.A[c][c];
#pragma unroll
        for (int r = c + 1; r < D; r++) {
            L.A[r][c] *= ip;
#pragma unroll
            for (int m = c + 1; m < D; m++) L.A[r][m] -= L.A[r][c]*L.A[c][m];
        }
    }
    return ok;
}
template <int D> __device__ __forceinline__ void lu_solve_small(const LuSmall<D> &L, double (&x)[D])
{
#pragma unroll
    for (int c = 0; c < D; c++) {
#pragma unroll
        for (int r = c + 1; r < D; r++) { const double a = x[c], b = x[r]; x[c] = L.sw[c][r] ? b : a; x[r] = L.sw[c][r] ? a : b; }
#pragma unroll
        for (int r = c + 1; r < D; r++) x[r] -= L.A[r][c]*x[c];
    }
#pragma unroll
    for (int c = D - 1; c >= 0; c--) {
#pragma unroll
        for (int m = c + 1; m < D; m++) x[c] -= L.A[c][m]*x[m];
        x[c] /= L.A[c][c];
    }
}
/* components first ... of an array of jets (or the value of doubles) */
template <int D> __device__ __forceinline__ void lu_solve_small_comps(const LuSmall<D> &L, Jet (&R)[D], int first)
{
#pragma unroll
    for (int k = 0; k < 6; k++) {
        if (k < first) continue;
        double x[D];
#pragma unroll
        for (int j = 0; j < D; j++) x[j] = (k == 0) ? R[j].v : (k == 1) ? R[j].g0 : (k == 2) ? R[j].g1 : (k == 3) ? R[j].h00 : (k == 4) ? R[j].h01 : R[j].h11;
        lu_solve_small<D>(L, x);
#pragma unroll
        for (int j = 0; j < D; j++) {
            if (k == 0) R[j].v = x[j]; else if (k == 1) R[j].g0 = x[j]; else if (k == 2) R[j].g1 = x[j]; else if (k == 3) R[j].h00 = x[j]; else if (k == 4) R[j].h01 = x[j]; else R[j].h11 = x[j];
        }
    }
}
template <int D> __device__ __forceinline__ void lu_solve_small_comps(const LuSmall<D> &L, double (&R)[D], int first)
{
    if (first == 0) lu_solve_small<D>(L, R);
}

/* irk_b (below) for a compile-time number of collocation points */
template <class T, int D>
__device__ __forceinline__ T irk_b_small(const DevProb &P, T b0, T w, double G, double ds, double H, T *t)
{
    constexpr int ld = D + 1;
    const double *Cg = P.coll;
    double C[ld][ld], Dv[ld];      /* C[r][j], D[r]: uniform, read once */
#pragma unroll
    for (int r = 0; r < ld; r++) {
        Dv[r] = Cg[ld*ld + r];
#pragma unroll
        for (int j = 0; j < ld; j++) C[r][j] = Cg[r*ld + j];
    }
    const double dt = H/P.numSteps, wv = jval(w);
    T xb = b0, xt = t ? *t : jconst(T(), 0.0);
    for (int k = 0; k < P.numSteps; k++) {
        double v[D], F[D];
        LuSmall<D> L = {};      /* (defined also where the Newton loop below does not run at all: OptionsIRK.maxIter < 0 is rejected by the host, which the compiler cannot know) */
        const double xv = jval(xb);
        bool valid = true;
#pragma unroll
        for (int j = 0; j < D; j++) v[j] = xv;
        for (int it = 0; it <= P.newtonIters; it++) {
            double fmaxabs = 0;
#pragma unroll
            for (int j = 0; j < D; j++) {
                const double sv = sqrt(v[j]);
                const double f = 2*ds*(wv - (P.sr0 + P.sr1*sv + P.sr2*v[j]) - G), df = -2*ds*(0.5*P.sr1/sv + P.sr2);
                double p = C[0][j + 1]*xv;
#pragma unroll
                for (int r = 0; r < D; r++) { p += C[r + 1][j + 1]*v[r]; L.A[j][r] = -C[r + 1][j + 1]; }
                L.A[j][j] += dt*df;
                F[j] = dt*f - p;
                fmaxabs = fmax(fmaxabs, fabs(F[j]));
            }
            if (!lu_factor_small<D>(L)) { valid = false; break; }
            if (it == P.newtonIters || !isfinite(fmaxabs) || fmaxabs <= 1e-13*fmax(1.0, fabs(xv))) break;
            lu_solve_small<D>(L, F);
#pragma unroll
            for (int j = 0; j < D; j++) v[j] -= F[j];
        }
        if (!valid) { if (t) *t = jconst(T(), NAN); return jconst(T(), NAN); }
        T V[D], R[D];
#pragma unroll
        for (int j = 0; j < D; j++) V[j] = jconst(T(), v[j]);
        if (jcomps(xb) > 1) {
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
#pragma unroll
                for (int j = 0; j < D; j++) {
                    T p = xb*C[0][j + 1];
#pragma unroll
                    for (int r = 0; r < D; r++) p = p + V[r]*C[r + 1][j + 1];
                    R[j] = ode_b(P, V[j], w, G, ds)*dt - p;
                }
                lu_solve_small_comps<D>(L, R, 1);
#pragma unroll
                for (int j = 0; j < D; j++) { jcomp(R[j], 0) = 0.0; V[j] = V[j] - R[j]; }
            }
        }
        T nb = xb*Dv[0];
#pragma unroll
        for (int r = 0; r < D; r++) nb = nb + V[r]*Dv[r + 1];
        if (t) {
            LuSmall<D> M;
#pragma unroll
            for (int j = 0; j < D; j++)
#pragma unroll
                for (int r = 0; r < D; r++) M.A[j][r] = C[r + 1][j + 1];
            lu_factor_small<D>(M);
#pragma unroll
            for (int j = 0; j < D; j++) R[j] = xrecip(xsqrt(V[j]))*(dt*ds) - xt*C[0][j + 1];
            lu_solve_small_comps<D>(M, R, 0);
            T nt = xt*Dv[0];
#pragma unroll
            for (int r = 0; r < D; r++) nt = nt + R[r]*Dv[r + 1];
            xt = nt;
        }
        xb = nb;
    }
    if (t) *t = xt;
    return xb;
}

/*
 * casadi.simpleIRK(ode, numSteps, d, scheme, 'fast_newton') over [0, H] (train.py:310): per step of length dt = H/numSteps the d stage
 * values v solve  dt f(v_j) - (C[0][j] x + sum_r C[r][j] v_r) = 0  (x = start of the step; first guess v_j = x), the step ends at
 * D[0] x + sum_r D[r] v_r.  Newton's method runs on the values (at most OptionsIRK.maxIter iterations; like error_on_fail = False the
 * last iterate is used).  The derivatives follow from the implicit-function theorem, applied as two Newton corrections in jet
 * arithmetic with the Jacobian at the converged values: the first makes the first derivatives exact, the second the second ones.
 * t != nullptr: the time equation dt/dsigma = ds/sqrt(b) is integrated along (numApproxSteps = 0); its stage equations are linear in
 * the time stages.  tab: C[(d+1)*(d+1)] row r column j, then D[d+1] (device memory).
 */
template <class T>
__device__ inline T irk_b(const DevProb &P, T b0, T w, double G, double ds, double H, T *t)
{
    /* the usual orders with their Newton systems in registers */
    switch (P.collD) {
    case 1: return irk_b_small<T, 1>(P, b0, w, G, ds, H, t);
    case 2: return irk_b_small<T, 2>(P, b0, w, G, ds, H, t);
    case 3: return irk_b_small<T, 3>(P, b0, w, G, ds, H, t);
    default: break;
    }
    const int d = P.collD, ld = d + 1;
    const double *C = P.coll, *D = P.coll + ld*ld;
    const double dt = H/P.numSteps, wv = jval(w);
    T xb = b0, xt = t ? *t : jconst(T(), 0.0);
    for (int k = 0; k < P.numSteps; k++) {
        double v[COLL_MAX], A[COLL_MAX][COLL_MAX], F[COLL_MAX];
        int piv[COLL_MAX];
        const double xv = jval(xb);
        bool valid = true;
        for (int j = 0; j < d; j++) v[j] = xv;
        for (int it = 0; it <= P.newtonIters; it++) {
            double fmaxabs = 0;
            for (int j = 0; j < d; j++) {
                const double sv = sqrt(v[j]);
                const double f = 2*ds*(wv - (P.sr0 + P.sr1*sv + P.sr2*v[j]) - G), df = -2*ds*(0.5*P.sr1/sv + P.sr2);
                double p = C[j + 1]*xv;
                for (int r = 0; r < d; r++) { p += C[(r + 1)*ld + j + 1]*v[r]; A[j][r] = -C[(r + 1)*ld + j + 1]; }
                A[j][j] += dt*df;
                F[j] = dt*f - p;
                fmaxabs = fmax(fmaxabs, fabs(F[j]));
            }
            /* the Jacobian of the last pass is the one the derivatives use; a stage value that left the domain (speed squared below
             * zero) makes it NaN: the step is reported as NaN, which the line search treats as an invalid trial point */
            if (!lu_factor(d, A, piv)) { valid = false; break; }
            if (it == P.newtonIters || !isfinite(fmaxabs) || fmaxabs <= 1e-13*fmax(1.0, fabs(xv))) break;
            lu_solve(d, A, piv, F);
            for (int j = 0; j < d; j++) v[j] -= F[j];
        }
        if (!valid) { if (t) *t = jconst(T(), NAN); return jconst(T(), NAN); }
        T V[COLL_MAX], R[COLL_MAX];
        for (int j = 0; j < d; j++) V[j] = jconst(T(), v[j]);
        if (jcomps(xb) > 1) {
            for (int pass = 0; pass < 2; pass++) {
                for (int j = 0; j < d; j++) {
                    T p = xb*C[j + 1];
                    for (int r = 0; r < d; r++) p = p + V[r]*C[(r + 1)*ld + j + 1];
                    R[j] = ode_b(P, V[j], w, G, ds)*dt - p;
                }
                lu_solve_comps(d, A, piv, R, 1);
                for (int j = 0; j < d; j++) { jcomp(R[j], 0) = 0.0; V[j] = V[j] - R[j]; }
            }
        }
        T nb = xb*D[0];
        for (int r = 0; r < d; r++) nb = nb + V[r]*D[r + 1];
        if (t) {
            /* sum_r C[r][j] vt_r = dt ds/sqrt(V_j) - C[0][j] xt */
            double M[COLL_MAX][COLL_MAX];
            int pm[COLL_MAX];
            for (int j = 0; j < d; j++) for (int r = 0; r < d; r++) M[j][r] = C[(r + 1)*ld + j + 1];
            lu_factor(d, M, pm);
            for (int j = 0; j < d; j++) R[j] = xrecip(xsqrt(V[j]))*(dt*ds) - xt*C[j + 1];
            lu_solve_comps(d, M, pm, R, 0);
            T nt = xt*D[0];
            for (int r = 0; r < d; r++) nt = nt + R[r]*D[r + 1];
            xt = nt;
        }
        xb = nb;
    }
    if (t) *t = xt;
    return xb;
}

/*
 * (t, b) over the unit interval to the tolerances of OptionsCVODES (train.py:312-322, :521-534): Dormand-Prince 5(4), step-size
 * control on the values; the derivatives are those of the accepted steps (the discrete map), carried in jet arithmetic.
 */
template <class T>
__device__ inline void dopri_tb_plain(const DevProb &P, T b0, T w, double G, double ds, T &tau, T &bplus)
{
    constexpr double a21 = 1.0/5, a31 = 3.0/40, a32 = 9.0/40, a41 = 44.0/45, a42 = -56.0/15, a43 = 32.0/9,
                     a51 = 19372.0/6561, a52 = -25360.0/2187, a53 = 64448.0/6561, a54 = -212.0/729,
                     a61 = 9017.0/3168, a62 = -355.0/33, a63 = 46732.0/5247, a64 = 49.0/176, a65 = -5103.0/18656,
                     b1 = 35.0/384, b3 = 500.0/1113, b4 = 125.0/192, b5 = -2187.0/6784, b6 = 11.0/84,
                     e1 = 71.0/57600, e3 = -71.0/16695, e4 = 71.0/1920, e5 = -17253.0/339200, e6 = 22.0/525, e7 = -1.0/40;
    T yt = jconst(T(), 0.0), yb = b0;
    T kt[7], kb[7];
    auto rhs = [&](const T &bj, T &ot, T &ob) { ot = xrsqrt<MSD_FAST_MATH != 0>(bj)*ds; ob = ode_b<T, MSD_FAST_MATH != 0>(P, bj, w, G, ds); };      /* (msd_fastmath.hpp) */
    double sig = 0, h = 1.0;      /* the whole interval first (1.9 instead of 3.4 sets of stages per interval on the benchmark grid: a rejected first step lands on the right size) */
    rhs(yb, kt[0], kb[0]);
    for (int step = 0; step < 100000 && sig < 1.0; step++) {
        if (sig + h > 1.0) h = 1.0 - sig;
        T s;
        s = yb + kb[0]*(h*a21); rhs(s, kt[1], kb[1]);
        s = (yb + kb[0]*(h*a31)) + kb[1]*(h*a32); rhs(s, kt[2], kb[2]);
        s = ((yb + kb[0]*(h*a41)) + kb[1]*(h*a42)) + kb[2]*(h*a43); rhs(s, kt[3], kb[3]);
        s = (((yb + kb[0]*(h*a51)) + kb[1]*(h*a52)) + kb[2]*(h*a53)) + kb[3]*(h*a54); rhs(s, kt[4], kb[4]);
        s = ((((yb + kb[0]*(h*a61)) + kb[1]*(h*a62)) + kb[2]*(h*a63)) + kb[3]*(h*a64)) + kb[4]*(h*a65); rhs(s, kt[5], kb[5]);
        const T nt = ((((yt + kt[0]*(h*b1)) + kt[2]*(h*b3)) + kt[3]*(h*b4)) + kt[4]*(h*b5)) + kt[5]*(h*b6);
        const T nb = ((((yb + kb[0]*(h*b1)) + kb[2]*(h*b3)) + kb[3]*(h*b4)) + kb[4]*(h*b5)) + kb[5]*(h*b6);
        const bool finite = isfinite(jval(nt)) && isfinite(jval(nb)) && jval(nb) > 0;
        double err = 0;
        if (finite) {
            rhs(nb, kt[6], kb[6]);
            const double sct = P.intAtol + P.intRtol*fmax(fabs(jval(yt)), fabs(jval(nt))), scb = P.intAtol + P.intRtol*fmax(fabs(jval(yb)), fabs(jval(nb)));
            const double et = h*(e1*jval(kt[0]) + e3*jval(kt[2]) + e4*jval(kt[3]) + e5*jval(kt[4]) + e6*jval(kt[5]) + e7*jval(kt[6]));
            const double eb = h*(e1*jval(kb[0]) + e3*jval(kb[2]) + e4*jval(kb[3]) + e5*jval(kb[4]) + e6*jval(kb[5]) + e7*jval(kb[6]));
            err = fmax(fabs(et/sct), fabs(eb/scb));
        }
        if (finite && err <= 1.0) {
            sig += h;
            yt = nt; yb = nb; kt[0] = kt[6]; kb[0] = kb[6];     /* first same as last */
        }
        const double fac = !finite ? 0.2 : (err > 0) ? 0.9*pow_m02(err) : 5.0;
        h *= fmin(5.0, fmax(0.2, fac));
        if (h < 1e-14) break;      /* the step control has collapsed (the state left the model's domain) */
    }
    /* an integration that did not reach the end of the interval is no interval map: NaN, so that the line search rejects the point
     * (like the collocation step, irk_b) instead of taking a partially integrated state for it */
    if (!(sig >= 1.0)) { yt = jconst(T(), NAN); yb = jconst(T(), NAN); }
    tau = yt; bplus = yb;
}

/*
 * The same with the values going first (round 5): the step-size controller sees values only, so a set of stages is first run in value arithmetic --
 * a fifth of what it costs in jets -- and only an accepted step is run again with the jets (1.9 sets of stages per interval were jets before, one
 * of them a rejected first step); the stage after an accepted step ("first same as last") is evaluated as a jet only when another step follows.
 * The accepted steps, and with them the discrete map and its derivatives, are what dopri_tb_plain computes.
 */
#ifndef MSD_DOPRI_VALUES_FIRST
#define MSD_DOPRI_VALUES_FIRST 1
#endif
__device__ inline void dopri_tb_jet(const DevProb &P, Jet b0, Jet w, double G, double ds, Jet &tau, Jet &bplus)
{
    constexpr double a21 = 1.0/5, a31 = 3.0/40, a32 = 9.0/40, a41 = 44.0/45, a42 = -56.0/15, a43 = 32.0/9,
                     a51 = 19372.0/6561, a52 = -25360.0/2187, a53 = 64448.0/6561, a54 = -212.0/729,
                     a61 = 9017.0/3168, a62 = -355.0/33, a63 = 46732.0/5247, a64 = 49.0/176, a65 = -5103.0/18656,
                     b1 = 35.0/384, b3 = 500.0/1113, b4 = 125.0/192, b5 = -2187.0/6784, b6 = 11.0/84,
                     e1 = 71.0/57600, e3 = -71.0/16695, e4 = 71.0/1920, e5 = -17253.0/339200, e6 = 22.0/525, e7 = -1.0/40;
    constexpr bool FM = MSD_FAST_MATH != 0;
    const double wv = w.v;
    auto rhsv = [&](const double bj, double &ot, double &ob) { ot = xrsqrt<FM>(bj)*ds; ob = ode_b<double, FM>(P, bj, wv, G, ds); };
    auto rhsj = [&](const Jet &bj, Jet &ot, Jet &ob) { ot = xrsqrt<FM>(bj)*ds; ob = ode_b<Jet, FM>(P, bj, w, G, ds); };
    Jet yt = jconst(Jet(), 0.0), yb = b0;
    Jet kt[6], kb[6];
    double sig = 0, h = 1.0;
    rhsj(yb, kt[0], kb[0]);
    for (int step = 0; step < 100000 && sig < 1.0; step++) {
        if (sig + h > 1.0) h = 1.0 - sig;
        /* the step in values: error estimate, acceptance */
        double vt[7], vb[7], sv;
        const double ytv = yt.v, ybv = yb.v;
        vt[0] = kt[0].v; vb[0] = kb[0].v;
        sv = ybv + vb[0]*(h*a21); rhsv(sv, vt[1], vb[1]);
        sv = (ybv + vb[0]*(h*a31)) + vb[1]*(h*a32); rhsv(sv, vt[2], vb[2]);
        sv = ((ybv + vb[0]*(h*a41)) + vb[1]*(h*a42)) + vb[2]*(h*a43); rhsv(sv, vt[3], vb[3]);
        sv = (((ybv + vb[0]*(h*a51)) + vb[1]*(h*a52)) + vb[2]*(h*a53)) + vb[3]*(h*a54); rhsv(sv, vt[4], vb[4]);
        sv = ((((ybv + vb[0]*(h*a61)) + vb[1]*(h*a62)) + vb[2]*(h*a63)) + vb[3]*(h*a64)) + vb[4]*(h*a65); rhsv(sv, vt[5], vb[5]);
        const double ntv = ((((ytv + vt[0]*(h*b1)) + vt[2]*(h*b3)) + vt[3]*(h*b4)) + vt[4]*(h*b5)) + vt[5]*(h*b6);
        const double nbv = ((((ybv + vb[0]*(h*b1)) + vb[2]*(h*b3)) + vb[3]*(h*b4)) + vb[4]*(h*b5)) + vb[5]*(h*b6);
        const bool finite = isfinite(ntv) && isfinite(nbv) && nbv > 0;
        double err = 0;
        if (finite) {
            rhsv(nbv, vt[6], vb[6]);
            const double sct = P.intAtol + P.intRtol*fmax(fabs(ytv), fabs(ntv)), scb = P.intAtol + P.intRtol*fmax(fabs(ybv), fabs(nbv));
            const double et = h*(e1*vt[0] + e3*vt[2] + e4*vt[3] + e5*vt[4] + e6*vt[5] + e7*vt[6]);
            const double eb = h*(e1*vb[0] + e3*vb[2] + e4*vb[3] + e5*vb[4] + e6*vb[5] + e7*vb[6]);
            err = fmax(fabs(et/sct), fabs(eb/scb));
        }
        if (finite && err <= 1.0) {
            /* accepted: the same stages as jets */
            Jet s;
            s = yb + kb[0]*(h*a21); rhsj(s, kt[1], kb[1]);
            s = (yb + kb[0]*(h*a31)) + kb[1]*(h*a32); rhsj(s, kt[2], kb[2]);
            s = ((yb + kb[0]*(h*a41)) + kb[1]*(h*a42)) + kb[2]*(h*a43); rhsj(s, kt[3], kb[3]);
            s = (((yb + kb[0]*(h*a51)) + kb[1]*(h*a52)) + kb[2]*(h*a53)) + kb[3]*(h*a54); rhsj(s, kt[4], kb[4]);
            s = ((((yb + kb[0]*(h*a61)) + kb[1]*(h*a62)) + kb[2]*(h*a63)) + kb[3]*(h*a64)) + kb[4]*(h*a65); rhsj(s, kt[5], kb[5]);
            yt = ((((yt + kt[0]*(h*b1)) + kt[2]*(h*b3)) + kt[3]*(h*b4)) + kt[4]*(h*b5)) + kt[5]*(h*b6);
            yb = ((((yb + kb[0]*(h*b1)) + kb[2]*(h*b3)) + kb[3]*(h*b4)) + kb[4]*(h*b5)) + kb[5]*(h*b6);
            sig += h;
            if (sig < 1.0) rhsj(yb, kt[0], kb[0]);     /* first same as last */
        }
        const double fac = !finite ? 0.2 : (err > 0) ? 0.9*pow_m02(err) : 5.0;
        h *= fmin(5.0, fmax(0.2, fac));
        if (h < 1e-14) break;      /* the step control has collapsed (the state left the model's domain) */
    }
    if (!(sig >= 1.0)) { yt = jconst(Jet(), NAN); yb = jconst(Jet(), NAN); }
    tau = yt; bplus = yb;
}

/*
 * The pieces of dopri_tb_jet for the cooperative evaluation of Solver::evaluate_current (round 5).  On the benchmark grid two intervals of a horizon
 * -- the first and the last, at 1 m/s -- take 25 ... 28 accepted steps where the others take one or two, and the lanes of their waves wait while one
 * lane runs 26 sets of stages in jets.  The steps themselves are sequential, their derivatives are not: with the values at the step boundaries known
 * from the value pass, the local second-order jet of every step (seeded at its own starting value) can be evaluated by another lane, and the owner
 * composes the local jets by the chain rule.
 *   DopriValues: the controller in value arithmetic, one attempt per call (same arithmetic as the value stages of dopri_tb_jet / dopri_tb_plain<double>)
 *   dopri_step_jet: one step of size h in jets from the state yb; returns b+ and the step's share of the running time
 *   jet_after: chain rule for a local jet phi(b, w) behind the running jet B(b0, w)
 */
struct DopriValues {
    double sig, h, yt, yb, kt0, kb0;
    __device__ inline void start(const DevProb &P, double b0, double wv, double G, double ds)
    {
        sig = 0; h = 1.0; yt = 0; yb = b0;
        kt0 = xrsqrt<MSD_FAST_MATH != 0>(b0)*ds; kb0 = ode_b<double, MSD_FAST_MATH != 0>(P, b0, wv, G, ds);
    }
    __device__ inline bool finished() const { return sig >= 1.0; }
    /* one attempt: 1 accepted (h_used, y_before describe the step), 0 rejected, -1 the step control has collapsed */
    __device__ inline int attempt(const DevProb &P, double wv, double G, double ds, double &h_used, double &y_before)
    {
        constexpr double a21 = 1.0/5, a31 = 3.0/40, a32 = 9.0/40, a41 = 44.0/45, a42 = -56.0/15, a43 = 32.0/9,
                         a51 = 19372.0/6561, a52 = -25360.0/2187, a53 = 64448.0/6561, a54 = -212.0/729,
                         a61 = 9017.0/3168, a62 = -355.0/33, a63 = 46732.0/5247, a64 = 49.0/176, a65 = -5103.0/18656,
                         b1 = 35.0/384, b3 = 500.0/1113, b4 = 125.0/192, b5 = -2187.0/6784, b6 = 11.0/84,
                         e1 = 71.0/57600, e3 = -71.0/16695, e4 = 71.0/1920, e5 = -17253.0/339200, e6 = 22.0/525, e7 = -1.0/40;
        constexpr bool FM = MSD_FAST_MATH != 0;
        auto rhsv = [&](const double bj, double &ot, double &ob) { ot = xrsqrt<FM>(bj)*ds; ob = ode_b<double, FM>(P, bj, wv, G, ds); };
        if (sig + h > 1.0) h = 1.0 - sig;
        double vt[7], vb[7], sv;
        vt[0] = kt0; vb[0] = kb0;
        sv = yb + vb[0]*(h*a21); rhsv(sv, vt[1], vb[1]);
        sv = (yb + vb[0]*(h*a31)) + vb[1]*(h*a32); rhsv(sv, vt[2], vb[2]);
        sv = ((yb + vb[0]*(h*a41)) + vb[1]*(h*a42)) + vb[2]*(h*a43); rhsv(sv, vt[3], vb[3]);
        sv = (((yb + vb[0]*(h*a51)) + vb[1]*(h*a52)) + vb[2]*(h*a53)) + vb[3]*(h*a54); rhsv(sv, vt[4], vb[4]);
        sv = ((((yb + vb[0]*(h*a61)) + vb[1]*(h*a62)) + vb[2]*(h*a63)) + vb[3]*(h*a64)) + vb[4]*(h*a65); rhsv(sv, vt[5], vb[5]);
        const double ntv = ((((yt + vt[0]*(h*b1)) + vt[2]*(h*b3)) + vt[3]*(h*b4)) + vt[4]*(h*b5)) + vt[5]*(h*b6);
        const double nbv = ((((yb + vb[0]*(h*b1)) + vb[2]*(h*b3)) + vb[3]*(h*b4)) + vb[4]*(h*b5)) + vb[5]*(h*b6);
        const bool finite = isfinite(ntv) && isfinite(nbv) && nbv > 0;
        double err = 0;
        if (finite) {
            rhsv(nbv, vt[6], vb[6]);
            const double sct = P.intAtol + P.intRtol*fmax(fabs(yt), fabs(ntv)), scb = P.intAtol + P.intRtol*fmax(fabs(yb), fabs(nbv));
            const double et = h*(e1*vt[0] + e3*vt[2] + e4*vt[3] + e5*vt[4] + e6*vt[5] + e7*vt[6]);
            const double eb = h*(e1*vb[0] + e3*vb[2] + e4*vb[3] + e5*vb[4] + e6*vb[5] + e7*vb[6]);
            err = fmax(fabs(et/sct), fabs(eb/scb));
        }
        int rc = 0;
        if (finite && err <= 1.0) {
            h_used = h; y_before = yb;
            sig += h; yt = ntv; yb = nbv; kt0 = vt[6]; kb0 = vb[6];     /* first same as last */
            rc = 1;
        }
        const double fac = !finite ? 0.2 : (err > 0) ? 0.9*pow_m02(err) : 5.0;
        h *= fmin(5.0, fmax(0.2, fac));
        if (h < 1e-14 && !finished()) rc = -1;
        return rc;
    }
};

__device__ inline void dopri_step_jet(const DevProb &P, Jet &yb, Jet &dtau, const Jet w, double G, double ds, double h)
{
    constexpr double a21 = 1.0/5, a31 = 3.0/40, a32 = 9.0/40, a41 = 44.0/45, a42 = -56.0/15, a43 = 32.0/9,
                     a51 = 19372.0/6561, a52 = -25360.0/2187, a53 = 64448.0/6561, a54 = -212.0/729,
                     a61 = 9017.0/3168, a62 = -355.0/33, a63 = 46732.0/5247, a64 = 49.0/176, a65 = -5103.0/18656,
                     b1 = 35.0/384, b3 = 500.0/1113, b4 = 125.0/192, b5 = -2187.0/6784, b6 = 11.0/84;
    constexpr bool FM = MSD_FAST_MATH != 0;
    auto rhsj = [&](const Jet &bj, Jet &ot, Jet &ob) { ot = xrsqrt<FM>(bj)*ds; ob = ode_b<Jet, FM>(P, bj, w, G, ds); };
    Jet kt[6], kb[6], s;
    rhsj(yb, kt[0], kb[0]);
    s = yb + kb[0]*(h*a21); rhsj(s, kt[1], kb[1]);
    s = (yb + kb[0]*(h*a31)) + kb[1]*(h*a32); rhsj(s, kt[2], kb[2]);
    s = ((yb + kb[0]*(h*a41)) + kb[1]*(h*a42)) + kb[2]*(h*a43); rhsj(s, kt[3], kb[3]);
    s = (((yb + kb[0]*(h*a51)) + kb[1]*(h*a52)) + kb[2]*(h*a53)) + kb[3]*(h*a54); rhsj(s, kt[4], kb[4]);
    s = ((((yb + kb[0]*(h*a61)) + kb[1]*(h*a62)) + kb[2]*(h*a63)) + kb[3]*(h*a64)) + kb[4]*(h*a65); rhsj(s, kt[5], kb[5]);
    dtau = ((((kt[0]*(h*b1)) + kt[2]*(h*b3)) + kt[3]*(h*b4)) + kt[4]*(h*b5)) + kt[5]*(h*b6);
    yb = ((((yb + kb[0]*(h*b1)) + kb[2]*(h*b3)) + kb[3]*(h*b4)) + kb[4]*(h*b5)) + kb[5]*(h*b6);
}

/* phi(b, w) behind B(b0, w): value, gradient and Hessian of phi(B(b0, w), w) in (b0, w) */
__device__ inline Jet jet_after(const Jet &phi, const Jet &B)
{
    Jet r;
    r.v = phi.v;
    r.g0 = phi.g0*B.g0;
    r.g1 = phi.g0*B.g1 + phi.g1;
    r.h00 = phi.h00*B.g0*B.g0 + phi.g0*B.h00;
    r.h01 = phi.h00*B.g0*B.g1 + phi.h01*B.g0 + phi.g0*B.h01;
    r.h11 = phi.h00*B.g1*B.g1 + 2*phi.h01*B.g1 + phi.h11 + phi.g0*B.h11;
    return r;
}

__device__ inline void dopri_tb(const DevProb &P, double b0, double w, double G, double ds, double &tau, double &bplus) { dopri_tb_plain<double>(P, b0, w, G, ds, tau, bplus); }
__device__ inline void dopri_tb(const DevProb &P, Jet b0, Jet w, double G, double ds, Jet &tau, Jet &bplus)
{
    if (MSD_DOPRI_VALUES_FIRST) dopri_tb_jet(P, b0, w, G, ds, tau, bplus); else dopri_tb_plain<Jet>(P, b0, w, G, ds, tau, bplus);
}

/* one shooting interval with the integrator the problem names (P.integ: MSD_INTEGRATOR_ADAPTIVE or MSD_INTEGRATOR_COLLOCATION) */
template <class T>
__device__ inline void interval_map_general(const DevProb &P, double b0, double w0, double G, double ds, T &tau, T &bplus)
{
    const T b = make_var(T(), b0, 0), w = make_var(T(), w0, 1);
    if (P.integ == MSD_INTEGRATOR_ADAPTIVE) { dopri_tb(P, b, w, G, ds, tau, bplus); return; }     /* train.py:314: time integrated along */
    if (P.numApprox == 0) {
        T t = jconst(T(), 0.0);
        bplus = irk_b<T>(P, b, w, G, ds, 1.0, &t);
        tau = t;
        return;
    }
    const int ns = P.numApprox;                                   /* train.py:324-344 */
    T prev = b, acc = jconst(T(), 0.0);
    for (int j = 1; j <= ns; j++) {
        const T cur = irk_b<T>(P, b, w, G, ds, (double)j/ns, (T *)nullptr);
        acc = acc + xrecip(xsqrt(prev) + xsqrt(cur))*(2*ds*((double)j/ns - (double)(j - 1)/ns));
        prev = cur;
    }
    tau = acc; bplus = prev;
}
