/*
 * msd_lossint_table.hpp -- OptionsCasadiSolver.integrateLosses together with a loss TABLE (the dynamic loss model of mseetc/efficiency.py, or any loss
 * function tabulated by the host): the loss slack of an interval bounds
 *     E_k(v_i, dt, w, f) = int_0^dt L_k(f, v(t)) dt,    dv/dt = w - rr(v) - G,  v(0) = v_i,     k = traction part / regenerative-brake part
 * (reference: mseetc/ocp.py:118-120,231-241 -> TrainIntegrator.initLosses / calcLosses, mseetc/train.py:367-413: "energyTrDot = lossesTrFun(F, vel)/totalMass";
 * the switch sits at simulations/figure6.py:178).  L_k is the specific split loss power (utils.py:197-220; loss_split in msd_kernel.hpp), the loss power is
 * L(F, v)/M with F = f M like utils.py:261-289 integrates it in the post-processing.  (The reference's NLP path hands initLosses the SPECIFIC functions and
 * scales by the mass once more, train.py:376-377: exact for losses linear in F v -- constant efficiencies, msd_lossint.hpp -- and a force argument off by
 * a factor of the mass for a table: not reproduced)
 * Integration like msd_lossint.hpp: the adaptive Dormand-Prince 5(4) pair at CVODES' tolerances (train.py:396), step control on the values of
 * (v, E_tr, E_rgb), second-order jets in (v_i, dt, w, f) carried through the accepted steps.  Compiled into the kernels instantiated with DYN = 3 only.
 * Included by msd_kernel.hpp inside namespace msd, behind the dynamic loss model.
 */
#pragma once

struct Jet4 { double v, g[4], h[10]; };      /* variables (v0, dt, w, f); h: 00 01 02 03 11 12 13 22 23 33 */
__device__ __forceinline__ constexpr int j4a(int k) { return k < 4 ? 0 : k < 7 ? 1 : k < 9 ? 2 : 3; }
__device__ __forceinline__ constexpr int j4b(int k) { return k < 4 ? k : k < 7 ? k - 3 : k < 9 ? k - 5 : 3; }
__device__ __forceinline__ constexpr int j4h(int a, int b) { return a > b ? j4h(b, a) : a == 0 ? b : a == 1 ? 3 + b : a == 2 ? 5 + b : 9; }

__device__ __forceinline__ Jet4 operator+(Jet4 a, const Jet4 &b)
{
    a.v += b.v;
#pragma unroll
    for (int k = 0; k < 4; k++) a.g[k] += b.g[k];
#pragma unroll
    for (int k = 0; k < 10; k++) a.h[k] += b.h[k];
    return a;
}
__device__ __forceinline__ Jet4 operator*(Jet4 a, double s)
{
    a.v *= s;
#pragma unroll
    for (int k = 0; k < 4; k++) a.g[k] *= s;
#pragma unroll
    for (int k = 0; k < 10; k++) a.h[k] *= s;
    return a;
}
__device__ __forceinline__ Jet4 operator+(Jet4 a, double c) { a.v += c; return a; }
__device__ __forceinline__ Jet4 operator*(const Jet4 &a, const Jet4 &b)
{
    Jet4 r;
    r.v = a.v*b.v;
#pragma unroll
    for (int k = 0; k < 4; k++) r.g[k] = a.v*b.g[k] + b.v*a.g[k];
#pragma unroll
    for (int k = 0; k < 10; k++) r.h[k] = a.v*b.h[k] + b.v*a.h[k] + a.g[j4a(k)]*b.g[j4b(k)] + a.g[j4b(k)]*b.g[j4a(k)];
    return r;
}
__device__ __forceinline__ Jet4 j4var(Jet4, double v, int k)
{
    Jet4 r;
    r.v = v;
#pragma unroll
    for (int m = 0; m < 4; m++) r.g[m] = (m == k) ? 1.0 : 0.0;
#pragma unroll
    for (int m = 0; m < 10; m++) r.h[m] = 0;
    return r;
}
__device__ __forceinline__ double j4var(double, double v, int) { return v; }
__device__ __forceinline__ Jet4 j4const(Jet4, double c) { return j4var(Jet4(), c, -1); }
__device__ __forceinline__ double j4const(double, double c) { return c; }
__device__ __forceinline__ double j4val(const Jet4 &a) { return a.v; }
__device__ __forceinline__ double j4val(double a) { return a; }
/* L(f, v) along the jet v with f = variable 3: l = {L, L_f, L_v, L_ff, L_fv, L_vv} */
__device__ __forceinline__ Jet4 j4loss(const double (&l)[6], const Jet4 &v)
{
    Jet4 r;
    r.v = l[0];
#pragma unroll
    for (int a = 0; a < 4; a++) r.g[a] = l[2]*v.g[a] + (a == 3 ? l[1] : 0.0);
#pragma unroll
    for (int k = 0; k < 10; k++) {
        const int a = j4a(k), b = j4b(k);
        r.h[k] = l[2]*v.h[k] + l[5]*v.g[a]*v.g[b] + l[4]*((a == 3 ? v.g[b] : 0.0) + (b == 3 ? v.g[a] : 0.0)) + ((a == 3 && b == 3) ? l[3] : 0.0);
    }
    return r;
}
__device__ __forceinline__ double j4loss(const double (&l)[6], double) { return l[0]; }

/* E[0] = E_tr, E[1] = E_rgb; T = Jet4: with derivatives wrt (v0, dt, w, f), T = double: values only (same steps: the step control looks at values) */
template <class T>
__device__ inline void loss_energy(const DevProb &P, const DynLoss &D, double v0, double dt0, double w0, double f0, double G, T (&E)[2])
{
    constexpr double a21 = 1.0/5, a31 = 3.0/40, a32 = 9.0/40, a41 = 44.0/45, a42 = -56.0/15, a43 = 32.0/9,
                     a51 = 19372.0/6561, a52 = -25360.0/2187, a53 = 64448.0/6561, a54 = -212.0/729,
                     a61 = 9017.0/3168, a62 = -355.0/33, a63 = 46732.0/5247, a64 = 49.0/176, a65 = -5103.0/18656,
                     b1 = 35.0/384, b3 = 500.0/1113, b4 = 125.0/192, b5 = -2187.0/6784, b6 = 11.0/84,
                     e1 = 71.0/57600, e3 = -71.0/16695, e4 = 71.0/1920, e5 = -17253.0/339200, e6 = 22.0/525, e7 = -1.0/40;
    constexpr double atol = 1e-8, rtol = 1e-6;      /* train.py:396 */
    const T dt = j4var(T(), dt0, 1), w = j4var(T(), w0, 2);
    T y[3] = {j4var(T(), v0, 0), j4const(T(), 0.0), j4const(T(), 0.0)};
    T k[7][3];
    /* d(v, E_tr, E_rgb)/dsigma = dt (w - rr(v) - G, L_tr(f, v), L_rgb(f, v)) on the unit interval */
    auto rhs = [&](const T &vj, T (&o)[3]) {
        const T acc = ((vj*(-P.sr1) + (vj*vj)*(-P.sr2)) + w) + (-P.sr0 - G);
        o[0] = dt*acc;
        const double vv = j4val(vj);
        const int iy = table_speed_cell(D, vv);
        const Jet beta = spec_losses(D, true, 0.0, vv, iy);
#pragma unroll
        for (int r = 0; r < 2; r++) {
            double l[6];
            loss_split(D, r, f0, vv, beta, l, iy);
            o[1 + r] = dt*j4loss(l, vj);
        }
    };
    double sig = 0, h = 1.0;
    rhs(y[0], k[0]);
#pragma unroll 1
    for (int step = 0; step < 100000 && sig < 1.0; step++) {
        if (sig + h > 1.0) h = 1.0 - sig;
        T s;
        s = y[0] + k[0][0]*(h*a21); rhs(s, k[1]);
        s = (y[0] + k[0][0]*(h*a31)) + k[1][0]*(h*a32); rhs(s, k[2]);
        s = ((y[0] + k[0][0]*(h*a41)) + k[1][0]*(h*a42)) + k[2][0]*(h*a43); rhs(s, k[3]);
        s = (((y[0] + k[0][0]*(h*a51)) + k[1][0]*(h*a52)) + k[2][0]*(h*a53)) + k[3][0]*(h*a54); rhs(s, k[4]);
        s = ((((y[0] + k[0][0]*(h*a61)) + k[1][0]*(h*a62)) + k[2][0]*(h*a63)) + k[3][0]*(h*a64)) + k[4][0]*(h*a65); rhs(s, k[5]);
        T yn[3];
#pragma unroll
        for (int m = 0; m < 3; m++) yn[m] = ((((y[m] + k[0][m]*(h*b1)) + k[2][m]*(h*b3)) + k[3][m]*(h*b4)) + k[4][m]*(h*b5)) + k[5][m]*(h*b6);
        const bool finite = isfinite(j4val(yn[0])) && isfinite(j4val(yn[1])) && isfinite(j4val(yn[2])) && j4val(yn[0]) > 0;
        double err = 0;
        if (finite) {
            rhs(yn[0], k[6]);
#pragma unroll
            for (int m = 0; m < 3; m++) {
                const double sc = atol + rtol*fmax(fabs(j4val(y[m])), fabs(j4val(yn[m])));
                err = fmax(err, fabs(h*(e1*j4val(k[0][m]) + e3*j4val(k[2][m]) + e4*j4val(k[3][m]) + e5*j4val(k[4][m]) + e6*j4val(k[5][m]) + e7*j4val(k[6][m]))/sc));
            }
        }
        if (finite && err <= 1.0) {
            sig += h;
#pragma unroll
            for (int m = 0; m < 3; m++) { y[m] = yn[m]; k[0][m] = k[6][m]; }      /* first same as last */
        }
        const double fac = !finite ? 0.2 : (err > 0) ? 0.9*pow(err, -0.2) : 5.0;      /* (the oracle's arithmetic: the step sequences agree) */
        h *= fmin(5.0, fmax(0.2, fac));
        if (h < 1e-14) break;      /* the step control has collapsed */
    }
    if (!(sig >= 1.0)) { y[1] = j4const(T(), NAN); y[2] = j4const(T(), NAN); }      /* not integrated to the end: no value (the line search rejects the point) */
    E[0] = y[1]; E[1] = y[2];
}
