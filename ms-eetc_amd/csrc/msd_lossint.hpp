/*
 * msd_lossint.hpp -- OptionsCasadiSolver.integrateLosses (reference: mseetc/ocp.py:28,231-241 -> TrainIntegrator.initLosses / calcLosses,
 * mseetc/train.py:367-413): the loss slack of an interval bounds the loss POWER integrated over the interval's running time
 * dt = t_{i+1} - t_i, along the speed of the time-domain model dv/dt = w - rr(v) - G started at v_i.  With constant efficiencies the loss
 * power is (1-eta)/eta f v resp. -(1-eta_r) f v (train.py:199-212), so both integrals are multiples of the distance
 * X(v_i, dt, w) = int_0^dt v.  The reference integrates with CVODES at abstol 1e-8, reltol 1e-6 (train.py:396; SUNDIALS is third-party
 * code outside the reference repository): here the adaptive Dormand-Prince 5(4) pair at those tolerances, step control on the values,
 * first and second derivatives wrt (v_i, dt, w) carried through the accepted steps.  Compiled into the kernels instantiated with
 * DYN = 2 only.  Included by msd_kernel.hpp inside namespace msd.
 */
#pragma once

struct Jet3 { double v, g[3], h[6]; };      /* h: 00 01 02 11 12 22 */

__device__ __forceinline__ Jet3 operator+(Jet3 a, const Jet3 &b)
{
    a.v += b.v;
#pragma unroll
    for (int k = 0; k < 3; k++) a.g[k] += b.g[k];
#pragma unroll
    for (int k = 0; k < 6; k++) a.h[k] += b.h[k];
    return a;
}
__device__ __forceinline__ Jet3 operator*(Jet3 a, double s)
{
    a.v *= s;
#pragma unroll
    for (int k = 0; k < 3; k++) a.g[k] *= s;
#pragma unroll
    for (int k = 0; k < 6; k++) a.h[k] *= s;
    return a;
}
__device__ __forceinline__ Jet3 operator+(Jet3 a, double c) { a.v += c; return a; }
__device__ __forceinline__ Jet3 operator*(const Jet3 &a, const Jet3 &b)
{
    constexpr int A[6] = {0, 0, 0, 1, 1, 2}, B[6] = {0, 1, 2, 1, 2, 2};
    Jet3 r;
    r.v = a.v*b.v;
#pragma unroll
    for (int k = 0; k < 3; k++) r.g[k] = a.v*b.g[k] + b.v*a.g[k];
#pragma unroll
    for (int k = 0; k < 6; k++) r.h[k] = a.v*b.h[k] + b.v*a.h[k] + a.g[A[k]]*b.g[B[k]] + a.g[B[k]]*b.g[A[k]];
    return r;
}
__device__ __forceinline__ Jet3 j3var(Jet3, double v, int k) { Jet3 r = {v, {0, 0, 0}, {0, 0, 0, 0, 0, 0}}; r.g[k] = 1; return r; }
__device__ __forceinline__ double j3var(double, double v, int) { return v; }
__device__ __forceinline__ Jet3 j3const(Jet3, double c) { return {c, {0, 0, 0}, {0, 0, 0, 0, 0, 0}}; }
__device__ __forceinline__ double j3const(double, double c) { return c; }
__device__ __forceinline__ double j3val(const Jet3 &a) { return a.v; }
__device__ __forceinline__ double j3val(double a) { return a; }

/* X(v0, dt, w) = distance covered in the time dt; T = Jet3: with derivatives wrt (v0, dt, w), T = double: value only */
template <class T>
__device__ inline T loss_distance(const DevProb &P, double v0, double dt0, double w0, double G)
{
    constexpr double a21 = 1.0/5, a31 = 3.0/40, a32 = 9.0/40, a41 = 44.0/45, a42 = -56.0/15, a43 = 32.0/9,
                     a51 = 19372.0/6561, a52 = -25360.0/2187, a53 = 64448.0/6561, a54 = -212.0/729,
                     a61 = 9017.0/3168, a62 = -355.0/33, a63 = 46732.0/5247, a64 = 49.0/176, a65 = -5103.0/18656,
                     b1 = 35.0/384, b3 = 500.0/1113, b4 = 125.0/192, b5 = -2187.0/6784, b6 = 11.0/84,
                     e1 = 71.0/57600, e3 = -71.0/16695, e4 = 71.0/1920, e5 = -17253.0/339200, e6 = 22.0/525, e7 = -1.0/40;
    constexpr double atol = 1e-8, rtol = 1e-6;      /* train.py:396 */
    const T dt = j3var(T(), dt0, 1), w = j3var(T(), w0, 2);
    T yv = j3var(T(), v0, 0), yx = j3const(T(), 0.0);
    T kv[7], kx[7];
    /* d(v, X)/dsigma = dt (w - rr(v) - G, v) on the unit interval */
    auto rhs = [&](const T &vj, T &ov, T &ox) {
        const T acc = ((vj*(-P.sr1) + (vj*vj)*(-P.sr2)) + w) + (-P.sr0 - G);
        ov = dt*acc; ox = dt*vj;
    };
    double sig = 0, h = 1.0;      /* first try: the whole interval in one step (accepted on most intervals: the distance is smooth in its arguments; round 2 started at 0.05, three steps at least) */
    rhs(yv, kv[0], kx[0]);
    for (int step = 0; step < 100000 && sig < 1.0; step++) {
        if (sig + h > 1.0) h = 1.0 - sig;
        T s;
        s = yv + kv[0]*(h*a21); rhs(s, kv[1], kx[1]);
        s = (yv + kv[0]*(h*a31)) + kv[1]*(h*a32); rhs(s, kv[2], kx[2]);
        s = ((yv + kv[0]*(h*a41)) + kv[1]*(h*a42)) + kv[2]*(h*a43); rhs(s, kv[3], kx[3]);
        s = (((yv + kv[0]*(h*a51)) + kv[1]*(h*a52)) + kv[2]*(h*a53)) + kv[3]*(h*a54); rhs(s, kv[4], kx[4]);
        s = ((((yv + kv[0]*(h*a61)) + kv[1]*(h*a62)) + kv[2]*(h*a63)) + kv[3]*(h*a64)) + kv[4]*(h*a65); rhs(s, kv[5], kx[5]);
        const T nv = ((((yv + kv[0]*(h*b1)) + kv[2]*(h*b3)) + kv[3]*(h*b4)) + kv[4]*(h*b5)) + kv[5]*(h*b6);
        const T nx = ((((yx + kx[0]*(h*b1)) + kx[2]*(h*b3)) + kx[3]*(h*b4)) + kx[4]*(h*b5)) + kx[5]*(h*b6);
        const bool finite = isfinite(j3val(nv)) && isfinite(j3val(nx));
        double err = 0;
        if (finite) {
            rhs(nv, kv[6], kx[6]);
            const double scv = atol + rtol*fmax(fabs(j3val(yv)), fabs(j3val(nv))), scx = atol + rtol*fmax(fabs(j3val(yx)), fabs(j3val(nx)));
            const double ev = h*(e1*j3val(kv[0]) + e3*j3val(kv[2]) + e4*j3val(kv[3]) + e5*j3val(kv[4]) + e6*j3val(kv[5]) + e7*j3val(kv[6]));
            const double ex = h*(e1*j3val(kx[0]) + e3*j3val(kx[2]) + e4*j3val(kx[3]) + e5*j3val(kx[4]) + e6*j3val(kx[5]) + e7*j3val(kx[6]));
            err = fmax(fabs(ev/scv), fabs(ex/scx));
        }
        if (finite && err <= 1.0) {
            sig += h;
            yv = nv; yx = nx; kv[0] = kv[6]; kx[0] = kx[6];     /* first same as last */
        }
        const double fac = !finite ? 0.2 : (err > 0) ? 0.9*pow_m02(err) : 5.0;
        h *= fmin(5.0, fmax(0.2, fac));
        if (h < 1e-14) break;      /* the step control has collapsed */
    }
    if (!(sig >= 1.0)) yx = j3const(T(), NAN);      /* not integrated to the end: no value (the line search rejects the point) */
    return yx;
}
