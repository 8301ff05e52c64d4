/*
 * msd_integrators.hip -- the other two interval integrators of TrainIntegrator (mseetc/train.py:303-322), for arrays of
 * independent intervals (one thread per interval).  They serve TrainIntegrator.solve (train.py:347-364) the way
 * simulations/figure4.py uses it -- accuracy comparisons of single intervals -- not the OCP transcription, which stays on
 * the explicit Runge-Kutta map (msd_kernel.hpp).
 *
 *   adaptive  ('CVODES', train.py:314-322): the space-domain ODE over the unit interval,
 *                 dt/dsigma = ds/sqrt(b),  db/dsigma = 2 ds (w - sr0 - sr1 sqrt(b) - sr2 b - G)           (train.py:251-259)
 *             integrated to the caller's absolute/relative tolerances.  The ODE is non-stiff and has two states, so an
 *             adaptive Dormand-Prince 5(4) pair stands in for SUNDIALS' BDF code (same tolerances, same error norm per step).
 *   collocation ('IRK', train.py:303-310 -> casadi.simpleIRK): `numSteps` collocation steps of degree `order` on Radau or
 *             Legendre points; per step the equations  dt f(v_j) - sum_r C[r][j] x_r = 0  (x_0 = start of the step) are solved
 *             by Newton's method from v_j = x_0 (casadi's 'fast_newton' rootfinder, at most `maxIter` iterations), then
 *             x+ = sum_r D[r] x_r.  With numApproxSteps > 0 only b is integrated and the time follows from the trapezoidal
 *             rule on the sub-interval speeds, exactly like the explicit integrator (train.py:324-344).
 *             The interpolation matrices C, D come from the host (mseetc/train.py: collocationTables).
 */
#include <hip/hip_runtime.h>

#include <cmath>
#include <string>

#include "../../include/mseetc_hip.h"
#include "../../include/mseetc_aux.h"
#include "msd_fastmath.hpp"

namespace {

struct IvTrain { double sr0, sr1, sr2, g, rho; };

constexpr int MAX_ORDER = 9, MAX_SYS = 2*MAX_ORDER;

__device__ __forceinline__ double iv_resistance(const IvTrain &T, double grad, double curv)
{
    const double c = fabs(curv);
    const double cr = (c <= 1.0/300.0) ? T.g*0.5*c/(1 - 30*c) : T.g*0.65*c/(1 - 55*c);   /* train.py:252-253 as written */
    return T.g*grad*(1/T.rho) + cr*(1/T.rho);
}

/* f(t, b) over the unit interval and its derivative with respect to b (nothing depends on t) */
__device__ __forceinline__ void iv_rhs(const IvTrain &T, double ds, double w, double G, double b, double &ft, double &fb, double &dft, double &dfb)
{
    const double v = sqrt(b);
    ft = ds/v;
    fb = 2*ds*(w - (T.sr0 + T.sr1*v + T.sr2*b) - G);
    dft = -0.5*ds/(b*v);
    dfb = -2*ds*(0.5*T.sr1/v + T.sr2);
}

/* ---- adaptive Dormand-Prince 5(4) on (t, b) ---- */
__device__ int iv_dopri(const IvTrain &T, double ds, double w, double G, double &t, double &b, double atol, double rtol)
{
    const double a21 = 1.0/5, a31 = 3.0/40, a32 = 9.0/40, a41 = 44.0/45, a42 = -56.0/15, a43 = 32.0/9,
                 a51 = 19372.0/6561, a52 = -25360.0/2187, a53 = 64448.0/6561, a54 = -212.0/729,
                 a61 = 9017.0/3168, a62 = -355.0/33, a63 = 46732.0/5247, a64 = 49.0/176, a65 = -5103.0/18656,
                 b1 = 35.0/384, b3 = 500.0/1113, b4 = 125.0/192, b5 = -2187.0/6784, b6 = 11.0/84,
                 e1 = 71.0/57600, e3 = -71.0/16695, e4 = 71.0/1920, e5 = -17253.0/339200, e6 = 22.0/525, e7 = -1.0/40;
    double y[2] = {t, b}, k[7][2], yt[2], yn[2], d0, d1;
    auto f = [&](const double (&x)[2], double (&out)[2]) { iv_rhs(T, ds, w, G, x[1], out[0], out[1], d0, d1); };
    double tau = 0, h = 1.0;      /* the whole interval first, like the interval map of the NLP (msd_integ.hpp: dopri_tb) -- the two take the same steps */
    f(y, k[0]);
    for (int step = 0; step < 400000 && tau < 1.0; step++) {
        if (tau + h > 1.0) h = 1.0 - tau;
        for (int m = 0; m < 2; m++) yt[m] = y[m] + h*a21*k[0][m];
        f(yt, k[1]);
        for (int m = 0; m < 2; m++) yt[m] = y[m] + h*(a31*k[0][m] + a32*k[1][m]);
        f(yt, k[2]);
        for (int m = 0; m < 2; m++) yt[m] = y[m] + h*(a41*k[0][m] + a42*k[1][m] + a43*k[2][m]);
        f(yt, k[3]);
        for (int m = 0; m < 2; m++) yt[m] = y[m] + h*(a51*k[0][m] + a52*k[1][m] + a53*k[2][m] + a54*k[3][m]);
        f(yt, k[4]);
        for (int m = 0; m < 2; m++) yt[m] = y[m] + h*(a61*k[0][m] + a62*k[1][m] + a63*k[2][m] + a64*k[3][m] + a65*k[4][m]);
        f(yt, k[5]);
        for (int m = 0; m < 2; m++) yn[m] = y[m] + h*(b1*k[0][m] + b3*k[2][m] + b4*k[3][m] + b5*k[4][m] + b6*k[5][m]);
        f(yn, k[6]);
        double err = 0;
        for (int m = 0; m < 2; m++) {
            const double sc = atol + rtol*fmax(fabs(y[m]), fabs(yn[m]));
            err = fmax(err, fabs(h*(e1*k[0][m] + e3*k[2][m] + e4*k[3][m] + e5*k[4][m] + e6*k[5][m] + e7*k[6][m])/sc));
        }
        const bool finite = isfinite(yn[0]) && isfinite(yn[1]) && yn[1] > 0;
        if (finite && (err <= 1.0 || h < 1e-14)) {
            tau += h;
            for (int m = 0; m < 2; m++) { y[m] = yn[m]; k[0][m] = k[6][m]; }     /* first-same-as-last */
        }
        const double fac = !finite ? 0.2 : (err > 0) ? 0.9*pow(err, -0.2) : 5.0;
        h *= fmin(5.0, fmax(0.2, fac));
        if (h < 1e-300) return 1;
    }
    t = y[0]; b = y[1];
    return tau >= 1.0 ? 0 : 1;
}

/*
 * Energy dissipated by the rolling resistance over one interval (train.py:416-454): states (b, e) over the unit interval,
 * db/dsigma as above, de/dsigma = ds (sr0 + sr1 sqrt(b) + sr2 b); the same adaptive pair, error control on both states.
 */
__device__ int iv_rolling(const IvTrain &T, double ds, double w, double G, double &b, double &e, double atol, double rtol)
{
    const double a21 = 1.0/5, a31 = 3.0/40, a32 = 9.0/40, a41 = 44.0/45, a42 = -56.0/15, a43 = 32.0/9,
                 a51 = 19372.0/6561, a52 = -25360.0/2187, a53 = 64448.0/6561, a54 = -212.0/729,
                 a61 = 9017.0/3168, a62 = -355.0/33, a63 = 46732.0/5247, a64 = 49.0/176, a65 = -5103.0/18656,
                 b1 = 35.0/384, b3 = 500.0/1113, b4 = 125.0/192, b5 = -2187.0/6784, b6 = 11.0/84,
                 e1 = 71.0/57600, e3 = -71.0/16695, e4 = 71.0/1920, e5 = -17253.0/339200, e6 = 22.0/525, e7 = -1.0/40;
    double y[2] = {b, e}, k[7][2], yt[2], yn[2];
    auto f = [&](const double (&x)[2], double (&out)[2]) {
        const double v = sqrt(x[0]), rr = T.sr0 + T.sr1*v + T.sr2*x[0];
        out[0] = 2*ds*(w - rr - G); out[1] = ds*rr;
    };
    double tau = 0, h = 0.05;
    f(y, k[0]);
    for (int step = 0; step < 400000 && tau < 1.0; step++) {
        if (tau + h > 1.0) h = 1.0 - tau;
        for (int m = 0; m < 2; m++) yt[m] = y[m] + h*a21*k[0][m];
        f(yt, k[1]);
        for (int m = 0; m < 2; m++) yt[m] = y[m] + h*(a31*k[0][m] + a32*k[1][m]);
        f(yt, k[2]);
        for (int m = 0; m < 2; m++) yt[m] = y[m] + h*(a41*k[0][m] + a42*k[1][m] + a43*k[2][m]);
        f(yt, k[3]);
        for (int m = 0; m < 2; m++) yt[m] = y[m] + h*(a51*k[0][m] + a52*k[1][m] + a53*k[2][m] + a54*k[3][m]);
        f(yt, k[4]);
        for (int m = 0; m < 2; m++) yt[m] = y[m] + h*(a61*k[0][m] + a62*k[1][m] + a63*k[2][m] + a64*k[3][m] + a65*k[4][m]);
        f(yt, k[5]);
        for (int m = 0; m < 2; m++) yn[m] = y[m] + h*(b1*k[0][m] + b3*k[2][m] + b4*k[3][m] + b5*k[4][m] + b6*k[5][m]);
        f(yn, k[6]);
        double err = 0;
        for (int m = 0; m < 2; m++) {
            const double sc = atol + rtol*fmax(fabs(y[m]), fabs(yn[m]));
            err = fmax(err, fabs(h*(e1*k[0][m] + e3*k[2][m] + e4*k[3][m] + e5*k[4][m] + e6*k[5][m] + e7*k[6][m])/sc));
        }
        const bool finite = isfinite(yn[0]) && isfinite(yn[1]) && yn[0] > 0;
        if (finite && (err <= 1.0 || h < 1e-14)) {
            tau += h;
            for (int m = 0; m < 2; m++) { y[m] = yn[m]; k[0][m] = k[6][m]; }
        }
        const double fac = !finite ? 0.2 : (err > 0) ? 0.9*pow(err, -0.2) : 5.0;
        h *= fmin(5.0, fmax(0.2, fac));
        if (h < 1e-300) return 1;
    }
    b = y[0]; e = y[1];
    return tau >= 1.0 ? 0 : 1;
}

/* ---- collocation ---- */
struct Colloc { int d, numSteps, numApprox, maxIter; const double *C, *D; };     /* C[(d+1)*(d+1)] row r, column j; D[d+1] */

/* dense solve with partial pivoting, in place; returns false for a singular matrix */
__device__ bool iv_solve(int n, double (&A)[MAX_SYS][MAX_SYS], double (&rhs)[MAX_SYS])
{
    for (int c = 0; c < n; c++) {
        int piv = c; double big = fabs(A[c][c]);
        for (int r = c + 1; r < n; r++) if (fabs(A[r][c]) > big) { big = fabs(A[r][c]); piv = r; }
        if (!(big > 0)) return false;
        if (piv != c) { for (int m = 0; m < n; m++) { const double x = A[c][m]; A[c][m] = A[piv][m]; A[piv][m] = x; } const double x = rhs[c]; rhs[c] = rhs[piv]; rhs[piv] = x; }
        const double ip = 1.0/A[c][c];
        for (int r = c + 1; r < n; r++) {
            const double l = A[r][c]*ip;
            if (l == 0) continue;
            for (int m = c; m < n; m++) A[r][m] -= l*A[c][m];
            rhs[r] -= l*rhs[c];
        }
    }
    for (int c = n - 1; c >= 0; c--) {
        double x = rhs[c];
        for (int m = c + 1; m < n; m++) x -= A[c][m]*rhs[m];
        rhs[c] = x/A[c][c];
    }
    return true;
}

/*
 * casadi.simpleIRK over [0, h]: numSteps steps of length h/numSteps.  joint = integrate (t, b), otherwise b only.
 * Returns 0, or 1 when Newton's method did not reach the tolerance (the result is still written, like error_on_fail = false).
 */
__device__ int iv_irk(const IvTrain &T, const Colloc &K, double ds, double w, double G, bool joint, double h, double &t, double &b)
{
    const int d = K.d, nx = joint ? 2 : 1, n = d*nx;
    const double dt = h/K.numSteps;
    int flag = 0;
    double xt = t, xb = b;
    for (int k = 0; k < K.numSteps; k++) {
        double vt[MAX_ORDER], vb[MAX_ORDER];
        for (int j = 0; j < d; j++) { vt[j] = xt; vb[j] = xb; }
        bool converged = false;
        for (int it = 0; it < K.maxIter; it++) {
            double A[MAX_SYS][MAX_SYS], F[MAX_SYS];
            for (int r = 0; r < n; r++) for (int m = 0; m < n; m++) A[r][m] = 0;
            double fmaxabs = 0;
            for (int j = 0; j < d; j++) {
                double ft, fb, dft, dfb;
                iv_rhs(T, ds, w, G, vb[j], ft, fb, dft, dfb);
                /* xp_j = sum_r C[r][j+1] x_r with x_0 = start of the step */
                double pt = K.C[0*(d + 1) + j + 1]*xt, pb = K.C[0*(d + 1) + j + 1]*xb;
                for (int r = 0; r < d; r++) { const double c = K.C[(r + 1)*(d + 1) + j + 1]; pt += c*vt[r]; pb += c*vb[r]; }
                if (joint) {
                    F[2*j] = dt*ft - pt; F[2*j + 1] = dt*fb - pb;
                    for (int r = 0; r < d; r++) { const double c = K.C[(r + 1)*(d + 1) + j + 1]; A[2*j][2*r] -= c; A[2*j + 1][2*r + 1] -= c; }
                    A[2*j][2*j + 1] += dt*dft; A[2*j + 1][2*j + 1] += dt*dfb;
                    fmaxabs = fmax(fmaxabs, fmax(fabs(F[2*j]), fabs(F[2*j + 1])));
                } else {
                    F[j] = dt*fb - pb;
                    for (int r = 0; r < d; r++) A[j][r] -= K.C[(r + 1)*(d + 1) + j + 1];
                    A[j][j] += dt*dfb;
                    fmaxabs = fmax(fmaxabs, fabs(F[j]));
                }
            }
            if (!isfinite(fmaxabs)) break;
            if (fmaxabs <= 1e-13*fmax(1.0, fabs(xb))) { converged = true; break; }
            if (!iv_solve(n, A, F)) break;
            double stepmax = 0;
            for (int j = 0; j < d; j++) {
                if (joint) { vt[j] -= F[2*j]; vb[j] -= F[2*j + 1]; stepmax = fmax(stepmax, fmax(fabs(F[2*j]), fabs(F[2*j + 1]))); }
                else { vb[j] -= F[j]; stepmax = fmax(stepmax, fabs(F[j])); }
            }
            if (stepmax <= 1e-15*fmax(1.0, fabs(xb))) { converged = true; break; }
        }
        if (!converged) flag = 1;
        double nt = K.D[0]*xt, nb = K.D[0]*xb;
        for (int r = 0; r < d; r++) { nt += K.D[r + 1]*vt[r]; nb += K.D[r + 1]*vb[r]; }
        xt = nt; xb = nb;
    }
    t = xt; b = xb;
    return flag;
}

__global__ void adaptive_kernel(IvTrain T, int n, const double *t0, const double *b0, const double *ds, const double *w, const double *grad, const double *curv,
                                double atol, double rtol, double *t_out, double *b_out, int *status)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    double t = t0[k], b = b0[k];
    const int st = iv_dopri(T, ds[k], w[k], iv_resistance(T, grad[k], curv[k]), t, b, atol, rtol);
    t_out[k] = t; b_out[k] = b;
    if (status) status[k] = st;
}

/* the same by casadi.simpleRK(fun, numSteps, 4) -- TrainIntegrator.initRollingResistance(solver='RK'), train.py:428-432: classic RK4, equal steps */
__device__ int iv_rolling_rk(const IvTrain &T, double ds, double w, double G, double &b, double &e, int numSteps)
{
    auto f = [&](const double (&x)[2], double (&out)[2]) {
        const double v = sqrt(x[0]), rr = T.sr0 + T.sr1*v + T.sr2*x[0];
        out[0] = 2*ds*(w - rr - G); out[1] = ds*rr;
    };
    const double h = 1.0/numSteps;
    double y[2] = {b, e};
    for (int s = 0; s < numSteps; s++) {
        double k1[2], k2[2], k3[2], k4[2], yt[2];
        f(y, k1);
        for (int m = 0; m < 2; m++) yt[m] = y[m] + 0.5*h*k1[m];
        f(yt, k2);
        for (int m = 0; m < 2; m++) yt[m] = y[m] + 0.5*h*k2[m];
        f(yt, k3);
        for (int m = 0; m < 2; m++) yt[m] = y[m] + h*k3[m];
        f(yt, k4);
        for (int m = 0; m < 2; m++) y[m] += (h/6)*((k1[m] + 2*k2[m]) + (2*k3[m] + k4[m]));
    }
    b = y[0]; e = y[1];
    return (isfinite(b) && isfinite(e) && b > 0) ? 0 : 1;
}

__global__ void rolling_kernel(IvTrain T, int n, const double *b0, const double *ds, const double *w, const double *grad, const double *curv,
                               double atol, double rtol, double *e_out, double *b_out, int *status)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    double b = b0[k], e = 0.0;
    /* (atol < 0 with rtol = 0: -atol equal steps of classic RK4 instead of the adaptive pair) */
    const int st = (atol < 0) ? iv_rolling_rk(T, ds[k], w[k], iv_resistance(T, grad[k], curv[k]), b, e, (int)(-atol))
                              : iv_rolling(T, ds[k], w[k], iv_resistance(T, grad[k], curv[k]), b, e, atol, rtol);
    e_out[k] = e; b_out[k] = b;
    if (status) status[k] = st;
}

__global__ void colloc_kernel(IvTrain T, Colloc K, int n, const double *t0, const double *b0, const double *ds, const double *w, const double *grad, const double *curv,
                              double *t_out, double *b_out, int *status)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    const double G = iv_resistance(T, grad[k], curv[k]);
    double t = t0[k], b = b0[k];
    int st = 0;
    if (K.numApprox == 0) st = iv_irk(T, K, ds[k], w[k], G, true, 1.0, t, b);
    else {
        /* b at the evaluation points 0, 1/ns, ..., 1, each integrated from b0 (train.py:328-332); trapezoidal time (train.py:336-340) */
        const int ns = K.numApprox;
        double prev = b, dummy = 0, acc = t;
        for (int j = 1; j <= ns; j++) {
            double bj = b;
            st |= iv_irk(T, K, ds[k], w[k], G, false, (double)j/ns, dummy, bj);
            acc += 2*ds[k]*(1.0/ns)/(sqrt(prev) + sqrt(bj));
            prev = bj;
        }
        t = acc; b = prev;
    }
    t_out[k] = t; b_out[k] = b;
    if (status) status[k] = st;
}

/* the fused iteration's reciprocal and square root on a list of operands (msd_fastmath_probe: the GPU tests bound their error in ulps) */
__global__ void fastmath_kernel(int n, const double *x, double *rc, double *sq, double *rs)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= n) return;
    double r;
    rc[k] = msd::frcp(x[k]);
    sq[k] = msd::fsqrt2(x[k], r);
    rs[k] = r;
}

thread_local std::string g_iv_err;
int iv_fail(int code, const std::string &m) { g_iv_err = m; return code; }

}  // namespace

extern "C" {

const char *msd_interval_last_error(void) { return g_iv_err.c_str(); }

int msd_interval_integrate(int device, int n, const double *train5, int method, const double *params, int nparams,
                           const double *t0, const double *b0, const double *ds, const double *w, const double *grad, const double *curv,
                           double *t_out, double *b_out, int *status_out)
{
    if (n < 1 || !train5 || !params || !t0 || !b0 || !ds || !w || !grad || !curv || !t_out || !b_out) return iv_fail(MSD_E_INVALID, "bad argument");
    for (int k = 0; k < n; k++)
        if (!(b0[k] > 0) || !(ds[k] > 0)) return iv_fail(MSD_E_INVALID, "velocitySquared and ds must be positive");
    Colloc K = {0, 0, 0, 0, nullptr, nullptr};
    if (method == MSD_INTEGRATOR_ROLLING_RESISTANCE && nparams == 2 && params[1] == 0 && params[0] <= -1 && params[0] >= -1000 && params[0] == (double)(int)params[0]) {
        /* (-numSteps, 0): fixed steps of classic RK4 (solver='RK' of TrainIntegrator.initRollingResistance, train.py:428-432) */
    } else if (method == MSD_INTEGRATOR_ADAPTIVE || method == MSD_INTEGRATOR_ROLLING_RESISTANCE) {
        if (nparams != 2 || !(params[0] > 0) || !(params[1] > 0)) return iv_fail(MSD_E_INVALID, "adaptive integrator needs (abstol, reltol) > 0");
    } else if (method == MSD_INTEGRATOR_COLLOCATION) {
        if (nparams < 4) return iv_fail(MSD_E_INVALID, "collocation integrator needs (order, numSteps, numApproxSteps, maxIter, C, D)");
        K.d = (int)params[0]; K.numSteps = (int)params[1]; K.numApprox = (int)params[2]; K.maxIter = (int)params[3];
        if (K.d < 1 || K.d > MAX_ORDER) return iv_fail(MSD_E_INVALID, "Order of implicit Runge-Kutta should be a positive integer between 1 and 9!");
        if (K.numSteps < 1 || K.numApprox < 0 || K.maxIter < 1) return iv_fail(MSD_E_INVALID, "bad collocation options");
        if (nparams != 4 + (K.d + 1)*(K.d + 1) + (K.d + 1)) return iv_fail(MSD_E_INVALID, "collocation tables of the wrong size");
    } else return iv_fail(MSD_E_INVALID, "Unknown integration method!");
    if (hipSetDevice(device) != hipSuccess) return iv_fail(MSD_E_NODEVICE, "no such device");

    double *d = nullptr, *d_par = nullptr; int *d_st = nullptr;
    auto cleanup = [&]() { hipFree(d); hipFree(d_par); hipFree(d_st); };
#define IV_TRY(expr)                                                                                                   \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) { cleanup(); return iv_fail(MSD_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); } \
    } while (0)
    IV_TRY(hipMalloc((void **)&d, sizeof(double)*8*(size_t)n));
    IV_TRY(hipMalloc((void **)&d_st, sizeof(int)*(size_t)n));
    const double *src[6] = {t0, b0, ds, w, grad, curv};
    for (int a = 0; a < 6; a++) IV_TRY(hipMemcpy(d + (size_t)a*n, src[a], sizeof(double)*n, hipMemcpyHostToDevice));
    const IvTrain T = {train5[0], train5[1], train5[2], train5[3], train5[4]};
    const dim3 grid((n + 63)/64), block(64);
    if (method == MSD_INTEGRATOR_ROLLING_RESISTANCE) {
        /* t0 is ignored, t_out receives the specific energy [J/kg] */
        hipLaunchKernelGGL(rolling_kernel, grid, block, 0, 0, T, n, d + n, d + 2*(size_t)n, d + 3*(size_t)n, d + 4*(size_t)n, d + 5*(size_t)n,
                           params[0], params[1], d + 6*(size_t)n, d + 7*(size_t)n, d_st);
    } else if (method == MSD_INTEGRATOR_ADAPTIVE) {
        hipLaunchKernelGGL(adaptive_kernel, grid, block, 0, 0, T, n, d, d + n, d + 2*(size_t)n, d + 3*(size_t)n, d + 4*(size_t)n, d + 5*(size_t)n,
                           params[0], params[1], d + 6*(size_t)n, d + 7*(size_t)n, d_st);
    } else {
        IV_TRY(hipMalloc((void **)&d_par, sizeof(double)*(nparams - 4)));
        IV_TRY(hipMemcpy(d_par, params + 4, sizeof(double)*(nparams - 4), hipMemcpyHostToDevice));
        K.C = d_par; K.D = d_par + (K.d + 1)*(K.d + 1);
        hipLaunchKernelGGL(colloc_kernel, grid, block, 0, 0, T, K, n, d, d + n, d + 2*(size_t)n, d + 3*(size_t)n, d + 4*(size_t)n, d + 5*(size_t)n,
                           d + 6*(size_t)n, d + 7*(size_t)n, d_st);
    }
    IV_TRY(hipGetLastError());
    IV_TRY(hipMemcpy(t_out, d + 6*(size_t)n, sizeof(double)*n, hipMemcpyDeviceToHost));
    IV_TRY(hipMemcpy(b_out, d + 7*(size_t)n, sizeof(double)*n, hipMemcpyDeviceToHost));
    if (status_out) IV_TRY(hipMemcpy(status_out, d_st, sizeof(int)*n, hipMemcpyDeviceToHost));
#undef IV_TRY
    cleanup();
    return MSD_OK;
}

int msd_fastmath_probe(int device, int n, const double *x, double *rcp_out, double *sqrt_out, double *rsqrt_out)
{
    if (n < 1 || !x || !rcp_out || !sqrt_out || !rsqrt_out) return iv_fail(MSD_E_INVALID, "bad argument");
    if (hipSetDevice(device) != hipSuccess) return iv_fail(MSD_E_NODEVICE, "no such device");
    double *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(double)*4*(size_t)n) != hipSuccess) return iv_fail(MSD_E_HIP, "hipMalloc");
    hipError_t e = hipMemcpy(d, x, sizeof(double)*n, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(fastmath_kernel, dim3((n + 255)/256), dim3(256), 0, 0, n, d, d + n, d + 2*(size_t)n, d + 3*(size_t)n);
        e = hipGetLastError();
    }
    double *out[3] = {rcp_out, sqrt_out, rsqrt_out};
    for (int a = 0; a < 3 && e == hipSuccess; a++) e = hipMemcpy(out[a], d + (size_t)(a + 1)*n, sizeof(double)*n, hipMemcpyDeviceToHost);
    hipFree(d);
    return e == hipSuccess ? MSD_OK : iv_fail(MSD_E_HIP, hipGetErrorString(e));
}

}  // extern "C"
