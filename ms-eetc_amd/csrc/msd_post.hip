/*
 * msd_post.hip -- post-processing integrations of a solved trajectory on the GPU.
 *
 * The reference re-simulates every interval of a solution in the TIME domain with CVODES (mseetc/utils.py:110-194,
 * abstol 1e-12 / reltol 1e-14) and, for `integrateLosses=True`, integrates the traction / regenerative losses along
 * each interval (utils.py:261-289 -> train.py:367-413, abstol 1e-8 / reltol 1e-6).  The ODEs are non-stiff and tiny
 * (2 or 3 states), so an adaptive explicit Dormand-Prince 5(4) pair with the same tolerances replaces the BDF code:
 *   re-simulation:  ds/dtau = dt v,  dv/dtau = dt (f - sr0 - sr1 v - sr2 v^2 - g grad/rho - cr(curv)/rho)      (utils.py:124-131)
 *   losses:         dv/dtau = dt a(v),  deTr/dtau = dt Ltr(f M, v)/M,  deBr/dtau = dt Lrgb(f M, v)/M             (train.py:374-386)
 * One thread per scenario walks the intervals when errors accumulate (utils.py:164-194), one thread per
 * (scenario, interval) otherwise.
 */
#include <hip/hip_runtime.h>

#include <cstring>
#include <mutex>
#include <vector>

#include <cstdio>
#include <string>

#include "msd_kernel.hpp"

namespace {

struct PostTrain { double sr0, sr1, sr2, g, rho; };

struct PostLoss { int kind; double ct, cr; const double *table; };   /* 0 none, 1 static, 2 dynamic */

__device__ __forceinline__ double post_resistance(const PostTrain &T, double grad, double curv)
{
    const double c = fabs(curv);
    const double cr = (c <= 1.0/300.0) ? T.g*0.5*c/(1 - 30*c) : T.g*0.65*c/(1 - 55*c);   /* utils.py:126-127 as written */
    return T.g*grad*(1/T.rho) + cr*(1/T.rho);
}

/* right-hand side; y = (s, v) or (v, eTr, eBr) */
template <int DIM>
__device__ __forceinline__ void rhs(const PostTrain &T, const PostLoss &Ls, double f, double G, double dt, const double (&y)[DIM], double (&dy)[DIM])
{
    if (DIM == 2) {
        const double v = y[1];
        dy[0] = dt*v;
        dy[1] = dt*(f - (T.sr0 + T.sr1*v + T.sr2*v*v) - G);
    } else {
        const double v = y[0];
        dy[0] = dt*(f - (T.sr0 + T.sr1*v + T.sr2*v*v) - G);
        double ltr = 0, lrg = 0;
        if (Ls.kind == 1) { ltr = Ls.ct*f*v; lrg = -Ls.cr*f*v; }             /* train.py:203 through utils.py:197-220: linear in f */
        else if (Ls.kind == 2) {
            const msd::DynLoss D(Ls.table, nullptr, 0.0);
            double lr[2][6];
            msd::loss_rows(D, f, fmax(v, 1e-9), lr);                          /* rows are L/v */
            ltr = lr[0][0]*v; lrg = lr[1][0]*v;
        }
        dy[1] = dt*ltr; dy[2] = dt*lrg;
    }
}

/* Dormand-Prince 5(4) over tau in [0, 1] with step-size control on the mixed abs/rel error norm */
template <int DIM>
__device__ void dopri5(const PostTrain &T, const PostLoss &Ls, double f, double G, double dt, double (&y)[DIM], double atol, double rtol)
{
    const double a21 = 1.0/5, a31 = 3.0/40, a32 = 9.0/40, a41 = 44.0/45, a42 = -56.0/15, a43 = 32.0/9,
                 a51 = 19372.0/6561, a52 = -25360.0/2187, a53 = 64448.0/6561, a54 = -212.0/729,
                 a61 = 9017.0/3168, a62 = -355.0/33, a63 = 46732.0/5247, a64 = 49.0/176, a65 = -5103.0/18656,
                 b1 = 35.0/384, b3 = 500.0/1113, b4 = 125.0/192, b5 = -2187.0/6784, b6 = 11.0/84,
                 e1 = 71.0/57600, e3 = -71.0/16695, e4 = 71.0/1920, e5 = -17253.0/339200, e6 = 22.0/525, e7 = -1.0/40;
    double tau = 0, h = 0.05;
    double k1[DIM], k2[DIM], k3[DIM], k4[DIM], k5[DIM], k6[DIM], k7[DIM], yt[DIM], yn[DIM];
    rhs<DIM>(T, Ls, f, G, dt, y, k1);
    for (int step = 0; step < 200000 && tau < 1.0; step++) {
        if (tau + h > 1.0) h = 1.0 - tau;
        for (int m = 0; m < DIM; m++) yt[m] = y[m] + h*a21*k1[m];
        rhs<DIM>(T, Ls, f, G, dt, yt, k2);
        for (int m = 0; m < DIM; m++) yt[m] = y[m] + h*(a31*k1[m] + a32*k2[m]);
        rhs<DIM>(T, Ls, f, G, dt, yt, k3);
        for (int m = 0; m < DIM; m++) yt[m] = y[m] + h*(a41*k1[m] + a42*k2[m] + a43*k3[m]);
        rhs<DIM>(T, Ls, f, G, dt, yt, k4);
        for (int m = 0; m < DIM; m++) yt[m] = y[m] + h*(a51*k1[m] + a52*k2[m] + a53*k3[m] + a54*k4[m]);
        rhs<DIM>(T, Ls, f, G, dt, yt, k5);
        for (int m = 0; m < DIM; m++) yt[m] = y[m] + h*(a61*k1[m] + a62*k2[m] + a63*k3[m] + a64*k4[m] + a65*k5[m]);
        rhs<DIM>(T, Ls, f, G, dt, yt, k6);
        for (int m = 0; m < DIM; m++) yn[m] = y[m] + h*(b1*k1[m] + b3*k3[m] + b4*k4[m] + b5*k5[m] + b6*k6[m]);
        rhs<DIM>(T, Ls, f, G, dt, yn, k7);
        double err = 0;
        for (int m = 0; m < DIM; m++) {
            const double sc = atol + rtol*fmax(fabs(y[m]), fabs(yn[m]));
            const double e = h*(e1*k1[m] + e3*k3[m] + e4*k4[m] + e5*k5[m] + e6*k6[m] + e7*k7[m])/sc;
            err = fmax(err, fabs(e));
        }
        if (err <= 1.0 || h < 1e-14) {
            tau += h;
            for (int m = 0; m < DIM; m++) { y[m] = yn[m]; k1[m] = k7[m]; }     /* first-same-as-last */
        }
        const double fac = (err > 0) ? 0.9*pow(err, -0.2) : 5.0;
        h *= fmin(5.0, fmax(0.2, fac));
    }
}

/* accumulated re-simulation: one thread per scenario (utils.py:164-194 with accumulatedErrors=True) */
__global__ void resim_kernel(PostTrain T, int nscen, int N, const double *force, const double *dts, const double *grad, const double *curv,
                             const double *s0, const double *v0, double atol, double rtol, double *pos, double *vel)
{
    const int k = blockIdx.x*blockDim.x + threadIdx.x;
    if (k >= nscen) return;
    PostLoss none = {0, 0, 0, nullptr};
    double y[2] = {s0[k], v0[k]};
    pos[(size_t)k*(N + 1)] = y[0]; vel[(size_t)k*(N + 1)] = y[1];
    for (int i = 0; i < N; i++) {
        dopri5<2>(T, none, force[(size_t)k*N + i], post_resistance(T, grad[i], curv[i]), dts[(size_t)k*N + i], y, atol, rtol);
        pos[(size_t)k*(N + 1) + i + 1] = y[0]; vel[(size_t)k*(N + 1) + i + 1] = y[1];
    }
}

/* losses of every interval: one thread per (scenario, interval) (utils.py:261-289) */
__global__ void losses_kernel(PostTrain T, PostLoss Ls, int nscen, int N, const double *force_el, const double *force_pn, const double *dts, const double *grad,
                              const double *curv, const double *vstart, double atol, double rtol, double *eTr, double *eBr)
{
    const size_t id = (size_t)blockIdx.x*blockDim.x + threadIdx.x;
    if (id >= (size_t)nscen*N) return;
    const int i = (int)(id % N);
    double y[3] = {vstart[id], 0.0, 0.0};
    /* the loss functions see the electrical force only, the dynamics the total force (train.py:374-386) */
    PostLoss L = Ls;
    const double fel = force_el[id], ftot = fel + force_pn[id];
    /* integrate with the total force in the velocity equation and the electrical force in the loss equations */
    const double G = post_resistance(T, grad[i], curv[i]);
    /* the two forces differ only when the pneumatic brake acts: fold the difference into the constant resistance term */
    dopri5<3>(T, L, fel, G - (ftot - fel), dts[id], y, atol, rtol);
    eTr[id] = y[1]; eBr[id] = y[2];
}

thread_local std::string g_post_err;
int post_fail(int code, const std::string &m) { g_post_err = m; return code; }

#define POST_TRY(expr)                                                                                      \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess) { cleanup(); return post_fail(MSD_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); }  \
    } while (0)

}  // namespace

extern "C" {

const char *msd_post_last_error(void) { return g_post_err.c_str(); }

/* Scratch of the two entry points, per device, grow-only, kept for the life of the process: a device buffer and a page-locked staging buffer of the same size.  A call
 * packs its inputs into the staging buffer, copies them in one piece, launches, copies the results back in one piece -- eight synchronous copies, a hipMalloc and a
 * hipFree per call were 0.45 of the 0.66 ms that casadiSolver.solve() spent in its re-simulation (round 6) */
namespace {
struct PostScratch { int device; double *d, *h; size_t cap; };
std::mutex g_post_mu;
std::vector<PostScratch> g_post;
int post_scratch(int device, size_t doubles, double **d, double **h)
{
    for (PostScratch &p : g_post)
        if (p.device == device) {
            if (p.cap < doubles) {
                hipFree(p.d); hipHostFree(p.h); p.d = p.h = nullptr; p.cap = 0;
                if (hipMalloc((void **)&p.d, sizeof(double)*doubles) != hipSuccess || hipHostMalloc((void **)&p.h, sizeof(double)*doubles, hipHostMallocDefault) != hipSuccess) {
                    hipFree(p.d); p.d = nullptr; (void)hipGetLastError();
                    return post_fail(MSD_E_HIP, "no memory for the post-processing scratch");
                }
                p.cap = doubles;
            }
            *d = p.d; *h = p.h;
            return MSD_OK;
        }
    g_post.push_back(PostScratch{device, nullptr, nullptr, 0});
    return post_scratch(device, doubles, d, h);
}
}  // namespace

int msd_resimulate(int device, int nscen, int N, const double *train5, const double *force, const double *dts, const double *grad, const double *curv,
                   const double *s0, const double *v0, double abstol, double reltol, double *pos_out, double *vel_out)
{
    if (nscen < 1 || N < 1 || !train5 || !force || !dts || !grad || !curv || !s0 || !v0 || !pos_out || !vel_out) return post_fail(MSD_E_INVALID, "bad argument");
    if (hipSetDevice(device) != hipSuccess) return post_fail(MSD_E_NODEVICE, "no such device");
    const size_t nI = (size_t)nscen*N, nP = (size_t)nscen*(N + 1);
    const size_t nin = 2*nI + 2*N + 2*nscen, total = nin + 2*nP;
    std::lock_guard<std::mutex> lock(g_post_mu);
    double *d = nullptr, *h = nullptr;
    int rc = post_scratch(device, total, &d, &h);
    if (rc != MSD_OK) return rc;
    auto cleanup = [&]() {};
    double *d_force = d, *d_dt = d_force + nI, *d_grad = d_dt + nI, *d_curv = d_grad + N, *d_s0 = d_curv + N, *d_v0 = d_s0 + nscen, *d_pos = d_v0 + nscen, *d_vel = d_pos + nP;
    memcpy(h, force, sizeof(double)*nI); memcpy(h + nI, dts, sizeof(double)*nI); memcpy(h + 2*nI, grad, sizeof(double)*N); memcpy(h + 2*nI + N, curv, sizeof(double)*N);
    memcpy(h + 2*nI + 2*N, s0, sizeof(double)*nscen); memcpy(h + 2*nI + 2*N + nscen, v0, sizeof(double)*nscen);
    POST_TRY(hipMemcpy(d, h, sizeof(double)*nin, hipMemcpyHostToDevice));
    PostTrain T = {train5[0], train5[1], train5[2], train5[3], train5[4]};
    hipLaunchKernelGGL(resim_kernel, dim3((nscen + 63)/64), dim3(64), 0, 0, T, nscen, N, d_force, d_dt, d_grad, d_curv, d_s0, d_v0, abstol, reltol, d_pos, d_vel);
    POST_TRY(hipGetLastError());
    POST_TRY(hipMemcpy(h + nin, d_pos, sizeof(double)*2*nP, hipMemcpyDeviceToHost));
    memcpy(pos_out, h + nin, sizeof(double)*nP); memcpy(vel_out, h + nin + nP, sizeof(double)*nP);
    return MSD_OK;
}

int msd_integrate_losses(int device, int nscen, int N, const double *train5, int loss_kind, double ct, double cr, const double *loss_table, int loss_table_len,
                         const double *force_el, const double *force_pn, const double *dts, const double *grad, const double *curv, const double *vstart,
                         double abstol, double reltol, double *etr_out, double *ebr_out)
{
    if (nscen < 1 || N < 1 || !train5 || !force_el || !force_pn || !dts || !grad || !curv || !vstart || !etr_out || !ebr_out) return post_fail(MSD_E_INVALID, "bad argument");
    if (loss_kind < 0 || loss_kind > 2 || (loss_kind == 2 && (!loss_table || loss_table_len < 13))) return post_fail(MSD_E_INVALID, "bad loss model");
    if (hipSetDevice(device) != hipSuccess) return post_fail(MSD_E_NODEVICE, "no such device");
    const size_t nI = (size_t)nscen*N;
    double *d = nullptr, *d_tab = nullptr;
    auto cleanup = [&]() { hipFree(d); hipFree(d_tab); };
    POST_TRY(hipMalloc((void **)&d, sizeof(double)*(6*nI + 2*N)));
    double *d_fe = d, *d_fp = d_fe + nI, *d_dt = d_fp + nI, *d_vs = d_dt + nI, *d_etr = d_vs + nI, *d_ebr = d_etr + nI, *d_grad = d_ebr + nI, *d_curv = d_grad + N;
    POST_TRY(hipMemcpy(d_fe, force_el, sizeof(double)*nI, hipMemcpyHostToDevice));
    POST_TRY(hipMemcpy(d_fp, force_pn, sizeof(double)*nI, hipMemcpyHostToDevice));
    POST_TRY(hipMemcpy(d_dt, dts, sizeof(double)*nI, hipMemcpyHostToDevice));
    POST_TRY(hipMemcpy(d_vs, vstart, sizeof(double)*nI, hipMemcpyHostToDevice));
    POST_TRY(hipMemcpy(d_grad, grad, sizeof(double)*N, hipMemcpyHostToDevice));
    POST_TRY(hipMemcpy(d_curv, curv, sizeof(double)*N, hipMemcpyHostToDevice));
    if (loss_kind == 2) {
        POST_TRY(hipMalloc((void **)&d_tab, sizeof(double)*loss_table_len));
        POST_TRY(hipMemcpy(d_tab, loss_table, sizeof(double)*loss_table_len, hipMemcpyHostToDevice));
    }
    PostTrain T = {train5[0], train5[1], train5[2], train5[3], train5[4]};
    PostLoss Ls = {loss_kind, ct, cr, d_tab};
    hipLaunchKernelGGL(losses_kernel, dim3((unsigned)((nI + 63)/64)), dim3(64), 0, 0, T, Ls, nscen, N, d_fe, d_fp, d_dt, d_grad, d_curv, d_vs, abstol, reltol, d_etr, d_ebr);
    POST_TRY(hipGetLastError());
    POST_TRY(hipMemcpy(etr_out, d_etr, sizeof(double)*nI, hipMemcpyDeviceToHost));
    POST_TRY(hipMemcpy(ebr_out, d_ebr, sizeof(double)*nI, hipMemcpyDeviceToHost));
    cleanup();
    return MSD_OK;
}

}  // extern "C"
