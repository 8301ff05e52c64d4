/*
 * msd_fastmath.hpp -- reciprocal and square root of the fused interior-point iteration (device code, included by msd_kernel.hpp).
 */
#pragma once

namespace msd {

/* Reciprocal and square root of the fused iteration (round 5).  The compiler's IEEE sequences cost 11 (division) and 16 (square root) VALU
 * instructions, a quarter of them range scaling and special-case fix-up for operands the iteration never sees (slacks, multipliers, pivots and
 * squared speeds are normal numbers); 233 divisions and 92 square roots per lane and iteration made up a fifth of the first-pass kernel's issue
 * slots (profiles/r05/c1_phase_budget.txt).  frcp: v_rcp_f64 + two Newton steps (5 instructions, <= 1 ulp; 0, inf and NaN come out as NaN or
 * inf, which every caller's finiteness / positivity test catches as before).  fsqrt2: v_rsq_f64 + one coupled Goldschmidt step + two residual
 * corrections -- the compiler's own refinement without its scaling -- returning the root and, from the same registers plus one Newton step, its
 * reciprocal (<= 1 ulp each: tests/test_gpu_parity.py::test_fast_reciprocal_and_square_root).
 * The general iteration (follow-up kernels, restoration, watchdog) keeps the IEEE operations in its own passes (FM = false).  Two shared pieces take
 * frcp / fsqrt2 in every kernel: the 3 x 3 inverse of the scans' combines (msd_scan.hpp: inv3 -- one reciprocal of a determinant that the caller checks) and
 * the right-hand side and step-size factor of the adaptive shooting integrator (msd_integ.hpp).  So a follow-up kernel agrees with the host emulation (IEEE
 * throughout) to rounding, not bit for bit; frcp(0) is a NaN where 1/0 is an infinity -- either fails the callers' tests (fabs(det) > 0, isfinite). */
#ifndef MSD_FAST_MATH
#define MSD_FAST_MATH 1
#endif
#if defined(MSD_HOST_EMULATION) || !MSD_FAST_MATH
__device__ __forceinline__ double frcp(double x) { return 1.0/x; }
__device__ __forceinline__ double fsqrt2(double x, double &r) { const double s = sqrt(x); r = 1.0/s; return s; }
__device__ __forceinline__ double fsqrt(double x) { return sqrt(x); }
#else
__device__ __forceinline__ double frcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}
__device__ __forceinline__ double fsqrt2(double x, double &r)
{
    const double y = __builtin_amdgcn_rsq(x);
    double g = x*y, h = 0.5*y;
    const double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g); h = __builtin_fma(h, e, h);
    double d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    d = __builtin_fma(-g, g, x);
    g = __builtin_fma(d, h, g);
    /* the reciprocal to full precision too (one Newton step on 2h against the finished root): it scales the derivatives of sqrt(b) in the jets, and
     * an error of 2^-45 there -- what the coupled step leaves -- is fresh noise of |J| |lambda| 3e-14 in the dual residual of every iteration: a
     * short-horizon re-solve of config 4 with multipliers of 1e7 stalled at 1e-6 with it where the IEEE operations reach 5e-9 */
    r = h + h;
    r = __builtin_fma(r, __builtin_fma(-g, r, 1.0), r);
    return g;
}
__device__ __forceinline__ double fsqrt(double x) { double r; return fsqrt2(x, r); }
#endif

/* err^(-1/5) of the embedded Runge-Kutta pairs' step-size controller: it only proposes the next step size (the accept test is on err itself), a double
 * pow() is 150-odd instructions per step and lane -- about as much as the value-only stages of the step it follows */
#if defined(MSD_HOST_EMULATION) || !MSD_FAST_MATH
__device__ __forceinline__ double pow_m02(double err) { return pow(err, -0.2); }
#else
__device__ __forceinline__ double pow_m02(double err) { return (double)exp2f(-0.2f*log2f((float)err)); }
#endif

}  // namespace msd
