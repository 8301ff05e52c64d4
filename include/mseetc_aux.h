/*
 * mseetc_aux.h -- measurement aids of the C ABI (no counterpart in the reference: its only timing is IPOPT's t_wall_total, ocp.py:362).
 */
#ifndef MSEETC_AUX_H
#define MSEETC_AUX_H

#include "mseetc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/*
 * With `on`, every launch of the handle records HIP events around its first kernel alone (the first pass of a split solve, or the one kernel of
 * the others); msd_problem_first_pass_ms waits for the stream and returns the mean duration over the last launches (at most 64) -- the
 * dominant kernel's own time, next to the time of whole launches that msd_timer_begin / msd_timer_end bracket (bench.py: roofline).
 */
int msd_problem_time_first_pass(msd_handle h, int on);
int msd_problem_first_pass_ms(msd_handle h, float *mean_ms, int *launches);

/*
 * Page-locked host memory for the result arrays of msd_solve_batch: copies from the device into it run at the link's rate and need no staging
 * (into pageable memory they are staged and page-faulted: 34 MB of results per 8192 scenarios at N = 100 cost 2.8 ms instead of 0.7 ms).
 * mseetc/_device.py hands such arrays out as numpy arrays and reuses a buffer once nobody holds its array any more.
 */
int msd_host_alloc(unsigned long long bytes, void **ptr);
int msd_host_free(void *ptr);

/*
 * With `on`, a host-buffer call (msd_solve_batch, _ex, _warm, _multi) whose z_out -- and lam_out, if given -- lie in page-locked memory the device can
 * address (msd_host_alloc, hipHostMalloc, hipHostRegister) has the kernels store the results there themselves: every scenario's z* crosses the link when
 * that scenario is done, spread over the launch, and no copy waits behind the last one (config 1 on MI355X: 0.99 x the device-resident rate instead of
 * 0.89 x; same bits).  Arrays in pageable memory are served by copies as before.  The handle then keeps no device copy of such a solve:
 * msd_solve_batch_shifted right after it fails with MSD_E_INVALID -- a loop of shifted re-solves leaves the switch off (mseetc/mpc.py does); a handle with
 * msd_problem_keep_duals on is such a loop's and is served by copies whatever the switch says.  Default: off.
 * (The reference hands back host arrays, ocp.py:359-380: this is the boundary's own cost)
 */
int msd_problem_direct_results(msd_handle h, int on);

/*
 * Tuning switches of the kernel pickers for the problems created afterwards (process-wide; A/B measurements and one GPU test -- the library itself reads no
 * environment variable).  "no_full" != 0: the kernels without the structure of the NLP compiled in; "two_nodes_per_lane" != 0: the 64 x 2 geometry for
 * horizons of 64 ... 127 intervals of the shooting-integrator and integrateLosses families (default 128 x 1).  Unknown name: MSD_E_INVALID.
 */
int msd_tuning(const char *name, int value);

/*
 * Test hook: the reciprocal and the square root / reciprocal square root of the fused interior-point iteration (csrc/msd_fastmath.hpp: v_rcp_f64 /
 * v_rsq_f64 refined without the compiler's range scaling) evaluated on n operands, so that the GPU tests can bound their error against the IEEE
 * operations (tests/test_gpu_parity.py::test_fast_reciprocal_and_square_root: <= 1 ulp on normal operands).  Errors: msd_interval_last_error().
 */
int msd_fastmath_probe(int device, int n, const double *x, double *rcp_out, double *sqrt_out, double *rsqrt_out);

#ifdef __cplusplus
}
#endif
#endif
