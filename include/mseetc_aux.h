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

#ifdef __cplusplus
}
#endif
#endif
