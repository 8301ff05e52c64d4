/*
 * mseetc_mpc.h -- C ABI of the device-resident shrinking-horizon loop (BASELINE config 4).
 *
 * The reference has no MPC loop: its mechanism for a re-solve from the current position is Track.updateLimits(positionStart)
 * (track.py:420-450) + a new casadiSolver on the cropped track + solve(T, initialTime, initialVelocity) (ocp.py:310), always from a
 * cold start (ocp.py:325-339).  mseetc/mpc.py: shrinkingHorizon does that for a batch of scenarios with one launch per re-solve and the
 * bookkeeping on the host; this entry point runs the same loop with the bookkeeping on the device: the sequence of grids does not depend
 * on the solutions, so all problem records are uploaded once, and per re-solve the measured state (time and speed at node `stride` of the
 * previous solution, perturbed by the caller's noise draws), the scenario records, the failure handling (minimum running time from the
 * time-optimal twin, arrival time moved there, repeated solve) and the log are small kernels between the solver's launches on one stream.
 * Nothing is waited for and nothing crosses PCIe between the first launch and the end of the loop.
 */
#ifndef MSEETC_MPC_H
#define MSEETC_MPC_H

#include "mseetc_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct msd_mpc *msd_mpc_handle;

/* one record of the log per re-solve and scenario */
enum { MSD_MPC_T0 = 0,        /* measured time the re-solve starts from */
       MSD_MPC_V0,            /* measured speed (before the clipping of ocp.py:343) */
       MSD_MPC_T,             /* arrival time of the re-solve (moved where the measured state no longer allowed the one asked for) */
       MSD_MPC_STATUS, MSD_MPC_ITERS, MSD_MPC_OBJ,      /* MSD_ST_STATUS / _ITERS / _OBJ of the re-solve's last launch for the scenario */
       MSD_MPC_RELAXED,       /* > 0: the arrival time was moved in this re-solve -- 1 on the verdict of a converged time-optimal twin, 2 on that of a twin that ended next to its optimum without converging */
       MSD_MPC_COUNT };

typedef struct {
    int num_resolves;                      /* K: re-solve k runs on problems[k] */
    int stride;                            /* intervals the train advances between two re-solves (problems[k + 1] has `stride` intervals fewer) */
    const msd_problem_desc *problems;      /* [K] the energy problems on the cropped tracks (casadiSolver on Track.updateLimits(positionStart)) */
    const msd_problem_desc *twins;         /* [K] their time-optimal twins (ocp.py:146-150), or NULL: a failed re-solve is left as it is */
    const double *vlim_first;              /* [K] speed limit at the first node of grid k [m/s] (clipping of the initial speed, ocp.py:343) */
    const double *length;                  /* [K] length of grid k [m] */
    const unsigned char *tail;             /* [K] 1: grid k is grid k - 1 without its first `stride` intervals (the previous solutions are a
                                            * warm start for it on the device: msd_solve_batch_shifted); tail[0] is ignored */
    double vmin, vmax_train;               /* minimumVelocity; maximum speed of the train (loose running time of the twin) */
    double terminal_velocity;              /* clipped by the caller to [vmin, speed limit at the last node] (ocp.py:344) */
    int warm_start;                        /* 1: shifted primal-dual warm starts where tail[k]; 0: every re-solve from the problem's own starting point */
    double warm_mu, warm_push;             /* barrier parameter and interior push of a warm start (msd_solve_batch_shifted) */
    double noise;                          /* relative measurement noise: t <- max(t (1 + noise n1), 0), v <- v (1 + noise n2) */
    int relax_infeasible;                  /* 1: a re-solve that fails although ... see late_margin */
    double late_margin;                    /* a failed re-solve whose minimum running time tmin (twin) exceeds T - t0 is repeated with T = t0 + tmin (1 + m),
                                            * m = late_margin, then 4 m, then 16 m */
} msd_mpc_plan;

/*
 * Build the loop for one device handle pair: `h` is configured for problems[0] (its stream runs the loop), `twin` for twins[0] (may be NULL with
 * plan->twins).  Uploads every problem record; the descriptions may be freed afterwards.
 */
int msd_mpc_create(msd_handle h, msd_handle twin, const msd_mpc_plan *plan, msd_mpc_handle *out);
int msd_mpc_destroy(msd_mpc_handle m);

/*
 * Run the loop for nscen scenarios: arrival times T[nscen], common initial time and speed, draws n1/n2 [K - 1][nscen] (standard normal; the
 * state after re-solve k is perturbed with row k).  Host outputs (each may be NULL): log[K][nscen][MSD_MPC_COUNT]; z_log = the solutions of
 * every re-solve back to back ([nscen][nz_k] for k = 0 .. K - 1, nz_k = msd_mpc_nz(m, k)); loop_ms = device time of the whole loop (events around
 * it), solve_ms = sum of the solver's launches alone is not separable without waiting and is not reported.
 */
int msd_mpc_run(msd_mpc_handle m, int nscen, const double *T, double initial_time, double initial_velocity, const double *n1, const double *n2,
                double *log, double *z_log, float *loop_ms);
int msd_mpc_nz(msd_mpc_handle m, int k);

#ifdef __cplusplus
}
#endif
#endif
