/*
 * mseetc_hip.h -- C ABI of the MI355X (gfx950) multiple-shooting train-control solver.
 *
 * This is the drop-in boundary for the one hot path of dkouzoup/ms-eetc: what
 * `casadiSolver(train, track, opts).solve(...)` hands to `casadi.nlpsol('ipopt')`
 * (reference mseetc/ocp.py:288-290 construction, :359 the call) -- here solved for a whole
 * batch of independent scenarios by hand-written HIP kernels.  Plain pointers and sizes only;
 * the caller owns every buffer; every entry point returns 0 on success and a negative MSD_E_*
 * code otherwise (msd_last_error() gives the text); no exception crosses the boundary.
 * A handle may be used from one host thread at a time.
 *
 * Each entry point names the reference interface it replaces (paths relative to the reference
 * repository root).  The Python binding a maintainer would add is shown in INTEGRATION.md and
 * implemented in ms-eetc_amd/mseetc/_device.py.
 */
#ifndef MSEETC_HIP_H
#define MSEETC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define MSD_ABI_VERSION 6

/* error codes */
#define MSD_OK 0
#define MSD_E_INVALID (-1)     /* bad argument (ValueError on the Python side)            */
#define MSD_E_HIP (-2)         /* HIP runtime error                                       */
#define MSD_E_UNSUPPORTED (-3) /* option outside the device path (e.g. N too large)       */
#define MSD_E_NODEVICE (-4)

/* per-scenario solver status (stats[MSD_ST_STATUS]); >= 0 is success, like IPOPT's
 * Solve_Succeeded / Solved_To_Acceptable_Level (ocp.py:362-364) */
#define MSD_STATUS_SOLVED 0
#define MSD_STATUS_ACCEPTABLE 1
#define MSD_STATUS_MAXITER (-1)
#define MSD_STATUS_LINESEARCH (-2)
#define MSD_STATUS_REGULARIZATION (-3)
#define MSD_STATUS_NUMERIC (-4)
#define MSD_STATUS_TINY_STEP (-5)
/* IPOPT's 'Infeasible_Problem_Detected': the feasibility restoration phase (ocp.py:290,359 -> IPOPT; msd_resto.hpp) converged to a stationary
 * point of the infeasibility.  Also set by the host layer (mseetc/ocp.py: _classify_failures) from the minimum-running-time certificate of a
 * scenario that failed in another way (MSD_STATUS_LINESEARCH = IPOPT's 'Restoration_Failed': the restoration phase itself broke down) */
#define MSD_STATUS_INFEASIBLE (-6)

/*
 * Starting point of a solve.  REFERENCE: the reference's cold start (ocp.py:325-339: Fel 0.5, Fpb -0.1, s 1, t linear,
 * v 60 km/h).  PROFILE: a speed profile built from the limits, the running time and the end speeds with dynamically
 * consistent forces and zero constraint multipliers (msd_kernel.hpp: profile_start); it reaches the same optimum in about half the
 * iterations.  A scenario whose line search breaks down enters the feasibility restoration phase like in IPOPT (msd_resto.hpp: every loss
 * model, shooting integrator and horizon, in the follow-up kernel of the launch; unless no_restoration is set); one that still ends with a breakdown (any
 * failure but MSD_STATUS_INFEASIBLE and the iteration limit -- that one too when a restoration phase came before it) is solved again
 * from the other starting point inside the same launch.
 */
#define MSD_START_REFERENCE 0
#define MSD_START_PROFILE 1

/* stats record: MSD_ST_COUNT doubles per scenario */
enum {
    MSD_ST_STATUS = 0,
    MSD_ST_ITERS,        /* interior-point iterations ('IP iterations', ocp.py:362)         */
    MSD_ST_OBJ,          /* NLP objective: kWh (energy optimal) or scaled time               */
    MSD_ST_KKT,          /* final scaled optimality error                                    */
    MSD_ST_MU,
    MSD_ST_DUAL_INF, MSD_ST_CONSTR_VIOL, MSD_ST_COMPL,
    MSD_ST_N_REG, MSD_ST_N_SOC, MSD_ST_N_BACKTRACK,
    MSD_ST_CYC_TOTAL,    /* shader clock cycles the scenario's workgroup spent in the solve (telemetry)       */
    MSD_ST_CYC_KKT,      /* ... of which inside the serial stage recursion of the KKT solves (fallback path)  */
    MSD_ST_N_FALLBACK,   /* KKT solves that fell back from the stage-parallel scan to the serial sweep        */
    MSD_ST_N_RESTO,      /* restoration phases entered                                                        */
    MSD_ST_N_WATCHDOG,   /* watchdog procedures started (IPOPT: watchdog_shortened_iter_trigger = 10)          */
    MSD_ST_COUNT
};

/* scenario record: MSD_SC_COUNT doubles per scenario */
enum {
    MSD_SC_T0 = 0,       /* initialTime          (ocp.py:310,346-355)                        */
    MSD_SC_TEND,         /* terminalTime                                                     */
    MSD_SC_V0SQ,         /* clipped initialVelocity^2  (ocp.py:343)                          */
    MSD_SC_VNSQ,         /* clipped terminalVelocity^2 (ocp.py:344)                          */
    MSD_SC_COUNT
};

/* optional per-scenario rolling-stock record (BASELINE config 3: perturbed mass / Davis coefficients): MSD_OV_COUNT doubles.
 * It replaces the corresponding fields of msd_problem_desc for that scenario -- what constructing a new
 * Train(config={...}) + casadiSolver would change in the reference (train.py:44-62, ocp.py:96-116, 278). */
enum {
    MSD_OV_SR0 = 0, MSD_OV_SR1, MSD_OV_SR2,
    MSD_OV_F_MAX, MSD_OV_F_MIN, MSD_OV_F_MIN_PN,
    MSD_OV_PW_UPPER, MSD_OV_PW_LOWER,
    MSD_OV_OBJ_DEN,
    MSD_OV_TOTAL_MASS,   /* mass * rho [kg]: what the dynamic loss model turns specific forces into newtons with (efficiency.py:108); 0 = the problem's */
    MSD_OV_COUNT
};

/*
 * Problem description = everything casadiSolver.__init__ derives from (train, track, options)
 * (ocp.py:96-125, 134-284): specific force/power/acceleration bounds, the shooting grid and the
 * piecewise-constant track profile on it, the integrator options and the loss model.
 * "specific" = per kg of mass*rho (ocp.py:97).
 */
typedef struct msd_problem_desc {
    int abi_version;         /* MSD_ABI_VERSION */
    int num_intervals;       /* N (OptionsCasadiSolver.numIntervals, ocp.py:16)              */
    int with_pn_brake;       /* train.forceMinPn != 0 (ocp.py:102)                            */
    int has_power_rows;      /* powerMax or powerMin set (ocp.py:184)                         */
    int energy_optimal;      /* ocp.py:20                                                     */
    int num_steps;           /* OptionsRK.numSteps (train.py:463)                             */
    int num_approx_steps;    /* OptionsRK.numApproxSteps (train.py:465)                       */
    int loss_kind;           /* 0 none, 1 static efficiencies (train.py:199-212), 2 dynamic table (efficiency.py) */
    int max_iterations;      /* ocp.py:18,290                                                 */
    int start_kind;          /* MSD_START_REFERENCE (0) or MSD_START_PROFILE                  */
    int integrator;          /* shooting integrator (OptionsCasadiSolver.integrationMethod, ocp.py:26, train.py:294-322): 0 = 'RK' (simpleRK order 4),
                              * MSD_INTEGRATOR_COLLOCATION = 'IRK' (simpleIRK), MSD_INTEGRATOR_ADAPTIVE = 'CVODES' (integration to tolerances) */
    int coll_degree;         /* 'IRK': OptionsIRK.order = collocation points per step, 1..9 (train.py:485,502)      */
    int newton_iterations;   /* 'IRK': OptionsIRK.maxIter (train.py:493)                       */
    int integrate_losses;    /* OptionsCasadiSolver.integrateLosses (ocp.py:28,231-241): loss slacks [J/kg per interval] bound the loss power integrated
                              * over the running time of the interval; constant efficiencies (loss_kind 1), 'RK' shooting                          */
    int no_restoration;      /* 1: no feasibility restoration phase -- a solve whose line search breaks down ends with MSD_STATUS_LINESEARCH (after the
                              * restart from the other starting point), like ABI 4.  0 (default): IPOPT's behaviour                                         */
    int watchdog_trigger;    /* IPOPT's watchdog_shortened_iter_trigger (the reference leaves it at its default, ocp.py:290): 0 = that default, 10 successive
                              * shortened iterations start the watchdog procedure; < 0: no watchdog procedure; other values: tests                         */
    double sr0, sr1, sr2;    /* specific Davis coefficients (train.py:181-183)                */
    double g, rho;
    double f_max, f_min;     /* bounds of Fel (ocp.py:175-176; f_min = 0 without rg brake)    */
    double f_min_pn;         /* lower bound of Fpb                                            */
    double pw_upper, pw_lower; /* abs() bounds of the power rows (ocp.py:186-192)             */
    double acc_min, acc_max; /* ocp.py:113-114                                                */
    double loss_ct, loss_cr; /* static loss rows s >= ct*Fel, s >= -cr*Fel                    */
    double vmin_sq;          /* minimumVelocity^2 (ocp.py:22,271)                             */
    double obj_den;          /* scalingFactorObjective (ocp.py:278,282)                       */
    double tol;              /* IPOPT tol, 1e-8                                               */
    double int_abstol, int_reltol;   /* 'CVODES': OptionsCVODES.absTol, .relTol (train.py:525-526)     */
    double reserved_d[5];
    const double *ds;        /* [N]   interval lengths (ocp.py:125)                           */
    const double *grad;      /* [N]   gradient, permil/1000 (ocp.py:195)                      */
    const double *curv;      /* [N]   curvature 1/m (ocp.py:196)                              */
    const double *bmax;      /* [N+1] upper bound of b at interior nodes (ocp.py:266-269)     */
    /* loss_kind 2: parameter block of the dynamic loss model (efficiency.py:7-141): forceMax, powerMax, vTurn, vMin, vMax,
     * auxiliaries, (1-etaGear)/etaGear, 1-etaGear, R, V, totalMass, nx, ny, xb[nx+1], yb[ny+1], coef[nx][ny][4][4]
     * (bicubic patches about the cell centres of the not-a-knot spline of the measured motor + converter losses) */
    const double *loss_table;
    int loss_table_len;
    int reserved_tail;
    /* integrator = MSD_INTEGRATOR_COLLOCATION: the tables of casadi.simpleIRK, C[(coll_degree+1)^2] (row r, column j: derivative of the
     * Lagrange polynomial r at point j) then D[coll_degree+1] (the polynomials at 1), on the points {0} + casadi.collocation_points */
    const double *coll_tables;
} msd_problem_desc;

typedef struct msd_problem *msd_handle;

/* number of GPUs visible to the process */
int msd_device_count(void);

/* Replaces casadiSolver.__init__ (ocp.py:80-307): validates, uploads the grid/profile to `device`. */
int msd_problem_create(const msd_problem_desc *desc, int device, msd_handle *out);

int msd_problem_destroy(msd_handle h);
/* Load another problem (other grid, horizon, train, options) into an existing handle: its stream, events and device buffers are
 * kept and grown only when needed.  What constructing the next casadiSolver costs in a receding-horizon loop. */
int msd_problem_reconfigure(msd_handle h, const msd_problem_desc *desc);

/* nz = (4 + with_pn_brake)*N + 2 (ocp.py:166-272) and rows of g per interval (ocp.py:183-229) */
int msd_problem_nz(msd_handle h);
int msd_problem_rows_per_interval(msd_handle h);

/*
 * Replaces the body of casadiSolver.solve (ocp.py:325-362) for `nscen` scenarios at once: cold start
 * (ocp.py:325-339), interior-point solve (ocp.py:359), z* in the reference's layout and the statistics.
 * Host buffers; the call uploads, runs, downloads and synchronises.
 *   scen   [nscen][MSD_SC_COUNT]
 *   z_out  [nscen][nz]
 *   lam_out[nscen][rows_per_interval*N]  multipliers of g in the reference's row order, may be NULL
 *   stats  [nscen][MSD_ST_COUNT]
 *   kernel_ms: if not NULL receives the solve kernel's duration measured with HIP events on the launch stream
 */
int msd_solve_batch(msd_handle h, int nscen, const double *scen, double *z_out, double *lam_out, double *stats,
                    float *kernel_ms);

/*
 * Same with buffers already resident in device memory (the benchmark path): nothing is copied, the kernel is
 * enqueued on the handle's stream; msd_synchronize() waits.  Used to time throughput with inputs in HBM.
 */
int msd_solve_batch_device(msd_handle h, int nscen, const double *d_scen, double *d_z, double *d_lam, double *d_stats);
/* ... with per-scenario rolling-stock overrides resident in device memory (d_overrides[nscen][MSD_OV_COUNT], NULL = none) */
int msd_solve_batch_device_ex(msd_handle h, int nscen, const double *d_scen, const double *d_overrides, double *d_z, double *d_lam, double *d_stats);

/* launch geometry the handle's problem runs with: threads per workgroup (= per scenario) and shooting nodes per thread */
int msd_problem_geometry(msd_handle h, int *threads_per_scenario, int *nodes_per_thread);

/*
 * Split solves.  Every family solves a batch in two launches on the handle's stream: a first pass (the kernels with the structure of the reference's
 * rolling stock compiled in: the fused interior-point iteration alone; the others: the general iteration without its cold paths), and a follow-up
 * kernel (general iteration, restoration phase, watchdog procedure, second attempt) for the scenarios the first pass hands over through a list in
 * device memory -- in the reference all of that is inside the one call of IPOPT (ocp.py:359).
 * Telemetry, never reset: counts[0] scenarios handed over since the handle was created, counts[1 + why] by reason (0 no fused start for the
 * scenario, 1 wrong inertia or scan breakdown, 2 tiny step, 3 rejected first trial point with a second-order correction due, 4 line search
 * broke down, 5 breakdown that asks for the second attempt, 6 ten shortened iterations in a row: the watchdog procedure is due).  Waits for the
 * handle's stream.  n <= 8 entries are written.
 */
int msd_problem_follow_counts(msd_handle h, int *counts, int n);

/* msd_solve_batch with per-scenario rolling-stock overrides: overrides[nscen][MSD_OV_COUNT] (host), NULL = none */
int msd_solve_batch_ex(msd_handle h, int nscen, const double *scen, const double *overrides, double *z_out, double *lam_out, double *stats,
                       float *kernel_ms);

/*
 * msd_solve_batch_ex with a primal warm start (the reference always cold-starts,
 * ocp.py:325-339, so this has no reference counterpart: it reaches the same optimum in fewer iterations).
 *   z_guess    : [nscen][nz] starting points in the layout of z_out (e.g. the previous MPC solution moved to the new
 *                grid); NULL = cold start, mu_init/bound_push ignored
 *   mu_init    : initial barrier parameter (IPOPT option mu_init; cold start uses 0.1)
 *   bound_push : relative distance the guess keeps from bounds (IPOPT warm_start_bound_push/_frac; cold start 1e-2).
 * Bound and slack multipliers start at mu_init/slack, equality multipliers from the least-squares estimate.
 */
int msd_solve_batch_warm(msd_handle h, int nscen, const double *scen, const double *overrides, const double *z_guess, double mu_init,
                         double bound_push, double *z_out, double *lam_out, double *stats, float *kernel_ms);
/*
 * Warm start from the handle's own previous solve, kept on the device: the batch that msd_solve_batch* solved last on this handle
 * (same scenarios in the same order), `shift_intervals` intervals further down the horizon -- the shrinking-horizon re-solve of
 * BASELINE config 4, where the new grid is the tail of the old one (Track.updateLimits(positionStart), track.py:420-450, then
 * msd_problem_reconfigure with num_intervals smaller by shift_intervals).  The guess of a scenario is the tail of its stored z;
 * nothing is uploaded.  A scenario whose previous solve failed, or whose warm-started solve breaks down, is solved from the
 * problem's own starting point inside the same launch.
 */
int msd_solve_batch_shifted(msd_handle h, int nscen, const double *scen, const double *overrides, int shift_intervals, double mu_init,
                            double bound_push, double *z_out, double *lam_out, double *stats, float *kernel_ms);
/*
 * Primal-dual warm starts.  With `on` != 0 every solve of the handle records its multipliers on the device (MSD_DUAL_STRIDE doubles
 * per shooting node: dynamics multipliers 2, row multipliers 5, bound multipliers of the variables 5 + 5 and of the row slacks 5 + 5)
 * and msd_solve_batch_shifted starts from them as well: constraint multipliers as recorded (no least-squares estimate), bound and
 * slack multipliers too but not below 1e-3 of their central-path value mu_init / slack.  About 8 instead of 15 iterations per
 * re-solve of a shrinking horizon; mu_init = 1e-4 is a good choice then (1e-2 for primal-only warm starts).
 */
#define MSD_DUAL_STRIDE 27
int msd_problem_keep_duals(msd_handle h, int on);
int msd_synchronize(msd_handle h);

/*
 * The same batch over several handles -- one per device, created from the same problem description record. SURVEY 8b: `devices[]`;
 * 8e: one stream per device from a single process.  Handle k solves the contiguous slice [k nscen / n, (k + 1) nscen / n);
 * every slice is enqueued on its device before any is waited for; no data moves between devices.  kernel_ms receives the
 * longest of the per-device kernel times.  Buffers are host buffers laid out as for msd_solve_batch_warm (z_guess may be NULL).
 */
int msd_solve_batch_multi(const msd_handle *handles, int nhandles, int nscen, const double *scen, const double *overrides, const double *z_guess,
                          double mu_init, double bound_push, double *z_out, double *lam_out, double *stats, float *kernel_ms);

/*
 * The other two interval integrators of TrainIntegrator (mseetc/train.py:303-322) for n independent intervals -- what
 * TrainIntegrator(model, 'CVODES' | 'IRK', opts).solve(...) (train.py:347-364) evaluates, e.g. in simulations/figure4.py.
 *   train5 : sr0, sr1, sr2, g, rho
 *   method : MSD_INTEGRATOR_ADAPTIVE     params = {abstol, reltol}                         (OptionsCVODES, train.py:522-534)
 *            MSD_INTEGRATOR_COLLOCATION  params = {order, numSteps, numApproxSteps, maxIter, C[(order+1)^2], D[order+1]}
 *                                        (OptionsIRK, train.py:484-519; C[r][j] = dL_r/dtau(tau_j), D[r] = L_r(1) on the points
 *                                        {0} + casadi.collocation_points(order, 'radau' | 'legendre'))
 *   t0, b0, ds, w, grad, curv : [n] time, velocity squared, interval length, specific force Fel + Fpb, gradient/1000, curvature
 *   t_out, b_out : [n];  status_out (may be NULL): 0, or 1 where the integrator gave up (Newton not converged / step underflow)
 */
#define MSD_INTEGRATOR_ADAPTIVE 1
#define MSD_INTEGRATOR_COLLOCATION 2
/* TrainIntegrator.calcRollingResistance (train.py:416-454): params = {abstol, reltol}; t0 is ignored and t_out receives the specific
 * energy [J/kg] dissipated by the rolling resistance over the interval */
#define MSD_INTEGRATOR_ROLLING_RESISTANCE 3
int msd_interval_integrate(int device, int n, const double *train5, int method, const double *params, int nparams,
                           const double *t0, const double *b0, const double *ds, const double *w, const double *grad, const double *curv,
                           double *t_out, double *b_out, int *status_out);
const char *msd_interval_last_error(void);

/* device scratch management for callers without their own allocator (ctypes): */
int msd_device_alloc(msd_handle h, unsigned long long bytes, void **dptr);
int msd_device_free(msd_handle h, void *dptr);
int msd_copy_to_device(msd_handle h, void *dst, const void *src, unsigned long long bytes);
int msd_copy_to_host(msd_handle h, void *dst, const void *src, unsigned long long bytes);
/* HIP-event timing of everything enqueued on the handle's stream between begin and end (ms) */
int msd_timer_begin(msd_handle h);
int msd_timer_end(msd_handle h, float *ms);

/*
 * Replaces TrainIntegrator.solve (train.py:347-364) for `n` independent intervals: out[k][0..11] =
 * {t+ - t, b+, d/db, d/dw of both, second derivatives (bb, bw, ww) of both}, w = Fel + Fpb.
 */
int msd_stage_eval(msd_handle h, int n, const double *b, const double *w, const double *ds, const double *grad,
                   const double *curv, double *out12);

/* per-iteration log of scenario 0 of the next msd_solve_batch call: 8 doubles per iteration
 * (iter, objective, inf_pr, inf_du, lg(mu), |d|, alpha_du, alpha_pr); cap = number of rows. */
int msd_set_history(msd_handle h, double *host_hist, int cap);

const char *msd_last_error(void);

/*
 * Post-processing integrations (no problem handle needed).  train5 = {sr0, sr1, sr2, g, rho}; forces are specific [N/kg].
 *
 * msd_resimulate replaces simulateCVODES / IVP (mseetc/utils.py:110-194): time-domain re-simulation of every interval of
 * `nscen` trajectories with accumulated errors; force/dts are [nscen][N], grad/curv [N], outputs [nscen][N+1].
 * msd_integrate_losses replaces TrainIntegrator.initLosses/calcLosses as used by postProcessDataFrame(integrateLosses=True)
 * (mseetc/train.py:367-413, utils.py:261-289): energy lost in traction / regenerative braking over every interval [J/kg];
 * inputs and outputs are [nscen][N]; loss model as in msd_problem_desc.
 */
int msd_resimulate(int device, int nscen, int N, const double *train5, const double *force, const double *dts, const double *grad, const double *curv,
                   const double *s0, const double *v0, double abstol, double reltol, double *pos_out, double *vel_out);
int msd_integrate_losses(int device, int nscen, int N, const double *train5, int loss_kind, double ct, double cr, const double *loss_table, int loss_table_len,
                         const double *force_el, const double *force_pn, const double *dts, const double *grad, const double *curv, const double *vstart,
                         double abstol, double reltol, double *etr_out, double *ebr_out);
const char *msd_post_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
