/*
 * ms_oracle.h -- CPU ORACLE for the multiple-shooting train-control hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library.  The product path is the HIP library declared in include/mseetc_hip.h.
 *
 * What it restates (reference = dkouzoup/ms-eetc, paths relative to the reference):
 *   - the NLP transcription of mseetc/ocp.py:96-284 (variables, bounds, constraint rows,
 *     objective, scaling) on the grid of mseetc/track.py:91-107,
 *   - the interval integrator of mseetc/train.py:225-277,294-301,324-344 (RK4 via
 *     casadi.simpleRK semantics + trapezoidal time update), with first and second
 *     derivatives (CasADi AD in the reference),
 *   - the static loss rows of mseetc/train.py:199-216 + mseetc/utils.py:197-220 and the dynamic
 *     loss model of mseetc/efficiency.py:7-141 (table handed over as bicubic patches),
 *   - the NLP solver: the reference calls casadi.nlpsol('ipopt') (ocp.py:290,359), i.e.
 *     the third-party IPOPT bundled with casadi==3.6.3 (setup.py:12; IPOPT 3.14.x with
 *     MUMPS), which is NOT present under /root/reference and not installable here.  Its
 *     published algorithm (Waechter & Biegler, "On the implementation of an interior-point
 *     filter line-search algorithm for large-scale nonlinear programming", Math. Prog.
 *     106(1), 2006 -- Algorithm A with IPOPT's default option values) is restated in
 *     ms_oracle.c; the linear algebra (MUMPS LDL^T + inertia) is replaced by a Riccati
 *     recursion over the stages, whose pivots carry the same inertia information.
 *
 * PARITY STATUS: the NLP/integrator/grid restatement is pinned by the reference's own
 * stored numbers (tests/test_oracle_pins.py: GPOPS energies 440.14 kWh in the limit N -> inf,
 * figure5.py:96 minimumTime = 272.4726 s reproduced as 272.47254 s, figure4.py:22-23 speeds,
 * figure3.py:113-115 loss ratio) and by the reference's unit tests restated in
 * tests/test_reference_unit_tests.py.
 * At the IPOPT boundary parity is UNPINNED: the repository stores no casadiSolver output and
 * casadi cannot run here.  Substitute evidence: an independent numpy KKT +
 * second-order certificate of every oracle solution (tests/nlp_numpy.py: kkt_certificate).
 *
 * Two things here have no counterpart in the reference and exist so that the product's versions
 * of them can be checked iterate for iterate: the profile starting point (oracle_solve_start,
 * start = 1) and the primal warm start (oracle_solve_warm).  Both end at the optimum of the
 * reference's cold start (start = 0, ocp.py:325-339), which is what the pins above refer to.
 */
#ifndef MS_ORACLE_H
#define MS_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* integer parameters (ip[]) */
enum {
    OR_IP_N = 0,          /* number of shooting intervals                                    */
    OR_IP_WITH_PN,        /* pneumatic brake variable present (train.forceMinPn != 0)        */
    OR_IP_HAS_POWER,      /* power rows present (powerMax or powerMin set, ocp.py:184)       */
    OR_IP_ENERGY_OPT,     /* 1: energy optimal, 0: time optimal                              */
    OR_IP_NUM_STEPS,      /* RK4 steps per interval                                          */
    OR_IP_NUM_APPROX,     /* trapezoidal time sub-intervals (0: integrate time with RK4)     */
    OR_IP_LOSS_KIND,      /* 0 none, 1 static efficiencies, 2 dynamic table (oracle_set_loss_table)   */
    OR_IP_MAX_ITER,
    OR_IP_INTEGRATOR,     /* shooting integrator (train.py:280-322): 0 simpleRK order 4, 1 simpleIRK (collocation), 2 adaptive (CVODES' role) */
    OR_IP_COLL_DEGREE,    /* collocation points per step (OptionsIRK.order, train.py:485); tables via oracle_set_collocation */
    OR_IP_NEWTON_ITERS,   /* OptionsIRK.maxIter (train.py:493)                               */
    OR_IP_INTEGRATE_LOSSES, /* OptionsCasadiSolver.integrateLosses (ocp.py:28,231-241): loss slacks from the loss power integrated over the
                             * running time of the interval (static efficiencies only)                                    */
    OR_IP_WATCHDOG_TRIGGER, /* IPOPT's watchdog_shortened_iter_trigger: 0 = its default (10), < 0 = no watchdog procedure */
    OR_IP_COUNT
};

/* real parameters (dp[]) -- "specific" = per kg of mass*rho */
enum {
    OR_DP_SR0 = 0, OR_DP_SR1, OR_DP_SR2,  /* specific Davis coefficients (train.py:181-183)  */
    OR_DP_G, OR_DP_RHO,
    OR_DP_FMAX,           /* upper bound on Fel                                              */
    OR_DP_FMIN,           /* lower bound on Fel (0 without regenerative brake)               */
    OR_DP_FMIN_PN,        /* lower bound on Fpb                                              */
    OR_DP_PW_UPPER,       /* abs(upper) of the power rows (ocp.py:186,191)                   */
    OR_DP_PW_LOWER,       /* abs(lower) of the power rows (ocp.py:187,192)                   */
    OR_DP_ACC_MIN, OR_DP_ACC_MAX,
    OR_DP_LOSS_CT,        /* s >= ct*Fel   (static model, (1-eta_t)/eta_t)                   */
    OR_DP_LOSS_CR,        /* s >= -cr*Fel  (static model, 1-eta_r)                           */
    OR_DP_VMIN_SQ,        /* minimumVelocity^2                                               */
    OR_DP_OBJ_DEN,        /* scalingFactorObjective (ocp.py:278,282)                         */
    OR_DP_TOL,            /* IPOPT tol (1e-8)                                                */
    OR_DP_T0, OR_DP_TEND, OR_DP_V0SQ, OR_DP_VNSQ,   /* scenario (already clipped, ocp.py:343-344) */
    OR_DP_INT_ATOL, OR_DP_INT_RTOL,                 /* adaptive integrator: OptionsCVODES absTol, relTol (train.py:521-534) */
    OR_DP_COUNT
};

/* stats[] */
enum {
    OR_ST_STATUS = 0,     /* 0 solved, 1 solved to acceptable level, <0 failure              */
    OR_ST_ITERS,
    OR_ST_OBJ,            /* NLP objective (kWh for the energy problem, scaled time otherwise) */
    OR_ST_KKT,            /* final scaled NLP error E_0                                       */
    OR_ST_MU,
    OR_ST_DUAL_INF, OR_ST_CONSTR_VIOL, OR_ST_COMPL,
    OR_ST_N_REG,          /* iterations that needed inertia correction                       */
    OR_ST_N_SOC,          /* second-order corrections taken                                  */
    OR_ST_N_BACKTRACK,    /* total backtracking steps                                        */
    OR_ST_N_RESTO,        /* restoration phases entered                                      */
    OR_ST_N_WATCHDOG,     /* watchdog procedures started                                     */
    OR_ST_COUNT
};

#define OR_STATUS_SOLVED 0
#define OR_STATUS_ACCEPTABLE 1
#define OR_STATUS_MAXITER (-1)
#define OR_STATUS_LINESEARCH (-2)    /* the restoration phase failed (or was entered at an almost feasible point): IPOPT's Restoration_Failed */
#define OR_STATUS_REGULARIZATION (-3)
#define OR_STATUS_NUMERIC (-4)
#define OR_STATUS_TINY_STEP (-5)
#define OR_STATUS_INFEASIBLE (-6)    /* the restoration phase converged to a stationary point of the infeasibility: IPOPT's Infeasible_Problem_Detected */

/*
 * Solve one OCP.  ds/grad/curv have N entries (grad already divided by 1000, ocp.py:195),
 * bmax has N+1 entries (bmax[i] = min(vlim_i, vmax, vlim_{i-1})^2 for interior nodes,
 * ocp.py:266-269; entries 0 and N unused).
 *   z_out   : nz = (4+withPn)*N + 2 doubles in the reference's layout (ocp.py:166-272)
 *   lam_out : multipliers of the 7N (or fewer) constraint rows in the reference's row order, may be NULL
 *   hist    : optional iteration log, 8 doubles per iteration (iter, obj, inf_pr, inf_du, lg(mu), |d|, alpha_du, alpha_pr), may be NULL
 */
int oracle_solve(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                 const double *bmax, double *z_out, double *lam_out, double *stats, double *hist, int hist_cap);

/* Same with a primal warm start: guess in the layout of z_out (NULL = cold start), barrier parameter mu0, interior push `push`. */
int oracle_solve_warm(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                      const double *bmax, const double *guess, double mu0, double push,
                      double *z_out, double *lam_out, double *stats, double *hist, int hist_cap);

/*
 * Tables of casadi.simpleIRK for OR_IP_INTEGRATOR = 1: C[(d+1)*(d+1)] (row r, column j: derivative of the Lagrange polynomial r at point j)
 * and D[d+1] (the polynomials at 1) on the points {0} + collocation points; copied (d <= 9).
 */
void oracle_set_collocation(int d, const double *C, const double *D);

/* primal-dual warm start: multipliers of a node in OR_DUAL_STRIDE doubles -- lam (2), nu (5), zL (5), zU (5), zLs (5), zUs (5) */
#define OR_DUAL_STRIDE 27
int oracle_solve_dual(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                      const double *bmax, int start, const double *guess, const double *dual_guess, double mu0, double push,
                      double *z_out, double *lam_out, double *dual_out, double *stats);

/* start = 0: the reference's cold start (ocp.py:325-339); start = 1: profile start (see ms_oracle.c), repeated cold when it breaks down (any failure but the iteration limit) */
int oracle_solve_start(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                       const double *bmax, int start, double *z_out, double *lam_out, double *stats, double *hist, int hist_cap);
int oracle_solve_batch_start(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                             const double *bmax, int start, int nscen, const double *scen, double *z_out, double *stats, int nthreads);

/* Batch of scenarios (t0, T, v0sq, vNsq per scenario, 4 doubles each) with OpenMP over scenarios. */
int oracle_solve_batch(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                       const double *bmax, int nscen, const double *scen, double *z_out, double *stats, int nthreads);

/*
 * One shooting interval (train.py:347-364): out = {tau, bplus, dtau/db, dtau/dw, dbplus/db, dbplus/dw,
 * d2tau/dbdb, d2tau/dbdw, d2tau/dwdw, d2bplus/dbdb, d2bplus/dbdw, d2bplus/dwdw} where tau = t+ - t and w = Fel+Fpb.
 */
void oracle_stage_eval(const int *ip, const double *dp, double b, double w, double ds, double grad, double curv, double *out12);

/* parameter block of the dynamic loss model (efficiency.py), see ms_oracle.c; the pointer must stay valid */
void oracle_set_loss_table(const double *block);

/* feasibility restoration phase on (default) / off: off, a solve whose line search breaks down ends with OR_STATUS_LINESEARCH like before */
void oracle_set_restoration(int on);

/* longest run of successive iterations with a shortened (backtracked) step over the solves since the last reset: the quantity IPOPT's
 * watchdog compares with watchdog_shortened_iter_trigger = 10 (the watchdog itself is not restated) */
int oracle_max_shortened_run(int reset);
/* IPOPT's watchdog procedure (on by default, like in IPOPT); counts of procedures started / ended by an accepted trial point since the last reset */
void oracle_set_watchdog(int on);
void oracle_watchdog_counts(int *started, int *succeeded, int reset);
int oracle_watchdog_forced_steps(int reset);      /* trial points taken without the filter's consent while a procedure ran */
void oracle_loss_rows(const double *block, double f, double v, double *out12);

/* NLP functions at z (reference layout): objective and the constraint rows in the reference's order. */
void oracle_nlp_eval(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                     const double *z, double *obj, double *g_out);

#ifdef __cplusplus
}
#endif
#endif
