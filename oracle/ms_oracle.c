/*
 * ms_oracle.c -- CPU ORACLE (test infrastructure, not product code; see ms_oracle.h).
 *
 * Serial, double precision, written for clarity: every interval is assembled as a dense
 * 8x8 local system and the stage recursion uses plain dense 3x3 / 6x6 loops.  The HIP
 * product (ms-eetc_amd/csrc) is a separate implementation with a different decomposition
 * (one thread per stage, sparsity-exploiting Riccati); the two only share the mathematics.
 *
 * Reference citations are relative to the reference repository root.
 *
 * What pins it (ms_oracle.h has the list): the reference's stored solutions and figures for the NLP and its optimum.  The interior-point
 * algorithm itself is IPOPT's, an un-vendored dependency of the reference (ocp.py:290,359): restated from the published algorithm (Waechter &
 * Biegler 2006) and implementation -- for the restoration phase and the watchdog procedure no vector of the reference exists: PARITY UNPINNED
 * for those two (their sections below say what they were restated from and what the tests check instead).
 */
#include "ms_oracle.h"

#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------------
 * second-order forward-mode "jets" in two inputs (b, w): value, gradient, Hessian (00,01,11).
 * The reference gets these derivatives from CasADi's AD of the same expressions.
 * ---------------------------------------------------------------------------------------- */
typedef struct { double v, g0, g1, h00, h01, h11; } jet;

static jet j_const(double c) { jet r = {c, 0, 0, 0, 0, 0}; return r; }
static jet j_var(double v, int k) { jet r = {v, k == 0, k == 1, 0, 0, 0}; return r; }
static jet j_add(jet a, jet b) { jet r = {a.v + b.v, a.g0 + b.g0, a.g1 + b.g1, a.h00 + b.h00, a.h01 + b.h01, a.h11 + b.h11}; return r; }
static jet j_sub(jet a, jet b) { jet r = {a.v - b.v, a.g0 - b.g0, a.g1 - b.g1, a.h00 - b.h00, a.h01 - b.h01, a.h11 - b.h11}; return r; }
static jet j_scale(jet a, double s) { jet r = {a.v*s, a.g0*s, a.g1*s, a.h00*s, a.h01*s, a.h11*s}; return r; }
static jet j_axpy(double s, jet a, jet y) { return j_add(j_scale(a, s), y); }
/* outer function F(a) with F' = f1, F'' = f2 */
static jet j_chain(jet a, double F, double f1, double f2)
{
    jet r;
    r.v = F;
    r.g0 = f1*a.g0;
    r.g1 = f1*a.g1;
    r.h00 = f1*a.h00 + f2*a.g0*a.g0;
    r.h01 = f1*a.h01 + f2*a.g0*a.g1;
    r.h11 = f1*a.h11 + f2*a.g1*a.g1;
    return r;
}
static jet j_sqrt(jet a) { double s = sqrt(a.v); return j_chain(a, s, 0.5/s, -0.25/(a.v*s)); }
static jet j_recip(jet a) { double r = 1.0/a.v; return j_chain(a, r, -r*r, 2*r*r*r); }

static jet j_mul(jet a, jet b)
{
    jet r;
    r.v = a.v*b.v;
    r.g0 = a.v*b.g0 + b.v*a.g0;
    r.g1 = a.v*b.g1 + b.v*a.g1;
    r.h00 = a.v*b.h00 + 2*a.g0*b.g0 + b.v*a.h00;
    r.h01 = a.v*b.h01 + a.g0*b.g1 + a.g1*b.g0 + b.v*a.h01;
    r.h11 = a.v*b.h11 + 2*a.g1*b.g1 + b.v*a.h11;
    return r;
}

/* ------------------------------------------------------------------------------------------
 * dynamic loss model (reference: mseetc/efficiency.py:7-141, utils.py:197-220, train.py:214-217).
 * Parameter block (doubles): forceMax, powerMax, vTurn, vMin, vMax, auxiliaries, cgT = (1-etaG)/etaG, cgB = 1-etaG, R, V,
 * totalMass, nx, ny, xb[nx+1], yb[ny+1], coef[nx][ny][4][4] (bicubic patches about the cell centres, ascending powers).
 * vTurn = 0 marks a direct table of the total losses over (signed force [N], speed), see spec_losses.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    double Fmax, Pmax, vTurn, vMin, vMax, aux, cgT, cgB, R, V, M;
    int nx, ny;
    const double *xb, *yb, *coef;
} DynLoss;

static const double *g_loss_block = NULL;
void oracle_set_loss_table(const double *block) { g_loss_block = block; }

static void dyn_init(DynLoss *D, const double *b)
{
    D->Fmax = b[0]; D->Pmax = b[1]; D->vTurn = b[2]; D->vMin = b[3]; D->vMax = b[4]; D->aux = b[5]; D->cgT = b[6]; D->cgB = b[7];
    D->R = b[8]; D->V = b[9]; D->M = b[10]; D->nx = (int)b[11]; D->ny = (int)b[12];
    D->xb = b + 13; D->yb = D->xb + D->nx + 1; D->coef = D->yb + D->ny + 1;
}

/* bicubic table: value and derivatives up to second order wrt (x, y); all zero outside the x range (efficiency.py:137) */
static void table_eval(const DynLoss *D, double x, double y, double out[6])
{
    for (int k = 0; k < 6; k++) out[k] = 0;
    if (x < D->xb[0] || x > D->xb[D->nx]) return;
    int ix = 0, iy = 0;
    while (ix + 1 < D->nx && x >= D->xb[ix + 1]) ix++;
    while (iy + 1 < D->ny && y >= D->yb[iy + 1]) iy++;
    const double dx = x - 0.5*(D->xb[ix] + D->xb[ix + 1]), dy = y - 0.5*(D->yb[iy] + D->yb[iy + 1]);
    const double *c = D->coef + 16*(ix*D->ny + iy);
    const double X[4] = {1, dx, dx*dx, dx*dx*dx}, X1[4] = {0, 1, 2*dx, 3*dx*dx}, X2[4] = {0, 0, 2, 6*dx};
    const double Y[4] = {1, dy, dy*dy, dy*dy*dy}, Y1[4] = {0, 1, 2*dy, 3*dy*dy}, Y2[4] = {0, 0, 2, 6*dy};
    for (int p = 0; p < 4; p++)
        for (int q = 0; q < 4; q++) {
            const double cc = c[4*p + q];
            out[0] += cc*X[p]*Y[q]; out[1] += cc*X1[p]*Y[q]; out[2] += cc*X[p]*Y1[q];
            out[3] += cc*X2[p]*Y[q]; out[4] += cc*X1[p]*Y1[q]; out[5] += cc*X[p]*Y2[q];
        }
}

/* specific total losses [W/kg] of one branch as a jet in (f, v): traction branch for f >= 0, braking branch for f < 0 */
static jet spec_losses(const DynLoss *D, int traction, double f, double v)
{
    jet F = j_scale(j_var(f, 0), D->M), vv = j_var(v, 1);
    jet vc = (v >= D->vMin && v <= D->vMax) ? vv : j_const(v < D->vMin ? D->vMin : D->vMax);   /* efficiency.py:40 */
    /* vTurn = 0: direct table of the total losses [W] over (signed force [N], speed): a user-supplied L(F, v) (train.py:190-219) tabulated
     * by the caller; no load conversion, no gear / auxiliaries / transformer terms, no zeroing */
    const int direct = !(D->vTurn > 0);
    jet absF = (traction || direct) ? F : j_scale(F, -1);
    jet load = direct ? absF : (vc.v <= D->vTurn) ? j_scale(absF, 100/D->Fmax) : j_scale(j_mul(absF, vc), 100/D->Pmax);   /* efficiency.py:7-12 */
    double t[6];
    table_eval(D, load.v, vc.v, t);
    jet motor;
    motor.v = t[0];
    motor.g0 = t[1]*load.g0 + t[2]*vc.g0;
    motor.g1 = t[1]*load.g1 + t[2]*vc.g1;
    motor.h00 = t[1]*load.h00 + t[2]*vc.h00 + t[3]*load.g0*load.g0 + 2*t[4]*load.g0*vc.g0 + t[5]*vc.g0*vc.g0;
    motor.h01 = t[1]*load.h01 + t[2]*vc.h01 + t[3]*load.g0*load.g1 + t[4]*(load.g0*vc.g1 + load.g1*vc.g0) + t[5]*vc.g0*vc.g1;
    motor.h11 = t[1]*load.h11 + t[2]*vc.h11 + t[3]*load.g1*load.g1 + 2*t[4]*load.g1*vc.g1 + t[5]*vc.g1*vc.g1;
    if (direct) return j_scale(motor, 1/D->M);                                   /* train.py:216 */
    if (!(motor.v > 0)) return j_const(0);                                       /* efficiency.py:137 */
    jet pW = traction ? j_mul(F, vv) : j_scale(j_mul(F, vv), -1);               /* efficiency.py:108-109 */
    jet gear = j_scale(pW, traction ? D->cgT : D->cgB);                          /* efficiency.py:112-116 */
    jet Pm, inner;
    if (traction) { Pm = j_add(j_add(pW, gear), j_add(motor, j_const(D->aux))); inner = j_sub(j_const(D->V*D->V), j_scale(Pm, 4*D->R)); }
    else { Pm = j_sub(j_sub(pW, gear), j_add(motor, j_const(D->aux))); inner = j_add(j_const(D->V*D->V), j_scale(Pm, 4*D->R)); }
    jet dif = j_sub(j_const(D->V), j_sqrt(inner));
    jet trafo = j_scale(j_mul(dif, dif), 1/(4*D->R));                            /* efficiency.py:127-130 */
    jet total = j_add(j_add(gear, motor), j_add(j_const(D->aux), trafo));
    return j_scale(total, 1/D->M);                                               /* train.py:216 */
}

/*
 * The two loss rows of ocp.py:225-226 as functions of (f, vbar): g = L(f, v)/v for the traction part (row 0) and the
 * regenerative-brake part (row 1), each extended linearly through f = 0 (utils.py:197-220: slope = dL/df at +-1e-10,
 * intercept = L(0, v)).  out[row] = {g, g_f, g_v, g_ff, g_fv, g_vv}.  In the linear-extension branch the third
 * derivative d3L/df dv2 that g_vv would need is dropped (the branch is never active at a solution; only the
 * curvature used by Newton's method is affected, not the NLP).
 */
/* the split loss power itself (utils.py:197-220; specific, W/kg): out = {L, L_f, L_v, L_ff, L_fv, L_vv} of the traction part (row 0) or the
 * regenerative-brake part (row 1) at (f, v); beta = spec_losses(D, 1, 0, v) */
static void loss_split(const DynLoss *D, int row, double f, double v, const jet *beta, double out[6])
{
    const double tol = 1e-10;
    const int traction = (row == 0);
    const int truth = traction ? (f >= 0) : (f < 0);
    if (truth) {
        jet s = spec_losses(D, traction, f, v);
        out[0] = s.v; out[1] = s.g0; out[2] = s.g1; out[3] = s.h00; out[4] = s.h01; out[5] = s.h11;
    } else {
        jet a = spec_losses(D, traction, traction ? tol : -tol, v);    /* slope alpha(v) = a.g0, alpha'(v) = a.h01 */
        out[0] = a.g0*f + beta->v; out[1] = a.g0; out[2] = a.h01*f + beta->g1; out[3] = 0; out[4] = a.h01; out[5] = beta->h11;
    }
}

static void loss_rows(const DynLoss *D, double f, double v, double out[2][6])
{
    jet beta = spec_losses(D, 1, 0.0, v);
    for (int row = 0; row < 2; row++) {
        double l[6];
        loss_split(D, row, f, v, &beta, l);
        const double L = l[0], Lf = l[1], Lv = l[2], Lff = l[3], Lfv = l[4], Lvv = l[5];
        const double iv = 1/v;
        out[row][0] = L*iv;
        out[row][1] = Lf*iv;
        out[row][2] = Lv*iv - L*iv*iv;
        out[row][3] = Lff*iv;
        out[row][4] = Lfv*iv - Lf*iv*iv;
        out[row][5] = Lvv*iv - 2*Lv*iv*iv + 2*L*iv*iv*iv;
    }
}

/* test hook: the two loss rows and their derivatives at (f, v) for a parameter block */
void oracle_loss_rows(const double *block, double f, double v, double *out12)
{
    DynLoss D; dyn_init(&D, block);
    double lr[2][6];
    loss_rows(&D, f, v, lr);
    for (int k = 0; k < 6; k++) { out12[k] = lr[0][k]; out12[6 + k] = lr[1][k]; }
}

/* ------------------------------------------------------------------------------------------
 * problem data
 * ---------------------------------------------------------------------------------------- */
enum { VT = 0, VB = 1, VF = 2, VP = 3, VS = 4, NV = 5 };          /* variables of a stage          */
enum { RPW0 = 0, RPW1 = 1, RACC = 2, RLTR = 3, RLRG = 4, NR = 5 }; /* inequality rows of an interval */
/* local ordering of an interval: t_i b_i q_i f_i p_i s_i | t_{i+1} b_{i+1};  q_i := f_{i-1} */
enum { LT = 0, LB = 1, LQ = 2, LF = 3, LP = 4, LS = 5, LT1 = 6, LB1 = 7, NL = 8 };
static const int var2loc[NV] = {LT, LB, LF, LP, LS};
enum { WD_TRIGGER_DEFAULT = 10 };      /* IPOPT: watchdog_shortened_iter_trigger */

typedef struct {
    int N, withPn, hasPower, energyOpt, numSteps, numApprox, lossKind, maxIter;
    int integ, collD, newtonIters, intLosses;
    int wdTrigger;          /* watchdog_shortened_iter_trigger (<= 0: no watchdog procedure) */
    double intAtol, intRtol;
    const double *ds, *grad, *curv, *bmax;
    double sr0, sr1, sr2, g, rho, fmax, fmin, fminPn, pwU, pwL, accMin, accMax, ct, cr, vminSq, objDen, tol;
    double t0, tEnd, v0sq, vNsq;
    DynLoss dyn;
} Prob;

static void prob_init(Prob *P, const int *ip, const double *dp, const double *ds, const double *grad, const double *curv, const double *bmax)
{
    P->N = ip[OR_IP_N]; P->withPn = ip[OR_IP_WITH_PN]; P->hasPower = ip[OR_IP_HAS_POWER]; P->energyOpt = ip[OR_IP_ENERGY_OPT];
    P->numSteps = ip[OR_IP_NUM_STEPS]; P->numApprox = ip[OR_IP_NUM_APPROX]; P->lossKind = ip[OR_IP_LOSS_KIND]; P->maxIter = ip[OR_IP_MAX_ITER];
    P->intLosses = ip[OR_IP_INTEGRATE_LOSSES];
    P->wdTrigger = ip[OR_IP_WATCHDOG_TRIGGER] == 0 ? WD_TRIGGER_DEFAULT : ip[OR_IP_WATCHDOG_TRIGGER];
    P->integ = ip[OR_IP_INTEGRATOR]; P->collD = ip[OR_IP_COLL_DEGREE]; P->newtonIters = ip[OR_IP_NEWTON_ITERS];
    P->intAtol = dp[OR_DP_INT_ATOL]; P->intRtol = dp[OR_DP_INT_RTOL];
    P->ds = ds; P->grad = grad; P->curv = curv; P->bmax = bmax;
    P->sr0 = dp[OR_DP_SR0]; P->sr1 = dp[OR_DP_SR1]; P->sr2 = dp[OR_DP_SR2]; P->g = dp[OR_DP_G]; P->rho = dp[OR_DP_RHO];
    P->fmax = dp[OR_DP_FMAX]; P->fmin = dp[OR_DP_FMIN]; P->fminPn = dp[OR_DP_FMIN_PN];
    P->pwU = dp[OR_DP_PW_UPPER]; P->pwL = dp[OR_DP_PW_LOWER]; P->accMin = dp[OR_DP_ACC_MIN]; P->accMax = dp[OR_DP_ACC_MAX];
    P->ct = dp[OR_DP_LOSS_CT]; P->cr = dp[OR_DP_LOSS_CR]; P->vminSq = dp[OR_DP_VMIN_SQ]; P->objDen = dp[OR_DP_OBJ_DEN]; P->tol = dp[OR_DP_TOL];
    P->t0 = dp[OR_DP_T0]; P->tEnd = dp[OR_DP_TEND]; P->v0sq = dp[OR_DP_V0SQ]; P->vNsq = dp[OR_DP_VNSQ];
    if (P->lossKind == 2 && g_loss_block) dyn_init(&P->dyn, g_loss_block);
}

/* velocity-independent specific resistance of interval i: g*grad/rho + cr(curv)/rho (train.py:252-254) */
static double track_resistance(const Prob *P, double grad, double curv)
{
    double c = fabs(curv);
    double cr = (c <= 1.0/300.0) ? P->g*0.5*c/(1 - 30*c) : P->g*0.65*c/(1 - 55*c);
    return P->g*grad*(1/P->rho) + cr*(1/P->rho);
}

/* d(b)/d(sigma) on the unit interval: 2*ds*(w - rr(b) - G)  (train.py:251,254,256,259) */
static jet ode_b(const Prob *P, jet b, jet w, double G, double ds)
{
    jet rr = j_add(j_const(P->sr0), j_add(j_scale(j_sqrt(b), P->sr1), j_scale(b, P->sr2)));
    jet acc = j_sub(j_sub(w, rr), j_const(G));
    return j_scale(acc, 2*ds);
}

/* casadi.simpleRK(f, numSteps, 4) applied to the b-equation only, total step H (train.py:298-301) */
static jet rk4_b(const Prob *P, jet b, jet w, double G, double ds, double H)
{
    double h = H/P->numSteps;
    for (int s = 0; s < P->numSteps; s++) {
        jet k1 = ode_b(P, b, w, G, ds);
        jet k2 = ode_b(P, j_axpy(0.5*h, k1, b), w, G, ds);
        jet k3 = ode_b(P, j_axpy(0.5*h, k2, b), w, G, ds);
        jet k4 = ode_b(P, j_axpy(h, k3, b), w, G, ds);
        jet sum = j_add(j_add(k1, j_scale(k2, 2)), j_add(j_scale(k3, 2), k4));
        b = j_axpy(h/6, sum, b);
    }
    return b;
}

/* ------------------------------------------------------------------------------------------
 * the other two integrators TrainIntegrator offers for the shooting intervals (train.py:303-322)
 * ---------------------------------------------------------------------------------------- */
#define COLL_MAX 9
static int g_coll_d = 0;
static double g_coll_C[(COLL_MAX + 1)*(COLL_MAX + 1)], g_coll_D[COLL_MAX + 1];
void oracle_set_collocation(int d, const double *C, const double *D)
{
    if (d < 1 || d > COLL_MAX) { g_coll_d = 0; return; }
    g_coll_d = d;
    memcpy(g_coll_C, C, sizeof(double)*(d + 1)*(d + 1));
    memcpy(g_coll_D, D, sizeof(double)*(d + 1));
}

/* LU factorisation with partial pivoting (in place) and the solve with it; 0 = singular */
static int lu_factor(int n, double A[COLL_MAX][COLL_MAX], int piv[COLL_MAX])
{
    for (int c = 0; c < n; c++) {
        int p = c; double big = fabs(A[c][c]);
        for (int r = c + 1; r < n; r++) if (fabs(A[r][c]) > big) { big = fabs(A[r][c]); p = r; }
        if (!(big > 0)) return 0;
        piv[c] = p;
        if (p != c) for (int m = 0; m < n; m++) { double x = A[c][m]; A[c][m] = A[p][m]; A[p][m] = x; }
        for (int r = c + 1; r < n; r++) {
            A[r][c] /= A[c][c];
            for (int m = c + 1; m < n; m++) A[r][m] -= A[r][c]*A[c][m];
        }
    }
    return 1;
}
static void lu_solve(int n, double A[COLL_MAX][COLL_MAX], const int piv[COLL_MAX], double x[COLL_MAX])
{
    for (int c = 0; c < n; c++) if (piv[c] != c) { double y = x[c]; x[c] = x[piv[c]]; x[piv[c]] = y; }
    for (int c = 0; c < n; c++) for (int r = c + 1; r < n; r++) x[r] -= A[r][c]*x[c];
    for (int c = n - 1; c >= 0; c--) {
        for (int m = c + 1; m < n; m++) x[c] -= A[c][m]*x[m];
        x[c] /= A[c][c];
    }
}
/* the same solve for every derivative component of an array of jets (the values are left alone) */
static void lu_solve_jets(int n, double A[COLL_MAX][COLL_MAX], const int piv[COLL_MAX], jet R[COLL_MAX], int with_value)
{
    double x[COLL_MAX];
    for (int comp = with_value ? 0 : 1; comp < 6; comp++) {
        for (int j = 0; j < n; j++) x[j] = (&R[j].v)[comp];
        lu_solve(n, A, piv, x);
        for (int j = 0; j < n; j++) (&R[j].v)[comp] = x[j];
    }
}

/*
 * casadi.simpleIRK(ode, numSteps, d, scheme, 'fast_newton') over [0, H] (train.py:310): per step of length dt = H/numSteps the d
 * stage values v solve  dt f(v_j) - (C[0][j] x + sum_r C[r][j] v_r) = 0  (x = start of the step; initial guess v_j = x) and the
 * step ends at D[0] x + sum_r D[r] v_r.  Newton's method runs on the values (at most OptionsIRK.maxIter iterations; like
 * error_on_fail = False the last iterate is used); the derivatives follow from the implicit-function theorem, applied as two
 * Newton corrections in jet arithmetic with the Jacobian at the converged values (the first makes the first derivatives exact,
 * the second the second derivatives).  t != NULL: the time equation dt/dsigma = ds/sqrt(b) is integrated along (numApproxSteps = 0);
 * its stage equations are linear in the time stages.
 */
static jet irk_b(const Prob *P, jet b0, jet w, double G, double ds, double H, jet *t)
{
    const int d = g_coll_d, ld = d + 1;
    const double *C = g_coll_C, *D = g_coll_D;
    const double dt = H/P->numSteps;
    jet xb = b0, xt = t ? *t : j_const(0);
    for (int k = 0; k < P->numSteps; k++) {
        double v[COLL_MAX], A[COLL_MAX][COLL_MAX], F[COLL_MAX];
        int piv[COLL_MAX];
        int valid = 1;
        for (int j = 0; j < d; j++) v[j] = xb.v;
        for (int it = 0; it <= P->newtonIters; it++) {
            double fmaxabs = 0;
            for (int j = 0; j < d; j++) {
                const double sv = sqrt(v[j]);
                const double f = 2*ds*(w.v - (P->sr0 + P->sr1*sv + P->sr2*v[j]) - G), df = -2*ds*(0.5*P->sr1/sv + P->sr2);
                double p = C[0*ld + j + 1]*xb.v;
                for (int r = 0; r < d; r++) { p += C[(r + 1)*ld + j + 1]*v[r]; A[j][r] = -C[(r + 1)*ld + j + 1]; }
                A[j][j] += dt*df;
                F[j] = dt*f - p;
                fmaxabs = fmax(fmaxabs, fabs(F[j]));
            }
            /* the Jacobian of the last pass is the one the derivatives use; a stage value outside the domain makes it NaN: the step is NaN */
            if (!lu_factor(d, A, piv)) { valid = 0; break; }
            if (it == P->newtonIters || !isfinite(fmaxabs) || fmaxabs <= 1e-13*fmax(1.0, fabs(xb.v))) break;
            lu_solve(d, A, piv, F);
            for (int j = 0; j < d; j++) v[j] -= F[j];
        }
        if (!valid) { if (t) *t = j_const(NAN); return j_const(NAN); }
        jet V[COLL_MAX], R[COLL_MAX];
        for (int j = 0; j < d; j++) V[j] = j_const(v[j]);
        for (int pass = 0; pass < 2; pass++) {
            for (int j = 0; j < d; j++) {
                jet p = j_scale(xb, C[0*ld + j + 1]);
                for (int r = 0; r < d; r++) p = j_axpy(C[(r + 1)*ld + j + 1], V[r], p);
                R[j] = j_sub(j_scale(ode_b(P, V[j], w, G, ds), dt), p);
            }
            lu_solve_jets(d, A, piv, R, 0);
            for (int j = 0; j < d; j++) { R[j].v = 0; V[j] = j_sub(V[j], R[j]); }
        }
        jet nb = j_scale(xb, D[0]);
        for (int r = 0; r < d; r++) nb = j_axpy(D[r + 1], V[r], nb);
        if (t) {
            /* sum_r C[r][j] vt_r = dt ds/sqrt(V_j) - C[0][j] xt */
            double M[COLL_MAX][COLL_MAX];
            int pm[COLL_MAX];
            for (int j = 0; j < d; j++) for (int r = 0; r < d; r++) M[j][r] = C[(r + 1)*ld + j + 1];
            lu_factor(d, M, pm);
            for (int j = 0; j < d; j++) R[j] = j_sub(j_scale(j_recip(j_sqrt(V[j])), dt*ds), j_scale(xt, C[0*ld + j + 1]));
            lu_solve_jets(d, M, pm, R, 1);
            jet nt = j_scale(xt, D[0]);
            for (int r = 0; r < d; r++) nt = j_axpy(D[r + 1], R[r], nt);
            xt = nt;
        }
        xb = nb;
    }
    if (t) *t = xt;
    return xb;
}

/*
 * Adaptive integration of (t, b) over the unit interval to the tolerances of OptionsCVODES (train.py:312-322 uses SUNDIALS' CVODES,
 * a third-party code that is not available here): Dormand-Prince 5(4) with the usual step-size controller acting on the values;
 * the derivatives are those of the accepted steps (the discrete map), carried in jet arithmetic.
 */
static void dopri_tb(const Prob *P, jet b0, jet w, double G, double ds, jet *tau, jet *bplus)
{
    static const double a21 = 1.0/5, a31 = 3.0/40, a32 = 9.0/40, a41 = 44.0/45, a42 = -56.0/15, a43 = 32.0/9,
                 a51 = 19372.0/6561, a52 = -25360.0/2187, a53 = 64448.0/6561, a54 = -212.0/729,
                 a61 = 9017.0/3168, a62 = -355.0/33, a63 = 46732.0/5247, a64 = 49.0/176, a65 = -5103.0/18656,
                 b1 = 35.0/384, b3 = 500.0/1113, b4 = 125.0/192, b5 = -2187.0/6784, b6 = 11.0/84,
                 e1 = 71.0/57600, e3 = -71.0/16695, e4 = 71.0/1920, e5 = -17253.0/339200, e6 = 22.0/525, e7 = -1.0/40;
    jet y[2] = {j_const(0), b0}, k[7][2], yn[2];
#define RHS(bj, out) do { (out)[0] = j_scale(j_recip(j_sqrt(bj)), ds); (out)[1] = ode_b(P, (bj), w, G, ds); } while (0)
    double sig = 0, h = 1.0;      /* the whole interval first: most intervals of a shooting grid need one or two steps (a rejected first step costs one set of stages and lands on the right size) */
    RHS(y[1], k[0]);
    for (int step = 0; step < 100000 && sig < 1.0; step++) {
        if (sig + h > 1.0) h = 1.0 - sig;
        jet yb;
        yb = j_axpy(h*a21, k[0][1], y[1]); RHS(yb, k[1]);
        yb = j_axpy(h*a32, k[1][1], j_axpy(h*a31, k[0][1], y[1])); RHS(yb, k[2]);
        yb = j_axpy(h*a43, k[2][1], j_axpy(h*a42, k[1][1], j_axpy(h*a41, k[0][1], y[1]))); RHS(yb, k[3]);
        yb = j_axpy(h*a54, k[3][1], j_axpy(h*a53, k[2][1], j_axpy(h*a52, k[1][1], j_axpy(h*a51, k[0][1], y[1])))); RHS(yb, k[4]);
        yb = j_axpy(h*a65, k[4][1], j_axpy(h*a64, k[3][1], j_axpy(h*a63, k[2][1], j_axpy(h*a62, k[1][1], j_axpy(h*a61, k[0][1], y[1]))))); RHS(yb, k[5]);
        for (int m = 0; m < 2; m++)
            yn[m] = j_axpy(h*b6, k[5][m], j_axpy(h*b5, k[4][m], j_axpy(h*b4, k[3][m], j_axpy(h*b3, k[2][m], j_axpy(h*b1, k[0][m], y[m])))));
        const int finite = isfinite(yn[0].v) && isfinite(yn[1].v) && yn[1].v > 0;
        double err = 0;
        if (finite) {
            RHS(yn[1], k[6]);
            for (int m = 0; m < 2; m++) {
                const double sc = P->intAtol + P->intRtol*fmax(fabs(y[m].v), fabs(yn[m].v));
                err = fmax(err, fabs(h*(e1*k[0][m].v + e3*k[2][m].v + e4*k[3][m].v + e5*k[4][m].v + e6*k[5][m].v + e7*k[6][m].v)/sc));
            }
        }
        if (finite && err <= 1.0) {
            sig += h;
            for (int m = 0; m < 2; m++) { y[m] = yn[m]; k[0][m] = k[6][m]; }     /* first same as last */
        }
        const double fac = !finite ? 0.2 : (err > 0) ? 0.9*pow(err, -0.2) : 5.0;
        h *= fmin(5.0, fmax(0.2, fac));
        if (h < 1e-14) break;      /* the step control has collapsed (the state left the model's domain) */
    }
#undef RHS
    /* not integrated to the end of the interval: no interval map (NaN: the line search rejects the point) */
    if (!(sig >= 1.0)) { y[0] = j_const(NAN); y[1] = j_const(NAN); }
    *tau = y[0]; *bplus = y[1];
}

/* ------------------------------------------------------------------------------------------
 * integrateLosses (ocp.py:231-241 -> TrainIntegrator.initLosses/calcLosses, train.py:367-413): the loss slack of an interval bounds the
 * loss POWER integrated over the interval's running time dt = t_{i+1} - t_i, along the speed of the time-domain model
 * dv/dt = w - rr(v) - G started at v_i.  With constant efficiencies the loss power is (1-eta)/eta f v resp. -(1-eta_r) f v (train.py:199-212),
 * so both integrals are multiples of the distance X(v_i, dt, w) = int_0^dt v.  The reference integrates with CVODES at abstol 1e-8,
 * reltol 1e-6 (train.py:396; third-party SUNDIALS): here the adaptive Dormand-Prince pair at those tolerances, step control on the
 * values, first and second derivatives wrt (v_i, dt, w) carried through the accepted steps.
 * ---------------------------------------------------------------------------------------- */
typedef struct { double v, g[3], h[6]; } jet3;      /* h: 00 01 02 11 12 22 */
static const int J3A[6] = {0, 0, 0, 1, 1, 2}, J3B[6] = {0, 1, 2, 1, 2, 2};
static jet3 j3_const(double c) { jet3 r; memset(&r, 0, sizeof r); r.v = c; return r; }
static jet3 j3_var(double v, int k) { jet3 r = j3_const(v); r.g[k] = 1; return r; }
static jet3 j3_axpy(double s, jet3 a, jet3 y) { y.v += s*a.v; for (int k = 0; k < 3; k++) y.g[k] += s*a.g[k]; for (int k = 0; k < 6; k++) y.h[k] += s*a.h[k]; return y; }
static jet3 j3_scale(jet3 a, double s) { return j3_axpy(s, a, j3_const(0)); }
static jet3 j3_mul(jet3 a, jet3 b)
{
    jet3 r; r.v = a.v*b.v;
    for (int k = 0; k < 3; k++) r.g[k] = a.v*b.g[k] + b.v*a.g[k];
    for (int k = 0; k < 6; k++) r.h[k] = a.v*b.h[k] + b.v*a.h[k] + a.g[J3A[k]]*b.g[J3B[k]] + a.g[J3B[k]]*b.g[J3A[k]];
    return r;
}

static jet3 loss_distance(const Prob *P, double v0, double dt0, double w0, double G)
{
    static const double a21 = 1.0/5, a31 = 3.0/40, a32 = 9.0/40, a41 = 44.0/45, a42 = -56.0/15, a43 = 32.0/9,
                 a51 = 19372.0/6561, a52 = -25360.0/2187, a53 = 64448.0/6561, a54 = -212.0/729,
                 a61 = 9017.0/3168, a62 = -355.0/33, a63 = 46732.0/5247, a64 = 49.0/176, a65 = -5103.0/18656,
                 b1 = 35.0/384, b3 = 500.0/1113, b4 = 125.0/192, b5 = -2187.0/6784, b6 = 11.0/84,
                 e1 = 71.0/57600, e3 = -71.0/16695, e4 = 71.0/1920, e5 = -17253.0/339200, e6 = 22.0/525, e7 = -1.0/40;
    const double atol = 1e-8, rtol = 1e-6;      /* train.py:396 */
    const jet3 dt = j3_var(dt0, 1), w = j3_var(w0, 2);
    jet3 y[2] = {j3_var(v0, 0), j3_const(0)}, k[7][2], yn[2];
    /* d(v, X)/dsigma = dt (w - rr(v) - G, v) on the unit interval */
#define LRHS(vj, out) do { jet3 acc_ = j3_axpy(-P->sr1, (vj), j3_axpy(-P->sr2, j3_mul((vj), (vj)), j3_axpy(1.0, w, j3_const(-P->sr0 - G)))); \
                           (out)[0] = j3_mul(dt, acc_); (out)[1] = j3_mul(dt, (vj)); } while (0)
    double sig = 0, h = 1.0;      /* first try: the whole interval in one step */
    LRHS(y[0], k[0]);
    for (int step = 0; step < 100000 && sig < 1.0; step++) {
        if (sig + h > 1.0) h = 1.0 - sig;
        jet3 s;
        s = j3_axpy(h*a21, k[0][0], y[0]); LRHS(s, k[1]);
        s = j3_axpy(h*a32, k[1][0], j3_axpy(h*a31, k[0][0], y[0])); LRHS(s, k[2]);
        s = j3_axpy(h*a43, k[2][0], j3_axpy(h*a42, k[1][0], j3_axpy(h*a41, k[0][0], y[0]))); LRHS(s, k[3]);
        s = j3_axpy(h*a54, k[3][0], j3_axpy(h*a53, k[2][0], j3_axpy(h*a52, k[1][0], j3_axpy(h*a51, k[0][0], y[0])))); LRHS(s, k[4]);
        s = j3_axpy(h*a65, k[4][0], j3_axpy(h*a64, k[3][0], j3_axpy(h*a63, k[2][0], j3_axpy(h*a62, k[1][0], j3_axpy(h*a61, k[0][0], y[0]))))); LRHS(s, k[5]);
        for (int m = 0; m < 2; m++)
            yn[m] = j3_axpy(h*b6, k[5][m], j3_axpy(h*b5, k[4][m], j3_axpy(h*b4, k[3][m], j3_axpy(h*b3, k[2][m], j3_axpy(h*b1, k[0][m], y[m])))));
        const int finite = isfinite(yn[0].v) && isfinite(yn[1].v);
        double err = 0;
        if (finite) {
            LRHS(yn[0], k[6]);
            for (int m = 0; m < 2; m++) {
                const double sc = atol + rtol*fmax(fabs(y[m].v), fabs(yn[m].v));
                err = fmax(err, fabs(h*(e1*k[0][m].v + e3*k[2][m].v + e4*k[3][m].v + e5*k[4][m].v + e6*k[5][m].v + e7*k[6][m].v)/sc));
            }
        }
        if (finite && err <= 1.0) {
            sig += h;
            for (int m = 0; m < 2; m++) { y[m] = yn[m]; k[0][m] = k[6][m]; }
        }
        const double fac = !finite ? 0.2 : (err > 0) ? 0.9*pow(err, -0.2) : 5.0;
        h *= fmin(5.0, fmax(0.2, fac));
        if (h < 1e-14) break;
    }
#undef LRHS
    if (!(sig >= 1.0)) y[1] = j3_const(NAN);
    return y[1];
}

/* test hook: X and its nine derivatives (v, dt, w; vv, v dt, v w, dt dt, dt w, w w) */
void oracle_loss_distance(const int *ip, const double *dp, double v0, double dt, double w, double grad, double curv, double *out10)
{
    Prob P; prob_init(&P, ip, dp, NULL, NULL, NULL, NULL);
    const jet3 X = loss_distance(&P, v0, dt, w, track_resistance(&P, grad, curv));
    out10[0] = X.v;
    for (int k = 0; k < 3; k++) out10[1 + k] = X.g[k];
    for (int k = 0; k < 6; k++) out10[4 + k] = X.h[k];
}

/* ------------------------------------------------------------------------------------------
 * integrateLosses with a loss TABLE (dynamic loss model of efficiency.py, or any tabulated loss function): the loss slack bounds
 *     E_k(v_i, dt, w, f) = int_0^dt L_k(f, v(t)) dt,   dv/dt = w - rr(v) - G,  v(0) = v_i,     k = traction part / regenerative-brake part
 * (ocp.py:231-241 -> TrainIntegrator.initLosses / calcLosses, train.py:367-413: "energyTrDot = lossesTrFun(F, vel)/totalMass").  L_k is the specific
 * split loss power (loss_split).  Intended semantics: the reference hands initLosses the SPECIFIC functions of train.powerLossesFuns() (ocp.py:99,118-120)
 * and initLosses scales force and result by the mass once more (train.py:376-377) -- exact for losses linear in F v (constant efficiencies: what the
 * static rows above reproduce), off by a factor of the mass in the force argument for a table; the reference's own switch for this combination sits
 * commented out at simulations/figure6.py:178.  Here the loss power is L(F, v)/M with F = f M, like utils.py:261-289 (postProcessDataFrame) integrates it.
 * Same adaptive Dormand-Prince pair as loss_distance at CVODES' tolerances (train.py:396), step control on the values of (v, E_tr, E_rgb); second-order
 * jets in (v_i, dt, w, f) through the accepted steps.
 * ---------------------------------------------------------------------------------------- */
typedef struct { double v, g[4], h[10]; } jet4;      /* variables (v0, dt, w, f); h: 00 01 02 03 11 12 13 22 23 33 */
static const int J4A[10] = {0, 0, 0, 0, 1, 1, 1, 2, 2, 3}, J4B[10] = {0, 1, 2, 3, 1, 2, 3, 2, 3, 3};
static int j4h(int a, int b) { if (a > b) { int t = a; a = b; b = t; } return a == 0 ? b : a == 1 ? 3 + b : a == 2 ? 5 + b : 9; }
static jet4 j4_const(double c) { jet4 r; memset(&r, 0, sizeof r); r.v = c; return r; }
static jet4 j4_var(double v, int k) { jet4 r = j4_const(v); r.g[k] = 1; return r; }
static jet4 j4_axpy(double s, jet4 a, jet4 y) { y.v += s*a.v; for (int k = 0; k < 4; k++) y.g[k] += s*a.g[k]; for (int k = 0; k < 10; k++) y.h[k] += s*a.h[k]; return y; }
static jet4 j4_mul(jet4 a, jet4 b)
{
    jet4 r; r.v = a.v*b.v;
    for (int k = 0; k < 4; k++) r.g[k] = a.v*b.g[k] + b.v*a.g[k];
    for (int k = 0; k < 10; k++) r.h[k] = a.v*b.h[k] + b.v*a.h[k] + a.g[J4A[k]]*b.g[J4B[k]] + a.g[J4B[k]]*b.g[J4A[k]];
    return r;
}
/* L(f, v) with v a jet and f variable 3: l = {L, L_f, L_v, L_ff, L_fv, L_vv} */
static jet4 j4_loss(const double l[6], jet4 v)
{
    jet4 r; r.v = l[0];
    for (int a = 0; a < 4; a++) r.g[a] = l[2]*v.g[a] + (a == 3 ? l[1] : 0.0);
    for (int k = 0; k < 10; k++) {
        const int a = J4A[k], b = J4B[k];
        r.h[k] = l[2]*v.h[k] + l[5]*v.g[a]*v.g[b] + l[4]*((a == 3 ? v.g[b] : 0.0) + (b == 3 ? v.g[a] : 0.0)) + ((a == 3 && b == 3) ? l[3] : 0.0);
    }
    return r;
}

#define DP54_TABLEAU \
    static const double a21 = 1.0/5, a31 = 3.0/40, a32 = 9.0/40, a41 = 44.0/45, a42 = -56.0/15, a43 = 32.0/9, \
                 a51 = 19372.0/6561, a52 = -25360.0/2187, a53 = 64448.0/6561, a54 = -212.0/729, \
                 a61 = 9017.0/3168, a62 = -355.0/33, a63 = 46732.0/5247, a64 = 49.0/176, a65 = -5103.0/18656, \
                 b1 = 35.0/384, b3 = 500.0/1113, b4 = 125.0/192, b5 = -2187.0/6784, b6 = 11.0/84, \
                 e1 = 71.0/57600, e3 = -71.0/16695, e4 = 71.0/1920, e5 = -17253.0/339200, e6 = 22.0/525, e7 = -1.0/40;

/* E[0], E[1] = E_tr, E_rgb with derivatives (order 2) or their values only (order 0: E[k].v; same steps, the step control looks at values only) */
static double g_le_atol = 1e-8, g_le_rtol = 1e-6;      /* train.py:396 (the test hook below tightens them to check the derivatives by differences) */
static void loss_energy(const Prob *P, double v0, double dt0, double w0, double f0, double G, jet4 E[2], int order)
{
    DP54_TABLEAU
    const double atol = g_le_atol, rtol = g_le_rtol;
    const DynLoss *D = &P->dyn;
    const jet4 dt = order ? j4_var(dt0, 1) : j4_const(dt0), w = order ? j4_var(w0, 2) : j4_const(w0);
    jet4 y[3] = {order ? j4_var(v0, 0) : j4_const(v0), j4_const(0), j4_const(0)}, k[7][3], yn[3];
    /* d(v, E_tr, E_rgb)/dsigma = dt (w - rr(v) - G, L_tr(f, v), L_rgb(f, v)) on the unit interval */
#define ERHS(vj, out) do { const jet4 vj_ = (vj); \
        const jet4 acc_ = j4_axpy(-P->sr1, vj_, j4_axpy(-P->sr2, j4_mul(vj_, vj_), j4_axpy(1.0, w, j4_const(-P->sr0 - G)))); \
        (out)[0] = j4_mul(dt, acc_); \
        const jet beta_ = spec_losses(D, 1, 0.0, vj_.v); \
        for (int r_ = 0; r_ < 2; r_++) { double l_[6]; loss_split(D, r_, f0, vj_.v, &beta_, l_); \
            if (!order) { l_[1] = l_[2] = l_[3] = l_[4] = l_[5] = 0; } \
            (out)[1 + r_] = j4_mul(dt, j4_loss(l_, vj_)); } } while (0)
    double sig = 0, h = 1.0;
    ERHS(y[0], k[0]);
    for (int step = 0; step < 100000 && sig < 1.0; step++) {
        if (sig + h > 1.0) h = 1.0 - sig;
        jet4 s;
        s = j4_axpy(h*a21, k[0][0], y[0]); ERHS(s, k[1]);
        s = j4_axpy(h*a32, k[1][0], j4_axpy(h*a31, k[0][0], y[0])); ERHS(s, k[2]);
        s = j4_axpy(h*a43, k[2][0], j4_axpy(h*a42, k[1][0], j4_axpy(h*a41, k[0][0], y[0]))); ERHS(s, k[3]);
        s = j4_axpy(h*a54, k[3][0], j4_axpy(h*a53, k[2][0], j4_axpy(h*a52, k[1][0], j4_axpy(h*a51, k[0][0], y[0])))); ERHS(s, k[4]);
        s = j4_axpy(h*a65, k[4][0], j4_axpy(h*a64, k[3][0], j4_axpy(h*a63, k[2][0], j4_axpy(h*a62, k[1][0], j4_axpy(h*a61, k[0][0], y[0]))))); ERHS(s, k[5]);
        for (int m = 0; m < 3; m++)
            yn[m] = j4_axpy(h*b6, k[5][m], j4_axpy(h*b5, k[4][m], j4_axpy(h*b4, k[3][m], j4_axpy(h*b3, k[2][m], j4_axpy(h*b1, k[0][m], y[m])))));
        const int finite = isfinite(yn[0].v) && isfinite(yn[1].v) && isfinite(yn[2].v) && yn[0].v > 0;
        double err = 0;
        if (finite) {
            ERHS(yn[0], k[6]);
            for (int m = 0; m < 3; m++) {
                const double sc = atol + rtol*fmax(fabs(y[m].v), fabs(yn[m].v));
                err = fmax(err, fabs(h*(e1*k[0][m].v + e3*k[2][m].v + e4*k[3][m].v + e5*k[4][m].v + e6*k[5][m].v + e7*k[6][m].v)/sc));
            }
        }
        if (finite && err <= 1.0) {
            sig += h;
            for (int m = 0; m < 3; m++) { y[m] = yn[m]; k[0][m] = k[6][m]; }
        }
        const double fac = !finite ? 0.2 : (err > 0) ? 0.9*pow(err, -0.2) : 5.0;
        h *= fmin(5.0, fmax(0.2, fac));
        if (h < 1e-14) break;
    }
#undef ERHS
    if (!(sig >= 1.0)) { y[1] = j4_const(NAN); y[2] = j4_const(NAN); }
    E[0] = y[1]; E[1] = y[2];
}

/* test hook: E_tr, E_rgb and their derivatives wrt (v0, dt, w, f): out[k][15] = value, 4 first, 10 second (00 01 02 03 11 12 13 22 23 33);
 * atol, rtol > 0: those tolerances instead of CVODES' */
void oracle_loss_energy(const int *ip, const double *dp, double v0, double dt, double w, double f, double grad, double curv, double atol, double rtol, double *out30)
{
    Prob P; prob_init(&P, ip, dp, NULL, NULL, NULL, NULL);
    jet4 E[2];
    const double a0 = g_le_atol, r0 = g_le_rtol;
    if (atol > 0 && rtol > 0) { g_le_atol = atol; g_le_rtol = rtol; }      /* (tests only, single-threaded) */
    loss_energy(&P, v0, dt, w, f, track_resistance(&P, grad, curv), E, 2);
    g_le_atol = a0; g_le_rtol = r0;
    for (int r = 0; r < 2; r++) {
        out30[15*r] = E[r].v;
        for (int k = 0; k < 4; k++) out30[15*r + 1 + k] = E[r].g[k];
        for (int k = 0; k < 10; k++) out30[15*r + 5 + k] = E[r].h[k];
    }
}

/*
 * One shooting interval: tau = t+ - t and b+ as jets in (b, w), w = Fel + Fpb.
 * numApprox == 0: RK4 on (t, b) jointly (train.py:296-301, dt/ds = ds/sqrt(b) :255,258).
 * numApprox  > 0: RK4 on b, evaluated from b0 at h = j/ns, trapezoidal time (train.py:324-344).
 */
static void interval_map(const Prob *P, double b0, double w0, double G, double ds, jet *tau, jet *bplus)
{
    jet b = j_var(b0, 0), w = j_var(w0, 1);

    if (P->integ == 2) { dopri_tb(P, b, w, G, ds, tau, bplus); return; }       /* train.py:314: numApproxSteps = 0 */
    if (P->integ == 1 && P->numApprox == 0) {
        jet t = j_const(0);
        *bplus = irk_b(P, b, w, G, ds, 1.0, &t);
        *tau = t;
        return;
    }

    if (P->numApprox == 0) {
        double h = 1.0/P->numSteps;
        jet t = j_const(0);
        for (int s = 0; s < P->numSteps; s++) {
            jet k1b = ode_b(P, b, w, G, ds);
            jet k1t = j_scale(j_recip(j_sqrt(b)), ds);
            jet b2 = j_axpy(0.5*h, k1b, b);
            jet k2b = ode_b(P, b2, w, G, ds);
            jet k2t = j_scale(j_recip(j_sqrt(b2)), ds);
            jet b3 = j_axpy(0.5*h, k2b, b);
            jet k3b = ode_b(P, b3, w, G, ds);
            jet k3t = j_scale(j_recip(j_sqrt(b3)), ds);
            jet b4 = j_axpy(h, k3b, b);
            jet k4b = ode_b(P, b4, w, G, ds);
            jet k4t = j_scale(j_recip(j_sqrt(b4)), ds);
            b = j_axpy(h/6, j_add(j_add(k1b, j_scale(k2b, 2)), j_add(j_scale(k3b, 2), k4b)), b);
            t = j_axpy(h/6, j_add(j_add(k1t, j_scale(k2t, 2)), j_add(j_scale(k3t, 2), k4t)), t);
        }
        *tau = t; *bplus = b;
        return;
    }

    int ns = P->numApprox;
    jet prev = b, acc = j_const(0);
    for (int j = 1; j <= ns; j++) {
        jet cur = (P->integ == 1) ? irk_b(P, b, w, G, ds, (double)j/ns, NULL) : rk4_b(P, b, w, G, ds, (double)j/ns);
        /* 2*ds*(e_{j} - e_{j-1})/(v_{j-1} + v_j) */
        jet den = j_add(j_sqrt(prev), j_sqrt(cur));
        acc = j_add(acc, j_scale(j_recip(den), 2*ds*((double)j/ns - (double)(j - 1)/ns)));
        prev = cur;
    }
    *tau = acc; *bplus = prev;
}

void oracle_stage_eval(const int *ip, const double *dp, double b, double w, double ds, double grad, double curv, double *out)
{
    Prob P; prob_init(&P, ip, dp, NULL, NULL, NULL, NULL);
    jet tau, bp;
    interval_map(&P, b, w, track_resistance(&P, grad, curv), ds, &tau, &bp);
    out[0] = tau.v; out[1] = bp.v; out[2] = tau.g0; out[3] = tau.g1; out[4] = bp.g0; out[5] = bp.g1;
    out[6] = tau.h00; out[7] = tau.h01; out[8] = tau.h11; out[9] = bp.h00; out[10] = bp.h01; out[11] = bp.h11;
}

/* ------------------------------------------------------------------------------------------
 * NLP functions in the reference's layout (ocp.py:166-284)
 * ---------------------------------------------------------------------------------------- */
static int rows_per_interval(const Prob *P) { return (P->hasPower ? 2 : 0) + 1 + 2 + (P->energyOpt ? 2 : 0); }

void oracle_nlp_eval(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                     const double *z, double *obj, double *gout)
{
    Prob P; prob_init(&P, ip, dp, ds, grad, curv, NULL);
    int N = P.N, stp = 4 + P.withPn, rpi = rows_per_interval(&P);
    double J = 0;

    for (int i = 0; i < N; i++) {
        const double *zi = z + stp*i;
        double f = zi[0], p = P.withPn ? zi[1] : 0, s = zi[1 + P.withPn], t = zi[2 + P.withPn], b = zi[3 + P.withPn];
        double t1 = (i + 1 < N) ? z[stp*(i + 1) + 2 + P.withPn] : z[stp*N];
        double b1 = (i + 1 < N) ? z[stp*(i + 1) + 3 + P.withPn] : z[stp*N + 1];
        double G = track_resistance(&P, grad[i], curv[i]);
        jet tau, bp;
        interval_map(&P, b, f + p, G, ds[i], &tau, &bp);
        double *g = gout ? gout + rpi*i : NULL;
        int r = 0;
        if (g) {
            if (P.hasPower) { g[r++] = f*sqrt(b); g[r++] = f*sqrt(b1); }
            g[r++] = f + p - (P.sr0 + P.sr1*sqrt(b) + P.sr2*b) - G;
            g[r++] = t1 - (t + tau.v);
            g[r++] = b1 - bp.v;
            if (P.energyOpt) {
                if (P.lossKind == 2) {
                    double lr[2][6];
                    loss_rows(&P.dyn, f, 0.5*(sqrt(b) + sqrt(b1)), lr);      /* ocp.py:221 mid-point speed */
                    g[r++] = s - lr[0][0]; g[r++] = s - lr[1][0];
                } else if (P.intLosses) {
                    const jet3 X = loss_distance(&P, sqrt(b), t1 - t, f + p, G);
                    g[r++] = s - P.ct*f*X.v; g[r++] = s + P.cr*f*X.v;
                } else { g[r++] = s - P.ct*f; g[r++] = s + P.cr*f; }
            }
        }
        if (P.energyOpt) {
            J += P.intLosses ? ds[i]*f + s : ds[i]*(f + s);
            if (i > 0) { double fp = z[stp*(i - 1)]; J += 1e-3*(f - fp)*(f - fp); }
        } else {
            J += 1e-4*(f*f + p*p);
        }
    }
    if (!P.energyOpt) J += z[stp*N];
    *obj = J/P.objDen;
}

/* ------------------------------------------------------------------------------------------
 * interior-point workspace
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    double x[NV];        /* t b f p s  (f p s unused at the terminal stage) */
    double sig[NR];      /* slacks of the inequality rows of interval i */
    double lam[2];       /* dynamics multipliers (t, b) of interval i */
    double nu[NR];       /* multipliers of d(x) - sigma = 0 */
    double zL[NV], zU[NV], zLs[NR], zUs[NR];
} StageIt;

typedef struct {
    double lb[NV], ub[NV];
    int on[NV], hasL[NV], hasU[NV];   /* on: free variable of the NLP (exists and not fixed) */
} StageBd;

typedef struct {
    /* function values */
    double tau, bplus, c[2], d[NR];
    double tg[2], bg[2], th[3], bh[3];     /* derivatives of tau, bplus wrt (b, w) */
    /* row derivatives in local ordering */
    double gr[NR][NL], hr[NR][NL][NL];
    double objg[NL], objh[NL][NL];
} StageEv;

typedef struct {
    double H[6][6], h[6], E[6][3];         /* condensed stage system, cross term y_i^T E x_{i+1} */
    double F[3][6], r[3];
    double K[3][3], k[3], Pn[3][3], pn[3]; /* feedback, and the (P,p) of stage i+1 used at i */
    double ef[6], e0; int je;               /* terminal elimination: d(control je) = ef . y~ + e0 (last interval only); je = 3 (Fel) or 4 (Fpb) */
    double M[2][2], Dt, Db;                 /* restoration: (P2 + D^-1)^-1 on the relaxed rows (t, b), their D = n/zn + p/zp (unscaled rows) */
    double Ghat[6][6], ghat[6];             /* G, g before elimination (for multiplier recovery) */
    double oa, ob;                          /* the two forces' own curvatures H_ff - H_fp, H_pp - H_fp, accumulated on their own (compute_direction) */
} StageKkt;

typedef struct {
    double dx[NV], dsig[NR], dlam[2], dnu[NR], dzL[NV], dzU[NV], dzLs[NR], dzUs[NR];
} StageDir;

/*
 * Feasibility restoration (IPOPT's MinC_1NrmRestorationPhase, Waechter & Biegler 2006 section 3.3; reached by the reference through
 * ocp.py:290,359 and surfaced at :362-370).  Every equality row of the barrier problem -- the two dynamics rows of an interval
 * (scaled) and the rows d(x) - sigma = 0 -- is relaxed with a pair of non-negative variables: row + n - p = 0.
 */
enum { NC = 2 + NR };
typedef struct { double n[NC], p[NC], zn[NC], zp[NC]; } StageRs;
typedef struct { double dn[NC], dp[NC], dzn[NC], dzp[NC]; } StageRsDir;
typedef struct Resto {
    double rho, eta;            /* penalty parameter (resto_penalty_parameter = 1000), proximity weight sqrt(mu) */
    StageRs *v, *trial;         /* per interval */
    StageRsDir *d;
    double (*xR)[NV], (*dr)[NV];/* reference point and the scaling D_R = 1/max(1, |x_R|) of the proximity term */
} Resto;

typedef struct {
    Prob P;
    struct Resto *resto;     /* non-null while the restoration problem is being solved: compute_direction() then builds its Newton system */
    int N, rowOn[NR];
    double dL[NR], dU[NR]; int rhasL[NR], rhasU[NR];
    double rs[NR];          /* inequality row scaling (gradient based) */
    double sf;              /* objective scaling */
    double *sct, *scb;      /* dynamics row scaling (enters norms only) */
    double *G;              /* track resistance per interval */
    StageIt *it, *trial;
    StageBd *bd;
    StageEv *ev;
    StageKkt *kk;
    StageDir *dir, *soc;
    double HN[3][3], hN[3]; /* terminal stage */
    double mu, tau;
    /* filter */
    double filt_theta[512], filt_phi[512]; int nfilt;
    double theta_max, theta_min;
    /* inertia correction memory */
    double delta_last;
    /* stats */
    int n_reg, n_soc, n_back, n_resto, n_wd;
} Ws;

/* IPOPT default option values (Waechter & Biegler 2006, section 3 + IPOPT 3.14 defaults) */
static const double K_BOUND_RELAX = 1e-8;
static const double K_PUSH = 1e-2;                          /* bound_push = bound_frac (kappa_1, kappa_2) */
static const double K_MU_INIT = 0.1, K_EPS = 10.0;          /* mu_init, barrier_tol_factor (kappa_eps) */
static const double K_MU_LIN = 0.2, K_MU_SUP = 1.5;         /* kappa_mu, theta_mu */
static const double K_TAU_MIN = 0.99;
static const double K_SMAX = 100.0;
static const double K_SIGMA = 1e10;                         /* kappa_Sigma */
static const double K_D = 1e-5;                             /* kappa_d */
static const double G_THETA = 1e-5, G_PHI = 1e-8;           /* gamma_theta, gamma_phi */
static const double K_DELTA = 1.0, S_THETA = 1.1, S_PHI = 2.3, ETA_PHI = 1e-8;
static const double K_SOC = 0.99; static const int P_MAX_SOC = 4;
static const double ALPHA_MIN_FRAC = 0.05;
static const double DW_MIN = 1e-20, DW_0 = 1e-4, DW_MAX = 1e40, KW_MINUS = 1.0/3.0, KW_PLUS = 8.0, KW_PLUS_BAR = 100.0;
static const double LAM_INIT_MAX = 1e3;
static const double ACC_TOL = 1e-6; static const int ACC_ITER = 15;

static int var_exists(const Ws *W, int i, int k)
{
    if (k == VT || k == VB) return 1;
    if (i >= W->N) return 0;
    if (k == VP) return W->P.withPn;
    return 1;
}

static void setup_bounds(Ws *W)
{
    const Prob *P = &W->P; int N = W->N;
    for (int i = 0; i <= N; i++) {
        StageBd *B = &W->bd[i];
        for (int k = 0; k < NV; k++) { B->lb[k] = -INFINITY; B->ub[k] = INFINITY; B->on[k] = var_exists(W, i, k); }
        /* ocp.py:175-181 controls and slack */
        B->lb[VF] = P->fmin; B->ub[VF] = P->fmax;
        B->lb[VP] = P->fminPn; B->ub[VP] = 0;
        B->lb[VS] = 0; B->ub[VS] = INFINITY;
        /* ocp.py:247-272 states */
        if (i == 0) { B->lb[VT] = B->ub[VT] = P->t0; B->lb[VB] = B->ub[VB] = P->v0sq; }
        else if (i == N) { B->lb[VT] = P->t0; B->ub[VT] = P->tEnd; B->lb[VB] = B->ub[VB] = P->vNsq; }
        else { B->lb[VT] = P->t0; B->ub[VT] = P->tEnd; B->lb[VB] = P->vminSq; B->ub[VB] = P->bmax[i]; }
        for (int k = 0; k < NV; k++) {
            if (!B->on[k]) continue;
            if (B->lb[k] == B->ub[k]) { B->on[k] = 0; continue; }   /* fixed_variable_treatment = make_parameter */
            B->hasL[k] = isfinite(B->lb[k]); B->hasU[k] = isfinite(B->ub[k]);
            if (B->hasL[k]) B->lb[k] -= K_BOUND_RELAX*fmax(1.0, fabs(B->lb[k]));
            if (B->hasU[k]) B->ub[k] += K_BOUND_RELAX*fmax(1.0, fabs(B->ub[k]));
        }
    }
    /* rows (ocp.py:184-201, 225-229) */
    for (int r = 0; r < NR; r++) { W->rowOn[r] = 0; W->dL[r] = -INFINITY; W->dU[r] = INFINITY; W->rs[r] = 1; }
    if (P->hasPower) {
        W->rowOn[RPW0] = W->rowOn[RPW1] = 1;
        W->dL[RPW0] = W->dL[RPW1] = -fabs(P->pwL); W->dU[RPW0] = W->dU[RPW1] = fabs(P->pwU);
    }
    W->rowOn[RACC] = 1; W->dL[RACC] = P->accMin; W->dU[RACC] = P->accMax;
    if (P->energyOpt) { W->rowOn[RLTR] = W->rowOn[RLRG] = 1; W->dL[RLTR] = W->dL[RLRG] = 0; }
}

static void relax_row_bounds(Ws *W)
{
    for (int r = 0; r < NR; r++) {
        if (!W->rowOn[r]) continue;
        W->dL[r] *= W->rs[r]; W->dU[r] *= W->rs[r];
        W->rhasL[r] = isfinite(W->dL[r]); W->rhasU[r] = isfinite(W->dU[r]);
        if (W->rhasL[r]) W->dL[r] -= K_BOUND_RELAX*fmax(1.0, fabs(W->dL[r]));
        if (W->rhasU[r]) W->dU[r] += K_BOUND_RELAX*fmax(1.0, fabs(W->dU[r]));
    }
}

/* ------------------------------------------------------------------------------------------
 * function evaluation at an iterate; order 0: values only, 2: with derivatives
 * ---------------------------------------------------------------------------------------- */
static void eval_interval(const Ws *W, const StageIt *it, int i, StageEv *e, int order)
{
    const Prob *P = &W->P;
    const double *x = it[i].x, *x1 = it[i + 1].x;
    double b = x[VB], f = x[VF], p = P->withPn ? x[VP] : 0.0, s = x[VS];
    jet tau, bp;
    interval_map(P, b, f + p, W->G[i], P->ds[i], &tau, &bp);
    e->tau = tau.v; e->bplus = bp.v;
    e->c[0] = x1[VT] - (x[VT] + tau.v);
    e->c[1] = x1[VB] - bp.v;
    double sb = sqrt(b), sb1 = sqrt(x1[VB]);
    e->d[RPW0] = W->rs[RPW0]*f*sb;
    e->d[RPW1] = W->rs[RPW1]*f*sb1;
    e->d[RACC] = W->rs[RACC]*(f + p - (P->sr0 + P->sr1*sb + P->sr2*b) - W->G[i]);
    double lr[2][6];
    jet3 X = j3_const(0);
    jet4 E4[2];
    const int tabInt = P->lossKind == 2 && P->intLosses && P->energyOpt;      /* loss table integrated over the running time (loss_energy) */
    if (tabInt) {
        loss_energy(P, sb, x1[VT] - x[VT], f + p, f, W->G[i], E4, order ? 2 : 0);      /* ocp.py:233 */
        e->d[RLTR] = W->rs[RLTR]*(s - E4[0].v);
        e->d[RLRG] = W->rs[RLRG]*(s - E4[1].v);
    } else if (P->lossKind == 2) {
        loss_rows(&P->dyn, f, 0.5*(sb + sb1), lr);
        e->d[RLTR] = W->rs[RLTR]*(s - lr[0][0]);
        e->d[RLRG] = W->rs[RLRG]*(s - lr[1][0]);
    } else if (P->intLosses && P->energyOpt) {
        X = loss_distance(P, sb, x1[VT] - x[VT], f + p, W->G[i]);                /* ocp.py:233 */
        e->d[RLTR] = W->rs[RLTR]*(s - P->ct*f*X.v);
        e->d[RLRG] = W->rs[RLRG]*(s + P->cr*f*X.v);
    } else {
        e->d[RLTR] = W->rs[RLTR]*(s - P->ct*f);
        e->d[RLRG] = W->rs[RLRG]*(s + P->cr*f);
    }
    if (order == 0) return;

    e->tg[0] = tau.g0; e->tg[1] = tau.g1; e->th[0] = tau.h00; e->th[1] = tau.h01; e->th[2] = tau.h11;
    e->bg[0] = bp.g0; e->bg[1] = bp.g1; e->bh[0] = bp.h00; e->bh[1] = bp.h01; e->bh[2] = bp.h11;

    memset(e->gr, 0, sizeof e->gr); memset(e->hr, 0, sizeof e->hr);
    memset(e->objg, 0, sizeof e->objg); memset(e->objh, 0, sizeof e->objh);

    /* Fel*sqrt(b_i), Fel*sqrt(b_{i+1}) (ocp.py:189) */
    e->gr[RPW0][LF] = sb; e->gr[RPW0][LB] = 0.5*f/sb;
    e->hr[RPW0][LF][LB] = e->hr[RPW0][LB][LF] = 0.5/sb; e->hr[RPW0][LB][LB] = -0.25*f/(b*sb);
    e->gr[RPW1][LF] = sb1; e->gr[RPW1][LB1] = 0.5*f/sb1;
    e->hr[RPW1][LF][LB1] = e->hr[RPW1][LB1][LF] = 0.5/sb1; e->hr[RPW1][LB1][LB1] = -0.25*f/(x1[VB]*sb1);
    /* acceleration at the interval start (ocp.py:199, train.py:251-254) */
    e->gr[RACC][LF] = 1; e->gr[RACC][LP] = P->withPn ? 1 : 0; e->gr[RACC][LB] = -(0.5*P->sr1/sb + P->sr2);
    e->hr[RACC][LB][LB] = 0.25*P->sr1/(b*sb);
    /* static loss rows (ocp.py:225-226 with train.py:203 / utils.py:197-220) */
    if (P->lossKind == 2 && !tabInt) {
        /* rows s - g(f, vbar(b, b1)), vbar = (sqrt(b) + sqrt(b1))/2 */
        const double vb = 0.25/sb, vb1 = 0.25/sb1, vbb = -0.125/(b*sb), vb1b1 = -0.125/(x1[VB]*sb1);
        for (int k = 0; k < 2; k++) {
            const int r = k == 0 ? RLTR : RLRG;
            const double gf = lr[k][1], gv = lr[k][2], gff = lr[k][3], gfv = lr[k][4], gvv = lr[k][5];
            e->gr[r][LS] = 1; e->gr[r][LF] = -gf; e->gr[r][LB] = -gv*vb; e->gr[r][LB1] = -gv*vb1;
            e->hr[r][LF][LF] = -gff;
            e->hr[r][LF][LB] = e->hr[r][LB][LF] = -gfv*vb;
            e->hr[r][LF][LB1] = e->hr[r][LB1][LF] = -gfv*vb1;
            e->hr[r][LB][LB] = -(gvv*vb*vb + gv*vbb);
            e->hr[r][LB1][LB1] = -(gvv*vb1*vb1 + gv*vb1b1);
            e->hr[r][LB][LB1] = e->hr[r][LB1][LB] = -gvv*vb*vb1;
        }
    } else if (P->intLosses && P->energyOpt) {
        /* rows s + kappa_k phi_k(b, f, p, dt), dt = t1 - t.  Constant efficiencies: phi = f X(v(b), dt, f + p) for both rows, kappa = -ct, +cr.
         * Loss table: phi_k = E_k(v(b), dt, f + p, f), kappa = -1.  Chain rule onto the local variables (b, f, p | t, t1) */
        const double vb = 0.5/sb, vbb = -0.25/(b*sb);
        enum { QB = 0, QF = 1, QP = 2, QD = 3 };      /* b, f, p, dt */
        double ph1[2][4], ph2[2][4][4], kapv[2];
        if (tabInt) {
            for (int k = 0; k < 2; k++) {
                const jet4 *E = &E4[k];
                kapv[k] = -1.0;
                ph1[k][QB] = E->g[0]*vb; ph1[k][QF] = E->g[2] + E->g[3]; ph1[k][QP] = E->g[2]; ph1[k][QD] = E->g[1];
                ph2[k][QB][QB] = E->h[j4h(0, 0)]*vb*vb + E->g[0]*vbb;
                ph2[k][QB][QF] = (E->h[j4h(0, 2)] + E->h[j4h(0, 3)])*vb; ph2[k][QB][QP] = E->h[j4h(0, 2)]*vb; ph2[k][QB][QD] = E->h[j4h(0, 1)]*vb;
                ph2[k][QF][QF] = E->h[j4h(2, 2)] + 2*E->h[j4h(2, 3)] + E->h[j4h(3, 3)];
                ph2[k][QF][QP] = E->h[j4h(2, 2)] + E->h[j4h(2, 3)]; ph2[k][QP][QP] = E->h[j4h(2, 2)];
                ph2[k][QF][QD] = E->h[j4h(1, 2)] + E->h[j4h(1, 3)]; ph2[k][QP][QD] = E->h[j4h(1, 2)];
                ph2[k][QD][QD] = E->h[j4h(1, 1)];
                for (int a = 0; a < 4; a++) for (int c = 0; c < a; c++) ph2[k][a][c] = ph2[k][c][a];
            }
        } else {
            double X1[4], X2[4][4];
            X1[QB] = X.g[0]*vb; X1[QF] = X.g[2]; X1[QP] = X.g[2]; X1[QD] = X.g[1];
            X2[QB][QB] = X.h[0]*vb*vb + X.g[0]*vbb;
            X2[QB][QF] = X2[QB][QP] = X.h[2]*vb; X2[QB][QD] = X.h[1]*vb;
            X2[QF][QF] = X2[QF][QP] = X2[QP][QP] = X.h[5];
            X2[QF][QD] = X2[QP][QD] = X.h[4];
            X2[QD][QD] = X.h[3];
            for (int a = 0; a < 4; a++) for (int c = 0; c < a; c++) X2[a][c] = X2[c][a];
            kapv[0] = -P->ct; kapv[1] = P->cr;
            for (int k = 0; k < 2; k++)
                for (int a = 0; a < 4; a++) {                         /* phi = f X */
                    ph1[k][a] = f*X1[a] + (a == QF ? X.v : 0.0);
                    for (int c = 0; c < 4; c++) ph2[k][a][c] = f*X2[a][c] + (a == QF ? X1[c] : 0.0) + (c == QF ? X1[a] : 0.0);
                }
        }
        /* local columns of (b, f, p, dt): dt = t1 - t enters with +1 at LT1 and -1 at LT */
        const int col[4] = {LB, LF, LP, -1};
        for (int k = 0; k < 2; k++) {
            const int r = k == 0 ? RLTR : RLRG;
            const double kap = kapv[k];
            e->gr[r][LS] = 1;
            for (int a = 0; a < 4; a++) {
                if (a == QP && !P->withPn) continue;
                if (col[a] >= 0) e->gr[r][col[a]] += kap*ph1[k][a];
                else { e->gr[r][LT1] += kap*ph1[k][a]; e->gr[r][LT] -= kap*ph1[k][a]; }
                for (int c = 0; c < 4; c++) {
                    if (c == QP && !P->withPn) continue;
                    const double hv = kap*ph2[k][a][c];
                    const int na = col[a] >= 0 ? 1 : 2, nc = col[c] >= 0 ? 1 : 2;
                    const int ia[2] = {col[a] >= 0 ? col[a] : LT1, LT}, ic[2] = {col[c] >= 0 ? col[c] : LT1, LT};
                    const double sa[2] = {1, -1};
                    for (int m = 0; m < na; m++) for (int q = 0; q < nc; q++) e->hr[r][ia[m]][ic[q]] += (na == 2 ? sa[m] : 1)*(nc == 2 ? sa[q] : 1)*hv;
                }
            }
        }
    } else {
        e->gr[RLTR][LS] = 1; e->gr[RLTR][LF] = -P->ct;
        e->gr[RLRG][LS] = 1; e->gr[RLRG][LF] = P->cr;
    }
    for (int r = 0; r < NR; r++)
        if (W->rs[r] != 1.0)
            for (int a = 0; a < NL; a++) { e->gr[r][a] *= W->rs[r]; for (int c = 0; c < NL; c++) e->hr[r][a][c] *= W->rs[r]; }

    /* objective of the interval (ocp.py:146-150, 223, 245, 276-284) */
    double sc = W->sf/P->objDen;
    if (P->energyOpt) {
        e->objg[LF] = sc*P->ds[i]; e->objg[LS] = P->intLosses ? sc : sc*P->ds[i];      /* ocp.py:223 resp. :235 */
        if (i > 0) {
            double q = it[i - 1].x[VF];
            e->objg[LF] += sc*2e-3*(f - q); e->objg[LQ] = -sc*2e-3*(f - q);
            e->objh[LF][LF] = sc*2e-3; e->objh[LQ][LQ] = sc*2e-3; e->objh[LF][LQ] = e->objh[LQ][LF] = -sc*2e-3;
        }
    } else {
        e->objg[LF] = sc*2e-4*f; e->objh[LF][LF] = sc*2e-4;
        if (P->withPn) { e->objg[LP] = sc*2e-4*p; e->objh[LP][LP] = sc*2e-4; }
        if (i == W->N - 1) e->objg[LT1] = sc;
    }
}

static double objective_value(const Ws *W, const StageIt *it)
{
    const Prob *P = &W->P; double J = 0;
    for (int i = 0; i < W->N; i++) {
        double f = it[i].x[VF], p = P->withPn ? it[i].x[VP] : 0.0, s = it[i].x[VS];
        if (P->energyOpt) {
            J += P->intLosses ? P->ds[i]*f + s : P->ds[i]*(f + s);
            if (i > 0) { double q = it[i - 1].x[VF]; J += 1e-3*(f - q)*(f - q); }
        } else J += 1e-4*(f*f + p*p);
    }
    if (!P->energyOpt) J += it[W->N].x[VT];
    return W->sf*J/P->objDen;
}

/* barrier function value and constraint violation (1-norm of the scaled rows) at an iterate */
static void merit_terms(const Ws *W, const StageIt *it, double mu, double *theta, double *phi, int *ok)
{
    int N = W->N; double th = 0, bar = 0; *ok = 1;
    StageEv e;
    for (int i = 0; i < N; i++) {
        eval_interval(W, it, i, &e, 0);
        th += W->sct[i]*fabs(e.c[0]) + W->scb[i]*fabs(e.c[1]);
        for (int r = 0; r < NR; r++) {
            if (!W->rowOn[r]) continue;
            double sg = it[i].sig[r];
            th += fabs(e.d[r] - sg);
            if (W->rhasL[r]) { double s = sg - W->dL[r]; if (s <= 0) *ok = 0; else bar -= mu*log(s); }
            if (W->rhasU[r]) { double s = W->dU[r] - sg; if (s <= 0) *ok = 0; else bar -= mu*log(s); }
            if (W->rhasL[r] && !W->rhasU[r]) bar += K_D*mu*(sg - W->dL[r]);
            if (!W->rhasL[r] && W->rhasU[r]) bar += K_D*mu*(W->dU[r] - sg);
        }
    }
    for (int i = 0; i <= N; i++) {
        const StageBd *B = &W->bd[i];
        for (int k = 0; k < NV; k++) {
            if (!B->on[k]) continue;
            double x = it[i].x[k];
            if (B->hasL[k]) { double s = x - B->lb[k]; if (s <= 0) *ok = 0; else bar -= mu*log(s); }
            if (B->hasU[k]) { double s = B->ub[k] - x; if (s <= 0) *ok = 0; else bar -= mu*log(s); }
            if (B->hasL[k] && !B->hasU[k]) bar += K_D*mu*(x - B->lb[k]);
            if (!B->hasL[k] && B->hasU[k]) bar += K_D*mu*(B->ub[k] - x);
        }
    }
    if (!isfinite(th) || !isfinite(bar)) *ok = 0;
    *theta = th; *phi = objective_value(W, it) + bar;
}

/* ------------------------------------------------------------------------------------------
 * optimality error E_mu (W&B eq. (5)) of the scaled problem + unscaled infeasibilities
 * ---------------------------------------------------------------------------------------- */
typedef struct { double dual, primal, compl_, sd, sc, E, dual_u, primal_u, compl_u; } Err;

static void kkt_error(Ws *W, double mu, Err *R)
{
    int N = W->N; const Prob *P = &W->P;
    double dual = 0, prim = 0, comp = 0, prim_u = 0, comp0 = 0;
    double sumlam = 0, sumz = 0; int nlam = 0, nz = 0;
    /* gradient of the Lagrangian, accumulated per stage variable */
    double (*gl)[NV] = calloc(N + 1, sizeof *gl);
    for (int i = 0; i < N; i++) {
        StageEv *e = &W->ev[i]; StageIt *I = &W->it[i];
        double loc[NL];
        for (int a = 0; a < NL; a++) loc[a] = e->objg[a];
        for (int r = 0; r < NR; r++) if (W->rowOn[r]) for (int a = 0; a < NL; a++) loc[a] += I->nu[r]*e->gr[r][a];
        /* dynamics: c_t = t1 - t - tau, c_b = b1 - bplus */
        loc[LT1] += I->lam[0]; loc[LT] -= I->lam[0];
        loc[LB] -= I->lam[0]*e->tg[0]; loc[LF] -= I->lam[0]*e->tg[1]; loc[LP] -= P->withPn ? I->lam[0]*e->tg[1] : 0;
        loc[LB1] += I->lam[1];
        loc[LB] -= I->lam[1]*e->bg[0]; loc[LF] -= I->lam[1]*e->bg[1]; loc[LP] -= P->withPn ? I->lam[1]*e->bg[1] : 0;
        gl[i][VT] += loc[LT]; gl[i][VB] += loc[LB]; gl[i][VF] += loc[LF]; gl[i][VP] += loc[LP]; gl[i][VS] += loc[LS];
        if (i > 0) gl[i - 1][VF] += loc[LQ];
        gl[i + 1][VT] += loc[LT1]; gl[i + 1][VB] += loc[LB1];
        prim = fmax(prim, fmax(W->sct[i]*fabs(e->c[0]), W->scb[i]*fabs(e->c[1])));
        prim_u = fmax(prim_u, fmax(fabs(e->c[0]), fabs(e->c[1])));
        sumlam += fabs(I->lam[0])/W->sct[i] + fabs(I->lam[1])/W->scb[i]; nlam += 2;
        for (int r = 0; r < NR; r++) {
            if (!W->rowOn[r]) continue;
            double sg = I->sig[r];
            prim = fmax(prim, fabs(e->d[r] - sg));
            prim_u = fmax(prim_u, fabs(e->d[r] - sg)/W->rs[r]);
            /* violation of the original row bounds by d(x) is covered by d - sigma and the slack bounds */
            sumlam += fabs(I->nu[r]); nlam++;
            /* stationarity wrt the slack: -nu - zL + zU */
            double gs = -I->nu[r];
            if (W->rhasL[r]) { gs -= I->zLs[r]; double c = (sg - W->dL[r])*I->zLs[r]; comp = fmax(comp, fabs(c - mu)); comp0 = fmax(comp0, fabs(c)); sumz += I->zLs[r]; nz++; }
            if (W->rhasU[r]) { gs += I->zUs[r]; double c = (W->dU[r] - sg)*I->zUs[r]; comp = fmax(comp, fabs(c - mu)); comp0 = fmax(comp0, fabs(c)); sumz += I->zUs[r]; nz++; }
            dual = fmax(dual, fabs(gs));
        }
    }
    for (int i = 0; i <= N; i++) {
        StageBd *B = &W->bd[i]; StageIt *I = &W->it[i];
        for (int k = 0; k < NV; k++) {
            if (!B->on[k]) continue;
            double g = gl[i][k];
            if (B->hasL[k]) { g -= I->zL[k]; double c = (I->x[k] - B->lb[k])*I->zL[k]; comp = fmax(comp, fabs(c - mu)); comp0 = fmax(comp0, fabs(c)); sumz += I->zL[k]; nz++; }
            if (B->hasU[k]) { g += I->zU[k]; double c = (B->ub[k] - I->x[k])*I->zU[k]; comp = fmax(comp, fabs(c - mu)); comp0 = fmax(comp0, fabs(c)); sumz += I->zU[k]; nz++; }
            dual = fmax(dual, fabs(g));
        }
    }
    free(gl);
    R->sd = fmax(K_SMAX, (sumlam + sumz)/fmax(1, nlam + nz))/K_SMAX;
    R->sc = fmax(K_SMAX, sumz/fmax(1, nz))/K_SMAX;
    R->dual = dual; R->primal = prim; R->compl_ = comp;
    R->E = fmax(dual/R->sd, fmax(prim, comp/R->sc));
    R->dual_u = dual/W->sf; R->primal_u = prim_u; R->compl_u = comp0/W->sf;
}

/* ------------------------------------------------------------------------------------------
 * small dense helpers
 * ---------------------------------------------------------------------------------------- */
static int chol3p(double A[3][3], int n, double L[3][3], double piv1);
static int chol3(double A[3][3], int n, double L[3][3]) { return chol3p(A, n, L, NAN); }
/* piv1: when finite, the second pivot A11 - A01^2/A00 computed by the caller without the difference of large numbers (compute_direction) */
static int chol3p(double A[3][3], int n, double L[3][3], double piv1)
{
    memset(L, 0, 9*sizeof(double));
    for (int j = 0; j < n; j++) {
        double d = A[j][j];
        for (int k = 0; k < j; k++) d -= L[j][k]*L[j][k];
        if (j == 1 && isfinite(piv1)) d = piv1;
        if (!(d > 0) || !isfinite(d)) return 0;
        L[j][j] = sqrt(d);
        for (int i = j + 1; i < n; i++) {
            double s = A[i][j];
            for (int k = 0; k < j; k++) s -= L[i][k]*L[j][k];
            L[i][j] = s/L[j][j];
        }
    }
    return 1;
}
static void chol3_solve(double L[3][3], int n, const double *b, double *x)
{
    double y[3];
    for (int i = 0; i < n; i++) { double s = b[i]; for (int k = 0; k < i; k++) s -= L[i][k]*y[k]; y[i] = s/L[i][i]; }
    for (int i = n - 1; i >= 0; i--) { double s = y[i]; for (int k = i + 1; k < n; k++) s -= L[k][i]*x[k]; x[i] = s/L[i][i]; }
}

/* Sigma and barrier gradient of one bounded scalar */
static void bar_terms(double x, double lb, double ub, int hasL, int hasU, double zL, double zU, double mu, double *Sigma, double *gphi)
{
    double S = 0, g = 0;
    if (hasL) { S += zL/(x - lb); g -= mu/(x - lb); }
    if (hasU) { S += zU/(ub - x); g += mu/(ub - x); }
    if (hasL && !hasU) g += K_D*mu;
    if (!hasL && hasU) g -= K_D*mu;
    *Sigma = S; *gphi = g;
}

/* ------------------------------------------------------------------------------------------
 * primal-dual direction: condensed stage systems + Riccati recursion.
 *   res_c[i][2], res_d[i][NR]: the constraint values used as right-hand side (c and d - sigma;
 *   replaced by the accumulated values in a second-order correction).
 * returns 1 if the inertia is correct (all stage pivots positive), 0 otherwise.
 * ---------------------------------------------------------------------------------------- */
static double rs_D(const StageRs *v, int j) { return v->n[j]/v->zn[j] + v->p[j]/v->zp[j]; }

/*
 * With W->resto set this is the Newton system of the restoration problem with (n, p, z_n, z_p) eliminated: every relaxed row reads
 *   (row gradient) dx + rhat = D y+,   D = n/z_n + p/z_p,   rhat = row + n - p + (mu - rho n)/z_n - (mu - rho p)/z_p
 * (res_c / res_d then carry rhat; the dynamics rows scaled).  An inequality row condenses like before with Sigma replaced by
 * 1/(D + 1/Sigma); a dynamics row becomes x+ = F y + r + D lam+: the value function of stage i+1 is used through
 * M = (P2 + D^-1)^-1 on the relaxed rows.  The objective is the proximity term eta/2 |D_R (x - x_R)|^2; b_N stays a parameter,
 * the last interval's b row is a penalty on F_b y + r_b instead of an elimination.
 */
static int compute_direction(Ws *W, double mu, double dw, const double (*res_c)[2], const double (*res_d)[NR], StageDir *D)
{
    int N = W->N; const Prob *P = &W->P;
    const Resto *R = W->resto;

    memset(W->HN, 0, sizeof W->HN); memset(W->hN, 0, sizeof W->hN);
    for (int i = 0; i < N; i++) { memset(W->kk[i].H, 0, sizeof W->kk[i].H); memset(W->kk[i].h, 0, sizeof W->kk[i].h); memset(W->kk[i].E, 0, sizeof W->kk[i].E); W->kk[i].oa = W->kk[i].ob = 0; }

    /* --- assemble ------------------------------------------------------------------- */
    for (int i = 0; i < N; i++) {
        StageEv *e = &W->ev[i]; StageIt *I = &W->it[i]; StageKkt *Kk = &W->kk[i];
        double Hl[NL][NL], hl[NL];
        for (int a = 0; a < NL; a++) { hl[a] = R ? 0.0 : e->objg[a]; for (int c = 0; c < NL; c++) Hl[a][c] = R ? 0.0 : e->objh[a][c]; }
        /* The two forces' own curvatures, H_ff - H_fp and H_pp - H_fp, term by term (round 5; msd_kernel.hpp: S_OA, S_OB).  Where both brakes are free
         * and the acceleration row is active its barrier term Sigma g g^T (1e12 at convergence) sits in H_ff, H_fp and H_pp alike, and the curvature
         * of the split between the two forces (1e-4 in a time-optimal problem) is lost in the second pivot of the control block, G_pp - G_fp^2/G_ff:
         * rounding noise of either sign decided the inertia there */
        double oa = R ? 0.0 : e->objh[LF][LF] - e->objh[LF][LP], ob = R ? 0.0 : e->objh[LP][LP] - e->objh[LF][LP];
        /* -lam_t * hess(tau) - lam_b * hess(bplus) on (b, w) with w = f + p */
        {
            double hbb = -(I->lam[0]*e->th[0] + I->lam[1]*e->bh[0]);
            double hbw = -(I->lam[0]*e->th[1] + I->lam[1]*e->bh[1]);
            double hww = -(I->lam[0]*e->th[2] + I->lam[1]*e->bh[2]);
            Hl[LB][LB] += hbb; Hl[LB][LF] += hbw; Hl[LF][LB] += hbw; Hl[LF][LF] += hww;
            if (P->withPn) { Hl[LB][LP] += hbw; Hl[LP][LB] += hbw; Hl[LF][LP] += hww; Hl[LP][LF] += hww; Hl[LP][LP] += hww; }
            else oa += hww;
        }
        for (int r = 0; r < NR; r++) {
            if (!W->rowOn[r]) continue;
            double Sg, gphi;
            bar_terms(I->sig[r], W->dL[r], W->dU[r], W->rhasL[r], W->rhasU[r], I->zLs[r], I->zUs[r], mu, &Sg, &gphi);
            Sg += dw;
            double coef = Sg*res_d[i][r] + gphi;
            if (R) { const double St = 1.0/(rs_D(&R->v[i], 2 + r) + 1.0/Sg); coef = St*res_d[i][r] + gphi*St/Sg; Sg = St; }
            for (int a = 0; a < NL; a++) {
                hl[a] += e->gr[r][a]*coef;
                for (int c = 0; c < NL; c++) Hl[a][c] += I->nu[r]*e->hr[r][a][c] + Sg*e->gr[r][a]*e->gr[r][c];
            }
            oa += I->nu[r]*(e->hr[r][LF][LF] - e->hr[r][LF][LP]) + Sg*e->gr[r][LF]*(e->gr[r][LF] - e->gr[r][LP]);
            ob += I->nu[r]*(e->hr[r][LP][LP] - e->hr[r][LF][LP]) + Sg*e->gr[r][LP]*(e->gr[r][LP] - e->gr[r][LF]);
        }
        Kk->oa += oa; Kk->ob += ob;
        for (int a = 0; a < 6; a++) { Kk->h[a] += hl[a]; for (int c = 0; c < 6; c++) Kk->H[a][c] += Hl[a][c]; }
        for (int a = 0; a < 6; a++) { Kk->E[a][0] += Hl[a][LT1]; Kk->E[a][1] += Hl[a][LB1]; }
        if (i + 1 < N) {
            StageKkt *Kn = &W->kk[i + 1];
            Kn->h[0] += hl[LT1]; Kn->h[1] += hl[LB1];
            Kn->H[0][0] += Hl[LT1][LT1]; Kn->H[0][1] += Hl[LT1][LB1]; Kn->H[1][0] += Hl[LB1][LT1]; Kn->H[1][1] += Hl[LB1][LB1];
        } else {
            W->hN[0] += hl[LT1]; W->hN[1] += hl[LB1];
            W->HN[0][0] += Hl[LT1][LT1]; W->HN[0][1] += Hl[LT1][LB1]; W->HN[1][0] += Hl[LB1][LT1]; W->HN[1][1] += Hl[LB1][LB1];
        }
        /* dynamics linearisation */
        memset(Kk->F, 0, sizeof Kk->F);
        Kk->F[0][0] = 1; Kk->F[0][1] = e->tg[0]; Kk->F[0][3] = e->tg[1]; Kk->F[0][4] = P->withPn ? e->tg[1] : 0;
        Kk->F[1][1] = e->bg[0]; Kk->F[1][3] = e->bg[1]; Kk->F[1][4] = P->withPn ? e->bg[1] : 0;
        Kk->F[2][3] = 1;
        Kk->r[0] = -res_c[i][0]; Kk->r[1] = -res_c[i][1]; Kk->r[2] = 0;
        if (R) { Kk->r[0] /= W->sct[i]; Kk->r[1] /= W->scb[i]; }
    }
    /* variable bounds + regularisation */
    for (int i = 0; i <= N; i++) {
        StageBd *B = &W->bd[i]; StageIt *I = &W->it[i];
        for (int k = 0; k < NV; k++) {
            if (!B->on[k]) continue;
            double Sg, gphi;
            bar_terms(I->x[k], B->lb[k], B->ub[k], B->hasL[k], B->hasU[k], I->zL[k], I->zU[k], mu, &Sg, &gphi);
            int a = var2loc[k];
            if (R) { const double w = R->eta*R->dr[i][k]*R->dr[i][k]; Sg += w; gphi += w*(I->x[k] - R->xR[i][k]); }
            if (i < N) { W->kk[i].H[a][a] += Sg + dw; W->kk[i].h[a] += gphi; if (k == VF) W->kk[i].oa += Sg + dw; if (k == VP) W->kk[i].ob += Sg + dw; }
            else { W->HN[a][a] += Sg + dw; W->hN[a] += gphi; }
        }
    }

    /* --- backward recursion ------------------------------------------------------------ */
    double Pm[3][3], pv[3];
    memcpy(Pm, W->HN, sizeof Pm); memcpy(pv, W->hN, sizeof pv);
    /* b_N is a parameter: its row/column never enters */
    for (int a = 0; a < 3; a++) { Pm[1][a] = Pm[a][1] = 0; } pv[1] = 0;

    for (int i = N - 1; i >= 0; i--) {
        StageKkt *Kk = &W->kk[i];
        memcpy(Kk->Pn, Pm, sizeof Pm); memcpy(Kk->pn, pv, sizeof pv);
        if (i == N - 1) for (int a = 0; a < 6; a++) Kk->E[a][1] = 0;   /* db_N = 0: no coupling with the parameter b_N */
        double G[6][6], g[6], PF[3][6], Pr[3];
        for (int a = 0; a < 3; a++) {
            for (int c = 0; c < 6; c++) { double s = 0; for (int m = 0; m < 3; m++) s += Pm[a][m]*Kk->F[m][c]; PF[a][c] = s; }
            double s = pv[a]; for (int m = 0; m < 3; m++) s += Pm[a][m]*Kk->r[m]; Pr[a] = s;
        }
        for (int a = 0; a < 6; a++) {
            for (int c = 0; c < 6; c++) {
                double s = Kk->H[a][c];
                for (int m = 0; m < 3; m++) s += Kk->F[m][a]*PF[m][c] + Kk->E[a][m]*Kk->F[m][c] + Kk->F[m][a]*Kk->E[c][m];
                G[a][c] = s;
            }
            double s = Kk->h[a];
            for (int m = 0; m < 3; m++) s += Kk->F[m][a]*Pr[m] + Kk->E[a][m]*Kk->r[m];
            g[a] = s;
        }
        if (R) {
            const double Dt = rs_D(&R->v[i], 0)/(W->sct[i]*W->sct[i]), Db = rs_D(&R->v[i], 1)/(W->scb[i]*W->scb[i]);
            Kk->Dt = Dt; Kk->Db = Db; memset(Kk->M, 0, sizeof Kk->M);
            if (i == N - 1) Kk->M[0][0] = Dt/(1 + Pm[0][0]*Dt);
            else {
                /* M = (P2 + D^-1)^-1 = D^1/2 (I + D^1/2 P2 D^1/2)^-1 D^1/2 */
                const double st = sqrt(Dt), sb = sqrt(Db);
                const double ma = 1 + st*Pm[0][0]*st, mb = st*Pm[0][1]*sb, mc = 1 + sb*Pm[1][1]*sb, det = ma*mc - mb*mb;
                if (!(det > 0) || !(ma > 0)) return 0;      /* P2 + D^-1 not positive definite: wrong inertia */
                Kk->M[0][0] = st*(mc/det)*st; Kk->M[0][1] = Kk->M[1][0] = -st*(mb/det)*sb; Kk->M[1][1] = sb*(ma/det)*sb;
            }
            if (!(1 + Pm[0][0]*Dt > 0)) return 0;
            double Q[2][6];
            for (int m = 0; m < 2; m++) for (int c = 0; c < 6; c++) Q[m][c] = PF[m][c] + Kk->E[c][m];
            for (int a = 0; a < 6; a++) {
                double qa[2] = {Kk->M[0][0]*Q[0][a] + Kk->M[1][0]*Q[1][a], Kk->M[0][1]*Q[0][a] + Kk->M[1][1]*Q[1][a]};
                for (int c = 0; c < 6; c++) G[a][c] -= qa[0]*Q[0][c] + qa[1]*Q[1][c];
                g[a] -= qa[0]*Pr[0] + qa[1]*Pr[1];
            }
        }
        if (!P->withPn) { for (int a = 0; a < 6; a++) G[4][a] = G[a][4] = 0; G[4][4] = 1; g[4] = 0; }
        memcpy(Kk->Ghat, G, sizeof G); memcpy(Kk->ghat, g, sizeof g);

        if (i == N - 1) {
            /* b_N fixed: the b-row of the dynamics is an equality in (x, u), Bb db + Bw (df + dp) + rb = 0: one of the two forces is eliminated
             * through it -- the one with the smaller curvature.  (Round 5.  Rounds 1-4 always eliminated df: with Fel on a bound its barrier
             * curvature Sigma_f ~ 1e11 then sits in every reduced entry and the value function of stage N-1 comes out as a difference of two such
             * numbers -- P_bb = eb^2 (G_ff - G_ff^2/(G_ff + G_pp - 2 G_fp)) -- whose rounding error, times Bw^2, decided the inertia of
             * stage N-2 on degenerate problems: random sweep seed 176, profiles/r04.  Eliminating the softer force leaves the stiff one as an
             * ordinary pivot: at most one bit is lost.)  Restoration: the row is relaxed, F_b y + r_b = -D_b lam_b+ with the penalty
             * D_b lam_b+^2 / 2; in terms of v = sqrt(D_b) lam_b+ (slot of the eliminated force, unit curvature) the system stays well
             * conditioned as D_b -> 0 */
            double Bw = Kk->F[1][3];
            const int je = (P->withPn && G[4][4] < G[3][3]) ? 4 : 3, jk = 7 - je;
            Kk->je = je;
            double T[6][6]; memset(T, 0, sizeof T);
            for (int a = 0; a < 6; a++) T[a][a] = 1;
            T[je][je] = R ? -sqrt(Kk->Db)/Bw : 0;
            T[je][1] = -Kk->F[1][1]/Bw; T[je][jk] = -Kk->F[1][jk]/Bw;
            double y0f = -Kk->r[1]/Bw;
            memset(Kk->ef, 0, sizeof Kk->ef); Kk->ef[1] = T[je][1]; Kk->ef[jk] = T[je][jk]; Kk->e0 = y0f;
            /* G~ = T^T G T, g~ = T^T (G y0 + g); column/row je of T is zero -> that force decoupled, make it an identity pivot */
            double GT[6][6], G2[6][6], gy[6], g2[6];
            for (int a = 0; a < 6; a++) { for (int c = 0; c < 6; c++) { double s = 0; for (int m = 0; m < 6; m++) s += G[a][m]*T[m][c]; GT[a][c] = s; } gy[a] = g[a] + G[a][je]*y0f; }
            for (int a = 0; a < 6; a++) { for (int c = 0; c < 6; c++) { double s = 0; for (int m = 0; m < 6; m++) s += T[m][a]*GT[m][c]; G2[a][c] = s; } double s = 0; for (int m = 0; m < 6; m++) s += T[m][a]*gy[m]; g2[a] = s; }
            if (R) G2[je][je] += 1;
            else { for (int a = 0; a < 6; a++) G2[je][a] = G2[a][je] = 0; G2[je][je] = 1; g2[je] = 0; }
            /* pivot of the force that is kept: G_kk - 2 G_ek + G_ee = the sum of the two own curvatures (the value function of stage N has no q entries);
             * not with cross terms in t_N (integrated loss rows) */
            if (!R && P->withPn && !P->intLosses) G2[jk][jk] = Kk->oa + Kk->ob;
            memcpy(G, G2, sizeof G); memcpy(g, g2, sizeof g);
        }

        double Guu[3][3], L[3][3];
        for (int a = 0; a < 3; a++) for (int c = 0; c < 3; c++) Guu[a][c] = G[3 + a][3 + c];
        double piv_p = NAN;      /* second pivot of the control block from the own curvatures (regular stages of the original problem) */
        if (!R && P->withPn && i < N - 1) {
            double ex = 0;       /* cross terms: (E_f - E_p) F_w */
            for (int m = 0; m < 2; m++) ex += (Kk->E[3][m] - Kk->E[4][m])*Kk->F[m][3];
            const double alpha = Kk->oa + PF[2][3] + ex, beta = Kk->ob - PF[2][4] - ex;      /* G_ff - G_fp, G_pp - G_fp */
            piv_p = beta + (Guu[0][1]/Guu[0][0])*alpha;
        }
        if (!chol3p(Guu, 3, L, piv_p)) return 0;
        for (int c = 0; c < 3; c++) {
            double rhs[3], sol[3];
            for (int a = 0; a < 3; a++) rhs[a] = -G[3 + a][c];
            chol3_solve(L, 3, rhs, sol);
            for (int a = 0; a < 3; a++) Kk->K[a][c] = sol[a];
        }
        { double rhs[3]; for (int a = 0; a < 3; a++) rhs[a] = -g[3 + a]; chol3_solve(L, 3, rhs, Kk->k); }
        for (int a = 0; a < 3; a++) {
            for (int c = 0; c < 3; c++) { double s = G[a][c]; for (int m = 0; m < 3; m++) s += G[a][3 + m]*Kk->K[m][c]; Pm[a][c] = s; }
            double s = g[a]; for (int m = 0; m < 3; m++) s += G[a][3 + m]*Kk->k[m]; pv[a] = s;
        }
        for (int a = 0; a < 3; a++) for (int c = a + 1; c < 3; c++) { double m = 0.5*(Pm[a][c] + Pm[c][a]); Pm[a][c] = Pm[c][a] = m; }
    }

    /* --- forward rollout ----------------------------------------------------------------- */
    double dxs[3] = {0, 0, 0};   /* (dt, db, dq) of stage i; x_0 is a parameter */
    for (int i = 0; i < N; i++) {
        StageKkt *Kk = &W->kk[i]; StageDir *d = &D[i];
        double du[3], y[6];
        for (int a = 0; a < 3; a++) { double s = Kk->k[a]; for (int c = 0; c < 3; c++) s += Kk->K[a][c]*dxs[c]; du[a] = s; }
        for (int a = 0; a < 3; a++) { y[a] = dxs[a]; y[3 + a] = du[a]; }
        const int je = (i == N - 1) ? Kk->je : 3;
        const double v_last = y[je];
        if (i == N - 1) { double s = Kk->e0; for (int a = 0; a < 6; a++) if (a != je) s += Kk->ef[a]*y[a]; y[je] = s; if (R) y[je] -= sqrt(Kk->Db)/Kk->F[1][3]*v_last; }
        if (!P->withPn) y[4] = 0;
        double xn[3];
        for (int a = 0; a < 3; a++) { double s = Kk->r[a]; for (int c = 0; c < 6; c++) s += Kk->F[a][c]*y[c]; xn[a] = s; }
        if (i == N - 1) xn[1] = 0;
        double lam_r[2] = {0, 0};
        if (R) {
            /* x+ = a + D lam+, lam+ = -(I + P2 D)^-1 (P a + p + E^T y) on the relaxed rows (the last interval: t only, b_N is a
             * parameter): the multipliers from the relaxed rows themselves, so that the rows' linearisation holds to rounding */
            double gg[2];
            for (int m = 0; m < 2; m++) { double s = Kk->pn[m]; for (int c = 0; c < 3; c++) s += Kk->Pn[m][c]*xn[c]; for (int c = 0; c < 6; c++) s += Kk->E[c][m]*y[c]; gg[m] = s; }
            if (i == N - 1) { lam_r[0] = -gg[0]/(1 + Kk->Dt*Kk->Pn[0][0]); lam_r[1] = v_last/sqrt(Kk->Db); xn[0] += Kk->Dt*lam_r[0]; }
            else {
                const double a00 = 1 + Kk->Pn[0][0]*Kk->Dt, a01 = Kk->Pn[0][1]*Kk->Db, a10 = Kk->Pn[1][0]*Kk->Dt, a11 = 1 + Kk->Pn[1][1]*Kk->Db;
                const double det = a00*a11 - a01*a10;
                lam_r[0] = -(a11*gg[0] - a01*gg[1])/det; lam_r[1] = -(a00*gg[1] - a10*gg[0])/det;
                xn[0] += Kk->Dt*lam_r[0]; xn[1] += Kk->Db*lam_r[1];
            }
        }
        memset(d, 0, sizeof *d);
        d->dx[VT] = y[0]; d->dx[VB] = y[1]; d->dx[VF] = y[3]; d->dx[VP] = y[4]; d->dx[VS] = y[5];
        /* new dynamics multipliers: lam+ = -(P+ x+ + p+ + E^T y) */
        double lp[3];
        for (int a = 0; a < 3; a++) { double s = Kk->pn[a]; for (int c = 0; c < 3; c++) s += Kk->Pn[a][c]*xn[c]; for (int c = 0; c < 6; c++) s += Kk->E[c][a]*y[c]; lp[a] = -s; }
        if (i == N - 1 && !R) {
            /* multiplier of the eliminated row from stationarity wrt the eliminated force: (G y + g)_e - lam_b * Bw = 0 */
            double s = Kk->ghat[je]; for (int c = 0; c < 6; c++) s += Kk->Ghat[je][c]*y[c];
            lp[1] = s/Kk->F[1][3];
        }
        if (R) { lp[0] = lam_r[0]; lp[1] = lam_r[1]; }
        d->dlam[0] = lp[0] - W->it[i].lam[0]; d->dlam[1] = lp[1] - W->it[i].lam[1];
        dxs[0] = xn[0]; dxs[1] = xn[1]; dxs[2] = xn[2];
    }
    memset(&D[N], 0, sizeof D[N]);
    D[N].dx[VT] = dxs[0]; D[N].dx[VB] = 0;

    /* --- slacks, inequality multipliers, bound multipliers ----------------------------------- */
    for (int i = 0; i < N; i++) {
        StageEv *e = &W->ev[i]; StageIt *I = &W->it[i]; StageDir *d = &D[i];
        double dl[NL];
        dl[LT] = d->dx[VT]; dl[LB] = d->dx[VB]; dl[LQ] = i > 0 ? D[i - 1].dx[VF] : 0; dl[LF] = d->dx[VF]; dl[LP] = d->dx[VP]; dl[LS] = d->dx[VS];
        dl[LT1] = D[i + 1].dx[VT]; dl[LB1] = D[i + 1].dx[VB];
        for (int r = 0; r < NR; r++) {
            if (!W->rowOn[r]) continue;
            double Sg, gphi;
            bar_terms(I->sig[r], W->dL[r], W->dU[r], W->rhasL[r], W->rhasU[r], I->zLs[r], I->zUs[r], mu, &Sg, &gphi);
            double dsg = res_d[i][r];
            for (int a = 0; a < NL; a++) dsg += e->gr[r][a]*dl[a];
            d->dsig[r] = dsg;
            d->dnu[r] = (Sg + dw)*dsg + gphi - I->nu[r];
            if (R) {
                const double Sw = Sg + dw, St = 1.0/(rs_D(&R->v[i], 2 + r) + 1.0/Sw), nup = St*dsg + gphi*St/Sw;
                d->dsig[r] = (nup - gphi)/Sw; d->dnu[r] = nup - I->nu[r];
                dsg = d->dsig[r];
            }
            if (W->rhasL[r]) { double s = I->sig[r] - W->dL[r]; d->dzLs[r] = mu/s - I->zLs[r] - I->zLs[r]/s*dsg; }
            if (W->rhasU[r]) { double s = W->dU[r] - I->sig[r]; d->dzUs[r] = mu/s - I->zUs[r] + I->zUs[r]/s*dsg; }
        }
    }
    for (int i = 0; i <= N; i++) {
        StageBd *B = &W->bd[i]; StageIt *I = &W->it[i]; StageDir *d = &D[i];
        for (int k = 0; k < NV; k++) {
            if (!B->on[k]) { d->dx[k] = 0; continue; }
            if (B->hasL[k]) { double s = I->x[k] - B->lb[k]; d->dzL[k] = mu/s - I->zL[k] - I->zL[k]/s*d->dx[k]; }
            if (B->hasU[k]) { double s = B->ub[k] - I->x[k]; d->dzU[k] = mu/s - I->zU[k] + I->zU[k]/s*d->dx[k]; }
        }
    }
    if (R)
        for (int i = 0; i < N; i++) {
            const StageRs *v = &R->v[i]; StageRsDir *q = &R->d[i]; const StageIt *I = &W->it[i];
            memset(q, 0, sizeof *q);
            for (int j = 0; j < NC; j++) {
                if (j >= 2 && !W->rowOn[j - 2]) continue;
                const double sc = j == 0 ? W->sct[i] : j == 1 ? W->scb[i] : 1.0;
                const double y = (j < 2 ? I->lam[j] : I->nu[j - 2])/sc, dy = (j < 2 ? D[i].dlam[j] : D[i].dnu[j - 2])/sc;
                q->dn[j] = (mu - v->n[j]*(R->rho + y))/v->zn[j] - v->n[j]/v->zn[j]*dy;
                q->dp[j] = (mu - v->p[j]*(R->rho - y))/v->zp[j] + v->p[j]/v->zp[j]*dy;
                q->dzn[j] = R->rho + y + dy - v->zn[j];
                q->dzp[j] = R->rho - y - dy - v->zp[j];
            }
        }
    return 1;
}

/* maximum residual of the un-condensed Newton equations for a computed direction (debug aid) */
static double direction_residual(Ws *W, double mu, double dw, const double (*res_c)[2], const double (*res_d)[NR], StageDir *D)
{
    int N = W->N; const Prob *P = &W->P; double worst = 0;
    const Resto *R = W->resto;
    double (*rx)[NV] = calloc(N + 1, sizeof *rx);
    for (int i = 0; i < N; i++) {
        StageEv *e = &W->ev[i]; StageIt *I = &W->it[i];
        double dl[NL], loc[NL];
        dl[LT] = D[i].dx[VT]; dl[LB] = D[i].dx[VB]; dl[LQ] = i > 0 ? D[i - 1].dx[VF] : 0; dl[LF] = D[i].dx[VF]; dl[LP] = D[i].dx[VP]; dl[LS] = D[i].dx[VS];
        dl[LT1] = D[i + 1].dx[VT]; dl[LB1] = D[i + 1].dx[VB];
        double lt = I->lam[0] + D[i].dlam[0], lb = I->lam[1] + D[i].dlam[1];
        /* W d (with current multipliers) + grad f + J^T (new multipliers) */
        for (int a = 0; a < NL; a++) {
            double s = R ? 0.0 : e->objg[a];
            for (int c = 0; c < NL; c++) {
                double w = R ? 0.0 : e->objh[a][c];
                for (int r = 0; r < NR; r++) if (W->rowOn[r]) w += I->nu[r]*e->hr[r][a][c];
                s += w*dl[c];
            }
            for (int r = 0; r < NR; r++) if (W->rowOn[r]) s += (I->nu[r] + D[i].dnu[r])*e->gr[r][a];
            loc[a] = s;
        }
        double hbb = -(I->lam[0]*e->th[0] + I->lam[1]*e->bh[0]), hbw = -(I->lam[0]*e->th[1] + I->lam[1]*e->bh[1]), hww = -(I->lam[0]*e->th[2] + I->lam[1]*e->bh[2]);
        double dw_ = dl[LF] + (P->withPn ? dl[LP] : 0);
        loc[LB] += hbb*dl[LB] + hbw*dw_;
        loc[LF] += hbw*dl[LB] + hww*dw_;
        if (P->withPn) loc[LP] += hbw*dl[LB] + hww*dw_;
        loc[LT1] += lt; loc[LT] -= lt; loc[LB] -= lt*e->tg[0]; loc[LF] -= lt*e->tg[1]; if (P->withPn) loc[LP] -= lt*e->tg[1];
        loc[LB1] += lb; loc[LB] -= lb*e->bg[0]; loc[LF] -= lb*e->bg[1]; if (P->withPn) loc[LP] -= lb*e->bg[1];
        rx[i][VT] += loc[LT]; rx[i][VB] += loc[LB]; rx[i][VF] += loc[LF]; rx[i][VP] += loc[LP]; rx[i][VS] += loc[LS];
        if (i > 0) rx[i - 1][VF] += loc[LQ];
        rx[i + 1][VT] += loc[LT1]; rx[i + 1][VB] += loc[LB1];
        /* linearised dynamics */
        double rt = dl[LT1] - dl[LT] - e->tg[0]*dl[LB] - e->tg[1]*dw_ + res_c[i][0];
        double rb = dl[LB1] - e->bg[0]*dl[LB] - e->bg[1]*dw_ + res_c[i][1];
        if (R) {
            rt = dl[LT1] - dl[LT] - e->tg[0]*dl[LB] - e->tg[1]*dw_ + res_c[i][0]/W->sct[i] - rs_D(&R->v[i], 0)/(W->sct[i]*W->sct[i])*lt;
            rb = dl[LB1] - e->bg[0]*dl[LB] - e->bg[1]*dw_ + res_c[i][1]/W->scb[i] - rs_D(&R->v[i], 1)/(W->scb[i]*W->scb[i])*lb;
        }
        worst = fmax(worst, fmax(fabs(rt), fabs(rb)));
        if (getenv("ORACLE_DBGCAT")) fprintf(stderr, "   i=%d dyn res %g %g\n", i, rt, rb);
        for (int r = 0; r < NR; r++) {
            if (!W->rowOn[r]) continue;
            double lin = res_d[i][r] - D[i].dsig[r];
            for (int a = 0; a < NL; a++) lin += e->gr[r][a]*dl[a];
            if (R) lin -= rs_D(&R->v[i], 2 + r)*(I->nu[r] + D[i].dnu[r]);
            worst = fmax(worst, fabs(lin));
            if (getenv("ORACLE_DBGCAT")) fprintf(stderr, "   i=%d row %d lin res %g\n", i, r, lin);
            /* slack stationarity: dw*dsig - nu+ - zL+ + zU+ = 0 */
            double s = dw*D[i].dsig[r] - (I->nu[r] + D[i].dnu[r]);
            if (W->rhasL[r]) { s -= I->zLs[r] + D[i].dzLs[r]; double sl = I->sig[r] - W->dL[r]; worst = fmax(worst, fabs(I->zLs[r]*D[i].dsig[r] + sl*D[i].dzLs[r] - (mu - sl*I->zLs[r]))); }
            if (W->rhasU[r]) { s += I->zUs[r] + D[i].dzUs[r]; double su = W->dU[r] - I->sig[r]; worst = fmax(worst, fabs(-I->zUs[r]*D[i].dsig[r] + su*D[i].dzUs[r] - (mu - su*I->zUs[r]))); }
            if (W->rhasL[r] && !W->rhasU[r]) s += K_D*mu;
            if (!W->rhasL[r] && W->rhasU[r]) s -= K_D*mu;
            worst = fmax(worst, fabs(s));
            if (getenv("ORACLE_DBGCAT")) fprintf(stderr, "   i=%d row %d slackstat res %g\n", i, r, s);
        }
    }
    for (int i = 0; i <= N; i++) {
        StageBd *B = &W->bd[i]; StageIt *I = &W->it[i];
        for (int k = 0; k < NV; k++) {
            if (!B->on[k]) continue;
            double s = rx[i][k] + dw*D[i].dx[k];
            if (R) s += R->eta*R->dr[i][k]*R->dr[i][k]*(I->x[k] + D[i].dx[k] - R->xR[i][k]);
            if (B->hasL[k]) s -= I->zL[k] + D[i].dzL[k];
            if (B->hasU[k]) s += I->zU[k] + D[i].dzU[k];
            if (B->hasL[k] && !B->hasU[k]) s += K_D*mu;
            if (!B->hasL[k] && B->hasU[k]) s -= K_D*mu;
            if (getenv("ORACLE_DBGCAT")) fprintf(stderr, "   i=%d var %d (x %.6e dx %.3e slackL %.3e slackU %.3e zL %.3e zU %.3e dzL %.3e dzU %.3e rx %.3e) stat res %g\n", i, k, I->x[k], D[i].dx[k], I->x[k] - B->lb[k], B->ub[k] - I->x[k], I->zL[k], I->zU[k], D[i].dzL[k], D[i].dzU[k], rx[i][k], s);
            worst = fmax(worst, fabs(s));
        }
    }
    free(rx);
    return worst;
}

/* fraction to the boundary for the primal variables of a direction */
static double alpha_primal_max(const Ws *W, const StageDir *D, double tau)
{
    double a = 1.0; int N = W->N;
    for (int i = 0; i <= N; i++) {
        const StageBd *B = &W->bd[i]; const StageIt *I = &W->it[i];
        for (int k = 0; k < NV; k++) {
            if (!B->on[k]) continue;
            double dx = D[i].dx[k];
            if (B->hasL[k] && dx < 0) a = fmin(a, -tau*(I->x[k] - B->lb[k])/dx);
            if (B->hasU[k] && dx > 0) a = fmin(a, tau*(B->ub[k] - I->x[k])/dx);
        }
        if (i == N) break;
        for (int r = 0; r < NR; r++) {
            if (!W->rowOn[r]) continue;
            double dsg = D[i].dsig[r];
            if (W->rhasL[r] && dsg < 0) a = fmin(a, -tau*(I->sig[r] - W->dL[r])/dsg);
            if (W->rhasU[r] && dsg > 0) a = fmin(a, tau*(W->dU[r] - I->sig[r])/dsg);
        }
    }
    return a;
}

static double alpha_dual_max(const Ws *W, const StageDir *D, double tau)
{
    double a = 1.0; int N = W->N;
    for (int i = 0; i <= N; i++) {
        const StageBd *B = &W->bd[i]; const StageIt *I = &W->it[i];
        for (int k = 0; k < NV; k++) {
            if (!B->on[k]) continue;
            if (B->hasL[k] && D[i].dzL[k] < 0) a = fmin(a, -tau*I->zL[k]/D[i].dzL[k]);
            if (B->hasU[k] && D[i].dzU[k] < 0) a = fmin(a, -tau*I->zU[k]/D[i].dzU[k]);
        }
        if (i == N) break;
        for (int r = 0; r < NR; r++) {
            if (!W->rowOn[r]) continue;
            if (W->rhasL[r] && D[i].dzLs[r] < 0) a = fmin(a, -tau*I->zLs[r]/D[i].dzLs[r]);
            if (W->rhasU[r] && D[i].dzUs[r] < 0) a = fmin(a, -tau*I->zUs[r]/D[i].dzUs[r]);
        }
    }
    return a;
}

static void make_trial(Ws *W, const StageDir *D, double alpha)
{
    int N = W->N;
    for (int i = 0; i <= N; i++) {
        W->trial[i] = W->it[i];
        for (int k = 0; k < NV; k++) if (W->bd[i].on[k]) W->trial[i].x[k] += alpha*D[i].dx[k];
        if (i < N) for (int r = 0; r < NR; r++) if (W->rowOn[r]) W->trial[i].sig[r] += alpha*D[i].dsig[r];
    }
}

static int filter_ok(const Ws *W, double theta, double phi)
{
    for (int j = 0; j < W->nfilt; j++)
        if (theta >= W->filt_theta[j] && phi >= W->filt_phi[j]) return 0;
    return 1;
}

/* lhs <= rhs up to round-off relative to basval (IPOPT's Compare_le) */
static int cmp_le(double lhs, double rhs, double basval) { return lhs - rhs <= 10.0*DBL_EPSILON*fabs(basval); }

/* directional derivative of the barrier function along D, largest component of the step, largest relative component (tiny-step test) */
static void step_measures(const Ws *W, double mu, const StageDir *D, double *gphid_out, double *dnorm_out, double *rel_out)
{
    const int N = W->N;
    double gphid = 0, dnorm = 0, rel_step = 0;
    for (int i = 0; i <= N; i++) {
        const StageBd *B = &W->bd[i]; const StageIt *I = &W->it[i];
        for (int k = 0; k < NV; k++) {
            if (!B->on[k]) continue;
            double Sg, gp; bar_terms(I->x[k], B->lb[k], B->ub[k], B->hasL[k], B->hasU[k], I->zL[k], I->zU[k], mu, &Sg, &gp);
            double go = 0;
            if (i < N) go = W->ev[i].objg[var2loc[k]];
            if (i > 0 && k == VT) go += W->ev[i - 1].objg[LT1];
            if (i > 0 && k == VB) go += W->ev[i - 1].objg[LB1];
            if (k == VF && i + 1 < N) go += W->ev[i + 1].objg[LQ];
            gphid += (go + gp)*D[i].dx[k];
            dnorm = fmax(dnorm, fabs(D[i].dx[k]));
            rel_step = fmax(rel_step, fabs(D[i].dx[k])/(1 + fabs(I->x[k])));
        }
        if (i == N) break;
        for (int r = 0; r < NR; r++) {
            if (!W->rowOn[r]) continue;
            double Sg, gp; bar_terms(I->sig[r], W->dL[r], W->dU[r], W->rhasL[r], W->rhasU[r], I->zLs[r], I->zUs[r], mu, &Sg, &gp);
            gphid += gp*D[i].dsig[r];
            dnorm = fmax(dnorm, fabs(D[i].dsig[r]));
            rel_step = fmax(rel_step, fabs(D[i].dsig[r])/(1 + fabs(I->sig[r])));
        }
    }
    *gphid_out = gphid; *dnorm_out = dnorm; *rel_out = rel_step;
}

/* StopWatchDog: the stored iterate and the stored direction back in place, with the derivatives and residuals of that point */
static void watchdog_restore(Ws *W, const StageIt *wd_it, const StageDir *wd_dir, double (*res_c)[2], double (*res_d)[NR])
{
    const int N = W->N;
    memcpy(W->it, wd_it, (N + 1)*sizeof(StageIt)); memcpy(W->dir, wd_dir, (N + 1)*sizeof(StageDir));
    for (int i = 0; i < N; i++) eval_interval(W, W->it, i, &W->ev[i], 2);
    for (int i = 0; i < N; i++) { res_c[i][0] = W->ev[i].c[0]; res_c[i][1] = W->ev[i].c[1]; for (int r = 0; r < NR; r++) res_d[i][r] = W->rowOn[r] ? W->ev[i].d[r] - W->it[i].sig[r] : 0; }
}

/* ------------------------------------------------------------------------------------------
 * driver
 * ---------------------------------------------------------------------------------------- */
static void ws_alloc(Ws *W, int N)
{
    W->N = N;
    W->it = calloc(N + 1, sizeof(StageIt)); W->trial = calloc(N + 1, sizeof(StageIt));
    W->bd = calloc(N + 1, sizeof(StageBd)); W->ev = calloc(N + 1, sizeof(StageEv)); W->kk = calloc(N + 1, sizeof(StageKkt));
    W->dir = calloc(N + 1, sizeof(StageDir)); W->soc = calloc(N + 1, sizeof(StageDir));
    W->sct = calloc(N + 1, sizeof(double)); W->scb = calloc(N + 1, sizeof(double)); W->G = calloc(N + 1, sizeof(double));
}
static void ws_free(Ws *W)
{
    free(W->it); free(W->trial); free(W->bd); free(W->ev); free(W->kk); free(W->dir); free(W->soc); free(W->sct); free(W->scb); free(W->G);
}

/* kp: bound_push and bound_frac (both 1e-2 for a cold start; warm_start_bound_push/_frac for a warm one) */
static double push_in(double x, double lb, double ub, int hasL, int hasU, double kp)
{
    if (hasL && hasU) {
        double pL = fmin(kp*fmax(1.0, fabs(lb)), kp*(ub - lb));
        double pU = fmin(kp*fmax(1.0, fabs(ub)), kp*(ub - lb));
        if (x < lb + pL) x = lb + pL;
        if (x > ub - pU) x = ub - pU;
    } else if (hasL) { double pL = kp*fmax(1.0, fabs(lb)); if (x < lb + pL) x = lb + pL; }
    else if (hasU) { double pU = kp*fmax(1.0, fabs(ub)); if (x > ub - pU) x = ub - pU; }
    return x;
}

static int debug_level(void) { const char *s = getenv("ORACLE_DEBUG"); return s ? atoi(s) : 0; }

/* ------------------------------------------------------------------------------------------
 * feasibility restoration phase
 * ---------------------------------------------------------------------------------------- */
static int g_resto = 1;
void oracle_set_restoration(int on) { g_resto = on; }

/*
 * IPOPT's watchdog procedure (IpBacktrackingLineSearch.cpp: StartWatchDog / StopWatchDog, FilterLSAcceptor::StartWatchDog / StopWatchDog;
 * options watchdog_shortened_iter_trigger = 10, watchdog_trial_iter_max = 3 -- the defaults the reference runs with, ocp.py:290).  IPOPT's
 * sources are not part of the reference tree or of this image: the procedure is restated from the published implementation, parity unpinned.
 *   - an iteration whose accepted step needed more than one backtracking step counts as shortened, one that took its first trial step
 *     resets the count (n_steps > 1 / n_steps == 0 in FindAcceptableTrialPoint); a change of the barrier parameter resets it too
 *     (BacktrackingLineSearch::Reset) and ends a running watchdog;
 *   - after 10 shortened iterations the iterate, the search direction and (theta, phi, grad phi^T d) of that point are stored; from then on
 *     only the full fraction-to-the-boundary step is tried, tested against the stored reference values; a trial point that passes ends the
 *     procedure (the filter is augmented with the reference point), one that does not is taken all the same, without touching the filter;
 *   - after 3 such trial iterations without success (or a tiny step, or a trial point that cannot be evaluated) the stored iterate comes
 *     back and the ordinary backtracking line search runs on the stored direction, starting from half the maximal step.
 * oracle_max_shortened_run: telemetry of round 3 -- the longest run of successive iterations with at least one backtracking step.
 */
static int g_watchdog = 1;
void oracle_set_watchdog(int on) { g_watchdog = on; }
static const int WD_TRIAL_MAX = 3;
static int g_wd_started = 0, g_wd_succeeded = 0, g_wd_forced = 0;      /* procedures started, ended by an accepted trial point; trial points taken without the filter's consent */
void oracle_watchdog_counts(int *started, int *succeeded, int reset)
{
#pragma omp critical(msd_shortened)
    { *started = g_wd_started; *succeeded = g_wd_succeeded; if (reset) g_wd_started = g_wd_succeeded = 0; }
}
int oracle_watchdog_forced_steps(int reset)
{
    int v;
#pragma omp critical(msd_shortened)
    { v = g_wd_forced; if (reset) g_wd_forced = 0; }
    return v;
}
static void note_watchdog(int started, int succeeded, int forced_steps)
{
#pragma omp critical(msd_shortened)
    { g_wd_started += started; g_wd_succeeded += succeeded; g_wd_forced += forced_steps; }
}
static int g_max_shortened = 0;
int oracle_max_shortened_run(int reset) { int v; 
#pragma omp critical(msd_shortened)
    { v = g_max_shortened; if (reset) g_max_shortened = 0; }
    return v; }
static void note_shortened_run(int run) {
#pragma omp critical(msd_shortened)
    { if (run > g_max_shortened) g_max_shortened = run; } }

static const double RESTO_RHO = 1000.0;        /* resto_penalty_parameter */
static const double RESTO_KAPPA = 0.9;         /* required_infeasibility_reduction */
static const double RESTO_THETA_MAX_FACT = 1e8;/* resto.theta_max_fact */
static const double BOUND_MULT_RESET = 1e3;    /* bound_mult_reset_threshold */
static const int RESTO_MAX_ITER = 100;         /* iterations of one restoration phase (not IPOPT's default, which is unlimited: a phase that has not found an
                                                * acceptable point by then is given up -- the phases that succeed take 1 to 30 -- and the solve ends with
                                                * Restoration_Failed, after which it is repeated from the other starting point) */

/* value of relaxed row j of interval i at an evaluated point (without n - p) */
static double rs_row(const Ws *W, const StageIt *it, int i, const StageEv *e, int j)
{
    return j == 0 ? W->sct[i]*e->c[0] : j == 1 ? W->scb[i]*e->c[1] : e->d[j - 2] - it[i].sig[j - 2];
}
static int rs_on(const Ws *W, int j) { return j < 2 || W->rowOn[j - 2]; }

/* theta_R = 1-norm of the relaxed rows, phi_R = rho sum(n + p) + eta/2 |D_R (x - x_R)|^2 + barrier terms of (x, sigma, n, p) */
static void resto_merit(const Ws *W, const StageIt *it, const StageRs *rv, double mu, double *theta, double *phi, int *ok)
{
    const Resto *R = W->resto; const int N = W->N; double th = 0, bar = 0, f = 0; *ok = 1;
    StageEv e;
    for (int i = 0; i < N; i++) {
        eval_interval(W, it, i, &e, 0);
        for (int j = 0; j < NC; j++) {
            if (!rs_on(W, j)) continue;
            const double n = rv[i].n[j], p = rv[i].p[j];
            th += fabs(rs_row(W, it, i, &e, j) + n - p);
            if (n <= 0 || p <= 0) *ok = 0; else bar -= mu*(log(n) + log(p));
            bar += K_D*mu*(n + p); f += R->rho*(n + p);
        }
        for (int r = 0; r < NR; r++) {
            if (!W->rowOn[r]) continue;
            double sg = it[i].sig[r];
            if (W->rhasL[r]) { double s = sg - W->dL[r]; if (s <= 0) *ok = 0; else bar -= mu*log(s); }
            if (W->rhasU[r]) { double s = W->dU[r] - sg; if (s <= 0) *ok = 0; else bar -= mu*log(s); }
            if (W->rhasL[r] && !W->rhasU[r]) bar += K_D*mu*(sg - W->dL[r]);
            if (!W->rhasL[r] && W->rhasU[r]) bar += K_D*mu*(W->dU[r] - sg);
        }
    }
    for (int i = 0; i <= N; i++) {
        const StageBd *B = &W->bd[i];
        for (int k = 0; k < NV; k++) {
            if (!B->on[k]) continue;
            double x = it[i].x[k];
            if (B->hasL[k]) { double s = x - B->lb[k]; if (s <= 0) *ok = 0; else bar -= mu*log(s); }
            if (B->hasU[k]) { double s = B->ub[k] - x; if (s <= 0) *ok = 0; else bar -= mu*log(s); }
            if (B->hasL[k] && !B->hasU[k]) bar += K_D*mu*(x - B->lb[k]);
            if (!B->hasL[k] && B->hasU[k]) bar += K_D*mu*(B->ub[k] - x);
            const double q = R->dr[i][k]*(x - R->xR[i][k]);
            f += 0.5*R->eta*q*q;
        }
    }
    if (!isfinite(th) || !isfinite(bar) || !isfinite(f)) *ok = 0;
    *theta = th; *phi = f + bar;
}

/* optimality error of the restoration problem (W->ev holds the derivatives at W->it) */
static void resto_kkt_error(Ws *W, double mu, Err *Rr)
{
    const Resto *R = W->resto; int N = W->N; const Prob *P = &W->P;
    double dual = 0, prim = 0, comp = 0, comp0 = 0, sumlam = 0, sumz = 0; int nlam = 0, nz = 0;
    double (*gl)[NV] = calloc(N + 1, sizeof *gl);
    for (int i = 0; i < N; i++) {
        StageEv *e = &W->ev[i]; StageIt *I = &W->it[i]; const StageRs *v = &R->v[i];
        double loc[NL];
        for (int a = 0; a < NL; a++) loc[a] = 0;
        for (int r = 0; r < NR; r++) if (W->rowOn[r]) for (int a = 0; a < NL; a++) loc[a] += I->nu[r]*e->gr[r][a];
        loc[LT1] += I->lam[0]; loc[LT] -= I->lam[0];
        loc[LB] -= I->lam[0]*e->tg[0]; loc[LF] -= I->lam[0]*e->tg[1]; loc[LP] -= P->withPn ? I->lam[0]*e->tg[1] : 0;
        loc[LB1] += I->lam[1];
        loc[LB] -= I->lam[1]*e->bg[0]; loc[LF] -= I->lam[1]*e->bg[1]; loc[LP] -= P->withPn ? I->lam[1]*e->bg[1] : 0;
        gl[i][VT] += loc[LT]; gl[i][VB] += loc[LB]; gl[i][VF] += loc[LF]; gl[i][VP] += loc[LP]; gl[i][VS] += loc[LS];
        if (i > 0) gl[i - 1][VF] += loc[LQ];
        gl[i + 1][VT] += loc[LT1]; gl[i + 1][VB] += loc[LB1];
        for (int j = 0; j < NC; j++) {
            if (!rs_on(W, j)) continue;
            const double sc = j == 0 ? W->sct[i] : j == 1 ? W->scb[i] : 1.0;
            const double y = (j < 2 ? I->lam[j] : I->nu[j - 2])/sc;
            prim = fmax(prim, fabs(rs_row(W, W->it, i, e, j) + v->n[j] - v->p[j]));
            sumlam += fabs(y); nlam++;
            dual = fmax(dual, fmax(fabs(R->rho + y - v->zn[j]), fabs(R->rho - y - v->zp[j])));
            const double cn = v->n[j]*v->zn[j], cp = v->p[j]*v->zp[j];
            comp = fmax(comp, fmax(fabs(cn - mu), fabs(cp - mu))); comp0 = fmax(comp0, fmax(fabs(cn), fabs(cp)));
            sumz += v->zn[j] + v->zp[j]; nz += 2;
        }
        for (int r = 0; r < NR; r++) {
            if (!W->rowOn[r]) continue;
            double sg = I->sig[r], gs = -I->nu[r];
            if (W->rhasL[r]) { gs -= I->zLs[r]; double c = (sg - W->dL[r])*I->zLs[r]; comp = fmax(comp, fabs(c - mu)); comp0 = fmax(comp0, fabs(c)); sumz += I->zLs[r]; nz++; }
            if (W->rhasU[r]) { gs += I->zUs[r]; double c = (W->dU[r] - sg)*I->zUs[r]; comp = fmax(comp, fabs(c - mu)); comp0 = fmax(comp0, fabs(c)); sumz += I->zUs[r]; nz++; }
            dual = fmax(dual, fabs(gs));
        }
    }
    for (int i = 0; i <= N; i++) {
        StageBd *B = &W->bd[i]; StageIt *I = &W->it[i];
        for (int k = 0; k < NV; k++) {
            if (!B->on[k]) continue;
            double g = gl[i][k] + R->eta*R->dr[i][k]*R->dr[i][k]*(I->x[k] - R->xR[i][k]);
            if (B->hasL[k]) { g -= I->zL[k]; double c = (I->x[k] - B->lb[k])*I->zL[k]; comp = fmax(comp, fabs(c - mu)); comp0 = fmax(comp0, fabs(c)); sumz += I->zL[k]; nz++; }
            if (B->hasU[k]) { g += I->zU[k]; double c = (B->ub[k] - I->x[k])*I->zU[k]; comp = fmax(comp, fabs(c - mu)); comp0 = fmax(comp0, fabs(c)); sumz += I->zU[k]; nz++; }
            dual = fmax(dual, fabs(g));
        }
    }
    free(gl);
    memset(Rr, 0, sizeof *Rr);
    Rr->sd = fmax(K_SMAX, (sumlam + sumz)/fmax(1, nlam + nz))/K_SMAX;
    Rr->sc = fmax(K_SMAX, sumz/fmax(1, nz))/K_SMAX;
    Rr->dual = dual; Rr->primal = prim; Rr->compl_ = comp;
    Rr->E = fmax(dual/Rr->sd, fmax(prim, comp/Rr->sc));
    Rr->dual_u = dual; Rr->primal_u = prim; Rr->compl_u = comp0;
}

/*
 * The restoration phase proper: the same filter interior-point iteration on
 *   min rho sum(n + p) + sqrt(mu)/2 |D_R (x - x_R)|^2   s.t.  rows(x, sigma) + n - p = 0,  bounds on x and sigma,  n, p >= 0
 * started at the current point x_R with mu = max(mu, |rows|_inf), (n, p) from the closed-form minimiser, z = mu/(n, p), zero row
 * multipliers and the bound multipliers cut at rho (IpRestoIterateInitializer).  It ends as soon as an iterate reduces the
 * infeasibility of the original problem to 90 % and is acceptable to the original filter and to (theta_ref, phi_ref)
 * (IpRestoFilterConvCheck); if instead the restoration problem itself converges, the original problem is locally infeasible there.
 * On success the original iterate takes (x, sigma), its bound multipliers a step towards mu/slack at the new point (reset to 1 when
 * one exceeds 1000) and zero row multipliers (constr_mult_reset_threshold = 0).  No second-order correction inside.
 * Returns 1 restored, 0 failed, -1 locally infeasible, -2 iteration limit; *nit = iterations taken.
 */
static int restoration(Ws *W, double mu_orig, double theta_ref, double phi_ref, int iter0, int *nit, double *hist, int hist_cap, int dbg)
{
    const Prob *P = &W->P; const int N = W->N;
    Resto Rs; Resto *R = &Rs; memset(R, 0, sizeof *R);
    R->rho = RESTO_RHO;
    R->v = calloc(N + 1, sizeof(StageRs)); R->trial = calloc(N + 1, sizeof(StageRs)); R->d = calloc(N + 1, sizeof(StageRsDir));
    R->xR = calloc(N + 1, sizeof *R->xR); R->dr = calloc(N + 1, sizeof *R->dr);
    StageIt *save = malloc((N + 1)*sizeof(StageIt)); memcpy(save, W->it, (N + 1)*sizeof(StageIt));
    double (*res_c)[2] = calloc(N + 1, sizeof *res_c);
    double (*res_d)[NR] = calloc(N + 1, sizeof *res_d);
    double ftheta[512], fphi[512]; int nf = 0;
    int ret = 0, k = 0;

    for (int i = 0; i <= N; i++) for (int kk = 0; kk < NV; kk++) { R->xR[i][kk] = W->it[i].x[kk]; R->dr[i][kk] = 1.0/fmax(1.0, fabs(W->it[i].x[kk])); }
    double cmax = 0;
    for (int i = 0; i < N; i++) {
        eval_interval(W, W->it, i, &W->ev[i], 0);
        for (int j = 0; j < NC; j++) if (rs_on(W, j)) cmax = fmax(cmax, fabs(rs_row(W, W->it, i, &W->ev[i], j)));
    }
    double mu = fmax(mu_orig, cmax), tau = fmax(K_TAU_MIN, 1 - mu);
    R->eta = sqrt(mu);
    for (int i = 0; i < N; i++) {
        StageRs *v = &R->v[i];
        for (int j = 0; j < NC; j++) {
            if (!rs_on(W, j)) { v->n[j] = v->p[j] = v->zn[j] = v->zp[j] = 1; continue; }
            const double c = rs_row(W, W->it, i, &W->ev[i], j), a = (mu - R->rho*c)/(2*R->rho);
            v->n[j] = a + sqrt(a*a + mu*c/(2*R->rho)); v->p[j] = c + v->n[j];
            v->zn[j] = mu/v->n[j]; v->zp[j] = mu/v->p[j];
        }
    }
    for (int i = 0; i <= N; i++) {
        StageIt *I = &W->it[i];
        for (int kk = 0; kk < NV; kk++) { I->zL[kk] = fmin(R->rho, I->zL[kk]); I->zU[kk] = fmin(R->rho, I->zU[kk]); }
        for (int r = 0; r < NR; r++) { I->zLs[r] = fmin(R->rho, I->zLs[r]); I->zUs[r] = fmin(R->rho, I->zUs[r]); I->nu[r] = 0; }
        I->lam[0] = I->lam[1] = 0;
    }
    W->resto = R;

    double thR, phR; int okR;
    resto_merit(W, W->it, R->v, mu, &thR, &phR, &okR);
    const double thmax = RESTO_THETA_MAX_FACT*fmax(1.0, thR), thmin = 1e-4*fmax(1.0, thR);
    double delta_last = 0, alpha_pr = 0, alpha_du = 0, dnorm = 0;
    int tiny_count = 0;
    const double mu_floor = fmin(P->tol, 1e-4)/(K_EPS + 1.0);
    Err Er;

    for (k = 0; ; k++) {
        for (int i = 0; i < N; i++) eval_interval(W, W->it, i, &W->ev[i], 2);
        resto_kkt_error(W, 0.0, &Er);
        double th_o, ph_o; int ok_o;
        merit_terms(W, W->it, mu_orig, &th_o, &ph_o, &ok_o);
        if (hist && k > 0 && iter0 + k < hist_cap) {
            double *hh = hist + 8*(iter0 + k);
            hh[0] = iter0 + k; hh[1] = objective_value(W, W->it)/W->sf; hh[2] = th_o; hh[3] = Er.dual; hh[4] = log10(mu); hh[5] = dnorm; hh[6] = alpha_du; hh[7] = alpha_pr;
        }
        if (dbg) fprintf(stderr, "[oracle] it %3dr obj %.8e theta %.2e | resto inf_pr %.2e inf_du %.2e compl %.2e lg(mu) %5.1f |d| %.2e a_du %.2e a_pr %.2e\n",
                         iter0 + k, objective_value(W, W->it)/W->sf, th_o, Er.primal, Er.dual, Er.compl_, log10(mu), dnorm, alpha_du, alpha_pr);
        /* back to the original problem? (not before one step has been taken) */
        if (k >= 1 && ok_o && th_o <= RESTO_KAPPA*theta_ref && th_o <= W->theta_max && filter_ok(W, th_o, ph_o)
            && (cmp_le(th_o, (1 - G_THETA)*theta_ref, theta_ref) || cmp_le(ph_o - phi_ref, -G_PHI*theta_ref, phi_ref))) { ret = 1; break; }
        /* the restoration problem itself solved: a stationary point of the infeasibility */
        if (Er.E <= P->tol && Er.dual_u <= 1.0 && Er.primal_u <= 1e-4 && Er.compl_u <= 1e-4) {
            double pinf = 0;
            for (int i = 0; i < N; i++) {
                pinf = fmax(pinf, fmax(fabs(W->ev[i].c[0]), fabs(W->ev[i].c[1])));
                for (int r = 0; r < NR; r++) if (W->rowOn[r]) pinf = fmax(pinf, fabs(W->ev[i].d[r] - W->it[i].sig[r])/W->rs[r]);
            }
            ret = (pinf <= 1e-4) ? 0 : -1;
            break;
        }
        if (iter0 + k >= P->maxIter) { ret = -2; break; }
        if (k >= RESTO_MAX_ITER) { ret = 0; break; }

        /* barrier parameter of the restoration problem (monotone); the proximity weight follows it */
        {
            Err Rm; resto_kkt_error(W, mu, &Rm);
            int changed = 0;
            while (Rm.E <= K_EPS*mu && mu > mu_floor) {
                double nm = fmax(mu_floor, fmin(K_MU_LIN*mu, pow(mu, K_MU_SUP)));
                if (nm >= mu) break;
                mu = nm; tau = fmax(K_TAU_MIN, 1 - mu); changed = 1;
                resto_kkt_error(W, mu, &Rm);
            }
            if (changed) { nf = 0; R->eta = sqrt(mu); }
        }
        resto_merit(W, W->it, R->v, mu, &thR, &phR, &okR);

        for (int i = 0; i < N; i++) {
            const StageRs *v = &R->v[i];
            for (int j = 0; j < NC; j++) {
                double rh = 0;
                if (rs_on(W, j)) rh = rs_row(W, W->it, i, &W->ev[i], j) + v->n[j] - v->p[j] + (mu - R->rho*v->n[j])/v->zn[j] - (mu - R->rho*v->p[j])/v->zp[j];
                if (j < 2) res_c[i][j] = rh; else res_d[i][j - 2] = rh;
            }
        }
        double dw = 0; int ok = compute_direction(W, mu, 0.0, (const double (*)[2])res_c, (const double (*)[NR])res_d, W->dir);
        if (!ok) {
            dw = (delta_last == 0) ? DW_0 : fmax(DW_MIN, KW_MINUS*delta_last);
            for (;;) {
                ok = compute_direction(W, mu, dw, (const double (*)[2])res_c, (const double (*)[NR])res_d, W->dir);
                if (ok) break;
                dw *= (delta_last == 0) ? KW_PLUS_BAR : KW_PLUS;
                if (dw > DW_MAX) break;
            }
            if (!ok) { if (dbg) fprintf(stderr, "[oracle] resto: regularisation failed\n"); ret = 0; break; }
            delta_last = dw;
        }

        if (dbg >= 2) fprintf(stderr, "[oracle]    resto dw %.2e newton residual %.3e\n", dw, direction_residual(W, mu, dw, (const double (*)[2])res_c, (const double (*)[NR])res_d, W->dir));
        /* directional derivative of phi_R, step norms, fraction to the boundary */
        double gphid = 0, rel_step = 0; dnorm = 0;
        double amax = alpha_primal_max(W, W->dir, tau); alpha_du = alpha_dual_max(W, W->dir, tau);
        for (int i = 0; i <= N; i++) {
            StageBd *B = &W->bd[i]; StageIt *I = &W->it[i];
            for (int kk = 0; kk < NV; kk++) {
                if (!B->on[kk]) continue;
                double Sg, gp; bar_terms(I->x[kk], B->lb[kk], B->ub[kk], B->hasL[kk], B->hasU[kk], I->zL[kk], I->zU[kk], mu, &Sg, &gp);
                gphid += (R->eta*R->dr[i][kk]*R->dr[i][kk]*(I->x[kk] - R->xR[i][kk]) + gp)*W->dir[i].dx[kk];
                dnorm = fmax(dnorm, fabs(W->dir[i].dx[kk]));
                rel_step = fmax(rel_step, fabs(W->dir[i].dx[kk])/(1 + fabs(I->x[kk])));
            }
            if (i == N) break;
            for (int r = 0; r < NR; r++) {
                if (!W->rowOn[r]) continue;
                double Sg, gp; bar_terms(I->sig[r], W->dL[r], W->dU[r], W->rhasL[r], W->rhasU[r], I->zLs[r], I->zUs[r], mu, &Sg, &gp);
                gphid += gp*W->dir[i].dsig[r];
                dnorm = fmax(dnorm, fabs(W->dir[i].dsig[r]));
                rel_step = fmax(rel_step, fabs(W->dir[i].dsig[r])/(1 + fabs(I->sig[r])));
            }
            const StageRs *v = &R->v[i]; const StageRsDir *q = &R->d[i];
            for (int j = 0; j < NC; j++) {
                if (!rs_on(W, j)) continue;
                gphid += (R->rho - mu/v->n[j] + K_D*mu)*q->dn[j] + (R->rho - mu/v->p[j] + K_D*mu)*q->dp[j];
                dnorm = fmax(dnorm, fmax(fabs(q->dn[j]), fabs(q->dp[j])));
                rel_step = fmax(rel_step, fmax(fabs(q->dn[j])/(1 + v->n[j]), fabs(q->dp[j])/(1 + v->p[j])));
                if (q->dn[j] < 0) amax = fmin(amax, -tau*v->n[j]/q->dn[j]);
                if (q->dp[j] < 0) amax = fmin(amax, -tau*v->p[j]/q->dp[j]);
                if (q->dzn[j] < 0) alpha_du = fmin(alpha_du, -tau*v->zn[j]/q->dzn[j]);
                if (q->dzp[j] < 0) alpha_du = fmin(alpha_du, -tau*v->zp[j]/q->dzp[j]);
            }
        }

        int tiny = rel_step < 10*DBL_EPSILON;
        double alpha = amax; int accepted = 0, ftype_armijo = 0;
        if (tiny) {
            accepted = 1;
            if (++tiny_count >= 2 && mu <= mu_floor*(1 + 1e-12)) { if (dbg) fprintf(stderr, "[oracle] resto: tiny steps\n"); ret = 0; break; }
        } else tiny_count = 0;
        double amin;
        if (gphid < 0) {
            amin = G_THETA;
            amin = fmin(amin, G_PHI*thR/(-gphid));
            if (thR <= thmin) amin = fmin(amin, K_DELTA*pow(thR, S_THETA)/pow(-gphid, S_PHI));
        } else amin = G_THETA;
        amin *= ALPHA_MIN_FRAC;
        for (;;) {
            make_trial(W, W->dir, alpha);
            for (int i = 0; i < N; i++) for (int j = 0; j < NC; j++) { R->trial[i].n[j] = R->v[i].n[j] + alpha*R->d[i].dn[j]; R->trial[i].p[j] = R->v[i].p[j] + alpha*R->d[i].dp[j]; }
            if (accepted) break;
            double th_t, ph_t; int okt;
            resto_merit(W, W->trial, R->trial, mu, &th_t, &ph_t, &okt);
            int ftype = (gphid < 0) && (alpha*pow(-gphid, S_PHI) > K_DELTA*pow(thR, S_THETA));
            int acc = 0;
            if (okt && th_t <= thmax) {
                if (ftype && thR <= thmin) acc = cmp_le(ph_t - phR, ETA_PHI*alpha*gphid, phR);
                else acc = cmp_le(th_t, (1 - G_THETA)*thR, thR) || cmp_le(ph_t - phR, -G_PHI*thR, phR);
                if (acc) for (int m = 0; m < nf; m++) if (th_t >= ftheta[m] && ph_t >= fphi[m]) { acc = 0; break; }
            }
            if (acc) { accepted = 1; ftype_armijo = ftype && cmp_le(ph_t - phR, ETA_PHI*alpha*gphid, phR); break; }
            if (dbg >= 3) fprintf(stderr, "[oracle]      resto trial alpha %.3e theta %.6e (%.6e) phi %.10e (%.10e) ok %d ftype %d\n", alpha, th_t, thR, ph_t, phR, okt, ftype);
            alpha *= 0.5; W->n_back++;
            if (alpha < amin) break;
        }
        if (!accepted) { if (dbg) fprintf(stderr, "[oracle] resto: line search failed (alpha %.3e < %.3e, gphid %.3e thetaR %.3e)\n", alpha, amin, gphid, thR); ret = 0; break; }
        alpha_pr = alpha;
        if (!tiny && !ftype_armijo && nf < 512) { ftheta[nf] = (1 - G_THETA)*thR; fphi[nf] = phR - G_PHI*thR; nf++; }

        for (int i = 0; i <= N; i++) {
            StageIt *I = &W->it[i]; StageBd *B = &W->bd[i]; const StageDir *d = &W->dir[i];
            for (int kk = 0; kk < NV; kk++) {
                if (!B->on[kk]) continue;
                I->x[kk] = W->trial[i].x[kk];
                if (B->hasL[kk]) { I->zL[kk] += alpha_du*d->dzL[kk]; double s = I->x[kk] - B->lb[kk]; I->zL[kk] = fmax(fmin(I->zL[kk], K_SIGMA*mu/s), mu/(K_SIGMA*s)); }
                if (B->hasU[kk]) { I->zU[kk] += alpha_du*d->dzU[kk]; double s = B->ub[kk] - I->x[kk]; I->zU[kk] = fmax(fmin(I->zU[kk], K_SIGMA*mu/s), mu/(K_SIGMA*s)); }
            }
            if (i == N) break;
            for (int r = 0; r < NR; r++) {
                if (!W->rowOn[r]) continue;
                I->sig[r] = W->trial[i].sig[r];
                I->nu[r] += alpha_pr*d->dnu[r];
                if (W->rhasL[r]) { I->zLs[r] += alpha_du*d->dzLs[r]; double s = I->sig[r] - W->dL[r]; I->zLs[r] = fmax(fmin(I->zLs[r], K_SIGMA*mu/s), mu/(K_SIGMA*s)); }
                if (W->rhasU[r]) { I->zUs[r] += alpha_du*d->dzUs[r]; double s = W->dU[r] - I->sig[r]; I->zUs[r] = fmax(fmin(I->zUs[r], K_SIGMA*mu/s), mu/(K_SIGMA*s)); }
            }
            I->lam[0] += alpha_pr*d->dlam[0]; I->lam[1] += alpha_pr*d->dlam[1];
            StageRs *v = &R->v[i]; const StageRsDir *q = &R->d[i];
            for (int j = 0; j < NC; j++) {
                if (!rs_on(W, j)) continue;
                v->n[j] = R->trial[i].n[j]; v->p[j] = R->trial[i].p[j];
                v->zn[j] += alpha_du*q->dzn[j]; v->zn[j] = fmax(fmin(v->zn[j], K_SIGMA*mu/v->n[j]), mu/(K_SIGMA*v->n[j]));
                v->zp[j] += alpha_du*q->dzp[j]; v->zp[j] = fmax(fmin(v->zp[j], K_SIGMA*mu/v->p[j]), mu/(K_SIGMA*v->p[j]));
            }
        }
    }
    W->resto = NULL;

    if (ret == 1) {
        /* bound multipliers of the original problem: z + alpha dz with dz = (mu - z slack_new)/slack_old (MinC_1NrmRestorationPhase::
         * ComputeBoundMultiplierStep), alpha from the fraction-to-the-boundary rule; all of them 1 when one ends above 1000 */
        const double tau_o = fmax(K_TAU_MIN, 1 - mu_orig);
        double a = 1.0, zmax = 0;
#define RS_DZ(z, sn, so) ((mu_orig - (z)*(sn))/(so))
        for (int pass = 0; pass < 2; pass++)
            for (int i = 0; i <= N; i++) {
                StageIt *I = &W->it[i]; const StageIt *O = &save[i]; const StageBd *B = &W->bd[i];
                for (int kk = 0; kk < NV; kk++) {
                    if (!B->on[kk]) continue;
                    if (B->hasL[kk]) { double dz = RS_DZ(O->zL[kk], I->x[kk] - B->lb[kk], O->x[kk] - B->lb[kk]); if (!pass) { if (dz < 0) a = fmin(a, -tau_o*O->zL[kk]/dz); } else { I->zL[kk] = O->zL[kk] + a*dz; zmax = fmax(zmax, I->zL[kk]); } }
                    if (B->hasU[kk]) { double dz = RS_DZ(O->zU[kk], B->ub[kk] - I->x[kk], B->ub[kk] - O->x[kk]); if (!pass) { if (dz < 0) a = fmin(a, -tau_o*O->zU[kk]/dz); } else { I->zU[kk] = O->zU[kk] + a*dz; zmax = fmax(zmax, I->zU[kk]); } }
                }
                if (i == N) break;
                for (int r = 0; r < NR; r++) {
                    if (!W->rowOn[r]) continue;
                    if (W->rhasL[r]) { double dz = RS_DZ(O->zLs[r], I->sig[r] - W->dL[r], O->sig[r] - W->dL[r]); if (!pass) { if (dz < 0) a = fmin(a, -tau_o*O->zLs[r]/dz); } else { I->zLs[r] = O->zLs[r] + a*dz; zmax = fmax(zmax, I->zLs[r]); } }
                    if (W->rhasU[r]) { double dz = RS_DZ(O->zUs[r], W->dU[r] - I->sig[r], W->dU[r] - O->sig[r]); if (!pass) { if (dz < 0) a = fmin(a, -tau_o*O->zUs[r]/dz); } else { I->zUs[r] = O->zUs[r] + a*dz; zmax = fmax(zmax, I->zUs[r]); } }
                }
            }
#undef RS_DZ
        for (int i = 0; i <= N; i++) {
            StageIt *I = &W->it[i]; const StageBd *B = &W->bd[i];
            if (zmax > BOUND_MULT_RESET) {
                for (int kk = 0; kk < NV; kk++) if (B->on[kk]) { if (B->hasL[kk]) I->zL[kk] = 1; if (B->hasU[kk]) I->zU[kk] = 1; }
                for (int r = 0; r < NR; r++) if (i < N && W->rowOn[r]) { if (W->rhasL[r]) I->zLs[r] = 1; if (W->rhasU[r]) I->zUs[r] = 1; }
            }
            I->lam[0] = I->lam[1] = 0;
            for (int r = 0; r < NR; r++) I->nu[r] = 0;
        }
    } else {
        /* the original iterate keeps its bound multipliers (row multipliers zero); (x, sigma) are left where the restoration phase ended */
        for (int i = 0; i <= N; i++) {
            StageIt *I = &W->it[i]; const StageIt *O = &save[i];
            memcpy(I->zL, O->zL, sizeof I->zL); memcpy(I->zU, O->zU, sizeof I->zU); memcpy(I->zLs, O->zLs, sizeof I->zLs); memcpy(I->zUs, O->zUs, sizeof I->zUs);
            I->lam[0] = I->lam[1] = 0;
            for (int r = 0; r < NR; r++) I->nu[r] = 0;
        }
    }
    if (dbg) fprintf(stderr, "[oracle] restoration phase: %d iterations, outcome %d\n", k, ret);
    free(R->v); free(R->trial); free(R->d); free(R->xR); free(R->dr); free(save); free(res_c); free(res_d);
    *nit = k;
    return ret;
}

int oracle_solve(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                 const double *bmax, double *z_out, double *lam_out, double *stats, double *hist, int hist_cap)
{
    return oracle_solve_warm(ip, dp, ds, grad, curv, bmax, NULL, 0.0, 0.0, z_out, lam_out, stats, hist, hist_cap);
}

/*
 * Warm start (not in the reference, which always cold-starts, ocp.py:325-339): the primal guess replaces the
 * cold-start values, is pushed into the interior with `push` instead of bound_push/bound_frac, the bound and slack
 * multipliers start on the central path of mu0 (z = mu0/slack) and the barrier parameter starts at mu0.
 * Scaling factors are still computed at the (warm) starting point like IPOPT does.
 */
static int solve_core(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                      const double *bmax, const double *guess, double mu0, double push, int skip_lsq,
                      double *z_out, double *lam_out, double *stats, double *hist, int hist_cap,
                      const double *dual_guess, double *dual_out);

int oracle_solve_warm(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                      const double *bmax, const double *guess, double mu0, double push,
                      double *z_out, double *lam_out, double *stats, double *hist, int hist_cap)
{
    return solve_core(ip, dp, ds, grad, curv, bmax, guess, mu0, push, 0, z_out, lam_out, stats, hist, hist_cap, NULL, NULL);
}

static void profile_guess(const Prob *P, double *z);

/* primal-dual warm start; dual_guess / dual_out: [N + 1][OR_DUAL_STRIDE] (either may be NULL); start = the starting point of a plain solve
 * (guess == NULL), so that one entry serves the first solve of a sequence, which only records its multipliers */
int oracle_solve_dual(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                      const double *bmax, int start, const double *guess, const double *dual_guess, double mu0, double push,
                      double *z_out, double *lam_out, double *dual_out, double *stats)
{
    if (guess) return solve_core(ip, dp, ds, grad, curv, bmax, guess, mu0, push, 0, z_out, lam_out, stats, NULL, 0, dual_guess, dual_out);
    if (start == 1) {
        Prob P;
        prob_init(&P, ip, dp, ds, grad, curv, bmax);
        const int nz = (4 + (P.withPn ? 1 : 0))*P.N + 2;
        double *g = malloc(nz*sizeof(double));
        profile_guess(&P, g);
        int st = solve_core(ip, dp, ds, grad, curv, bmax, g, K_MU_INIT, K_PUSH, 1, z_out, lam_out, stats, NULL, 0, NULL, dual_out);
        free(g);
        return st;
    }
    return solve_core(ip, dp, ds, grad, curv, bmax, NULL, 0.0, 0.0, 0, z_out, lam_out, stats, NULL, 0, NULL, dual_out);
}

/* skip_lsq: start with zero constraint multipliers instead of the least-squares estimate (profile start: the estimate costs a
 * KKT solve and buys no iterations there) */
static int solve_core(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                      const double *bmax, const double *guess, double mu0, double push, int skip_lsq,
                      double *z_out, double *lam_out, double *stats, double *hist, int hist_cap,
                      const double *dual_guess, double *dual_out)
{
    const int warm = guess != NULL;
    const double kp = warm ? push : K_PUSH;
    Ws Wst; Ws *W = &Wst; memset(W, 0, sizeof *W);
    prob_init(&W->P, ip, dp, ds, grad, curv, bmax);
    const Prob *P = &W->P; int N = P->N; int dbg = debug_level();
    ws_alloc(W, N);
    setup_bounds(W);
    for (int i = 0; i < N; i++) W->G[i] = track_resistance(P, grad[i], curv[i]);

    /* ---- starting point (ocp.py:325-339): Fel 0.5, Fpb -0.1, s 1, t linear, b (60/3.6)^2 ------------ */
    {
        double dt = (P->tEnd - P->t0)/N, vel0 = (60/3.6)*(60/3.6);
        for (int i = 0; i <= N; i++) {
            StageIt *I = &W->it[i];
            I->x[VT] = P->t0 + dt*i; I->x[VB] = vel0; I->x[VF] = 0.5; I->x[VP] = P->withPn ? -0.1 : 0.0; I->x[VS] = 1;
        }
        if (warm) {
            const int nu = 1 + (P->withPn ? 1 : 0), stp = nu + 3;
            for (int i = 0; i <= N; i++) {
                StageIt *I = &W->it[i];
                const double *q = guess + stp*i;
                if (i < N) { I->x[VF] = q[0]; I->x[VP] = P->withPn ? q[1] : 0.0; I->x[VS] = q[nu]; I->x[VT] = q[nu + 1]; I->x[VB] = q[nu + 2]; }
                else { I->x[VT] = q[0]; I->x[VB] = q[1]; }
            }
        }
    }
    /* ---- gradient-based scaling at the user's starting point (nlp_scaling_method = gradient-based, max gradient 100) ---- */
    W->sf = 1;
    for (int i = 0; i <= N; i++) { W->sct[i] = 1; W->scb[i] = 1; }
    {
        /* fixed variables take their values before scaling/evaluation */
        W->it[0].x[VT] = P->t0; W->it[0].x[VB] = P->v0sq; W->it[N].x[VB] = P->vNsq;
        double gmax = 0, rmax[NR] = {0, 0, 0, 0, 0};
        for (int i = 0; i < N; i++) {
            eval_interval(W, W->it, i, &W->ev[i], 2);
            StageEv *e = &W->ev[i];
            for (int a = 0; a < NL; a++) if (a != LQ) gmax = fmax(gmax, fabs(e->objg[a]));
            double mb = 1.0;    /* entry of b_{i+1} (or none when it is a parameter) */
            if (i == N - 1) mb = 0;
            if (i > 0) mb = fmax(mb, fabs(e->bg[0]));
            mb = fmax(mb, fabs(e->bg[1]));
            W->scb[i] = mb > 100 ? 100/mb : 1;
            double mt = 1.0;
            if (i > 0) mt = fmax(mt, fabs(e->tg[0]));
            mt = fmax(mt, fabs(e->tg[1]));
            W->sct[i] = mt > 100 ? 100/mt : 1;
            for (int r = 0; r < NR; r++) for (int a = 0; a < NL; a++) {
                if (i == 0 && (a == LT || a == LB)) continue;
                if (i == N - 1 && a == LB1) continue;
                rmax[r] = fmax(rmax[r], fabs(e->gr[r][a]));
            }
        }
        gmax *= 1.0;   /* sf is still 1 here */
        if (gmax > 100) W->sf = 100/gmax;
        for (int r = 0; r < NR; r++) if (W->rowOn[r] && rmax[r] > 100) W->rs[r] = 100/rmax[r];
    }
    relax_row_bounds(W);

    /* ---- push the starting point into the interior, initialise slacks and multipliers ------------------------------ */
    for (int i = 0; i <= N; i++) {
        StageBd *B = &W->bd[i]; StageIt *I = &W->it[i];
        for (int k = 0; k < NV; k++) {
            if (!B->on[k]) continue;
            I->x[k] = push_in(I->x[k], B->lb[k], B->ub[k], B->hasL[k], B->hasU[k], kp);
            I->zL[k] = B->hasL[k] ? (warm ? mu0/(I->x[k] - B->lb[k]) : 1) : 0;
            I->zU[k] = B->hasU[k] ? (warm ? mu0/(B->ub[k] - I->x[k]) : 1) : 0;
        }
    }
    for (int i = 0; i < N; i++) {
        eval_interval(W, W->it, i, &W->ev[i], 2);
        for (int r = 0; r < NR; r++) {
            if (!W->rowOn[r]) continue;
            W->it[i].sig[r] = push_in(W->ev[i].d[r], W->dL[r], W->dU[r], W->rhasL[r], W->rhasU[r], kp);
            W->it[i].zLs[r] = W->rhasL[r] ? (warm ? mu0/(W->it[i].sig[r] - W->dL[r]) : 1) : 0;
            W->it[i].zUs[r] = W->rhasU[r] ? (warm ? mu0/(W->dU[r] - W->it[i].sig[r]) : 1) : 0;
        }
    }

    /* primal-dual warm start (OR_DUAL_STRIDE doubles per node: lam 2, nu 5, zL 5, zU 5, zLs 5, zUs 5): the constraint multipliers are
     * taken as given (no least-squares estimate), the bound and slack multipliers too but not below 1e-3 of their central-path
     * value mu0/slack at the pushed point -- a multiplier that was zero at the old solution must be able to grow */
    if (warm && dual_guess) {
        for (int i = 0; i <= N; i++) {
            const double *q = dual_guess + OR_DUAL_STRIDE*i; StageIt *I = &W->it[i]; StageBd *B = &W->bd[i];
            for (int k = 0; k < NV; k++) {
                if (!B->on[k]) continue;
                if (B->hasL[k]) I->zL[k] = fmax(q[7 + k], 1e-3*mu0/(I->x[k] - B->lb[k]));
                if (B->hasU[k]) I->zU[k] = fmax(q[12 + k], 1e-3*mu0/(B->ub[k] - I->x[k]));
            }
            if (i == N) break;
            I->lam[0] = q[0]; I->lam[1] = q[1];
            for (int r = 0; r < NR; r++) {
                if (!W->rowOn[r]) { I->nu[r] = 0; continue; }
                I->nu[r] = q[2 + r];
                if (W->rhasL[r]) I->zLs[r] = fmax(q[17 + r], 1e-3*mu0/(I->sig[r] - W->dL[r]));
                if (W->rhasU[r]) I->zUs[r] = fmax(q[22 + r], 1e-3*mu0/(W->dU[r] - I->sig[r]));
            }
        }
        skip_lsq = 2;
    }
    double mu = warm ? mu0 : K_MU_INIT, tau = fmax(K_TAU_MIN, 1 - mu);
    double (*res_c)[2] = calloc(N + 1, sizeof *res_c);
    double (*res_d)[NR] = calloc(N + 1, sizeof *res_d);
    double (*soc_c)[2] = calloc(N + 1, sizeof *soc_c);
    double (*soc_d)[NR] = calloc(N + 1, sizeof *soc_d);

    /* least-squares multiplier estimate (W&B section 3.6): solve with W = 0, Sigma = I, no barrier terms.
     * Realised with the same machinery: a temporary iterate whose Sigma are 1 and whose gradient is
     * grad f - zL + zU; constraint residuals zero. */
    if (skip_lsq == 2) {
        /* multipliers given */
    } else if (skip_lsq) {
        for (int i = 0; i < N; i++) { W->it[i].lam[0] = W->it[i].lam[1] = 0; for (int r = 0; r < NR; r++) W->it[i].nu[r] = 0; }
    } else {
        /* build h = grad f - zL + zU by tricking bar_terms: use a private assembly */
        StageIt *save = malloc((N + 1)*sizeof(StageIt)); memcpy(save, W->it, (N + 1)*sizeof(StageIt));
        /* zero multipliers -> W = objective Hessian only; we want W = 0 exactly, so use a copy of ev with zero obj Hessian */
        StageEv *evs = malloc((N + 1)*sizeof(StageEv)); memcpy(evs, W->ev, (N + 1)*sizeof(StageEv));
        for (int i = 0; i < N; i++) { memset(W->ev[i].objh, 0, sizeof W->ev[i].objh); memset(W->ev[i].hr, 0, sizeof W->ev[i].hr); W->it[i].lam[0] = W->it[i].lam[1] = 0; for (int r = 0; r < NR; r++) W->it[i].nu[r] = 0; }
        /* Sigma = 1 and gphi = -zL + zU: choose x - lb = 1 style values through direct assembly is awkward; instead
         * emulate: mu_ls = 0, and set z/(slack) = 1 by scaling z temporarily: zL' = (x - lb) * 1 (only one of the two if both exist) */
        for (int i = 0; i <= N; i++) {
            StageBd *B = &W->bd[i]; StageIt *I = &W->it[i];
            for (int k = 0; k < NV; k++) {
                if (!B->on[k]) continue;
                if (B->hasL[k]) { I->zL[k] = (I->x[k] - B->lb[k]); if (B->hasU[k]) I->zU[k] = 0; }
                else if (B->hasU[k]) I->zU[k] = (B->ub[k] - I->x[k]);
            }
            if (i == N) break;
            for (int r = 0; r < NR; r++) {
                if (!W->rowOn[r]) continue;
                if (W->rhasL[r]) { I->zLs[r] = (I->sig[r] - W->dL[r]); if (W->rhasU[r]) I->zUs[r] = 0; }
                else if (W->rhasU[r]) I->zUs[r] = (W->dU[r] - I->sig[r]);
            }
        }
        /* gradient part: with mu = 0 bar_terms gives gphi = 0; add -zL + zU (= -1 + 1 pattern) through objg of a scratch: handled by
         * adding to the direction rhs afterwards is not possible in the condensed form, so fold into ev.objg / a slack gradient array */
        for (int i = 0; i <= N; i++) {
            StageBd *B = &W->bd[i];
            for (int k = 0; k < NV; k++) {
                if (!B->on[k]) continue;
                double gadd = -(B->hasL[k] ? save[i].zL[k] : 0.0) + (B->hasU[k] ? save[i].zU[k] : 0.0);
                int a = var2loc[k];
                if (i < N) W->ev[i].objg[a] += gadd;
                else W->ev[N - 1].objg[k == VT ? LT1 : LB1] += gadd;
            }
        }
        /* slack gradient -zLs + zUs: enters as gphi in the row condensation; emulate through res_d: coef = Sg*res_d + gphi with Sg = 1 */
        for (int i = 0; i < N; i++) { res_c[i][0] = res_c[i][1] = 0; for (int r = 0; r < NR; r++) res_d[i][r] = W->rowOn[r] ? (-(W->rhasL[r] ? save[i].zLs[r] : 0.0) + (W->rhasU[r] ? save[i].zUs[r] : 0.0)) : 0; }
        int ok = compute_direction(W, 0.0, 0.0, (const double (*)[2])res_c, (const double (*)[NR])res_d, W->dir);
        double lmax = 0;
        if (ok) {
            for (int i = 0; i < N; i++) {
                /* dlam = lam+ (lam was zero); nu+ = Sg*dsig + gphi, with the emulation dsig already contains res_d -> nu+ = dsig */
                double lt = W->dir[i].dlam[0], lb = W->dir[i].dlam[1];
                lmax = fmax(lmax, fmax(fabs(lt)/W->sct[i], fabs(lb)/W->scb[i]));
                for (int r = 0; r < NR; r++) if (W->rowOn[r]) lmax = fmax(lmax, fabs(W->dir[i].dsig[r]));
            }
        }
        memcpy(W->ev, evs, (N + 1)*sizeof(StageEv)); free(evs);
        StageDir *Dl = W->dir;
        for (int i = 0; i <= N; i++) {
            double lt = ok ? Dl[i].dlam[0] : 0, lb = ok ? Dl[i].dlam[1] : 0, nus[NR];
            for (int r = 0; r < NR; r++) nus[r] = ok ? Dl[i].dsig[r] : 0;
            W->it[i] = save[i];
            if (i < N) {
                if (ok && lmax <= LAM_INIT_MAX && isfinite(lmax)) { W->it[i].lam[0] = lt; W->it[i].lam[1] = lb; for (int r = 0; r < NR; r++) W->it[i].nu[r] = W->rowOn[r] ? nus[r] : 0; }
                else { W->it[i].lam[0] = W->it[i].lam[1] = 0; for (int r = 0; r < NR; r++) W->it[i].nu[r] = 0; }
            }
        }
        free(save);
        if (dbg) fprintf(stderr, "[oracle] LS multipliers: ok=%d max=%g\n", ok, lmax);
    }

    /* ---- filter initialisation ------------------------------------------------------------------------------------------ */
    double theta, phi; int okp;
    merit_terms(W, W->it, mu, &theta, &phi, &okp);
    W->theta_max = 1e4*fmax(1.0, theta); W->theta_min = 1e-4*fmax(1.0, theta);
    W->nfilt = 0; W->delta_last = 0;

    int status = OR_STATUS_MAXITER, iter = 0, acc_count = 0, tiny_count = 0;
    int short_run = 0, short_max = 0;      /* successive shortened steps (watchdog telemetry) */
    int wd_short = 0, in_wd = 0, wd_trial = 0, wd_started = 0, wd_succeeded = 0, wd_forced_steps = 0;      /* watchdog */
    double wd_theta = 0, wd_phi = 0, wd_gphid = 0, wd_dw = 0;
    StageIt *wd_it = NULL; StageDir *wd_dir = NULL;
    Err R; memset(&R, 0, sizeof R);
    double alpha_pr = 0, alpha_du = 0, dnorm = 0;

    for (iter = 0; ; iter++) {
        for (int i = 0; i < N; i++) eval_interval(W, W->it, i, &W->ev[i], 2);
        kkt_error(W, 0.0, &R);
        if (hist && iter < hist_cap) {
            double *hh = hist + 8*iter;
            hh[0] = iter; hh[1] = objective_value(W, W->it)/W->sf; hh[2] = R.primal; hh[3] = R.dual; hh[4] = log10(mu); hh[5] = dnorm; hh[6] = alpha_du; hh[7] = alpha_pr;
        }
        if (dbg) fprintf(stderr, "[oracle] it %3d obj %.8e inf_pr %.2e inf_du %.2e compl %.2e lg(mu) %5.1f |d| %.2e a_du %.2e a_pr %.2e nf %d\n",
                         iter, objective_value(W, W->it)/W->sf, R.primal, R.dual, R.compl_, log10(mu), dnorm, alpha_du, alpha_pr, W->nfilt);
        /* convergence (scaled tol + unscaled dual_inf_tol = 1, constr_viol_tol = 1e-4, compl_inf_tol = 1e-4) */
        if (R.E <= P->tol && R.dual_u <= 1.0 && R.primal_u <= 1e-4 && R.compl_u <= 1e-4) { status = OR_STATUS_SOLVED; break; }
        if (R.E <= ACC_TOL && R.dual_u <= 1e10 && R.primal_u <= 1e-2 && R.compl_u <= 1e-2) { if (++acc_count >= ACC_ITER) { status = OR_STATUS_ACCEPTABLE; break; } }
        else acc_count = 0;
        if (iter >= P->maxIter) { status = OR_STATUS_MAXITER; break; }

        /* barrier parameter update (monotone Fiacco-McCormick, W&B eq. (7)) */
        {
            Err Rm; kkt_error(W, mu, &Rm);
            double mu_floor = fmin(P->tol, 1e-4)/(K_EPS + 1.0);
            int changed = 0;
            while (Rm.E <= K_EPS*mu && mu > mu_floor) {
                double nm = fmax(mu_floor, fmin(K_MU_LIN*mu, pow(mu, K_MU_SUP)));
                if (nm >= mu) break;
                mu = nm; tau = fmax(K_TAU_MIN, 1 - mu); changed = 1;
                kkt_error(W, mu, &Rm);
            }
            if (changed) { W->nfilt = 0; in_wd = 0; wd_short = 0; }      /* (a new barrier problem: filter and watchdog start afresh) */
        }
        merit_terms(W, W->it, mu, &theta, &phi, &okp);

        /* search direction with inertia correction (W&B Algorithm IC) */
        for (int i = 0; i < N; i++) { res_c[i][0] = W->ev[i].c[0]; res_c[i][1] = W->ev[i].c[1]; for (int r = 0; r < NR; r++) res_d[i][r] = W->rowOn[r] ? W->ev[i].d[r] - W->it[i].sig[r] : 0; }
        double dw = 0; int ok = compute_direction(W, mu, 0.0, (const double (*)[2])res_c, (const double (*)[NR])res_d, W->dir);
        if (!ok) {
            W->n_reg++;
            dw = (W->delta_last == 0) ? DW_0 : fmax(DW_MIN, KW_MINUS*W->delta_last);
            for (;;) {
                ok = compute_direction(W, mu, dw, (const double (*)[2])res_c, (const double (*)[NR])res_d, W->dir);
                if (ok) break;
                dw *= (W->delta_last == 0) ? KW_PLUS_BAR : KW_PLUS;
                if (dw > DW_MAX) break;
            }
            if (!ok) { status = OR_STATUS_REGULARIZATION; break; }
            W->delta_last = dw;
        }
        if (dbg >= 2) fprintf(stderr, "[oracle]    dw %.2e newton residual %.3e\n", dw, direction_residual(W, mu, dw, (const double (*)[2])res_c, (const double (*)[NR])res_d, W->dir));

        /* directional derivative of the barrier function and step norms */
        double gphid, rel_step;
        step_measures(W, mu, W->dir, &gphid, &dnorm, &rel_step);

        double amax = alpha_primal_max(W, W->dir, tau);
        alpha_du = alpha_dual_max(W, W->dir, tau);

        /* tiny step (IPOPT tiny_step_tol = 10 eps): accept the full step without line search */
        int tiny = rel_step < 10*DBL_EPSILON;

        /* watchdog (see above): a tiny step ends a running one -- everything resumes from the stored point with the stored direction */
        if (in_wd && tiny) {
            watchdog_restore(W, wd_it, wd_dir, res_c, res_d);
            in_wd = 0; wd_short = 0; dw = wd_dw; tiny = 0;
            merit_terms(W, W->it, mu, &theta, &phi, &okp);
            step_measures(W, mu, W->dir, &gphid, &dnorm, &rel_step);
            amax = alpha_primal_max(W, W->dir, tau); alpha_du = alpha_dual_max(W, W->dir, tau);
            if (dbg) fprintf(stderr, "[oracle]    watchdog stopped by a tiny step\n");
        }
        if (g_watchdog && P->wdTrigger > 0 && !in_wd && !tiny && wd_short >= P->wdTrigger) {
            if (!wd_it) { wd_it = malloc((N + 1)*sizeof(StageIt)); wd_dir = malloc((N + 1)*sizeof(StageDir)); }
            memcpy(wd_it, W->it, (N + 1)*sizeof(StageIt)); memcpy(wd_dir, W->dir, (N + 1)*sizeof(StageDir));
            wd_theta = theta; wd_phi = phi; wd_gphid = gphid; wd_dw = dw; wd_trial = 0; in_wd = 1; W->n_wd++; wd_started++;
            if (dbg) fprintf(stderr, "[oracle]    watchdog started (theta %.6e phi %.10e)\n", theta, phi);
        }

        double alpha = amax; int accepted = 0; const StageDir *Dacc = W->dir; int ftype_armijo = 0;
        if (tiny) {
            accepted = 1; make_trial(W, W->dir, alpha);
            if (++tiny_count >= 2 && mu <= fmin(P->tol, 1e-4)/(K_EPS + 1.0)*(1 + 1e-12)) { status = OR_STATUS_TINY_STEP; break; }
        } else tiny_count = 0;

        int ls = 0, skip_first = 0, forced = 0;
        double th_ref = theta, ph_ref = phi, gd_ref = gphid;      /* reference point of the acceptance tests: the current one, or the watchdog's */
        for (;;) {
            if (in_wd) { th_ref = wd_theta; ph_ref = wd_phi; gd_ref = wd_gphid; }
            else { th_ref = theta; ph_ref = phi; gd_ref = gphid; }
            /* alpha_min (W&B eq. (23)) */
            double amin;
            if (gd_ref < 0) {
                amin = G_THETA;
                amin = fmin(amin, G_PHI*th_ref/(-gd_ref));
                if (th_ref <= W->theta_min) amin = fmin(amin, K_DELTA*pow(th_ref, S_THETA)/pow(-gd_ref, S_PHI));
            } else amin = G_THETA;
            amin *= ALPHA_MIN_FRAC;

            alpha = skip_first ? 0.5*amax : amax; ls = 0;
            int okt_last = 1;
            while (!accepted) {
                make_trial(W, W->dir, alpha);
                double th_t, ph_t; int okt;
                merit_terms(W, W->trial, mu, &th_t, &ph_t, &okt);
                okt_last = okt;
                int acc = 0;
                int ftype = (gd_ref < 0) && (alpha*pow(-gd_ref, S_PHI) > K_DELTA*pow(th_ref, S_THETA));
                if (okt && th_t <= W->theta_max) {
                    if (ftype && th_ref <= W->theta_min) acc = cmp_le(ph_t - ph_ref, ETA_PHI*alpha*gd_ref, ph_ref);
                    else acc = cmp_le(th_t, (1 - G_THETA)*th_ref, th_ref) || cmp_le(ph_t - ph_ref, -G_PHI*th_ref, ph_ref);
                    if (acc) acc = filter_ok(W, th_t, ph_t);
                }
                if (acc) { accepted = 1; Dacc = W->dir; ftype_armijo = ftype && cmp_le(ph_t - ph_ref, ETA_PHI*alpha*gd_ref, ph_ref); break; }
                if (in_wd) break;      /* only the full step is tried while the watchdog runs */

                /* second-order correction (W&B section 2.4) on the first trial step if infeasibility grew */
                if (ls == 0 && !skip_first && okt && th_t >= th_ref) {
                    double th_old = th_ref, th_prev = th_t; int nsoc = 0;
                    for (int i = 0; i < N; i++) { soc_c[i][0] = W->ev[i].c[0]; soc_c[i][1] = W->ev[i].c[1]; for (int r = 0; r < NR; r++) soc_d[i][r] = res_d[i][r]; }
                    double alpha_soc = alpha;
                    while (nsoc < P_MAX_SOC) {
                        /* c_soc = alpha_soc * c_soc + c(trial) */
                        StageEv e;
                        for (int i = 0; i < N; i++) {
                            eval_interval(W, W->trial, i, &e, 0);
                            soc_c[i][0] = alpha_soc*soc_c[i][0] + e.c[0]; soc_c[i][1] = alpha_soc*soc_c[i][1] + e.c[1];
                            for (int r = 0; r < NR; r++) if (W->rowOn[r]) soc_d[i][r] = alpha_soc*soc_d[i][r] + (e.d[r] - W->trial[i].sig[r]);
                        }
                        if (!compute_direction(W, mu, dw, (const double (*)[2])soc_c, (const double (*)[NR])soc_d, W->soc)) break;
                        alpha_soc = alpha_primal_max(W, W->soc, tau);
                        make_trial(W, W->soc, alpha_soc);
                        double th_s, ph_s; int oks;
                        merit_terms(W, W->trial, mu, &th_s, &ph_s, &oks);
                        nsoc++; W->n_soc++;
                        int accs = 0;
                        if (oks && th_s <= W->theta_max) {
                            if (ftype && th_old <= W->theta_min) accs = cmp_le(ph_s - ph_ref, ETA_PHI*alpha*gd_ref, ph_ref);
                            else accs = cmp_le(th_s, (1 - G_THETA)*th_old, th_old) || cmp_le(ph_s - ph_ref, -G_PHI*th_old, ph_ref);
                            if (accs) accs = filter_ok(W, th_s, ph_s);
                        }
                        if (accs) { accepted = 1; Dacc = W->soc; ftype_armijo = ftype && cmp_le(ph_s - ph_ref, ETA_PHI*alpha*gd_ref, ph_ref); alpha = alpha_soc; break; }
                        if (!oks || th_s > K_SOC*th_prev) break;
                        th_prev = th_s;
                    }
                    if (accepted) break;
                }
                alpha *= 0.5; ls++; W->n_back++;
                if (alpha < amin) break;
            }
            if (!in_wd || tiny) break;
            if (accepted) {
                in_wd = 0; wd_succeeded++;
                if (dbg) fprintf(stderr, "[oracle]    watchdog: trial point accepted against the stored reference\n");
                break;
            }
            wd_trial++;
            if (okt_last && wd_trial <= WD_TRIAL_MAX) { accepted = 1; forced = 1; wd_forced_steps++; Dacc = W->dir; break; }      /* taken although the filter does not accept it */
            /* no success: back to the stored point, ordinary line search on the stored direction from half the maximal step */
            watchdog_restore(W, wd_it, wd_dir, res_c, res_d);
            in_wd = 0; wd_short = 0; dw = wd_dw; skip_first = 1;
            merit_terms(W, W->it, mu, &theta, &phi, &okp);
            step_measures(W, mu, W->dir, &gphid, &dnorm, &rel_step);
            amax = alpha_primal_max(W, W->dir, tau); alpha_du = alpha_dual_max(W, W->dir, tau);
            if (dbg) fprintf(stderr, "[oracle]    watchdog stopped after %d trial iterations: back to the stored point\n", wd_trial);
        }
        if (!accepted) {
            /* the step became too small: feasibility restoration (IpBacktrackingLineSearch: goto_resto).  Not from an almost feasible
             * point (resto_failure_feasibility_threshold = 100 tol); restated for the static loss rows, like the kernels (msd_resto.hpp) */
            /* IPOPT: "Restoration phase called at acceptable point" -- where the line search finds no step from a point that meets the acceptable
             * tolerances (acceptable_tol 1e-6 and its side conditions, without the count of acceptable_iter), the solve ends there with
             * Solved_To_Acceptable_Level (IpBacktrackingLineSearch: ACCEPTABLE_POINT_REACHED; restated from the published implementation like the watchdog).
             * A solve that sits on the rounding floor of its dual infeasibility -- 2e-8 ... 1e-7 against tol = 1e-8 on six-interval re-solves of
             * config 4 -- ends this way instead of breaking down */
            if (R.E <= ACC_TOL && R.dual_u <= 1e10 && R.primal_u <= 1e-2 && R.compl_u <= 1e-2) { status = OR_STATUS_ACCEPTABLE; break; }
            if (!g_resto || R.primal <= 1e2*P->tol) { status = OR_STATUS_LINESEARCH; break; }
            if (W->nfilt < 512) { W->filt_theta[W->nfilt] = (1 - G_THETA)*theta; W->filt_phi[W->nfilt] = phi - G_PHI*theta; W->nfilt++; }
            int nit = 0;
            const int rr = restoration(W, mu, theta, phi, iter, &nit, hist, hist_cap, dbg);
            W->n_resto++;
            if (rr == 1) { iter += nit - 1; acc_count = 0; tiny_count = 0; alpha_pr = alpha_du = dnorm = 0; continue; }
            iter += nit;
            for (int i = 0; i < N; i++) eval_interval(W, W->it, i, &W->ev[i], 2);
            kkt_error(W, 0.0, &R);
            status = rr == -1 ? OR_STATUS_INFEASIBLE : rr == -2 ? OR_STATUS_MAXITER : OR_STATUS_LINESEARCH;
            break;
        }
        alpha_pr = alpha;
        if (!tiny) { short_run = (ls > 0) ? short_run + 1 : 0; if (short_run > short_max) short_max = short_run; }
        if (ls == 0) wd_short = 0;      /* (n_steps == 0: the first trial step was taken) */
        if (ls > 1) wd_short++;         /* (n_steps > 1: a shortened iteration) */

        /* filter augmentation (W&B eq. (22)), with the reference point of the tests */
        if (!tiny && !forced && !ftype_armijo) {
            if (W->nfilt < 512) { W->filt_theta[W->nfilt] = (1 - G_THETA)*th_ref; W->filt_phi[W->nfilt] = ph_ref - G_PHI*th_ref; W->nfilt++; }
        }

        /* accept: primal from the trial point, equality multipliers with the primal step, bound multipliers with the
         * fraction-to-the-boundary step of the direction that was actually taken (the corrected one after a SOC) */
        if (Dacc != W->dir) alpha_du = alpha_dual_max(W, Dacc, tau);
        for (int i = 0; i <= N; i++) {
            StageIt *I = &W->it[i]; StageBd *B = &W->bd[i]; const StageDir *d = &Dacc[i];
            for (int k = 0; k < NV; k++) if (B->on[k]) I->x[k] = W->trial[i].x[k];
            for (int k = 0; k < NV; k++) {
                if (!B->on[k]) continue;
                if (B->hasL[k]) I->zL[k] += alpha_du*d->dzL[k];
                if (B->hasU[k]) I->zU[k] += alpha_du*d->dzU[k];
            }
            if (i == N) break;
            for (int r = 0; r < NR; r++) {
                if (!W->rowOn[r]) continue;
                I->sig[r] = W->trial[i].sig[r];
                I->nu[r] += alpha_pr*d->dnu[r];
                if (W->rhasL[r]) I->zLs[r] += alpha_du*d->dzLs[r];
                if (W->rhasU[r]) I->zUs[r] += alpha_du*d->dzUs[r];
            }
            I->lam[0] += alpha_pr*d->dlam[0]; I->lam[1] += alpha_pr*d->dlam[1];
        }
        /* keep Sigma within [mu/(kappa_Sigma s), kappa_Sigma mu / s] (W&B eq. (16)) */
        for (int i = 0; i <= N; i++) {
            StageIt *I = &W->it[i]; StageBd *B = &W->bd[i];
            for (int k = 0; k < NV; k++) {
                if (!B->on[k]) continue;
                if (B->hasL[k]) { double s = I->x[k] - B->lb[k]; I->zL[k] = fmax(fmin(I->zL[k], K_SIGMA*mu/s), mu/(K_SIGMA*s)); }
                if (B->hasU[k]) { double s = B->ub[k] - I->x[k]; I->zU[k] = fmax(fmin(I->zU[k], K_SIGMA*mu/s), mu/(K_SIGMA*s)); }
            }
            if (i == N) break;
            for (int r = 0; r < NR; r++) {
                if (!W->rowOn[r]) continue;
                if (W->rhasL[r]) { double s = I->sig[r] - W->dL[r]; I->zLs[r] = fmax(fmin(I->zLs[r], K_SIGMA*mu/s), mu/(K_SIGMA*s)); }
                if (W->rhasU[r]) { double s = W->dU[r] - I->sig[r]; I->zUs[r] = fmax(fmin(I->zUs[r], K_SIGMA*mu/s), mu/(K_SIGMA*s)); }
            }
        }
    }

    /* ---- outputs ------------------------------------------------------------------------------------------------------------- */
    int stp = 4 + P->withPn;
    for (int i = 0; i < N; i++) {
        double *zi = z_out + stp*i; int c = 0;
        zi[c++] = W->it[i].x[VF]; if (P->withPn) zi[c++] = W->it[i].x[VP];
        zi[c++] = W->it[i].x[VS]; zi[c++] = W->it[i].x[VT]; zi[c++] = W->it[i].x[VB];
    }
    if (dual_out)
        for (int i = 0; i <= N; i++) {
            double *q = dual_out + OR_DUAL_STRIDE*i; const StageIt *I = &W->it[i];
            q[0] = (i < N) ? I->lam[0] : 0; q[1] = (i < N) ? I->lam[1] : 0;
            for (int r = 0; r < NR; r++) { q[2 + r] = (i < N) ? I->nu[r] : 0; q[17 + r] = (i < N) ? I->zLs[r] : 0; q[22 + r] = (i < N) ? I->zUs[r] : 0; }
            for (int k = 0; k < NV; k++) { q[7 + k] = I->zL[k]; q[12 + k] = I->zU[k]; }
        }
    z_out[stp*N] = W->it[N].x[VT]; z_out[stp*N + 1] = W->it[N].x[VB];
    if (lam_out) {
        int rpi = rows_per_interval(P);
        for (int i = 0; i < N; i++) {
            double *l = lam_out + rpi*i; int c = 0;
            if (P->hasPower) { l[c++] = W->it[i].nu[RPW0]*W->rs[RPW0]/W->sf; l[c++] = W->it[i].nu[RPW1]*W->rs[RPW1]/W->sf; }
            l[c++] = W->it[i].nu[RACC]*W->rs[RACC]/W->sf;
            l[c++] = W->it[i].lam[0]/W->sf; l[c++] = W->it[i].lam[1]/W->sf;
            if (P->energyOpt) { l[c++] = W->it[i].nu[RLTR]*W->rs[RLTR]/W->sf; l[c++] = W->it[i].nu[RLRG]*W->rs[RLRG]/W->sf; }
        }
    }
    if (stats) {
        stats[OR_ST_STATUS] = status; stats[OR_ST_ITERS] = iter; stats[OR_ST_OBJ] = objective_value(W, W->it)/W->sf;
        stats[OR_ST_KKT] = R.E; stats[OR_ST_MU] = mu; stats[OR_ST_DUAL_INF] = R.dual_u; stats[OR_ST_CONSTR_VIOL] = R.primal_u; stats[OR_ST_COMPL] = R.compl_u;
        stats[OR_ST_N_REG] = W->n_reg; stats[OR_ST_N_SOC] = W->n_soc; stats[OR_ST_N_BACKTRACK] = W->n_back; stats[OR_ST_N_RESTO] = W->n_resto; stats[OR_ST_N_WATCHDOG] = W->n_wd;
    }
    note_shortened_run(short_max); note_watchdog(wd_started, wd_succeeded, wd_forced_steps);
    free(wd_it); free(wd_dir);
    free(res_c); free(res_d); free(soc_c); free(soc_d);
    ws_free(W);
    return status;
}

/*
 * Profile start (this build's alternative to the reference's cold start, ocp.py:325-339): a speed profile that respects
 * the limits, accelerates from v0 and brakes to vN with 0.3 m/s^2 and cruises at the speed that uses up the running time;
 * times from the trapezoidal rule; forces from the acceleration the profile needs (with the profile floored at 5 m/s so
 * that the integrator stays away from b = 0), cut to the force and power limits; slacks just above the loss rows.
 * The interior-point iteration then starts like a cold one (mu = 0.1, push 1e-2) but needs about half the iterations.
 */
static void profile_guess(const Prob *P, double *z)
{
    const int N = P->N, nu = 1 + (P->withPn ? 1 : 0), stp = nu + 3;
    const double A = 0.3, MARG = 0.97*0.97, BFL = 25.0, S0 = 0.02, TFR = 0.995;
    double *pos = malloc((N + 1)*sizeof(double)), *b = malloc((N + 1)*sizeof(double)), *v = malloc((N + 1)*sizeof(double));
    pos[0] = 0;
    for (int i = 0; i < N; i++) pos[i + 1] = pos[i] + P->ds[i];
    const double L = pos[N], span = P->tEnd - P->t0;
    double cs = L/span;
    for (int it = 0; it < 3; it++) {
        for (int i = 0; i <= N; i++) {
            const double cap = (i >= 1 && i < N) ? MARG*P->bmax[i] : INFINITY;
            const double up = P->v0sq + 2*A*pos[i], dn = P->vNsq + 2*A*(L - pos[i]);
            double bi = fmin(fmin(cap, cs*cs), fmin(up, dn));
            if (i == 0) bi = P->v0sq;
            if (i == N) bi = P->vNsq;
            b[i] = bi; v[i] = sqrt(bi);
        }
        double tt = 0;
        for (int i = 0; i < N; i++) tt += 2*P->ds[i]/(v[i] + v[i + 1]);
        if (it < 2) cs *= tt/(span*TFR);
    }
    double acc = P->t0;
    for (int i = 0; i <= N; i++) {
        double *q = z + stp*i;
        const double ti = fmin(acc, P->tEnd);
        if (i == N) { q[0] = ti; q[1] = b[i]; break; }
        const double bs0 = fmax(b[i], BFL), bs1 = fmax(b[i + 1], BFL);
        const double vm = 0.5*(sqrt(bs0) + sqrt(bs1));
        const double f = (bs1 - bs0)/(2*P->ds[i]) + P->sr0 + P->sr1*vm + P->sr2*vm*vm + track_resistance(P, P->grad[i], P->curv[i]);
        double fel = fmin(fmax(f, P->fmin), P->fmax);
        if (P->hasPower) { const double vmx = fmax(v[i], v[i + 1]); fel = fmin(fmax(fel, -fabs(P->pwL)/vmx), fabs(P->pwU)/vmx); }
        double fpb = 0.0;
        if (P->withPn) {
            /* the interior push will move Fpb at least this far below its upper bound 0: start there and let Fel make up for it,
             * so that the pushed point still has the acceleration the profile needs (a start from standstill must not stall) */
            const double pb = K_PUSH*fmin(1.0, fabs(P->fminPn));
            fpb = fmin(fmin(fmax(f - fel, P->fminPn), 0.0), -pb);
            fel = fmin(fmax(f - fpb, P->fmin), P->fmax);
            if (P->hasPower) { const double vmx = fmax(v[i], v[i + 1]); fel = fmin(fmax(fel, -fabs(P->pwL)/vmx), fabs(P->pwU)/vmx); }
        }
        double sl;
        if (P->lossKind == 2) { double lr[2][6]; loss_rows(&P->dyn, fel, 0.5*(v[i] + v[i + 1]), lr); sl = fmax(lr[0][0], lr[1][0])*(P->intLosses ? P->ds[i] : 1.0) + S0; }
        else sl = fmax(P->ct*fel, -P->cr*fel)*(P->intLosses ? P->ds[i] : 1.0) + S0;      /* integrateLosses: the slack is an energy per interval */
        q[0] = fel; if (P->withPn) q[1] = fpb;
        q[nu] = sl; q[nu + 1] = ti; q[nu + 2] = b[i];
        acc += 2*P->ds[i]/(v[i] + v[i + 1]);
    }
    free(pos); free(b); free(v);
}

/* one solve from the profile start (nonzero return: failure status) */
static int solve_from_profile(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                              const double *bmax, double *z_out, double *lam_out, double *stats, double *hist, int hist_cap)
{
    Prob P;
    prob_init(&P, ip, dp, ds, grad, curv, bmax);
    const int nz = (4 + (P.withPn ? 1 : 0))*P.N + 2;
    double *guess = malloc(nz*sizeof(double));
    profile_guess(&P, guess);
    int st = solve_core(ip, dp, ds, grad, curv, bmax, guess, K_MU_INIT, K_PUSH, 1, z_out, lam_out, stats, hist, hist_cap, NULL, NULL);
    free(guess);
    return st;
}

/* start: 0 = the reference's cold start, 1 = profile start; a solve that breaks down (not: runs out of iterations) is repeated
 * from the other starting point, like the kernel does (msd_kernel.hpp: solve_kernel) */
int oracle_solve_start(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                       const double *bmax, int start, double *z_out, double *lam_out, double *stats, double *hist, int hist_cap)
{
    int st = (start == 1) ? solve_from_profile(ip, dp, ds, grad, curv, bmax, z_out, lam_out, stats, hist, hist_cap)
                          : oracle_solve(ip, dp, ds, grad, curv, bmax, z_out, lam_out, stats, hist, hist_cap);
    /* (the iteration limit after a restoration phase counts as a breakdown: the phase can leave the iterate where the original iteration only crawls) */
    if (st < 0 && st != OR_STATUS_INFEASIBLE && (st != OR_STATUS_MAXITER || stats[OR_ST_N_RESTO] > 0)) {
        const double spent = stats[OR_ST_ITERS];
        st = (start == 1) ? oracle_solve(ip, dp, ds, grad, curv, bmax, z_out, lam_out, stats, hist, hist_cap)
                          : solve_from_profile(ip, dp, ds, grad, curv, bmax, z_out, lam_out, stats, hist, hist_cap);
        stats[OR_ST_ITERS] += spent;
    }
    return st;
}

int oracle_solve_batch(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                       const double *bmax, int nscen, const double *scen, double *z_out, double *stats, int nthreads)
{
    return oracle_solve_batch_start(ip, dp, ds, grad, curv, bmax, 0, nscen, scen, z_out, stats, nthreads);
}

int oracle_solve_batch_start(const int *ip, const double *dp, const double *ds, const double *grad, const double *curv,
                             const double *bmax, int start, int nscen, const double *scen, double *z_out, double *stats, int nthreads)
{
    int N = ip[OR_IP_N], nz = (4 + ip[OR_IP_WITH_PN])*N + 2, nfail = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+:nfail)
    for (int k = 0; k < nscen; k++) {
        double dpl[OR_DP_COUNT];
        memcpy(dpl, dp, sizeof dpl);
        dpl[OR_DP_T0] = scen[4*k]; dpl[OR_DP_TEND] = scen[4*k + 1]; dpl[OR_DP_V0SQ] = scen[4*k + 2]; dpl[OR_DP_VNSQ] = scen[4*k + 3];
        int st = oracle_solve_start(ip, dpl, ds, grad, curv, bmax, start, z_out + (size_t)nz*k, NULL, stats + (size_t)OR_ST_COUNT*k, NULL, 0);
        if (st < 0) nfail++;
    }
    return nfail;
}
