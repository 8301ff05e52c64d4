/*
 * TEST-ONLY driver: runs msd::solve_kernel<NT> from ms-eetc_amd/csrc/msd_kernel.hpp on host threads
 * (see hip/hip_runtime.h in this directory).  Build: tests/hip_emu/build.sh.  Used by
 * tests/test_kernel_emulation.py to compare the kernel's logic with the oracle without a GPU.
 */
#include "emu_common.h"

thread_local emu_dim3 threadIdx, blockIdx, blockDim, gridDim;
thread_local emu_block *emu_blk;

/* primal-dual warm starts in the emulation: buffers for the next emu_solve_batch* call (dual_in already points at the first node used) */
static const double *g_dual_in = nullptr;
static double *g_dual_out = nullptr;
static long long g_dual_stride = 0;
extern "C" void emu_set_duals(const double *dual_in, long long stride, double *dual_out) { g_dual_in = dual_in; g_dual_stride = stride; g_dual_out = dual_out; }

extern "C" int emu_solve_batch_ex(const msd_problem_desc *d, int nscen, const double *scen, const double *ovr, double *z, double *lam, double *stats, double *hist, int cap);
extern "C" int emu_solve_batch(const msd_problem_desc *d, int nscen, const double *scen, double *z, double *lam, double *stats, double *hist, int cap)
{
    return emu_solve_batch_ex(d, nscen, scen, nullptr, z, lam, stats, hist, cap);
}
extern "C" int emu_solve_batch_warm(const msd_problem_desc *d, int nscen, const double *scen, const double *ovr, const double *guess, double mu0, double push,
                                   double *z, double *lam, double *stats, double *hist, int cap);
extern "C" int emu_solve_batch_ex(const msd_problem_desc *d, int nscen, const double *scen, const double *ovr, double *z, double *lam, double *stats, double *hist, int cap)
{
    return emu_solve_batch_warm(d, nscen, scen, ovr, nullptr, 0.0, 0.0, z, lam, stats, hist, cap);
}
extern "C" int emu_solve_batch_warm(const msd_problem_desc *d, int nscen, const double *scen, const double *ovr, const double *guess, double mu0, double push,
                                   double *z, double *lam, double *stats, double *hist, int cap)
{
    msd::DevProb P;
    P.guess = guess; P.guessStride = (4 + d->with_pn_brake)*d->num_intervals + 2; P.guessStatus = nullptr; P.warmMu = mu0; P.warmPush = push; P.start = d->start_kind; P.lossMass = 0; P.queue = nullptr; P.follow = nullptr; P.list = nullptr; P.socSeen = nullptr; P.dualOut = g_dual_out; P.dualIn = guess ? g_dual_in : nullptr; P.dualInStride = g_dual_stride; P.dualShift = 0;
    std::vector<double> pos(d->num_intervals + 1, 0.0);
    for (int i = 0; i < d->num_intervals; i++) pos[i + 1] = pos[i] + d->ds[i];
    P.pos = pos.data();
    P.N = d->num_intervals; P.withPn = d->with_pn_brake; P.hasPower = d->has_power_rows; P.energyOpt = d->energy_optimal;
    P.numSteps = d->num_steps; P.numApprox = d->num_approx_steps; P.lossKind = d->loss_kind; P.maxIter = d->max_iterations;
    P.sr0 = d->sr0; P.sr1 = d->sr1; P.sr2 = d->sr2; P.g = d->g; P.rho = d->rho; P.fmax = d->f_max; P.fmin = d->f_min; P.fminPn = d->f_min_pn;
    P.pwU = d->pw_upper; P.pwL = d->pw_lower; P.accMin = d->acc_min; P.accMax = d->acc_max; P.ct = d->loss_ct; P.cr = d->loss_cr;
    P.vminSq = d->vmin_sq; P.objDen = d->obj_den; P.tol = d->tol; P.ds = d->ds; P.grad = d->grad; P.curv = d->curv; P.bmax = d->bmax; P.loss = d->loss_table; P.lossCoef = nullptr;
    P.integ = d->integrator; P.collD = d->coll_degree; P.newtonIters = d->newton_iterations; P.intAtol = d->int_abstol; P.intRtol = d->int_reltol; P.coll = d->coll_tables;
    P.resto = d->no_restoration ? 0 : 1;
    P.wdTrigger = d->watchdog_trigger == 0 ? 10 : d->watchdog_trigger;
    P.oneAttempt = 0;
    if (d->integrator == MSD_INTEGRATOR_ADAPTIVE) P.numApprox = 0;
    const bool dyn = d->loss_kind == 2;
    const int nodes = P.N + 1;
    const EmuArgs a = {P, nscen, scen, ovr, z, lam, stats, hist, cap};
    if (d->integrate_losses) {      /* loss slacks from the integrated loss power (msd_lossint.hpp) */
        if (dyn) return (d->integrator == 0 && nodes <= 128 && emu_run_intloss_table(nodes <= 64 ? 64 : 128, 1, a)) ? 0 : -3;      /* the loss table integrated over the running time (msd_lossint_table.hpp) */
        if (d->integrator != 0) return (nodes <= 64 && emu_run_general_intloss(64, 1, a)) ? 0 : -3;      /* both options (msd_kernels_compose.hip) */
        return emu_run_intloss(64, nodes <= 64 ? 1 : 2, a) && nodes <= 128 ? 0 : -3;
    }
    if (d->integrator != 0) {       /* the kernels with the collocation / adaptive shooting integrators: two geometries are enough here */
        if (dyn || nodes > 128) return -3;
        return emu_run_general(64, nodes <= 64 ? 1 : 2, a) ? 0 : -3;
    }
    const char *force = getenv("EMU_GEOMETRY");     /* "NTxSPT" to test other geometries */
    int NT = 0, SPT = 0;
    if (force && !strcmp(force, "stream")) {      /* the long-horizon kernel (stage blocks in memory) at a thread count the emulation can afford */
        if (dyn || nodes > 128*5) return -3;
        {   /* the first pass with the structure of the rolling stock compiled in where the problem has it, like msd_api.hip: make_plan */
            const char *nf = getenv("EMU_NO_FULL");
            const bool st = P.hasPower && P.energyOpt && std::isfinite(P.accMin) && std::isfinite(P.accMax) && std::isfinite(P.pwU) && std::isfinite(P.pwL) && !(nf && *nf == '1');
            return emu_run_stream(a, st ? (P.withPn ? msd::FULL_BOTH : msd::FULL_RG) : 0) ? 0 : -3;
        }
    }
    if (force) sscanf(force, "%dx%d", &NT, &SPT);
    else if (nodes <= 64) { NT = 64; SPT = 1; }
    else if (nodes <= 128) { NT = dyn ? 128 : 64; SPT = dyn ? 1 : 2; }      /* (pick_geometry_t: the loss-table family takes one node per lane here) */
    else if (nodes <= 256) { NT = 128; SPT = 2; }
    else if (nodes <= 384) { NT = 192; SPT = 2; }
    else if (nodes <= 512) { NT = 256; SPT = 2; }
    else if (nodes <= 576 && !dyn) { NT = 192; SPT = 3; }      /* like pick_geometry_t: three nodes per lane on three waves */
    else { NT = 320; SPT = 2; }
    if (NT*SPT < nodes) return -3;
    /* the kernels with the structure of the NLP compiled in (msd_kernels_full.hip), chosen like msd_api.hip does; EMU_NO_FULL=1: the general ones */
    const char *nofull = getenv("EMU_NO_FULL");
    const bool full = !dyn && P.hasPower && P.energyOpt && std::isfinite(P.accMin) && std::isfinite(P.accMax) && std::isfinite(P.pwU) && std::isfinite(P.pwL) && !(nofull && *nofull == '1');
    if (full && emu_run_full(NT, SPT, a, P.withPn ? msd::FULL_BOTH : msd::FULL_RG)) return 0;
    /* the loss-table family with that structure compiled in, chosen like msd_api.hip does (first-pass kernels of msd_kernels_dynamic2.hip / 3.hip) */
    const bool structured = P.hasPower && P.energyOpt && std::isfinite(P.accMin) && std::isfinite(P.accMax) && std::isfinite(P.pwU) && std::isfinite(P.pwL) && !(nofull && *nofull == '1');
    if (dyn && structured && emu_run_dynamic(NT, SPT, a, P.withPn ? msd::FULL_BOTH : msd::FULL_RG)) return 0;
    /* the time-optimal problem on the same rolling stock: first-pass kernels with that structure compiled in (msd_kernels_time.hip) */
    const bool timed = !dyn && !P.energyOpt && P.hasPower && std::isfinite(P.accMin) && std::isfinite(P.accMax) && std::isfinite(P.pwU) && std::isfinite(P.pwL) && !(nofull && *nofull == '1');
    if (timed && emu_run_static(NT, SPT, a, P.withPn ? msd::FULL_TIME_BOTH : msd::FULL_TIME_RG)) return 0;
    return (dyn ? emu_run_dynamic(NT, SPT, a) : emu_run_static(NT, SPT, a)) ? 0 : -3;
}
