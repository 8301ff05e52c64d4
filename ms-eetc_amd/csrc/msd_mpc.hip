/*
 * msd_mpc.hip -- the shrinking-horizon loop of BASELINE config 4 with its bookkeeping on the device (include/mseetc_mpc.h).
 *
 * Reference mechanism: Track.updateLimits(positionStart) (track.py:420-450) + casadiSolver on the cropped track + solve(T, initialTime,
 * initialVelocity) (ocp.py:310); mseetc/mpc.py: shrinkingHorizon is the host-side statement of the loop this file runs without the host.
 * One stream (the energy problem's handle), per re-solve:  scenario records from the measured state -> solve (first pass + follow-up kernel,
 * warm-started from the previous solutions and multipliers where the new grid is a tail of the old one; one attempt per scenario when the loop
 * relaxes infeasible arrival times: WarmStart::one_attempt) -> failed scenarios collected in a list -> their minimum running times from the
 * time-optimal twin (complete kernel launched on the list) -> arrival times of the late ones moved, list re-solved (up to three times with
 * growing margins; what the twin does not declare late is on the first of these lists too, arrival time unchanged, and gets both attempts
 * there) -> log -> measured state for the next re-solve.  Every launch is unconditional; the
 * kernels launched on an empty list return at once.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "msd_handle.hpp"
#include "../../include/mseetc_mpc.h"

using msd_host::fail;

namespace {

constexpr int HDR = msd::FOLLOW_HDR;
constexpr int RING = 64;

__global__ void mpc_begin(int B, const double *T_in, double t0, double v0, double *T, double *tnow, double *vnow, int *flag)
{
    const int s = blockIdx.x*blockDim.x + threadIdx.x;
    if (s >= B) return;
    T[s] = T_in[s]; tnow[s] = t0; vnow[s] = v0; flag[s] = 0;
}

/* scenario records (t0, T, v0^2, vN^2) with the clipping of ocp.py:343-344 (mseetc/ocp.py: _scenarios), log of the measured state */
__global__ void mpc_scen(int B, const double *T, const double *tnow, const double *vnow, double vmin, double vlim0, double vNsq, double *scen, double *log)
{
    const int s = blockIdx.x*blockDim.x + threadIdx.x;
    if (s >= B) return;
    const double v0 = fmin(fmax(vnow[s], vmin), vlim0);
    double *q = scen + (size_t)MSD_SC_COUNT*s;
    q[MSD_SC_T0] = tnow[s]; q[MSD_SC_TEND] = T[s]; q[MSD_SC_V0SQ] = v0*v0; q[MSD_SC_VNSQ] = vNsq;
    double *l = log + (size_t)MSD_MPC_COUNT*s;
    l[MSD_MPC_T0] = tnow[s]; l[MSD_MPC_V0] = vnow[s]; l[MSD_MPC_RELAXED] = 0.0;
}

/* failed re-solves: into the list of the twin's launch, with a running time the twin can certainly meet (mseetc/ocp.py: minimumTime) */
__global__ void mpc_collect(int B, const double *stats, const double *scen, double loose, int *flag, int *list, double *scen_tw)
{
    const int s = blockIdx.x*blockDim.x + threadIdx.x;
    if (s >= B) return;
    const bool bad = stats[(size_t)MSD_ST_COUNT*s + MSD_ST_STATUS] < 0;
    flag[s] = bad ? 1 : 0;
    if (!bad) return;
    const double *q = scen + (size_t)MSD_SC_COUNT*s;
    double *w = scen_tw + (size_t)MSD_SC_COUNT*s;
    w[MSD_SC_T0] = q[MSD_SC_T0]; w[MSD_SC_TEND] = __dadd_rn(q[MSD_SC_T0], fmax(loose, __dmul_rn(3.0, q[MSD_SC_TEND] - q[MSD_SC_T0])));
    w[MSD_SC_V0SQ] = q[MSD_SC_V0SQ]; w[MSD_SC_VNSQ] = q[MSD_SC_VNSQ];
    const int k = atomicAdd(list, 1);
    list[HDR + 2*k] = s; list[HDR + 2*k + 1] = -1;
}

/*
 * attempt 0: a failed scenario whose minimum running time (twin) exceeds what its arrival time leaves is late: the arrival time moves to
 * t0 + tmin (1 + margin) and the scenario is solved again; the others keep their failure.  attempt > 0: a moved scenario that still fails
 * gets the next margin.  (mseetc/mpc.py: shrinkingHorizon, relaxInfeasible)
 */
__global__ void mpc_relax(int B, int attempt, const double *stats, const double *z_tw, int nz_tw, const double *st_tw, double *T, const double *tnow, double *tm,
                          int *flag, int *list, double *scen, double margin, double *log)
{
    const int s = blockIdx.x*blockDim.x + threadIdx.x;
    if (s >= B) return;
    bool again = false;
    if (attempt == 0) {
        if (flag[s] != 1) return;
        /* usable: the twin converged -- or it ended without converging on a feasible point NEXT TO its optimum: far down the central path (mu <= 1e-5) or
         * with an optimality error of 1e-4 (its primal point settles long before its multipliers do where both brakes share an active acceleration bound).
         * Round 6: feasibility alone is not enough.  The twin is given three times the running time asked for and its profile start uses that
         * time up, so a twin that breaks down EARLY sits on a feasible point whose time says nothing about the minimum: the scenario was declared late and
         * its arrival time moved by up to a factor of three although it may be feasible.  Such a scenario now keeps its arrival time and is solved again
         * with both attempts (flag 3).  The log tells the two kinds of verdict apart: 1 converged twin, 2 twin next to its optimum
         * (mseetc/ocp.py: minimumTime restates the rule) */
        const double *stw = st_tw + (size_t)MSD_ST_COUNT*s;
        const double viol = stw[MSD_ST_CONSTR_VIOL];
        const bool conv = stw[MSD_ST_STATUS] >= 0;
        const bool ok = conv || (isfinite(viol) && viol <= 1e-6 && (stw[MSD_ST_MU] <= 1e-5 || stw[MSD_ST_KKT] <= 1e-4));
        const double tmin = z_tw[(size_t)nz_tw*s + nz_tw - 2] - tnow[s];
        if (ok && tmin > T[s] - tnow[s]) { tm[s] = tmin; flag[s] = 2; again = true; log[(size_t)MSD_MPC_COUNT*s + MSD_MPC_RELAXED] = conv ? 1.0 : 2.0; }
        else {
            /* not late: the breakdown was the solver's.  The main launch makes one attempt per scenario (a late scenario would spend its second one, from
             * the other starting point, on a problem without a solution -- nine of 512 per re-solve, a third of the loop's time in round 4); the ones
             * the twin clears get both here, from the problem's own starting point, with their arrival time unchanged */
            flag[s] = 3;
            const int k = atomicAdd(list, 1);
            list[HDR + 2*k] = s; list[HDR + 2*k + 1] = -1;
            return;
        }
    } else {
        if (flag[s] != 2) return;
        if (stats[(size_t)MSD_ST_COUNT*s + MSD_ST_STATUS] < 0) again = true; else flag[s] = 0;
    }
    if (!again) return;
    T[s] = __dadd_rn(tnow[s], __dmul_rn(tm[s], 1 + margin));      /* (no contraction: the host loop's arithmetic, bit for bit) */
    scen[(size_t)MSD_SC_COUNT*s + MSD_SC_TEND] = T[s];
    const int k = atomicAdd(list, 1);
    list[HDR + 2*k] = s; list[HDR + 2*k + 1] = -1;
}

__global__ void mpc_log(int B, const double *stats, const double *T, double *log)
{
    const int s = blockIdx.x*blockDim.x + threadIdx.x;
    if (s >= B) return;
    const double *st = stats + (size_t)MSD_ST_COUNT*s;
    double *l = log + (size_t)MSD_MPC_COUNT*s;
    l[MSD_MPC_T] = T[s]; l[MSD_MPC_STATUS] = st[MSD_ST_STATUS]; l[MSD_MPC_ITERS] = st[MSD_ST_ITERS]; l[MSD_MPC_OBJ] = st[MSD_ST_OBJ];
}

/* the train advances `stride` intervals: time and speed at that node of the solution, perturbed; a failed scenario keeps its last measurement */
__global__ void mpc_advance(int B, const double *z, int nz, const double *stats, int off_t, int off_b, double noise, const double *n1, const double *n2,
                            double *tnow, double *vnow)
{
    const int s = blockIdx.x*blockDim.x + threadIdx.x;
    if (s >= B) return;
    if (stats[(size_t)MSD_ST_COUNT*s + MSD_ST_STATUS] < 0) return;
    const double t = z[(size_t)nz*s + off_t], v = sqrt(z[(size_t)nz*s + off_b]);
    tnow[s] = fmax(__dmul_rn(t, __dadd_rn(1.0, __dmul_rn(noise, n1[s]))), 0.0);      /* (no contraction: the host loop's arithmetic, bit for bit) */
    vnow[s] = __dmul_rn(v, __dadd_rn(1.0, __dmul_rn(noise, n2[s])));
}

}  // namespace

struct msd_mpc {
    msd_problem *h = nullptr, *twin = nullptr;
    int K = 0, stride = 0;
    std::vector<msd_host::Plan> pl, tw;
    std::vector<double> vlim0, length;
    std::vector<unsigned char> tail;
    double vmin = 1, vmax = 1, vN = 1, wmu = 0, wpush = 0, noise = 0, margin = 0;
    int warm = 0, relax = 0;
    double *d_prof = nullptr, *d_work = nullptr;
    double *d_tables = nullptr;       /* the loop's own copies of the loss table and the collocation tables of its problems and of their twins: the handles may be
                                       * reconfigured (which frees and regrows theirs) while the loop exists */
    int *d_ints = nullptr;            /* queue ring | follow | follow (twin) | list A | list B | flags */
    int *d_queue = nullptr, *d_follow = nullptr, *d_follow_tw = nullptr, *d_listA = nullptr, *d_listB = nullptr, *d_flag = nullptr;
    int queue_slot = 0;
    double *d_buf = nullptr;          /* every per-scenario array of a run, carved below */
    double *d_Tin = nullptr, *d_T = nullptr, *d_tnow = nullptr, *d_vnow = nullptr, *d_tm = nullptr, *d_scen = nullptr, *d_scen_tw = nullptr, *d_z[2] = {nullptr, nullptr},
           *d_st[2] = {nullptr, nullptr}, *d_dual[2] = {nullptr, nullptr}, *d_ztw = nullptr, *d_sttw = nullptr, *d_log = nullptr, *d_n1 = nullptr, *d_n2 = nullptr;
    double *d_zlog = nullptr; size_t cap_zlog = 0;
    int cap = 0;
    int nz_max = 0, nz_tw_max = 0, nodes_max = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
};

extern "C" {

int msd_mpc_destroy(msd_mpc_handle m)
{
    if (!m) return MSD_OK;
    if (m->h) { hipSetDevice(m->h->device); hipStreamSynchronize(m->h->stream); m->h->attached_loops--; }
    if (m->twin) m->twin->attached_loops--;
    hipFree(m->d_prof); hipFree(m->d_work); hipFree(m->d_tables); hipFree(m->d_ints); hipFree(m->d_buf); hipFree(m->d_zlog);
    if (m->e0) hipEventDestroy(m->e0);
    if (m->e1) hipEventDestroy(m->e1);
    delete m;
    return MSD_OK;
}

int msd_mpc_create(msd_handle h, msd_handle twin, const msd_mpc_plan *plan, msd_mpc_handle *out)
{
    if (!h || !plan || !out || plan->num_resolves < 1 || plan->stride < 1 || !plan->problems || !plan->vlim_first || !plan->length || !plan->tail)
        return fail(MSD_E_INVALID, "bad argument");
    if ((plan->twins != nullptr) != (twin != nullptr)) return fail(MSD_E_INVALID, "the time-optimal twins need a handle of their own (and only they)");
    if (plan->relax_infeasible && !plan->twins) return fail(MSD_E_INVALID, "relax_infeasible needs the time-optimal twins");
    if (twin && twin->device != h->device) return fail(MSD_E_INVALID, "both handles must live on one device");
    if (plan->warm_start && (!(plan->warm_mu > 0) || !(plan->warm_mu <= 1e3) || !(plan->warm_push > 0) || !(plan->warm_push <= 0.5)))
        return fail(MSD_E_INVALID, "warm start needs 0 < mu_init <= 1e3 and 0 < bound_push <= 0.5");
    HIP_TRY(hipSetDevice(h->device));
    msd_mpc *m = new msd_mpc();
    m->h = h; m->twin = twin; m->K = plan->num_resolves; m->stride = plan->stride;
    h->attached_loops++;      /* (the loop runs on the handle's stream: msd_problem_destroy refuses while a loop is attached) */
    if (twin) twin->attached_loops++;
    m->vmin = plan->vmin; m->vmax = plan->vmax_train; m->vN = plan->terminal_velocity; m->warm = plan->warm_start; m->wmu = plan->warm_mu; m->wpush = plan->warm_push;
    m->noise = plan->noise; m->relax = plan->relax_infeasible; m->margin = plan->late_margin;
    m->pl.resize(m->K); m->tw.resize(twin ? m->K : 0);
    m->vlim0.assign(plan->vlim_first, plan->vlim_first + m->K); m->length.assign(plan->length, plan->length + m->K); m->tail.assign(plan->tail, plan->tail + m->K);
    size_t prof = 0, work = 0;
    int rc = MSD_OK;
    for (int k = 0; k < m->K && rc == MSD_OK; k++) {
        const msd_problem_desc *d = plan->problems + k;
        rc = msd_host::check_desc(d);
        if (rc == MSD_OK && k > 0 && d->num_intervals != plan->problems[k - 1].num_intervals - m->stride) rc = fail(MSD_E_INVALID, "problem k + 1 must have `stride` intervals fewer than problem k");
        if (rc == MSD_OK && k > 0 && (d->loss_kind != plan->problems[0].loss_kind || d->integrator != plan->problems[0].integrator || d->loss_table_len != plan->problems[0].loss_table_len
                                       || (d->integrator == MSD_INTEGRATOR_COLLOCATION && d->coll_degree != plan->problems[0].coll_degree)))
            rc = fail(MSD_E_INVALID, "the problems of a loop share their loss model and integrator");
        if (rc == MSD_OK) rc = msd_host::make_plan(h->device, d, &m->pl[k]);
        if (rc == MSD_OK && twin) {
            const msd_problem_desc *t = plan->twins + k;
            rc = msd_host::check_desc(t);
            if (rc == MSD_OK && t->num_intervals != d->num_intervals) rc = fail(MSD_E_INVALID, "a twin shares the grid of its problem");
            if (rc == MSD_OK && k > 0 && (t->loss_kind != plan->twins[0].loss_kind || t->integrator != plan->twins[0].integrator)) rc = fail(MSD_E_INVALID, "the twins of a loop share their loss model and integrator");
            if (rc == MSD_OK) rc = msd_host::make_plan(h->device, t, &m->tw[k]);
        }
        if (rc != MSD_OK) break;
        prof += 5*(size_t)d->num_intervals + 2;
        const msd_host::Plan &p = m->pl[k];
        work = std::max(work, p.work_doubles*(size_t)std::max(p.max_grid, std::max(p.max_grid2, p.max_grid_lsq)));
        if (twin) { const msd_host::Plan &t = m->tw[k]; work = std::max(work, t.work_doubles*(size_t)std::max(t.max_grid, std::max(t.max_grid2, t.max_grid_lsq))); }
        m->nz_max = std::max(m->nz_max, p.nz); m->nodes_max = std::max(m->nodes_max, d->num_intervals + 1);
        if (twin) m->nz_tw_max = std::max(m->nz_tw_max, m->tw[k].nz);
    }
    if (rc != MSD_OK) { msd_mpc_destroy(m); return rc; }
    /* every profile in one buffer: ds | grad | curv | bmax | pos per re-solve (the twin shares the grid of its problem) */
    std::vector<double> stage(prof);
    if (hipMalloc((void **)&m->d_prof, sizeof(double)*prof) != hipSuccess || hipMalloc((void **)&m->d_work, sizeof(double)*work) != hipSuccess
        || hipEventCreate(&m->e0) != hipSuccess || hipEventCreate(&m->e1) != hipSuccess) { msd_mpc_destroy(m); return fail(MSD_E_HIP, "device allocation failed"); }
    size_t off = 0;
    for (int k = 0; k < m->K; k++) {
        const msd_problem_desc *d = plan->problems + k;
        const int N = d->num_intervals;
        double *ds = stage.data() + off, *grad = ds + N, *curv = grad + N, *bmax = curv + N, *pos = bmax + N + 1;
        std::copy(d->ds, d->ds + N, ds); std::copy(d->grad, d->grad + N, grad); std::copy(d->curv, d->curv + N, curv); std::copy(d->bmax, d->bmax + N + 1, bmax);
        pos[0] = 0;
        for (int i = 0; i < N; i++) pos[i + 1] = pos[i] + d->ds[i];
        for (msd_host::Plan *p : {&m->pl[k], twin ? &m->tw[k] : (msd_host::Plan *)nullptr}) {
            if (!p) continue;
            p->P.ds = m->d_prof + off; p->P.grad = p->P.ds + N; p->P.curv = p->P.grad + N; p->P.bmax = p->P.curv + N; p->P.pos = p->P.bmax + N + 1;
        }
        off += 5*(size_t)N + 2;
    }
    if (hipMemcpy(m->d_prof, stage.data(), sizeof(double)*prof, hipMemcpyHostToDevice) != hipSuccess) { msd_mpc_destroy(m); return fail(MSD_E_HIP, "profile upload failed"); }
    {
        /* loss table and collocation tables: the loop's own copies, from the first problem's (and the first twin's) description -- the problems of a
         * loop share them (checked above).  Rounds 3-4 pointed at the handles' buffers, which msd_problem_reconfigure frees and regrows */
        const msd_problem_desc *d0 = plan->problems, *t0 = twin ? plan->twins : nullptr;
        auto loss_len = [](const msd_problem_desc *d) { return (d && d->loss_kind == 2) ? (size_t)d->loss_table_len : (size_t)0; };
        auto coll_len = [](const msd_problem_desc *d) { return (d && d->integrator == MSD_INTEGRATOR_COLLOCATION) ? (size_t)(d->coll_degree + 1)*(d->coll_degree + 2) : (size_t)0; };
        const size_t n[4] = {loss_len(d0), coll_len(d0), loss_len(t0), coll_len(t0)};
        const double *src[4] = {d0->loss_table, d0->coll_tables, t0 ? t0->loss_table : nullptr, t0 ? t0->coll_tables : nullptr};
        const size_t total = n[0] + n[1] + n[2] + n[3];
        const double *dev[4] = {nullptr, nullptr, nullptr, nullptr};
        if (total) {
            if (hipMalloc((void **)&m->d_tables, sizeof(double)*total) != hipSuccess) { msd_mpc_destroy(m); return fail(MSD_E_HIP, "device allocation failed"); }
            size_t o = 0;
            for (int a = 0; a < 4; a++) {
                if (!n[a]) continue;
                if (hipMemcpy(m->d_tables + o, src[a], sizeof(double)*n[a], hipMemcpyHostToDevice) != hipSuccess) { msd_mpc_destroy(m); return fail(MSD_E_HIP, "table upload failed"); }
                dev[a] = m->d_tables + o; o += n[a];
            }
        }
        for (int k = 0; k < m->K; k++) {
            m->pl[k].P.loss = dev[0]; m->pl[k].P.coll = dev[1];
            if (twin) { m->tw[k].P.loss = dev[2]; m->tw[k].P.coll = dev[3]; }
        }
    }
    *out = m;
    return MSD_OK;
}

int msd_mpc_nz(msd_mpc_handle m, int k) { return (m && k >= 0 && k < m->K) ? m->pl[k].nz : 0; }

static int grow(msd_mpc *m, int B, bool want_z)
{
    if (B > m->cap) {
        hipFree(m->d_buf); hipFree(m->d_ints); m->d_buf = nullptr; m->d_ints = nullptr; m->cap = 0;
        const size_t b = (size_t)B;
        const size_t dual = (size_t)MSD_DUAL_STRIDE*m->nodes_max*b;
        const size_t doubles = 5*b /* Tin T tnow vnow tm */ + 2*MSD_SC_COUNT*b + 2*(size_t)m->nz_max*b + 2*MSD_ST_COUNT*b + 2*dual + (size_t)std::max(m->nz_tw_max, 1)*b + MSD_ST_COUNT*b
                               + (size_t)MSD_MPC_COUNT*m->K*b + 2*(size_t)m->K*b;
        HIP_TRY(hipMalloc((void **)&m->d_buf, sizeof(double)*doubles));
        double *p = m->d_buf;
        m->d_Tin = p; p += b; m->d_T = p; p += b; m->d_tnow = p; p += b; m->d_vnow = p; p += b; m->d_tm = p; p += b;
        m->d_scen = p; p += MSD_SC_COUNT*b; m->d_scen_tw = p; p += MSD_SC_COUNT*b;
        for (int i = 0; i < 2; i++) { m->d_z[i] = p; p += (size_t)m->nz_max*b; }
        for (int i = 0; i < 2; i++) { m->d_st[i] = p; p += MSD_ST_COUNT*b; }
        for (int i = 0; i < 2; i++) { m->d_dual[i] = p; p += dual; }
        m->d_ztw = p; p += (size_t)std::max(m->nz_tw_max, 1)*b; m->d_sttw = p; p += MSD_ST_COUNT*b;
        m->d_log = p; p += (size_t)MSD_MPC_COUNT*m->K*b; m->d_n1 = p; p += (size_t)m->K*b; m->d_n2 = p; p += (size_t)m->K*b;
        const size_t list = HDR + 2*b;
        const size_t ints = RING + 4*list + b;
        HIP_TRY(hipMalloc((void **)&m->d_ints, sizeof(int)*ints));
        HIP_TRY(hipMemset(m->d_ints, 0, sizeof(int)*ints));
        int *q = m->d_ints;
        m->d_queue = q; q += RING; m->d_follow = q; q += list; m->d_follow_tw = q; q += list; m->d_listA = q; q += list; m->d_listB = q; q += list; m->d_flag = q;
        m->cap = B;
    }
    if (want_z) {
        size_t need = 0;
        for (int k = 0; k < m->K; k++) need += (size_t)m->pl[k].nz*B;
        if (need > m->cap_zlog) {
            hipFree(m->d_zlog); m->d_zlog = nullptr; m->cap_zlog = 0;
            HIP_TRY(hipMalloc((void **)&m->d_zlog, sizeof(double)*need));
            m->cap_zlog = need;
        }
    }
    return MSD_OK;
}

int msd_mpc_run(msd_mpc_handle m, int nscen, const double *T, double initial_time, double initial_velocity, const double *n1, const double *n2,
                double *log, double *z_log, float *loop_ms)
{
    if (!m || nscen < 1 || !T || (m->K > 1 && m->noise != 0 && (!n1 || !n2))) return fail(MSD_E_INVALID, "bad argument");
    if (!(initial_time >= 0)) return fail(MSD_E_INVALID, "Initial time must be a positive number!");
    for (int s = 0; s < nscen; s++) if (!(T[s] > 0)) return fail(MSD_E_INVALID, "Terminal time must be a strictly positive number!");
    msd_problem *h = m->h;
    HIP_TRY(hipSetDevice(h->device));
    const int B = nscen, K = m->K;
    int rc = grow(m, B, z_log != nullptr);
    if (rc != MSD_OK) return rc;
    hipStream_t st = h->stream;
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(hipMemcpyAsync(m->d_Tin, T, sizeof(double)*B, hipMemcpyHostToDevice, st));
    if (K > 1 && n1 && n2) {
        HIP_TRY(hipMemcpyAsync(m->d_n1, n1, sizeof(double)*(size_t)(K - 1)*B, hipMemcpyHostToDevice, st));
        HIP_TRY(hipMemcpyAsync(m->d_n2, n2, sizeof(double)*(size_t)(K - 1)*B, hipMemcpyHostToDevice, st));
    } else if (K > 1) {
        HIP_TRY(hipMemsetAsync(m->d_n1, 0, sizeof(double)*(size_t)(K - 1)*B, st));
        HIP_TRY(hipMemsetAsync(m->d_n2, 0, sizeof(double)*(size_t)(K - 1)*B, st));
    }
    const dim3 tb(256), gb((B + 255)/256);
    auto queue = [&]() { int *q = m->d_queue + m->queue_slot; m->queue_slot = (m->queue_slot + 1) % RING; return q; };
    HIP_TRY(hipEventRecord(m->e0, st));
    hipLaunchKernelGGL(mpc_begin, gb, tb, 0, st, B, m->d_Tin, initial_time, initial_velocity, m->d_T, m->d_tnow, m->d_vnow, m->d_flag);
    int cur = 0;
    size_t zoff = 0;
    for (int k = 0; k < K; k++) {
        const msd_host::Plan &pl = m->pl[k];
        const int N = pl.P.N, stp = 4 + pl.P.withPn;
        double *logk = m->d_log + (size_t)MSD_MPC_COUNT*B*k;
        hipLaunchKernelGGL(mpc_scen, gb, tb, 0, st, B, m->d_T, m->d_tnow, m->d_vnow, m->vmin, m->vlim0[k], m->vN*m->vN, m->d_scen, logk);
        msd_host::WarmStart ws;
        const int prev = 1 - cur;
        if (m->warm && k > 0 && m->tail[k]) {
            /* the new grid is the tail of the old one: primal point and multipliers of the previous solutions, `stride` intervals down the horizon */
            const msd_host::Plan &pp = m->pl[k - 1];
            ws.d_guess = m->d_z[prev] + (size_t)stp*m->stride; ws.stride = pp.nz; ws.d_status = m->d_st[prev]; ws.mu = m->wmu; ws.push = m->wpush;
            ws.d_dual_in = m->d_dual[prev]; ws.dual_stride = (long long)MSD_DUAL_STRIDE*(pp.P.N + 1); ws.dual_shift = m->stride;
        }
        ws.d_dual_out = m->warm ? m->d_dual[cur] : nullptr;
        ws.one_attempt = m->relax;      /* (mpc_relax: what the twin does not declare late is solved again with both attempts) */
        ws.use_soc = true;              /* (second-order corrections are the rule in these re-solves: the first-pass kernel that has them, msd_kernel.hpp: SOCK) */
        rc = msd_host::launch_plan(pl, st, m->d_work, m->d_follow, queue(), B, m->d_scen, nullptr, m->d_z[cur], nullptr, m->d_st[cur], nullptr, 0, ws);
        if (rc != MSD_OK) { hipStreamSynchronize(st); return rc; }      /* (nothing of a loop that failed half-way stays queued on the handle's stream) */
        if (m->relax) {
            const msd_host::Plan &tw = m->tw[k];
            hipLaunchKernelGGL(mpc_collect, gb, tb, 0, st, B, m->d_st[cur], m->d_scen, 3*m->length[k]/m->vmax, m->d_flag, m->d_listA, m->d_scen_tw);
            rc = msd_host::launch_plan(tw, st, m->d_work, m->d_follow_tw, queue(), B, m->d_scen_tw, nullptr, m->d_ztw, nullptr, m->d_sttw, nullptr, 0, msd_host::WarmStart(), m->d_listA);
            if (rc != MSD_OK) { hipStreamSynchronize(st); return rc; }
            double margin = m->margin;
            for (int a = 0; a < 3; a++, margin *= 4) {
                hipLaunchKernelGGL(mpc_relax, gb, tb, 0, st, B, a, m->d_st[cur], m->d_ztw, tw.nz, m->d_sttw, m->d_T, m->d_tnow, m->d_tm, m->d_flag, m->d_listB, m->d_scen, margin, logk);
                msd_host::WarmStart cold;
                cold.d_dual_out = ws.d_dual_out;
                cold.use_soc = true;
                rc = msd_host::launch_plan(pl, st, m->d_work, m->d_follow, queue(), B, m->d_scen, nullptr, m->d_z[cur], nullptr, m->d_st[cur], nullptr, 0, cold, m->d_listB);
                if (rc != MSD_OK) { hipStreamSynchronize(st); return rc; }
            }
        }
        hipLaunchKernelGGL(mpc_log, gb, tb, 0, st, B, m->d_st[cur], m->d_T, logk);
        if (z_log) {
            HIP_TRY(hipMemcpyAsync(m->d_zlog + zoff, m->d_z[cur], sizeof(double)*(size_t)pl.nz*B, hipMemcpyDeviceToDevice, st));
            zoff += (size_t)pl.nz*B;
        }
        if (k + 1 < K && N - m->stride >= 1)
            hipLaunchKernelGGL(mpc_advance, gb, tb, 0, st, B, m->d_z[cur], pl.nz, m->d_st[cur], stp*m->stride + 2 + pl.P.withPn, stp*m->stride + 3 + pl.P.withPn, m->noise,
                               m->d_n1 + (size_t)B*k, m->d_n2 + (size_t)B*k, m->d_tnow, m->d_vnow);
        cur = prev;
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(m->e1, st));
    if (log) HIP_TRY(hipMemcpyAsync(log, m->d_log, sizeof(double)*(size_t)MSD_MPC_COUNT*K*B, hipMemcpyDeviceToHost, st));
    if (z_log) HIP_TRY(hipMemcpyAsync(z_log, m->d_zlog, sizeof(double)*zoff, hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (loop_ms) HIP_TRY(hipEventElapsedTime(loop_ms, m->e0, m->e1));
    return MSD_OK;
}

}  // extern "C"
