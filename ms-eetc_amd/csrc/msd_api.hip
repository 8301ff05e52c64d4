/*
 * msd_api.hip -- host side of the C ABI declared in include/mseetc_hip.h.
 *
 * A handle owns: the problem record with the grid/profile arrays resident in HBM, one HIP stream,
 * two HIP events, and grow-only device buffers for the host-buffer entry point.  Launch geometry:
 * one workgroup of NT = roundup(N + 1, 64) threads per scenario, grid = min(nscen, resident workgroups)
 * (workgroups stride through the batch), dynamic LDS = lds_doubles(N, NT) * 8 bytes.
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "msd_handle.hpp"

namespace {
thread_local std::string g_err;
}

namespace msd {
Tuning &tuning() { static Tuning t; return t; }
}

namespace msd_host {
int fail(int code, const std::string &msg) { g_err = msg; return code; }
}
using msd_host::fail;

namespace msd {
/* (msd_kernels_stream4.hip; declared here and not in msd_geometry.hpp, which every kernel unit depends on) */
Geometry pick_stream_geometry_general_dynamic(int N);
Geometry pick_stream_geometry_general_intloss(int N);
}

namespace msd_host {

/* argument checks shared by create and reconfigure */
int check_desc(const msd_problem_desc *d)
{
    if (d->abi_version != MSD_ABI_VERSION) return fail(MSD_E_INVALID, "ABI version mismatch");
    if (d->num_intervals < 1) return fail(MSD_E_INVALID, "Number of intervals must be a strictly positive integer!");
    if (d->max_iterations < 1) return fail(MSD_E_INVALID, "Maximum number of iterations must be a strictly positive integer!");
    if (d->num_steps < 1 || d->num_approx_steps < 0) return fail(MSD_E_INVALID, "bad integrator options");
    if (!d->ds || !d->grad || !d->curv || !d->bmax) return fail(MSD_E_INVALID, "null profile array");
    if (!(d->vmin_sq > 0) || !(d->obj_den > 0) || !(d->tol > 0)) return fail(MSD_E_INVALID, "vmin_sq, obj_den and tol must be positive");
    if (d->start_kind != MSD_START_REFERENCE && d->start_kind != MSD_START_PROFILE) return fail(MSD_E_INVALID, "unknown starting point");
    if (d->loss_kind < 0 || d->loss_kind > 2) return fail(MSD_E_UNSUPPORTED, "loss model not available on the device");
    if (d->loss_kind == 2) {
        if (!d->loss_table || d->loss_table_len < 13) return fail(MSD_E_INVALID, "dynamic loss model without its table");
        const int nx = (int)d->loss_table[11], ny = (int)d->loss_table[12];
        if (nx < 1 || ny < 1 || d->loss_table_len != 13 + (nx + 1) + (ny + 1) + 16*nx*ny) return fail(MSD_E_INVALID, "inconsistent loss table");
    }
    for (int i = 0; i < d->num_intervals; i++)
        if (!(d->ds[i] > 0)) return fail(MSD_E_INVALID, "interval lengths must be positive");
    if (d->integrator != 0 && d->integrator != MSD_INTEGRATOR_ADAPTIVE && d->integrator != MSD_INTEGRATOR_COLLOCATION)
        return fail(MSD_E_INVALID, "Unknown integration method!");
    if (d->integrator == MSD_INTEGRATOR_COLLOCATION) {
        if (d->coll_degree < 1 || d->coll_degree > 9) return fail(MSD_E_INVALID, "Order of implicit Runge-Kutta should be a positive integer between 1 and 9!");
        if (d->newton_iterations < 1) return fail(MSD_E_INVALID, "Maximum number of iterations must be a strictly positive integer!");
        if (!d->coll_tables) return fail(MSD_E_INVALID, "collocation integrator without its tables");
    }
    if (d->integrator == MSD_INTEGRATOR_ADAPTIVE && (!(d->int_abstol > 0) || !(d->int_reltol > 0))) return fail(MSD_E_INVALID, "tolerances of the adaptive integrator must be positive");
    if (d->integrate_losses && d->energy_optimal && d->loss_kind == 0)
        return fail(MSD_E_UNSUPPORTED, "integrateLosses needs a loss model (loss_kind 1 or 2)");
    if (d->integrate_losses && d->energy_optimal && d->loss_kind == 2 && d->integrator != 0)
        return fail(MSD_E_UNSUPPORTED, "integrateLosses with a loss table runs with explicit Runge-Kutta shooting (integrator 0)");
    return MSD_OK;
}


static int cu_count(int device, int *out)
{
    static std::mutex mu;
    static int cached[64] = {0};
    std::lock_guard<std::mutex> lock(mu);
    if (device < 64 && cached[device] > 0) { *out = cached[device]; return MSD_OK; }
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (device < 64) cached[device] = prop.multiProcessorCount;
    *out = prop.multiProcessorCount;
    return MSD_OK;
}

/*
 * Dynamic-LDS attribute and resident workgroups per compute unit of a kernel: asked of the runtime once per (device, kernel, LDS size)
 * -- a receding-horizon loop reconfigures its handle for every re-solve and the two queries cost more than the rest of it.
 */
static int kernel_limits(int device, const void *fn, int threads, size_t lds, int *per_cu)
{
    struct Info { size_t lds_attr = 0; std::map<size_t, int> occupancy; };
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, Info> cache;
    std::lock_guard<std::mutex> lock(mu);
    Info &info = cache[{device, fn}];
    if (lds > info.lds_attr) {
        HIP_TRY(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        info.lds_attr = lds;
    }
    /* occupancy by LDS size in steps of 8 KB, asked for the upper end of the step (never more workgroups than fit) */
    const size_t step = 8*1024, top = std::min<size_t>(((lds + step - 1)/step)*step, 160*1024);
    auto it = info.occupancy.find(top);
    if (it == info.occupancy.end()) {
        int n = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, fn, threads, top) != hipSuccess || n < 1) n = 1;
        it = info.occupancy.emplace(top, n).first;
    }
    *per_cu = it->second;
    return MSD_OK;
}


/* kernel geometry of a problem's horizon and structure, resident workgroups, problem record without the profile pointers */
int make_plan(int device, const msd_problem_desc *d, Plan *out)
{
    Plan &pl = *out;
    const int N = d->num_intervals;
    const bool dyn = d->loss_kind == 2;
    const bool gen = d->integrator != 0, intloss = d->integrate_losses != 0 && d->energy_optimal != 0;
    const bool wide = dyn || intloss;      /* stage blocks with the slack-b and slack-Fpb couplings */
    const bool itab = dyn && intloss;      /* the loss table integrated over the running time (msd_lossint_table.hpp: DYN = LOSS_INTEGRATED_TABLE) */
    /* both brakes, power rows (finite by construction: ocp.py:186-187), energy objective, finite acceleration bounds (ocp.py:113-114) */
    const bool full = d->with_pn_brake != 0 && d->has_power_rows != 0 && d->energy_optimal != 0 && std::isfinite(d->acc_min) && std::isfinite(d->acc_max)
                      && std::isfinite(d->pw_upper) && std::isfinite(d->pw_lower);
    /* the same without the pneumatic brake (forceMinPn = 0: the reference's scripts); static loss rows + explicit Runge-Kutta shooting only */
    const bool full_rg = d->with_pn_brake == 0 && d->has_power_rows != 0 && d->energy_optimal != 0 && std::isfinite(d->acc_min) && std::isfinite(d->acc_max)
                         && std::isfinite(d->pw_upper) && std::isfinite(d->pw_lower);
    /* the time-optimal problem on the same rolling stock (energyOptimal = False: minimumTime, the twins of msd_mpc.hip): power rows and acceleration row, no loss rows */
    const bool full_time = d->energy_optimal == 0 && d->has_power_rows != 0 && std::isfinite(d->acc_min) && std::isfinite(d->acc_max) && std::isfinite(d->pw_upper) && std::isfinite(d->pw_lower);
    const int structure = full ? msd::FULL_BOTH : full_rg ? msd::FULL_RG : full_time ? (d->with_pn_brake != 0 ? msd::FULL_TIME_BOTH : msd::FULL_TIME_RG) : 0;
    msd::Geometry geo = itab ? msd::pick_geometry_intloss_table(N) : (gen && dyn) ? msd::pick_geometry_general_dynamic(N) : (gen && intloss) ? msd::pick_geometry_general_intloss(N) : gen ? msd::pick_geometry_general(N, full) : intloss ? msd::pick_geometry_intloss(N, full)
                        : dyn ? msd::pick_geometry_dynamic(N, full ? msd::FULL_BOTH : full_rg ? msd::FULL_RG : 0) : msd::pick_geometry_static(N, structure);
    size_t lds = geo.fn ? sizeof(double)*(size_t)(msd::lds_doubles(N, geo.NT*geo.SPT, wide, geo.xch, geo.red) + msd::coop_doubles(geo.NT, gen) + geo.extra) : 0;
    if (!geo.fn || lds > 160*1024) {
        /* the stage blocks do not fit the LDS of a compute unit: the streamed kernels keep them in device memory */
        geo = itab ? msd::pick_stream_geometry_intloss_table(N) : (gen && dyn) ? msd::pick_stream_geometry_general_dynamic(N) : (gen && intloss) ? msd::pick_stream_geometry_general_intloss(N) : gen ? msd::pick_stream_geometry_general(N) : intloss ? msd::pick_stream_geometry_intloss(N)
              : dyn ? msd::pick_stream_geometry_dynamic(N) : msd::pick_stream_geometry_static(N, full ? msd::FULL_BOTH : full_rg ? msd::FULL_RG : 0);
        lds = sizeof(double)*(size_t)msd::lds_doubles_stream();
        if (!geo.fn)
            return fail(MSD_E_UNSUPPORTED, (gen || intloss || dyn) ? "numIntervals = " + std::to_string(N) + " exceeds the 1023 intervals of the streamed kernels for the dynamic loss model, the collocation / adaptive shooting integrators and integrateLosses"
                                           : "numIntervals = " + std::to_string(N) + " exceeds the 5119 intervals of the streamed kernel");
    }
    pl.NT = geo.NT; pl.SPT = geo.SPT; pl.lds_bytes = lds; pl.stream = geo.stream;
    msd::DevProb &P = pl.P;
    P.N = N; P.withPn = d->with_pn_brake != 0; P.hasPower = d->has_power_rows != 0; P.energyOpt = d->energy_optimal != 0;
    P.numSteps = d->num_steps; P.numApprox = d->num_approx_steps; P.lossKind = d->loss_kind; P.maxIter = d->max_iterations;
    P.sr0 = d->sr0; P.sr1 = d->sr1; P.sr2 = d->sr2; P.g = d->g; P.rho = d->rho; P.fmax = d->f_max; P.fmin = d->f_min; P.fminPn = d->f_min_pn;
    P.pwU = d->pw_upper; P.pwL = d->pw_lower; P.accMin = d->acc_min; P.accMax = d->acc_max; P.ct = d->loss_ct; P.cr = d->loss_cr;
    P.vminSq = d->vmin_sq; P.objDen = d->obj_den; P.tol = d->tol;
    P.guess = nullptr; P.guessStride = 0; P.guessStatus = nullptr; P.warmMu = 0; P.warmPush = 0; P.start = d->start_kind; P.lossMass = 0; P.queue = nullptr; P.follow = nullptr; P.list = nullptr; P.socSeen = nullptr; P.dualOut = nullptr; P.dualIn = nullptr; P.dualInStride = 0; P.dualShift = 0;
    P.ds = P.grad = P.curv = P.bmax = P.pos = nullptr;      /* (the owner of the profile buffer fills these) */
    P.loss = nullptr; P.lossCoef = nullptr;
    P.integ = d->integrator; P.collD = d->coll_degree; P.newtonIters = d->newton_iterations; P.intAtol = d->int_abstol; P.intRtol = d->int_reltol;
    P.coll = nullptr;
    P.resto = d->no_restoration ? 0 : 1;
    P.oneAttempt = 0;
    P.wdTrigger = d->watchdog_trigger == 0 ? 10 : d->watchdog_trigger;      /* IPOPT's default */
    if (d->integrator == MSD_INTEGRATOR_ADAPTIVE) P.numApprox = 0;      /* train.py:314 */

    int per_cu = 0, cus = 0;
    int rc = kernel_limits(device, (const void *)geo.fn, geo.NT, lds, &per_cu);
    if (rc != MSD_OK) return rc;
    rc = cu_count(device, &cus);
    if (rc != MSD_OK) return rc;
    pl.max_grid = per_cu*cus;
    pl.max_grid2 = 0;
    pl.NT2 = geo.NT; pl.SPT2 = geo.SPT; pl.lds_bytes2 = lds;
    if (geo.fn2 && geo.NT == 64 && geo.SPT == 1 && geo.xch == msd::XCH_FAST && full) {
        /* horizons of up to 63 intervals, both brakes: the first pass runs one node per lane, the follow-up kernel is the two-nodes-per-lane one (its second
         * node slots stay idle; the follow-up kernel restarts a scenario from its starting point, so nothing ties its geometry to the first pass's -- and the
         * family has no 64 x 1 follow-up instantiation).  The one-brake family has its 64 x 1 follow-up kernel back (round 6: msd_kernels_rg2.hip) */
        geo.fn2 = full ? msd::follow_kernel_full(64, 2) : msd::follow_kernel_full_rg(64, 2);
        pl.NT2 = 64; pl.SPT2 = 2;
        pl.lds_bytes2 = sizeof(double)*(size_t)msd::lds_doubles(N, 128, wide, geo.xch, geo.red);
    }
    size_t work2 = 0;
    if (!geo.fn2 && !geo.stream) {
        /* an LDS-resident kernel without a follow-up kernel of its own geometry (every family but the two with the structure of the reference's rolling
         * stock compiled in): a first-pass kernel -- the general iteration without the restoration phase and the watchdog procedure; a scenario that needs
         * either is followed up by the streamed kernel of the family (msd_kernel.hpp: FAMILY_HAS_RESTO, WD_HANDOVER).  The follow-up restarts the
         * scenario, so the two geometries need not agree */
        const msd::Geometry g2 = itab ? msd::pick_stream_geometry_intloss_table(N) : (gen && dyn) ? msd::pick_stream_geometry_general_dynamic(N) : (gen && intloss) ? msd::pick_stream_geometry_general_intloss(N) : gen ? msd::pick_stream_geometry_general(N)
                                 : intloss ? msd::pick_stream_geometry_intloss(N) : dyn ? msd::pick_stream_geometry_dynamic(N) : msd::pick_stream_geometry_static(N);
        if (!g2.fn2) return fail(MSD_E_UNSUPPORTED, "no follow-up kernel for numIntervals = " + std::to_string(N));
        geo.fn2 = g2.fn2; pl.NT2 = g2.NT; pl.SPT2 = g2.SPT;
        pl.lds_bytes2 = sizeof(double)*(size_t)msd::lds_doubles_stream();
        work2 = msd::stream_doubles(N, g2.NT*g2.SPT, wide);
    }
    if (geo.fn2) {
        rc = kernel_limits(device, (const void *)geo.fn2, pl.NT2, pl.lds_bytes2, &per_cu);
        if (rc != MSD_OK) return rc;
        pl.max_grid2 = per_cu*cus;
    }
    pl.max_grid_lsq = 0;
    if (geo.fn_lsq) {
        rc = kernel_limits(device, (const void *)geo.fn_lsq, geo.NT, lds, &per_cu);
        if (rc != MSD_OK) return rc;
        pl.max_grid_lsq = per_cu*cus;
    }
    pl.fused_family = geo.fn2 != nullptr && geo.xch == msd::XCH_FAST;
    pl.work_doubles = geo.stream ? msd::stream_doubles(N, geo.NT*geo.SPT, wide) : std::max(work2, msd::work_doubles(std::max(geo.NT*geo.SPT, work2 ? 0 : pl.NT2*pl.SPT2)));
    pl.nz = (4 + P.withPn)*N + 2; pl.nl = ((P.hasPower ? 2 : 0) + 3 + (P.energyOpt ? 2 : 0))*N;
    pl.kernel = geo.fn; pl.kernel2 = geo.fn2; pl.kernel_lsq = geo.fn_lsq;
    pl.kernel_soc = nullptr;
    if (geo.fn_soc) {
        /* (same launch as `kernel`; taken only when it is as resident) */
        int per_cu_soc = 0;
        if (kernel_limits(device, (const void *)geo.fn_soc, geo.NT, lds, &per_cu_soc) == MSD_OK && per_cu_soc*cus >= pl.max_grid) pl.kernel_soc = geo.fn_soc;
    }
    return MSD_OK;
}

int launch_plan(const Plan &pl, hipStream_t stream, double *d_work, int *d_follow, int *d_queue, int nscen, const double *d_scen, const double *d_ovr,
                double *d_z, double *d_lam, double *d_stats, double *d_hist, int hist_cap, const WarmStart &ws, int *d_list, hipEvent_t first_begin, hipEvent_t first_end)
{
    if (!pl.kernel) return fail(MSD_E_INVALID, "the handle holds no problem: its last (re)configuration failed");
    msd::DevProb P = pl.P;
    P.follow = nullptr; P.list = nullptr; P.queue = nullptr; P.socSeen = ws.d_soc_seen;
    P.guess = ws.d_guess; P.guessStride = ws.stride; P.guessStatus = ws.d_status; P.warmMu = ws.mu; P.warmPush = ws.push;
    P.dualIn = ws.d_dual_in; P.dualInStride = ws.dual_stride; P.dualShift = ws.dual_shift; P.dualOut = ws.d_dual_out;
    P.oneAttempt = ws.one_attempt ? 1 : 0;
    const bool split = pl.kernel2 != nullptr;
    const bool plain = !pl.fused_family || (ws.d_guess ? ws.d_dual_in != nullptr : P.start == MSD_START_PROFILE);      /* no multiplier estimate needed */
    const bool first_pass = split && (plain || pl.kernel_lsq != nullptr);
    if (d_list && !first_pass) {
        /* the scenarios of a list by one kernel: the follow-up kernel of a fused family whose start needs the least-squares estimate and has no first pass
         * for it, or a kernel that holds everything (an idle workgroup returns at once) */
        const msd::KernelFn fn = split ? pl.kernel2 : pl.kernel;
        const int cap = split ? pl.max_grid2 : pl.max_grid;
        P.list = d_list;
        hipLaunchKernelGGL(fn, dim3(std::min(nscen, cap)), dim3(split ? pl.NT2 : pl.NT), split ? pl.lds_bytes2 : pl.lds_bytes, stream, P, nscen, d_scen, d_ovr, d_z, d_lam, d_stats, d_hist, hist_cap, d_work);
        HIP_TRY(hipGetLastError());
        return MSD_OK;
    }
    P.list = d_list;      /* (not null: the first pass solves the scenarios of this list only -- the re-solves of msd_mpc.hip -- and hands over through `follow` as usual) */
    /* split solves (msd::Geometry::fn2): first pass + follow-up kernel behind it on the stream, the list of unfinished scenarios between them.
     * The first pass is the kernel without the least-squares multiplier estimate when every scenario can start without it (profile start,
     * primal-dual warm start), the one with it otherwise (the reference's starting point, a primal-only warm start) */
    const msd::KernelFn fn = (split && !first_pass) ? pl.kernel2 : plain ? ((ws.use_soc && pl.kernel_soc) ? pl.kernel_soc : pl.kernel) : pl.kernel_lsq;
    const int cap = (split && !first_pass) ? pl.max_grid2 : plain ? pl.max_grid : pl.max_grid_lsq;
    const int grid = nscen < cap ? nscen : cap;
    const int threads = (split && !first_pass) ? pl.NT2 : pl.NT;
    const size_t lds_bytes = (split && !first_pass) ? pl.lds_bytes2 : pl.lds_bytes;
    if (first_pass) {
        if (!d_follow) return fail(MSD_E_INVALID, "split solve without its list");
        P.follow = d_follow;
    }
    if (nscen > grid && !d_list) {
        /* more scenarios than resident workgroups: dynamic distribution through a counter (a list has its own) */
        if (!d_queue) return fail(MSD_E_INVALID, "launch without its scenario counter");
        P.queue = d_queue;
        HIP_TRY(hipMemsetAsync(P.queue, 0, sizeof(int), stream));
    }
    if (first_begin) HIP_TRY(hipEventRecord(first_begin, stream));
    hipLaunchKernelGGL(fn, dim3(grid), dim3(threads), lds_bytes, stream, P, nscen, d_scen, d_ovr, d_z, d_lam, d_stats, d_hist, hist_cap, d_work);
    HIP_TRY(hipGetLastError());
    if (first_end) HIP_TRY(hipEventRecord(first_end, stream));
#ifdef MSD_DEBUG_HOOKS      /* (diagnostic builds: MSD_DEBUG_NO_FOLLOW_UP=1 leaves the list as the first pass wrote it; the product library has no such switch) */
    const bool skip_follow_up = getenv("MSD_DEBUG_NO_FOLLOW_UP") && *getenv("MSD_DEBUG_NO_FOLLOW_UP") == '1';
#else
    constexpr bool skip_follow_up = false;
#endif
    if (first_pass && !skip_follow_up) {
        /* the follow-up kernel: usually nothing to do (0 of the 1024 + 8192 benchmark scenarios of configs 1 and 2) -- a workgroup that finds
         * the list empty returns at once, the others take scenarios off it until it is empty */
        const int grid2 = std::min(nscen, pl.max_grid2);
        P.queue = nullptr; P.list = d_follow; P.follow = nullptr;
        hipLaunchKernelGGL(pl.kernel2, dim3(grid2), dim3(pl.NT2), pl.lds_bytes2, stream, P, nscen, d_scen, d_ovr, d_z, d_lam, d_stats, d_hist, hist_cap, d_work);
        HIP_TRY(hipGetLastError());
    }
    return MSD_OK;
}

}  // namespace msd_host

using msd_host::check_desc;

extern "C" {

const char *msd_last_error(void) { return g_err.c_str(); }

int msd_tuning(const char *name, int value)
{
    if (!name) return fail(MSD_E_INVALID, "null argument");
    if (!strcmp(name, "no_full")) msd::tuning().no_full = value != 0;
    else if (!strcmp(name, "two_nodes_per_lane")) msd::tuning().two_nodes_per_lane = value != 0;
    else return fail(MSD_E_INVALID, std::string("unknown tuning switch: ") + name);
    return MSD_OK;
}

int msd_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

/*
 * Load a problem into a handle: its launch plan (make_plan) and the profile arrays in the handle's device buffer (grown when the horizon or
 * the loss table outgrows it).  Streams, events and the scenario buffers are kept.
 */
static int configure(msd_problem *h, const msd_problem_desc *d)
{
    const int N = d->num_intervals;
    msd_host::Plan pl;
    int rc = msd_host::make_plan(h->device, d, &pl);
    if (rc != MSD_OK) return rc;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));      /* nothing of the previous problem may still be running */
    h->kernel = nullptr;                           /* the handle holds no problem until every step below has succeeded (launch() checks) */

    /* ds | grad | curv | bmax | pos in one buffer */
    if (N > h->cap_N) {
        hipFree(h->d_prof); h->d_prof = nullptr; h->cap_N = 0;
        HIP_TRY(hipMalloc((void **)&h->d_prof, sizeof(double)*(5*(size_t)N + 2)));
        h->cap_N = N;
    }
    {
        /* through a pinned staging buffer of the handle, on the handle's stream: the launches that follow are ordered behind the copy and
         * nothing waits here (a synchronous copy from pageable memory costs most of a millisecond per re-solve of a receding horizon);
         * the stream was synchronised above, so the buffer is free */
        const size_t len = 5*(size_t)N + 2;
        if (len > h->cap_stage) {
            if (h->h_stage) hipHostFree(h->h_stage);
            h->h_stage = nullptr; h->cap_stage = 0;
            HIP_TRY(hipHostMalloc((void **)&h->h_stage, sizeof(double)*len, hipHostMallocDefault));
            h->cap_stage = len;
        }
        double *ds = h->h_stage, *grad = ds + N, *curv = grad + N, *bmax = curv + N, *pos = bmax + N + 1;
        memcpy(ds, d->ds, sizeof(double)*N); memcpy(grad, d->grad, sizeof(double)*N); memcpy(curv, d->curv, sizeof(double)*N);
        memcpy(bmax, d->bmax, sizeof(double)*(N + 1));
        pos[0] = 0;
        for (int i = 0; i < N; i++) pos[i + 1] = pos[i] + d->ds[i];
        HIP_TRY(hipMemcpyAsync(h->d_prof, h->h_stage, sizeof(double)*len, hipMemcpyHostToDevice, h->stream));
    }
    if (d->loss_kind == 2) {
        if (d->loss_table_len > h->cap_loss) {
            hipFree(h->d_loss); h->d_loss = nullptr; h->cap_loss = 0;
            HIP_TRY(hipMalloc((void **)&h->d_loss, sizeof(double)*d->loss_table_len));
            h->cap_loss = d->loss_table_len;
        }
        HIP_TRY(hipMemcpy(h->d_loss, d->loss_table, sizeof(double)*d->loss_table_len, hipMemcpyHostToDevice));
    }

    if (d->integrator == MSD_INTEGRATOR_COLLOCATION) {
        const int len = (d->coll_degree + 1)*(d->coll_degree + 2);
        if (!h->d_coll) HIP_TRY(hipMalloc((void **)&h->d_coll, sizeof(double)*10*11));
        HIP_TRY(hipMemcpy(h->d_coll, d->coll_tables, sizeof(double)*len, hipMemcpyHostToDevice));
    }

    h->NT = pl.NT; h->SPT = pl.SPT; h->lds_bytes = pl.lds_bytes; h->stream_kernel = pl.stream;
    h->P = pl.P;
    msd::DevProb &P = h->P;
    P.ds = h->d_prof; P.grad = P.ds + N; P.curv = P.grad + N; P.bmax = P.curv + N; P.pos = P.bmax + N + 1;
    P.loss = (d->loss_kind == 2) ? h->d_loss : nullptr;
    P.coll = (d->integrator == MSD_INTEGRATOR_COLLOCATION) ? h->d_coll : nullptr;
    h->max_grid = pl.max_grid; h->max_grid2 = pl.max_grid2; h->max_grid_lsq = pl.max_grid_lsq; h->fused_family = pl.fused_family;
    h->NT2 = pl.NT2; h->SPT2 = pl.SPT2; h->lds_bytes2 = pl.lds_bytes2;
    h->work_per_wg = pl.work_doubles;
    {
        const size_t need = pl.work_doubles*(size_t)std::max(h->max_grid, std::max(h->max_grid2, h->max_grid_lsq));
        if (need > h->cap_work) {
            hipFree(h->d_work); h->d_work = nullptr; h->cap_work = 0;
            HIP_TRY(hipMalloc((void **)&h->d_work, sizeof(double)*need));
            h->cap_work = need;
        }
    }
    /* the scenario buffers are sized by nz: a different layout invalidates them */
    const int nz = (4 + P.withPn)*N + 2, nl = ((P.hasPower ? 2 : 0) + 3 + (P.energyOpt ? 2 : 0))*N;
    if (nz > h->cap_nz || nl > h->cap_nl) {
        hipFree(h->d_scen); hipFree(h->d_ovr); hipFree(h->d_z); hipFree(h->d_lam); hipFree(h->d_stats); hipFree(h->d_guess); hipFree(h->d_z2); hipFree(h->d_stats2);
        h->d_scen = h->d_ovr = h->d_z = h->d_lam = h->d_stats = h->d_guess = h->d_z2 = h->d_stats2 = nullptr; h->cap_scen = 0; h->cap_guess = 0;
        h->prev_nscen = 0;
        h->cap_nz = nz; h->cap_nl = nl;
    }
    h->kernel = pl.kernel; h->kernel2 = pl.kernel2; h->kernel_lsq = pl.kernel_lsq; h->kernel_soc = pl.kernel_soc;
    if (h->kernel_soc && !h->h_soc_seen) {
        /* one word of page-locked host memory the kernels can write: "a launch of this handle handed a second-order correction over" */
        HIP_TRY(hipHostMalloc((void **)&h->h_soc_seen, sizeof(int), hipHostMallocMapped));
        HIP_TRY(hipHostGetDevicePointer((void **)&h->d_soc_seen, (void *)h->h_soc_seen, 0));
    }
    if (h->h_soc_seen) *h->h_soc_seen = 0;      /* (another problem: the verdict starts afresh; the stream was synchronised above) */
    return MSD_OK;
}

int msd_problem_create(const msd_problem_desc *d, int device, msd_handle *out)
{
    if (!d || !out) return fail(MSD_E_INVALID, "null argument");
    int rc = check_desc(d);
    if (rc != MSD_OK) return rc;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(MSD_E_NODEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return fail(MSD_E_INVALID, "device index out of range");
    HIP_TRY(hipSetDevice(device));
    msd_problem *h = new msd_problem();
    h->device = device;
    if (hipStreamCreate(&h->stream) != hipSuccess || hipEventCreate(&h->ev0) != hipSuccess || hipEventCreate(&h->ev1) != hipSuccess) {
        msd_problem_destroy(h);
        return fail(MSD_E_HIP, "stream/event creation failed");
    }
    rc = configure(h, d);
    if (rc != MSD_OK) { const std::string keep = g_err; msd_problem_destroy(h); g_err = keep; return rc; }
    *out = h;
    return MSD_OK;
}

int msd_problem_reconfigure(msd_handle h, const msd_problem_desc *d)
{
    if (!h || !d) return fail(MSD_E_INVALID, "null argument");
    int rc = check_desc(d);
    if (rc != MSD_OK) return rc;
    return configure(h, d);
}

int msd_problem_destroy(msd_handle h)
{
    if (!h) return MSD_OK;
    if (h->attached_loops > 0) return fail(MSD_E_INVALID, "a receding-horizon loop (msd_mpc_create) still runs on this handle: destroy the loop first");
    hipSetDevice(h->device);
    hipFree(h->d_prof); hipFree(h->d_loss); hipFree(h->d_work); hipFree(h->d_queue); hipFree(h->d_follow);
    hipFree(h->d_scen); hipFree(h->d_ovr); hipFree(h->d_z); hipFree(h->d_lam); hipFree(h->d_stats); hipFree(h->d_hist); hipFree(h->d_guess); hipFree(h->d_eval);
    hipFree(h->d_z2); hipFree(h->d_stats2); hipFree(h->d_dual); hipFree(h->d_dual2); hipFree(h->d_coll);
    if (h->h_stage) hipHostFree(h->h_stage);
    if (h->h_soc_seen) hipHostFree((void *)h->h_soc_seen);
    for (int k = 0; k < msd_problem::FP_RING; k++) { if (h->fp_beg[k]) hipEventDestroy(h->fp_beg[k]); if (h->fp_end[k]) hipEventDestroy(h->fp_end[k]); }
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->stream) hipStreamDestroy(h->stream);
    delete h;
    return MSD_OK;
}

int msd_problem_nz(msd_handle h) { return h ? (4 + h->P.withPn)*h->P.N + 2 : 0; }
int msd_problem_rows_per_interval(msd_handle h) { return h ? (h->P.hasPower ? 2 : 0) + 3 + (h->P.energyOpt ? 2 : 0) : 0; }

constexpr int QUEUE_RING = 64;

using msd_host::WarmStart;

/* the launch plan a configured handle holds */
static msd_host::Plan plan_of(const msd_problem *h)
{
    msd_host::Plan pl;
    pl.P = h->P; pl.NT = h->NT; pl.SPT = h->SPT; pl.lds_bytes = h->lds_bytes; pl.stream = h->stream_kernel;
    pl.kernel = h->kernel; pl.kernel_lsq = h->kernel_lsq; pl.kernel2 = h->kernel2; pl.kernel_soc = h->kernel_soc; pl.NT2 = h->NT2; pl.SPT2 = h->SPT2; pl.lds_bytes2 = h->lds_bytes2;
    pl.max_grid = h->max_grid; pl.max_grid_lsq = h->max_grid_lsq; pl.max_grid2 = h->max_grid2; pl.fused_family = h->fused_family;
    pl.work_doubles = h->work_per_wg; pl.nz = msd_problem_nz(const_cast<msd_problem *>(h)); pl.nl = msd_problem_rows_per_interval(const_cast<msd_problem *>(h))*h->P.N;
    return pl;
}

static int launch(msd_handle h, int nscen, const double *d_scen, const double *d_ovr, double *d_z, double *d_lam, double *d_stats, double *d_hist, int hist_cap,
                  const WarmStart &ws_in = WarmStart())
{
    if (!h->kernel) return fail(MSD_E_INVALID, "the handle holds no problem: its last (re)configuration failed");
    /* Second-order corrections (IPOPT's default behaviour, ocp.py:290): the first-pass kernel of the benchmark geometries hands a scenario that needs one to
     * the follow-up kernel, whose general iteration solves it again from its starting point -- one such scenario is the tail of its launch (config 3: 1.3 ms
     * behind a 6.3 ms first pass).  A handle whose launches have met one takes the first-pass kernel with the correction inside the fused iteration from
     * then on (msd_kernel.hpp: SOCK; 4 % slower per iteration, no tail); the kernels report it through a word of mapped host memory, read here without
     * waiting for anything */
    WarmStart ws = ws_in;
    if (h->kernel_soc) {
        ws.d_soc_seen = h->d_soc_seen;
        if (*h->h_soc_seen != 0) ws.use_soc = true;
    }
    if (h->kernel2) {
        const size_t need = msd::FOLLOW_HDR + 2*(size_t)nscen;
        if (need > h->cap_follow) {
            HIP_TRY(hipStreamSynchronize(h->stream));      /* (a follow-up kernel in flight reads the old list) */
            /* the telemetry behind the three list counters is documented as never reset: it moves to the new list */
            int keep[msd::FOLLOW_HDR] = {0};
            if (h->d_follow) HIP_TRY(hipMemcpy(keep, h->d_follow, sizeof(int)*msd::FOLLOW_HDR, hipMemcpyDeviceToHost));
            keep[0] = keep[1] = keep[2] = 0;
            hipFree(h->d_follow); h->d_follow = nullptr; h->cap_follow = 0;
            HIP_TRY(hipMalloc((void **)&h->d_follow, sizeof(int)*need));
            HIP_TRY(hipMemcpy(h->d_follow, keep, sizeof(int)*msd::FOLLOW_HDR, hipMemcpyHostToDevice));      /* afterwards the follow-up kernel leaves the three counters zeroed */
            h->cap_follow = need;
        }
    }
    /* one of QUEUE_RING scenario counters, so that launches queued back to back on the stream do not share it */
    if (!h->d_queue) HIP_TRY(hipMalloc((void **)&h->d_queue, sizeof(int)*QUEUE_RING));
    int *queue = h->d_queue + h->queue_slot;
    h->queue_slot = (h->queue_slot + 1) % QUEUE_RING;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (h->time_first_pass) {
        const int k = (int)(h->fp_count % msd_problem::FP_RING);
        if (!h->fp_beg[k]) { HIP_TRY(hipEventCreate(&h->fp_beg[k])); HIP_TRY(hipEventCreate(&h->fp_end[k])); }
        e0 = h->fp_beg[k]; e1 = h->fp_end[k];
        h->fp_count++;
    }
#ifdef MSD_DEBUG_HOOKS      /* (diagnostic builds: the device buffers of a launch, to place the address of a memory fault) */
    if (getenv("MSD_DEBUG_PTRS"))
        fprintf(stderr, "[msd] launch nscen %d  work %p (+%zu B)  follow %p  queue %p  prof %p  scen %p  z %p  lam %p  stats %p  hist %p  NT %d SPT %d NT2 %d SPT2 %d lds %zu lds2 %zu grid caps %d %d %d\n",
                nscen, (void *)h->d_work, sizeof(double)*h->cap_work, (void *)h->d_follow, (void *)queue, (void *)h->d_prof, (const void *)d_scen, (void *)d_z, (void *)d_lam, (void *)d_stats, (void *)d_hist,
                h->NT, h->SPT, h->NT2, h->SPT2, h->lds_bytes, h->lds_bytes2, h->max_grid, h->max_grid_lsq, h->max_grid2);
#endif
    return msd_host::launch_plan(plan_of(h), h->stream, h->d_work, h->d_follow, queue, nscen, d_scen, d_ovr, d_z, d_lam, d_stats, d_hist, hist_cap, ws, nullptr, e0, e1);
}

int msd_solve_batch_device(msd_handle h, int nscen, const double *d_scen, double *d_z, double *d_lam, double *d_stats)
{
    return msd_solve_batch_device_ex(h, nscen, d_scen, nullptr, d_z, d_lam, d_stats);
}

int msd_solve_batch_device_ex(msd_handle h, int nscen, const double *d_scen, const double *d_overrides, double *d_z, double *d_lam, double *d_stats)
{
    if (!h || nscen < 1 || !d_scen || !d_z || !d_stats) return fail(MSD_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    return launch(h, nscen, d_scen, d_overrides, d_z, d_lam, d_stats, nullptr, 0);
}

int msd_problem_geometry(msd_handle h, int *threads_per_scenario, int *nodes_per_thread)
{
    if (!h || !threads_per_scenario || !nodes_per_thread) return fail(MSD_E_INVALID, "bad argument");
    *threads_per_scenario = h->NT; *nodes_per_thread = h->SPT;
    return MSD_OK;
}

int msd_problem_follow_counts(msd_handle h, int *counts, int n)
{
    if (!h || !counts || n < 1) return fail(MSD_E_INVALID, "bad argument");
#ifdef MSD_DEBUG_HOOKS
    if (n > 8 && !(getenv("MSD_DEBUG_NO_FOLLOW_UP") && *getenv("MSD_DEBUG_NO_FOLLOW_UP") == '1')) n = 8;      /* (diagnostic builds: the list's entries behind the counters) */
#else
    if (n > 8) n = 8;
#endif
    if (n > (int)h->cap_follow - msd::FOLLOW_TOTAL && h->d_follow) n = (int)h->cap_follow - msd::FOLLOW_TOTAL;
    for (int k = 0; k < n; k++) counts[k] = 0;
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->d_follow) HIP_TRY(hipMemcpy(counts, h->d_follow + msd::FOLLOW_TOTAL, sizeof(int)*n, hipMemcpyDeviceToHost));
    return MSD_OK;
}

int msd_problem_time_first_pass(msd_handle h, int on)
{
    if (!h) return fail(MSD_E_INVALID, "null handle");
    h->time_first_pass = on != 0; h->fp_count = 0;
    return MSD_OK;
}

int msd_problem_first_pass_ms(msd_handle h, float *mean_ms, int *launches)
{
    if (!h || !mean_ms || !launches) return fail(MSD_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    const int n = (int)std::min<long long>(h->fp_count, msd_problem::FP_RING);
    double sum = 0;
    for (int k = 0; k < n; k++) { float ms = 0; HIP_TRY(hipEventElapsedTime(&ms, h->fp_beg[k], h->fp_end[k])); sum += ms; }
    *mean_ms = n ? (float)(sum/n) : 0.0f; *launches = n;
    return MSD_OK;
}

int msd_synchronize(msd_handle h)
{
    if (!h) return fail(MSD_E_INVALID, "null handle");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MSD_OK;
}

int msd_set_history(msd_handle h, double *host_hist, int cap)
{
    if (!h) return fail(MSD_E_INVALID, "null handle");
    h->h_hist = host_hist; h->hist_cap = host_hist ? cap : 0;
    return MSD_OK;
}

int msd_solve_batch(msd_handle h, int nscen, const double *scen, double *z_out, double *lam_out, double *stats, float *kernel_ms)
{
    return msd_solve_batch_ex(h, nscen, scen, nullptr, z_out, lam_out, stats, kernel_ms);
}

int msd_solve_batch_ex(msd_handle h, int nscen, const double *scen, const double *overrides, double *z_out, double *lam_out, double *stats, float *kernel_ms)
{
    return msd_solve_batch_warm(h, nscen, scen, overrides, nullptr, 0.0, 0.0, z_out, lam_out, stats, kernel_ms);
}

/* argument checks of the host-buffer entry points */
static int check_batch(msd_handle h, int nscen, const double *scen, const double *overrides, const double *z_guess, double mu_init, double bound_push,
                       const double *z_out, const double *stats)
{
    if (!h || nscen < 1 || !scen || !z_out || !stats) return fail(MSD_E_INVALID, "bad argument");
    if (z_guess && (!(mu_init > 0) || !(mu_init <= 1e3) || !(bound_push > 0) || !(bound_push <= 0.5)))
        return fail(MSD_E_INVALID, "warm start needs 0 < mu_init <= 1e3 and 0 < bound_push <= 0.5");
    if (overrides)
        for (int k = 0; k < nscen; k++) {
            const double *o = overrides + (size_t)MSD_OV_COUNT*k;
            if (!(o[MSD_OV_OBJ_DEN] > 0) || !(o[MSD_OV_F_MAX] > o[MSD_OV_F_MIN]) || !(o[MSD_OV_SR0] >= 0) || !(o[MSD_OV_SR1] >= 0) || !(o[MSD_OV_SR2] >= 0) ||
                !(o[MSD_OV_TOTAL_MASS] >= 0))
                return fail(MSD_E_INVALID, "invalid rolling-stock override");
            /* the kernels with the row structure compiled in take both power rows as two-sided (finite by construction in the problem record: ocp.py:186-187) */
            if (h->fused_family && (!std::isfinite(o[MSD_OV_PW_UPPER]) || !std::isfinite(o[MSD_OV_PW_LOWER])))
                return fail(MSD_E_INVALID, "rolling-stock override with an infinite power bound on a problem whose power rows are bounded");
        }
    for (int k = 0; k < nscen; k++) {
        const double *s = scen + (size_t)MSD_SC_COUNT*k;
        if (!(s[MSD_SC_T0] >= 0)) return fail(MSD_E_INVALID, "Initial time must be a positive number!");
        if (!(s[MSD_SC_TEND] > 0)) return fail(MSD_E_INVALID, "Terminal time must be a strictly positive number!");
        if (!(s[MSD_SC_V0SQ] > 0) || !(s[MSD_SC_VNSQ] > 0)) return fail(MSD_E_INVALID, "velocities must be positive");
    }
    if (z_guess) {
        const size_t nz = msd_problem_nz(h);
        for (size_t k = 0; k < nz*(size_t)nscen; k++)
            if (!std::isfinite(z_guess[k])) return fail(MSD_E_INVALID, "warm start guess must be finite");
    }
    return MSD_OK;
}

/* the address under which the device reaches a host array, or null: page-locked memory (msd_host_alloc, hipHostMalloc, hipHostRegister) has one */
static double *device_address(const double *p)
{
    hipPointerAttribute_t a;
    if (!p || hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return nullptr; }      /* (pageable memory: an error, cleared) */
    if (a.type != hipMemoryTypeHost || !a.devicePointer) return nullptr;
    return (double *)a.devicePointer;
}

/* the copies of a batch's results into the caller's arrays, on the handle's stream (last_direct: the kernels have written z* and the multipliers there) */
static int enqueue_downloads(msd_handle h, int nscen, double *z_out, double *lam_out, double *stats)
{
    HIP_TRY(hipSetDevice(h->device));
    const size_t nz = msd_problem_nz(h), nl = (size_t)msd_problem_rows_per_interval(h)*h->P.N;
    if (!h->last_direct) {
        HIP_TRY(hipMemcpyAsync(z_out, h->d_z, sizeof(double)*nz*nscen, hipMemcpyDeviceToHost, h->stream));
        if (lam_out) HIP_TRY(hipMemcpyAsync(lam_out, h->d_lam, sizeof(double)*nl*nscen, hipMemcpyDeviceToHost, h->stream));
    }
    HIP_TRY(hipMemcpyAsync(stats, h->d_stats, sizeof(double)*MSD_ST_COUNT*nscen, hipMemcpyDeviceToHost, h->stream));
    if (h->d_hist && h->h_hist && h->hist_cap > 0)
        HIP_TRY(hipMemcpyAsync(h->h_hist, h->d_hist, sizeof(double)*msd::HIST_COLS*h->hist_cap, hipMemcpyDeviceToHost, h->stream));
    return MSD_OK;
}

/* uploads, launch and (unless deferred) downloads of one batch on the handle's stream, nothing waited for: finish_batch() completes it.
 * defer_downloads: the caller issues enqueue_downloads() itself -- a copy into pageable host memory holds the calling thread until the
 * kernel in front of it has finished, so a caller that drives several devices launches on all of them first */
static int enqueue_batch(msd_handle h, int nscen, const double *scen, const double *overrides, const double *z_guess, double mu_init, double bound_push,
                         double *z_out, double *lam_out, double *stats, int shift = -1, bool defer_downloads = false)
{
    HIP_TRY(hipSetDevice(h->device));
    const size_t nz = msd_problem_nz(h), nl = (size_t)msd_problem_rows_per_interval(h)*h->P.N;
    if (nscen > h->cap_scen) {
        hipFree(h->d_scen); hipFree(h->d_ovr); hipFree(h->d_z); hipFree(h->d_lam); hipFree(h->d_stats); hipFree(h->d_z2); hipFree(h->d_stats2);
        h->d_scen = h->d_ovr = h->d_z = h->d_lam = h->d_stats = h->d_z2 = h->d_stats2 = nullptr; h->cap_scen = 0; h->prev_nscen = 0;
        HIP_TRY(hipMalloc((void **)&h->d_z2, sizeof(double)*(size_t)h->cap_nz*nscen));
        HIP_TRY(hipMalloc((void **)&h->d_stats2, sizeof(double)*MSD_ST_COUNT*nscen));
        HIP_TRY(hipMalloc((void **)&h->d_scen, sizeof(double)*MSD_SC_COUNT*nscen));
        HIP_TRY(hipMalloc((void **)&h->d_ovr, sizeof(double)*MSD_OV_COUNT*nscen));
        /* sized for the largest layout this handle has been configured for (msd_problem_reconfigure) */
        HIP_TRY(hipMalloc((void **)&h->d_z, sizeof(double)*(size_t)h->cap_nz*nscen));
        HIP_TRY(hipMalloc((void **)&h->d_lam, sizeof(double)*(size_t)h->cap_nl*nscen));
        HIP_TRY(hipMalloc((void **)&h->d_stats, sizeof(double)*MSD_ST_COUNT*nscen));
        h->cap_scen = nscen;
    }
    double *d_hist = nullptr;
    if (h->h_hist && h->hist_cap > 0) {
        hipFree(h->d_hist); h->d_hist = nullptr;
        HIP_TRY(hipMalloc((void **)&h->d_hist, sizeof(double)*msd::HIST_COLS*h->hist_cap));
        HIP_TRY(hipMemsetAsync(h->d_hist, 0, sizeof(double)*msd::HIST_COLS*h->hist_cap, h->stream));
        d_hist = h->d_hist;
    }
    WarmStart ws;
    ws.stride = (long long)nz;
    if (shift >= 0) {
        /* guesses = the previous solutions of this handle, `shift` intervals down the horizon: a tail of each stored z */
        const int stp = 4 + h->P.withPn;
        if (h->prev_nscen == 0 && h->last_direct)
            return fail(MSD_E_INVALID, "the previous solve of this handle stored its results in host memory directly (msd_problem_direct_results): the device holds no copy a shifted warm start could begin from");
        if (h->prev_nscen != nscen || h->prev_stp != stp || h->prev_nz - stp*shift != (int)nz)
            return fail(MSD_E_INVALID, "no previous solve of this handle matches the shifted warm start (same batch, horizon longer by `shift` intervals)");
        ws.d_guess = h->d_z + (size_t)stp*shift; ws.stride = h->prev_nz; ws.d_status = h->d_stats; ws.mu = mu_init; ws.push = bound_push;
        std::swap(h->d_z, h->d_z2); std::swap(h->d_stats, h->d_stats2);       /* results go to the other pair */
        if (h->keep_duals && h->prev_dual_nodes == h->P.N + 1 + shift) {
            ws.d_dual_in = h->d_dual; ws.dual_stride = (long long)MSD_DUAL_STRIDE*h->prev_dual_nodes; ws.dual_shift = shift;
            std::swap(h->d_dual, h->d_dual2);
        }
    }
    if (h->keep_duals) {
        const size_t need = (size_t)MSD_DUAL_STRIDE*(h->P.N + 1)*nscen;
        if (need > h->cap_dual) {
            if (ws.d_dual_in) return fail(MSD_E_INVALID, "the multiplier buffers cannot grow between a solve and its shifted re-solve");
            hipFree(h->d_dual); hipFree(h->d_dual2); h->d_dual = h->d_dual2 = nullptr; h->cap_dual = 0; h->prev_dual_nodes = 0;
            HIP_TRY(hipMalloc((void **)&h->d_dual, sizeof(double)*need));
            HIP_TRY(hipMalloc((void **)&h->d_dual2, sizeof(double)*need));
            h->cap_dual = need;
        }
        ws.d_dual_out = h->d_dual;
    }
    if (z_guess) {
        if (nscen > h->cap_guess) {
            hipFree(h->d_guess); h->d_guess = nullptr; h->cap_guess = 0;
            HIP_TRY(hipMalloc((void **)&h->d_guess, sizeof(double)*(size_t)h->cap_nz*nscen));
            h->cap_guess = nscen;
        }
        HIP_TRY(hipMemcpyAsync(h->d_guess, z_guess, sizeof(double)*nz*nscen, hipMemcpyHostToDevice, h->stream));
        ws.d_guess = h->d_guess; ws.mu = mu_init; ws.push = bound_push;
    }
    HIP_TRY(hipMemcpyAsync(h->d_scen, scen, sizeof(double)*MSD_SC_COUNT*nscen, hipMemcpyHostToDevice, h->stream));
    if (overrides) HIP_TRY(hipMemcpyAsync(h->d_ovr, overrides, sizeof(double)*MSD_OV_COUNT*nscen, hipMemcpyHostToDevice, h->stream));
    /* msd_problem_direct_results: result arrays the device can address are written by the kernels themselves -- every workgroup stores its z* when its
     * scenario is done, spread over the launch, instead of one copy behind the last of them (config 1: 0.99 x the device-resident rate instead of 0.89 x,
     * tools/zero_copy_probe.py).  The status records stay on the device (the follow-up kernel reads them) and are copied */
    double *kz = h->d_z, *kl = lam_out ? h->d_lam : nullptr;
    h->last_direct = false;
    if (h->direct_results && shift < 0 && !h->keep_duals) {      /* (keep_duals: a shifted re-solve is to follow, the solutions stay with the multipliers) */
        double *dz = device_address(z_out), *dl = lam_out ? device_address(lam_out) : nullptr;
        if (dz && (!lam_out || dl)) { kz = dz; kl = dl; h->last_direct = true; }
    }
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    int rc = launch(h, nscen, h->d_scen, overrides ? h->d_ovr : nullptr, kz, kl, h->d_stats, d_hist, h->hist_cap, ws);
    if (rc != MSD_OK) return rc;
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    if (!d_hist && h->d_hist) { hipFree(h->d_hist); h->d_hist = nullptr; }      /* (enqueue_downloads copies a history only when this launch wrote one) */
    h->prev_nscen = h->last_direct ? 0 : nscen; h->prev_nz = (int)nz; h->prev_stp = 4 + h->P.withPn;
    h->prev_dual_nodes = h->keep_duals ? h->P.N + 1 : 0;
    (void)nl;
    if (!defer_downloads) return enqueue_downloads(h, nscen, z_out, lam_out, stats);
    return MSD_OK;
}

static int finish_batch(msd_handle h, float *kernel_ms)
{
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (kernel_ms) HIP_TRY(hipEventElapsedTime(kernel_ms, h->ev0, h->ev1));
    return MSD_OK;
}

int msd_solve_batch_warm(msd_handle h, int nscen, const double *scen, const double *overrides, const double *z_guess, double mu_init, double bound_push,
                         double *z_out, double *lam_out, double *stats, float *kernel_ms)
{
    int rc = check_batch(h, nscen, scen, overrides, z_guess, mu_init, bound_push, z_out, stats);
    if (rc != MSD_OK) return rc;
    rc = enqueue_batch(h, nscen, scen, overrides, z_guess, mu_init, bound_push, z_out, lam_out, stats);
    if (rc != MSD_OK) return rc;
    return finish_batch(h, kernel_ms);
}

int msd_problem_direct_results(msd_handle h, int on)
{
    if (!h) return fail(MSD_E_INVALID, "null handle");
    h->direct_results = on != 0;
    return MSD_OK;
}

int msd_problem_keep_duals(msd_handle h, int on)
{
    if (!h) return fail(MSD_E_INVALID, "null handle");
    h->keep_duals = on != 0;
    if (!h->keep_duals) h->prev_dual_nodes = 0;
    return MSD_OK;
}

int msd_solve_batch_shifted(msd_handle h, int nscen, const double *scen, const double *overrides, int shift_intervals, double mu_init, double bound_push,
                            double *z_out, double *lam_out, double *stats, float *kernel_ms)
{
    if (shift_intervals < 0) return fail(MSD_E_INVALID, "shift_intervals must not be negative");
    int rc = check_batch(h, nscen, scen, overrides, nullptr, mu_init, bound_push, z_out, stats);
    if (rc != MSD_OK) return rc;
    if (!(mu_init > 0) || !(mu_init <= 1e3) || !(bound_push > 0) || !(bound_push <= 0.5))
        return fail(MSD_E_INVALID, "warm start needs 0 < mu_init <= 1e3 and 0 < bound_push <= 0.5");
    rc = enqueue_batch(h, nscen, scen, overrides, nullptr, mu_init, bound_push, z_out, lam_out, stats, shift_intervals);
    if (rc != MSD_OK) return rc;
    return finish_batch(h, kernel_ms);
}

/*
 * One batch over several handles (one per device, SURVEY 8b/8e: single process, one stream per device): handle k solves the
 * contiguous slice [k nscen / n, (k + 1) nscen / n) of the scenarios; all slices are enqueued before any is waited for.
 */
int msd_solve_batch_multi(const msd_handle *handles, int nhandles, int nscen, const double *scen, const double *overrides, const double *z_guess,
                          double mu_init, double bound_push, double *z_out, double *lam_out, double *stats, float *kernel_ms)
{
    if (!handles || nhandles < 1 || nscen < 1) return fail(MSD_E_INVALID, "bad argument");
    for (int k = 0; k < nhandles; k++) {
        if (!handles[k]) return fail(MSD_E_INVALID, "null handle");
        if (msd_problem_nz(handles[k]) != msd_problem_nz(handles[0]) || handles[k]->P.N != handles[0]->P.N ||
            msd_problem_rows_per_interval(handles[k]) != msd_problem_rows_per_interval(handles[0]))
            return fail(MSD_E_INVALID, "the handles of a multi-device solve must hold the same problem");
        for (int j = 0; j < k; j++)
            if (handles[j] == handles[k]) return fail(MSD_E_INVALID, "a handle appears twice");
    }
    int rc = check_batch(handles[0], nscen, scen, overrides, z_guess, mu_init, bound_push, z_out, stats);
    if (rc != MSD_OK) return rc;
    const size_t nz = msd_problem_nz(handles[0]), nl = (size_t)msd_problem_rows_per_interval(handles[0])*handles[0]->P.N;
    std::vector<int> lo(nhandles + 1);
    for (int k = 0; k <= nhandles; k++) lo[k] = (int)(((long long)nscen*k)/nhandles);
    int first_error = MSD_OK;
    std::string first_msg;
    std::vector<char> started(nhandles, 0);
    for (int k = 0; k < nhandles && first_error == MSD_OK; k++) {
        const int n = lo[k + 1] - lo[k];
        if (n < 1) continue;
        rc = enqueue_batch(handles[k], n, scen + (size_t)MSD_SC_COUNT*lo[k], overrides ? overrides + (size_t)MSD_OV_COUNT*lo[k] : nullptr,
                           z_guess ? z_guess + nz*lo[k] : nullptr, mu_init, bound_push, z_out + nz*lo[k], lam_out ? lam_out + nl*lo[k] : nullptr,
                           stats + (size_t)MSD_ST_COUNT*lo[k], -1, true);
        if (rc != MSD_OK) { first_error = rc; first_msg = g_err; } else started[k] = 1;
    }
    /* every device is running: now the result copies (each one waits for its own device only) */
    for (int k = 0; k < nhandles && first_error == MSD_OK; k++) {
        if (!started[k]) continue;
        rc = enqueue_downloads(handles[k], lo[k + 1] - lo[k], z_out + nz*lo[k], lam_out ? lam_out + nl*lo[k] : nullptr, stats + (size_t)MSD_ST_COUNT*lo[k]);
        if (rc != MSD_OK) { first_error = rc; first_msg = g_err; }
    }
    float worst = 0;
    for (int k = 0; k < nhandles; k++) {
        if (!started[k]) continue;
        float ms = 0;
        rc = finish_batch(handles[k], &ms);
        if (rc != MSD_OK && first_error == MSD_OK) { first_error = rc; first_msg = g_err; }
        if (ms > worst) worst = ms;
    }
    if (first_error != MSD_OK) { g_err = first_msg; return first_error; }
    if (kernel_ms) *kernel_ms = worst;
    return MSD_OK;
}

int msd_host_alloc(unsigned long long bytes, void **ptr)
{
    if (!ptr || bytes == 0) return fail(MSD_E_INVALID, "bad argument");
    HIP_TRY(hipHostMalloc(ptr, bytes, hipHostMallocDefault));
    return MSD_OK;
}
int msd_host_free(void *ptr)
{
    if (ptr) HIP_TRY(hipHostFree(ptr));
    return MSD_OK;
}

int msd_device_alloc(msd_handle h, unsigned long long bytes, void **dptr)
{
    if (!h || !dptr) return fail(MSD_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMalloc(dptr, bytes));
    return MSD_OK;
}
int msd_device_free(msd_handle h, void *dptr)
{
    if (!h) return fail(MSD_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipFree(dptr));
    return MSD_OK;
}
int msd_copy_to_device(msd_handle h, void *dst, const void *src, unsigned long long bytes)
{
    if (!h) return fail(MSD_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MSD_OK;
}
int msd_copy_to_host(msd_handle h, void *dst, const void *src, unsigned long long bytes)
{
    if (!h) return fail(MSD_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return MSD_OK;
}
int msd_timer_begin(msd_handle h)
{
    if (!h) return fail(MSD_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventRecord(h->ev0, h->stream));
    return MSD_OK;
}
int msd_timer_end(msd_handle h, float *ms)
{
    if (!h || !ms) return fail(MSD_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    HIP_TRY(hipEventRecord(h->ev1, h->stream));
    HIP_TRY(hipEventSynchronize(h->ev1));
    HIP_TRY(hipEventElapsedTime(ms, h->ev0, h->ev1));
    return MSD_OK;
}

int msd_stage_eval(msd_handle h, int n, const double *b, const double *w, const double *ds, const double *grad, const double *curv, double *out12)
{
    if (!h || n < 1 || !b || !w || !ds || !grad || !curv || !out12) return fail(MSD_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(h->device));
    if ((size_t)17*n > h->cap_eval) {      /* kept with the handle: no allocation per call */
        hipFree(h->d_eval); h->d_eval = nullptr; h->cap_eval = 0;
        HIP_TRY(hipMalloc((void **)&h->d_eval, sizeof(double)*17*(size_t)n));
        h->cap_eval = (size_t)17*n;
    }
    double *d_in = h->d_eval, *d_out = h->d_eval + 5*(size_t)n;
    const double *src[5] = {b, w, ds, grad, curv};
    for (int k = 0; k < 5; k++) hipMemcpyAsync(d_in + (size_t)k*n, src[k], sizeof(double)*n, hipMemcpyHostToDevice, h->stream);
    hipLaunchKernelGGL(msd::stage_eval_kernel<0>, dim3((n + 255)/256), dim3(256), 0, h->stream, h->P, n, d_in, d_in + n, d_in + 2*(size_t)n, d_in + 3*(size_t)n,
                       d_in + 4*(size_t)n, d_out);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(out12, d_out, sizeof(double)*12*n, hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
    if (e != hipSuccess) return fail(MSD_E_HIP, hipGetErrorString(e));
    return MSD_OK;
}

}  // extern "C"
