/*
 * msd_handle.hpp -- host-side internals shared by msd_api.hip and msd_mpc.hip (never included by the kernel units): the handle behind the
 * C ABI, the launch plan of a problem (kernel geometry, resident workgroups, problem record) and the launch of a plan on a stream.
 */
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "msd_kernel.hpp"
#include "msd_geometry.hpp"
#include "../../include/mseetc_aux.h"

namespace msd_host {

int fail(int code, const std::string &msg);      /* records the message of msd_last_error() and returns `code` */

#define HIP_TRY(expr)                                                                                               \
    do {                                                                                                            \
        hipError_t e_ = (expr);                                                                                     \
        if (e_ != hipSuccess) return msd_host::fail(MSD_E_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));  \
    } while (0)

/* what a launch of a problem needs besides buffers: the kernels of its horizon and structure, their limits, the problem record (the profile
 * pointers of P are the owner's to fill) */
struct Plan {
    msd::DevProb P;
    int NT = 0, SPT = 0;
    size_t lds_bytes = 0;
    bool stream = false;                              /* stage blocks in device memory (long horizons) */
    msd::KernelFn kernel = nullptr;                   /* complete kernel, or the first pass of a split solve (kernel2 != nullptr) */
    msd::KernelFn kernel_lsq = nullptr;               /* first pass for launches that need the least-squares multiplier estimate (msd::Geometry::fn_lsq) */
    msd::KernelFn kernel2 = nullptr;                  /* follow-up kernel of a split solve (msd::Geometry::fn2) */
    msd::KernelFn kernel_soc = nullptr;               /* `kernel` with the second-order correction inside the fused iteration (msd::Geometry::fn_soc): WarmStart::use_soc */
    int NT2 = 0, SPT2 = 0;                            /* its own launch geometry (it restarts a scenario from its starting point, so it need not share the first pass's) */
    size_t lds_bytes2 = 0;
    int max_grid = 0, max_grid_lsq = 0, max_grid2 = 0;      /* resident workgroups of the three */
    bool fused_family = false;                        /* `kernel` runs the fused iteration only (needs a profile start or a primal-dual warm start) */
    size_t work_doubles = 0;                          /* work area of one workgroup */
    int nz = 0, nl = 0;                               /* variables / constraint multipliers per scenario */
};

int check_desc(const msd_problem_desc *d);
int make_plan(int device, const msd_problem_desc *d, Plan *out);      /* (check_desc() must have accepted d) */

struct WarmStart { const double *d_guess = nullptr; long long stride = 0; const double *d_status = nullptr; double mu = 0, push = 0;
                   const double *d_dual_in = nullptr; long long dual_stride = 0; int dual_shift = 0; double *d_dual_out = nullptr;
                   bool one_attempt = false;         /* a solve that breaks down is not repeated from the other starting point (msd_mpc.hip: the loop certifies it first) */
                   bool use_soc = false;             /* the first-pass kernel with the second-order correction inside (Plan::kernel_soc), where the plan has one */
                   int *d_soc_seen = nullptr; };     /* DevProb::socSeen of the launch (device address of a mapped host word) */

/*
 * One batch on `stream`: the first pass + the follow-up kernel of a split solve, or the one kernel that holds everything.
 *   d_follow : list between the two kernels of a split solve (FOLLOW_HDR + 2 nscen ints, header zero)
 *   d_queue  : one int (zeroed here) for the dynamic distribution of more scenarios than resident workgroups
 *   d_list   : not null -- only the scenarios of this list (layout of DevProb::follow: [0] count, [1] [2] zero, pairs from FOLLOW_HDR on) are
 *              solved, by the complete kernel (the follow-up kernel of a split solve) from the problem's own starting point; the kernel
 *              leaves the header zeroed
 */
int launch_plan(const Plan &pl, hipStream_t stream, double *d_work, int *d_follow, int *d_queue, int nscen, const double *d_scen, const double *d_ovr,
                double *d_z, double *d_lam, double *d_stats, double *d_hist, int hist_cap, const WarmStart &ws, int *d_list = nullptr,
                hipEvent_t first_begin = nullptr, hipEvent_t first_end = nullptr);      /* (events recorded around the first kernel of the launch) */

}  // namespace msd_host

struct msd_problem {
    msd::DevProb P;
    int device = 0;
    int NT = 0;
    size_t lds_bytes = 0;
    int max_grid = 0;
    msd::KernelFn kernel = nullptr;
    msd::KernelFn kernel_lsq = nullptr;               /* first pass for launches that need the least-squares multiplier estimate (msd::Geometry::fn_lsq) */
    int max_grid_lsq = 0;
    msd::KernelFn kernel2 = nullptr;                  /* follow-up kernel of a split solve (msd::Geometry::fn2), its resident workgroups and the list between the two */
    msd::KernelFn kernel_soc = nullptr;               /* first pass with the second-order correction inside the fused iteration (msd::Geometry::fn_soc; same launch as `kernel`) */
    volatile int *h_soc_seen = nullptr;               /* mapped host word: a launch of the handle handed a second-order correction over (launch() then takes kernel_soc) */
    int *d_soc_seen = nullptr;                        /* its device address */
    int max_grid2 = 0;
    int NT2 = 0, SPT2 = 0; size_t lds_bytes2 = 0;     /* launch geometry of the follow-up kernel */
    int *d_follow = nullptr; size_t cap_follow = 0;
    bool fused_family = false;                        /* `kernel` runs the fused iteration only (needs a profile start or a primal-dual warm start) */
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    double *d_prof = nullptr, *d_loss = nullptr;      /* ds | grad | curv | bmax | pos; loss table */
    double *h_stage = nullptr; size_t cap_stage = 0;  /* pinned staging buffer for the profile upload */
    double *d_work = nullptr;                         /* private work areas of the resident workgroups (msd::work_doubles each) */
    double *d_eval = nullptr; size_t cap_eval = 0;    /* msd_stage_eval: inputs and outputs of n intervals (17 n doubles), grown on demand */
    int *d_queue = nullptr;                           /* scenario counters of the launches (a ring: launches in flight on the stream each own one) */
    int queue_slot = 0;
    size_t cap_work = 0;
    int SPT = 0;
    bool stream_kernel = false;                       /* the problem runs on a streamed kernel (stage blocks in device memory) */
    size_t work_per_wg = 0;                           /* doubles of work area per workgroup */
    int cap_N = 0, cap_loss = 0, cap_nz = 0, cap_nl = 0;
    /* grow-only scratch of the host-buffer entry point */
    double *d_scen = nullptr, *d_ovr = nullptr, *d_z = nullptr, *d_lam = nullptr, *d_stats = nullptr, *d_hist = nullptr, *d_guess = nullptr;
    /* second result buffers: a solve that warm-starts from the previous solve of the handle reads one pair while it writes the other */
    double *d_z2 = nullptr, *d_stats2 = nullptr;
    int prev_nscen = 0, prev_nz = 0, prev_stp = 0;      /* what d_z / d_stats hold (prev_nscen = 0: nothing) */
    /* msd_problem_direct_results: the kernels of a host-buffer call store z* (and the multipliers) in the caller's page-locked arrays themselves */
    bool direct_results = false, last_direct = false;
    /* multipliers of the solves (msd_problem_keep_duals): written to d_dual, read from it by a shifted warm start that writes d_dual2 */
    bool keep_duals = false;
    double *d_dual = nullptr, *d_dual2 = nullptr;
    double *d_coll = nullptr;                           /* tables of the collocation integrator */
    size_t cap_dual = 0;
    int prev_dual_nodes = 0;                            /* nodes per scenario of what d_dual holds (0: nothing) */
    int cap_scen = 0, cap_guess = 0;
    double *h_hist = nullptr;
    int hist_cap = 0;
    /* msd_problem_time_first_pass: events around the first kernel of the last launches (a ring), so that the dominant kernel's own duration can be
     * reported next to the time of a whole launch (first pass + follow-up kernel) */
    static constexpr int FP_RING = 64;
    bool time_first_pass = false;
    hipEvent_t fp_beg[FP_RING] = {}, fp_end[FP_RING] = {};
    long long fp_count = 0;
    int attached_loops = 0;                             /* receding-horizon loops (msd_mpc_create) that run on this handle's stream: it cannot be destroyed while one exists */
};

