"""
Benchmark of the hot path: full solves of a batch of independent train-control OCPs on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the HIP solver over one batch of synthetic scenarios that is already resident in HBM.
Workload = BASELINE.json configs[1]: B = 1024 scenarios per GPU, N = 100 shooting
intervals, track 00_var_speed_limit_100, train NL_Intercity_VIRM6 with the JSON defaults (both brakes), RK4 with
numSteps = 1 and trapezoidal time (numApproxSteps = 1), v0 = vN = 1 m/s, T_i = 1541 (1 + 0.15 u_i),
u = default_rng(20260612 + rank).random(B).

Multi-GPU: one process per GPU over torch.distributed (RCCL); scenarios are independent, so the ranks share nothing but the
barrier and the max over ranks (weak scaling, no collective on the data path).  `python bench.py --gpus N` with N > 1 and no
RANK in the environment starts the N ranks itself (torch.distributed.run as a child process, before anything touches the GPU);
under an external launcher (RANK set) it is one of the ranks.

Prints ONE JSON line on rank 0.  `roofline` prices the solve kernel with the streaming model S of SURVEY.md section 8d
(904 B per stage-iteration); `cpu_baseline` times the CPU oracle (a port, not the reference's CasADi/IPOPT, which cannot run
here) on the host cores and on one thread; `alt` (N = 1 only) carries the other workloads of SURVEY 8(d): the reference's
starting point on config 1, config 2, config 3 and config 4 at their per-GPU sizes.
"""

import argparse
import json
import os
import socket
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent

for p in (str(ROOT / 'ms-eetc_amd'), str(ROOT)):
    if p not in sys.path:
        sys.path.insert(0, p)

BYTES_PER_STAGE_ITER = 904.0   # SURVEY.md section 8d, streaming model S (nu = 2): 113 doubles
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)
FP64_VALU_PEAK_TFLOPS = 78.6   # vector double precision = half of the 157.3 TFLOPS FP32 vector spec of MI355X_MICROARCH.md (~78 TF/s); secondary bound


def usable_cores():
    "Cores this process may actually use: the affinity mask capped by the cgroup CPU quota (a 256-thread box may grant 16)."

    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)

    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]                      # cgroup v2
        if quota != 'max':
            n = min(n, max(1, int(float(quota)/float(period))))
    except Exception:
        try:
            quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())                      # cgroup v1
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0:
                n = min(n, max(1, quota//period))
        except Exception:
            pass

    return n


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=0, help='scenarios per GPU (default: the workload\'s per-GPU size)')
    ap.add_argument('--intervals', type=int, default=0, help='shooting intervals (default: the workload\'s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt', action='store_true', help='skip the extra workloads (reference start, configs 2-4)')
    ap.add_argument('--no-build', action='store_true', help='never compile (profiled runs): exit non-zero when the library is stale')
    ap.add_argument('--start', default='profile', choices=['profile', 'reference'],
                    help="starting point of every solve: 'profile' (library default, built on the device from the scenario) or 'reference' (cold start of ocp.py:325-339)")
    ap.add_argument('--workload', default='c1', choices=['c1', 'c2', 'c3'],
                    help='c1: BASELINE configs[1] (the metric); c2: N=200 on CH_StGallen_Wil; c3: config 1 with per-scenario rolling stock')
    return ap.parse_args(argv)


def free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def rank_command(args_list, gpus, port):
    "The child process that runs the ranks of a multi-GPU benchmark (torch.distributed.run, one process per GPU)."

    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(gpus), '--master-addr', '127.0.0.1',
            '--master-port', str(port), str(Path(__file__).resolve())] + list(args_list)


def launch_ranks(args):
    "Parent of a multi-GPU run: start the ranks as a child process and return its exit code.  Never touches the GPU."

    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    return subprocess.call(rank_command(sys.argv[1:], args.gpus, free_port()), env=env)


PER_GPU_BATCH = dict(c1=1024, c2=8192, c3=8192)


def measure(solver, scen, overrides, steps, warmup, barrier=None):
    "Timed region: `steps` launches over a batch resident in HBM.  Returns (elapsed s, kernel ms per launch, stats array)."

    import numpy as np
    from mseetc._device import ST, OV

    prob = solver.problem
    B = scen.shape[0]
    d_scen = prob.alloc(scen.nbytes)
    d_z = prob.alloc(8*prob.nz*B)
    d_st = prob.alloc(8*ST['COUNT']*B)
    prob.to_device(d_scen, scen)
    d_ov = None
    if overrides is not None:
        overrides = np.ascontiguousarray(overrides, dtype=np.float64).reshape(B, OV['COUNT'])
        d_ov = prob.alloc(overrides.nbytes)
        prob.to_device(d_ov, overrides)

    for _ in range(warmup):
        prob.solve_batch_device(B, d_scen, d_z, None, d_st, d_overrides=d_ov)
    prob.synchronize()

    if barrier:
        barrier()
    t0 = time.perf_counter()
    prob.timer_begin()                      # HIP events on the stream the kernel is launched on
    for _ in range(steps):
        prob.solve_batch_device(B, d_scen, d_z, None, d_st, d_overrides=d_ov)
    kernel_ms_total = prob.timer_end()      # waits for the last kernel
    prob.synchronize()
    if barrier:
        barrier(sync_only=True)
    elapsed = time.perf_counter() - t0
    if barrier:
        barrier()

    st = np.zeros((B, ST['COUNT']))
    prob.to_host(st, d_st)
    for d in (d_scen, d_z, d_st, d_ov):
        if d is not None:
            prob.free(d)
    return elapsed, kernel_ms_total/steps, st


def build_workload(name, B, N, rank, start, device):
    "(solver, scenarios (B,4), overrides or None, description)"

    from mseetc import workloads as wl
    from mseetc.ocp import casadiSolver

    train, track, N0 = wl.config(name)
    N = N or N0
    solver = casadiSolver(train, track, wl.options(N), device=device, startingPoint=start)
    overrides = None

    if name == 'c1':
        T = wl.c1_times(B, seed=20260612 + rank)
        text = "config 1: B={} scenarios per GPU, N={}, track 00_var_speed_limit_100, train NL_Intercity_VIRM6 (JSON defaults, both brakes), RK4 numSteps=1 numApproxSteps=1, v0=vN=1, T_i=1541(1+0.15u_i) seed 20260612+rank".format(B, N)
    elif name == 'c2':
        T = wl.c2_times(B, seed=20260613 + rank)
        text = "config 2: B={} scenarios per GPU, N={}, track CH_StGallen_Wil, train NL_Intercity_VIRM6, RK4 numSteps=1 numApproxSteps=1, v0=vN=1, T_i=1242(1+0.15u_i) seed 20260613+rank".format(B, N)
    else:
        T, pert = wl.c3_scenarios(B, train, seed=20260614 + rank)
        overrides = solver._overrides(B, pert['mass'], pert['r0'], pert['r1'], pert['r2'])
        text = "config 3: B={} scenarios per GPU, N={}, as config 1 (seed 20260614+rank) with mass, r0, r1, r2 perturbed per scenario by 5 % (clipped normal)".format(B, N)

    return solver, solver._scenarios(T, 0, 1, 1), overrides, text


def summarize(B, N, steps, elapsed, launch_ms, st):
    import numpy as np
    from mseetc._device import ST
    iters = st[:, ST['ITERS']]
    return {"solves_per_s": B*steps/elapsed, "launch_ms": launch_ms, "batch": B, "num_intervals": N, "converged": int(np.sum(st[:, ST['STATUS']] >= 0)),
            "ip_iterations_mean": float(np.mean(iters)), "ip_iterations_max": float(np.max(iters)),
            "kkt_fallbacks": int(np.sum(st[:, ST['N_FALLBACK']]))}


def hbm_traffic(entry):
    "Measured HBM bytes per launch of the headline kernel, if the committed measurement belongs to the library that is running."

    tf = ROOT / 'profiles' / 'hbm_traffic.json'
    if not tf.exists():
        return None, "no profiles/hbm_traffic.json", None
    try:
        rec = json.loads(tf.read_text())
    except Exception:
        return None, "unreadable profiles/hbm_traffic.json", None
    if rec.get('kernel_digest') != entry.hip_digest():
        return None, "profiles/hbm_traffic.json was measured on another build of the kernel (digest mismatch): re-run tools/profile_round.sh", None
    return rec.get('bytes_per_launch'), rec.get('source'), rec.get('issue')


def main():

    args = parse_args()

    # build (or check) the native library before anything touches the GPU: hipcc must never run in a process that has initialised it
    import __graft_entry__ as entry

    child = 'RANK' in os.environ

    if args.no_build:
        if entry.stale() and not os.environ.get('MSD_LIB'):       # MSD_LIB: a tuning build (tools/build_variant.py) is being measured
            raise SystemExit("bench.py --no-build: ms-eetc_amd/lib/libmseetc_hip.so is missing or stale; run `python3 __graft_entry__.py` first")
    else:
        entry.build()      # every rank may call it: an exclusive file lock serialises the ranks, all but the first find the library fresh

    import torch      # importing torch and counting devices does not initialise the GPU

    ndev = torch.cuda.device_count()

    # test aid for boxes with fewer GPUs than ranks: MSD_BENCH_SHARE_DEVICES=1 maps rank r to device r % (devices visible) and uses the gloo
    # backend for the barrier and the reductions (RCCL refuses two ranks on one device); the measured path is the same
    share = os.environ.get('MSD_BENCH_SHARE_DEVICES') == '1'

    if ndev < max(1, args.gpus) and not child and not (share and ndev >= 1):
        print("bench.py: {} HIP device(s) visible, {} requested -- nothing to measure here (the solver has no CPU fallback)".format(ndev, args.gpus), file=sys.stderr)
        return 0

    if args.gpus > 1 and not child:
        return launch_ranks(args)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))

    if world != args.gpus:
        raise SystemExit("--gpus {} but WORLD_SIZE {}".format(args.gpus, world))

    import numpy as np

    if share:
        local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    red_dev = 'cpu' if share else 'cuda'

    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if share:
            dist.init_process_group('gloo', rank=rank, world_size=world)
        else:
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    def barrier(sync_only=False):
        torch.cuda.synchronize()
        if world > 1 and not sync_only:
            dist.barrier()
            torch.cuda.synchronize()

    from mseetc._device import ST

    B = args.batch or PER_GPU_BATCH[args.workload]
    solver, scen, overrides, text = build_workload(args.workload, B, args.intervals, rank, args.start, local_rank)
    N = solver.numIntervals

    elapsed, launch_ms, st = measure(solver, scen, overrides, args.steps, args.warmup, barrier)

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_ok = int(np.sum(st[:, ST['STATUS']] >= 0))
    iters = st[:, ST['ITERS']]

    if world > 1:
        ok = torch.tensor([n_ok], dtype=torch.int64, device=red_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.SUM)
        n_ok_all = int(ok.item())
    else:
        n_ok_all = n_ok

    if rank == 0:

        total_solves = B*world*args.steps
        value = total_solves/elapsed

        stage_iters = float(N*np.sum(iters))                 # units one launch processes (this rank)
        achieved = BYTES_PER_STAGE_ITER*stage_iters/(launch_ms*1e-3)/1e9
        traffic, traffic_source, issue = hbm_traffic(entry)
        geo = solver.problem.geometry()

        start_text = ("every solve starts from the device-built speed profile (no information from earlier solves; same optimum as the reference's cold start)"
                      if args.start == 'profile' else "every solve cold-starts from the reference's point (ocp.py:325-339)")

        line = {
            "metric": "OCP solves/sec (N=100, VIRM6, var-speed-limit track)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3*elapsed/args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": text + ", " + start_text + ", KKT<=1e-8",
                       "batch_per_gpu": B, "num_intervals": N, "start": args.start, "converged": n_ok_all, "scenarios": B*world,
                       "kkt_fallbacks": int(np.sum(st[:, ST['N_FALLBACK']])), "cycles_per_solve_mean": float(np.mean(st[:, ST['CYC_TOTAL']])),
                       "ip_iterations_mean": float(np.mean(iters)), "ip_iterations_max": float(np.max(iters)), "parallelism": "scenarios sharded, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved/HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": traffic_source,
                         "model": "S = 904 B x N x sum(IP iterations) per launch (SURVEY 8d); iterate is LDS/register resident, so real HBM traffic is far below S",
                         "kernel": "msd::solve_kernel<{},{}> (one workgroup of {} threads per scenario, {} shooting nodes per lane)".format(geo[0], geo[1], geo[0], geo[1]),
                         "launch_ms": launch_ms, "stage_iterations_per_launch": stage_iters,
                         # what bounds the kernel in practice (SQ counters of the profiling pass, same digest rule as `traffic`): a latency-bound
                         # double-precision instruction stream, one wave per SIMD
                         "issue": issue, "fp64_valu_peak_tflops": FP64_VALU_PEAK_TFLOPS,
                         "fp64_valu_frac_model": (400.0*stage_iters/(launch_ms*1e-3)/1e12)/FP64_VALU_PEAK_TFLOPS},
        }

        solver.close()

        if world == 1 and not args.no_alt and args.workload == 'c1':
            line["alt"] = alt_workloads(args, local_rank)

        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(solver, scen, args.start, overrides is not None)

        print(json.dumps(line), flush=True)

    if world > 1:
        dist.destroy_process_group()

    return 0


def alt_workloads(args, device):
    "The other workloads of SURVEY 8(d) at their per-GPU sizes, a few launches each (N = 1 runs only)."

    import numpy as np
    from mseetc import workloads as wl
    from mseetc.mpc import shrinkingHorizon

    alt = {}
    k = max(3, args.steps//4)
    solver, scen, ov, text = build_workload('c1', PER_GPU_BATCH['c1'], 0, 0, 'reference', device)
    e, ms, st = measure(solver, scen, ov, k, 1)
    alt["reference_start"] = dict(summarize(scen.shape[0], solver.numIntervals, k, e, ms, st), workload=text + ", cold start of ocp.py:325-339")
    solver.close()

    solver, scen, ov, text = build_workload('c1', 8192, 0, 0, 'profile', device)
    e, ms, st = measure(solver, scen, ov, 3, 1)
    alt["c1_batch8192"] = dict(summarize(scen.shape[0], solver.numIntervals, 3, e, ms, st), workload=text)
    solver.close()

    for name in ('c2', 'c3'):
        solver, scen, ov, text = build_workload(name, PER_GPU_BATCH[name], 0, 0, 'profile', device)
        e, ms, st = measure(solver, scen, ov, 3, 1)
        alt[name] = dict(summarize(scen.shape[0], solver.numIntervals, 3, e, ms, st), workload=text)
        solver.close()

    # the other transcriptions of the reference's options on the config-1 batch (their own kernel instantiations, not tuned)
    from mseetc.ocp import casadiSolver
    train, track, N = wl.config('c1')
    T = wl.c1_times(PER_GPU_BATCH['c1'], seed=20260612)
    for name, extra, io in (("integrate_losses", dict(integrateLosses=True), dict(numSteps=1, numApproxSteps=1)),
                            ("irk_radau2", dict(integrationMethod='IRK'), dict(order=2, numSteps=1, numApproxSteps=1)),
                            ("cvodes_tolerances", dict(integrationMethod='CVODES'), dict())):
        solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationOptions=io, **extra), device=device, startingPoint='profile')
        scen = solver._scenarios(T, 0, 1, 1)
        e, ms, st = measure(solver, scen, None, 3, 1)
        alt[name] = dict(summarize(scen.shape[0], N, 3, e, ms, st), workload="config 1 batch with {} {}".format(extra, io))
        solver.close()

    # config 4: shrinking-horizon MPC, 512 scenarios per GPU (4096 over 8), 50 re-solves each: wall time of the whole loop
    train, track, N = wl.config('c4')
    T = wl.c1_times(512, seed=20260615)
    shrinkingHorizon(train, track, wl.options(N), T[:64], numResolves=2, noise=0.01, seed=1, device=device)      # warm up
    c4 = {}
    for warm in (False, True):
        t0 = time.perf_counter()
        log = shrinkingHorizon(train, track, wl.options(N), T, numResolves=50, noise=0.01, seed=1, warmStart=warm, device=device)
        wall = time.perf_counter() - t0
        n = sum(len(l['status']) for l in log)
        c4["warm" if warm else "cold"] = {"resolves_per_s": n/wall, "wall_s": wall, "resolves": len(log), "scenarios": 512,
                                          "ip_iterations_mean": float(np.mean([l['iterations'].mean() for l in log])),
                                          "failed": int(sum(int((l['status'] < 0).sum()) for l in log))}
    c4["workload"] = "config 4: 512 scenarios per GPU x 50 shrinking-horizon re-solves (stride 2 intervals, 1 % measurement noise), wall time of the host loop including transfers"
    alt["c4"] = c4
    return alt


def cpu_baseline(solver, scen, start, with_overrides):
    "The CPU oracle on the same workload: all usable host cores (bounded sample sized to about 12 s) and one thread."

    import numpy as np
    from oracle import oracle                      # the checker, timed here as the CPU baseline (kind: port)
    from mseetc.track import computeDiscretizationPoints

    train, track = solver.train, solver.track
    N = solver.numIntervals
    pts = computeDiscretizationPoints(track, N)
    opts = dict(numIntervals=N, maxIterations=500, energyOptimal=True, minimumVelocity=1, numSteps=1, numApproxSteps=1)
    oprob = oracle.pack_problem(train, pts, opts, 1, (1 - train.etaTraction)/train.etaTraction, 1 - train.etaRgBrake, track.length)

    ncores = usable_cores()
    base = scen[:min(len(scen), 1024)]
    t1 = time.perf_counter()
    oracle.solve_batch(oprob, base, nthreads=ncores, start=start)
    first = time.perf_counter() - t1
    reps = int(min(64, max(1, np.ceil(12.0/max(first, 1e-3)))))
    sc = np.tile(base, (reps, 1))
    t1 = time.perf_counter()
    zc, stc, nfail = oracle.solve_batch(oprob, sc, nthreads=ncores, start=start)
    dt = time.perf_counter() - t1
    # single thread: a short sample of the same batch
    m = min(len(base), 256)
    t1 = time.perf_counter()
    oracle.solve_batch(oprob, base[:m], nthreads=1, start=start)
    dt1 = time.perf_counter() - t1
    note = " (rolling-stock perturbations of config 3 not applied: nominal train)" if with_overrides else ""
    return {"value": sc.shape[0]/dt, "unit": "solves/s", "cores": ncores, "kind": "port",
            "sample": "{} solves (the first {} scenarios of the batch, repeated), CPU oracle (oracle/ms_oracle.c, same algorithm and starting point, gcc -O2, OpenMP over scenarios), {:.1f} s wall, {} failed{}".format(sc.shape[0], len(base), dt, nfail, note),
            "single_thread": {"value": m/dt1, "unit": "solves/s", "cores": 1, "sample": "{} solves, {:.1f} s".format(m, dt1)}}


if __name__ == '__main__':
    sys.exit(main())
