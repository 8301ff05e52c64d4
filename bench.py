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

`--single-process` drives the N devices from one process instead (one handle and one stream per device, SURVEY 8b `devices[]`,
no torch.distributed): the other dispatch style of SURVEY 8e, same sharding.

Prints ONE JSON line on rank 0.  `roofline` prices the solve kernel with the streaming model S of SURVEY.md section 8d
(904 B per stage-iteration) as the contract prescribes, and says next to it what the counters say: the compulsory bytes, the bytes
the TCC counters saw (profiles/hbm_traffic.json, per workload, digest-checked against the running library) and what really bounds
the kernel (`limiter`: the double-precision issue rate and exposed latency of one wave per SIMD).  `cpu_baseline` times the CPU
oracle (a port, not the reference's CasADi/IPOPT, which cannot run here) on the host cores and on one thread; `alt` (N = 1 only)
carries the other workloads of SURVEY 8(d): the reference's starting point on config 1, configs 2-4 at their per-GPU sizes, the
other transcriptions.  `--workload c4` measures the shrinking-horizon loop itself (a step = 50 re-solves of every scenario).
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


# the other transcriptions of the reference's options (ocp.py:26-28, train.py:303-322): solver options, integrator options
TRANSCRIPTIONS = dict(rk=(dict(), None),
                      integrate_losses=(dict(integrateLosses=True), dict(order=4, numSteps=1, numApproxSteps=1)),
                      irk_radau2=(dict(integrationMethod='IRK'), dict(order=2, numSteps=1, numApproxSteps=1)),
                      cvodes_tolerances=(dict(integrationMethod='CVODES'), dict()))


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
    ap.add_argument('--steps', type=int, default=None, help='timed steps (default per workload: c1 500, c2 50, c3 100, c4 5 -- about a second of launches)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed warm-up steps (default 10; c4: 1)')
    ap.add_argument('--batch', type=int, default=0, help='scenarios per GPU (default: the workload\'s per-GPU size)')
    ap.add_argument('--intervals', type=int, default=0, help='shooting intervals (default: the workload\'s)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-alt', action='store_true', help='skip the extra workloads (reference start, configs 2-4)')
    ap.add_argument('--no-build', action='store_true', help='never compile (profiled runs): exit non-zero when the library is stale')
    ap.add_argument('--start', default='profile', choices=['profile', 'reference'],
                    help="starting point of every solve: 'profile' (library default, built on the device from the scenario) or 'reference' (cold start of ocp.py:325-339)")
    ap.add_argument('--workload', default='c1', choices=['c1', 'c2', 'c3', 'c4'],
                    help='c1: BASELINE configs[1] (the metric); c2: N=200 on CH_StGallen_Wil; c3: config 1 with per-scenario rolling stock; '
                         'c4: shrinking-horizon MPC, 50 re-solves per scenario (a step = one whole loop, value = successful re-solves/s)')
    ap.add_argument('--transcription', default='rk', choices=sorted(TRANSCRIPTIONS),
                    help='the transcription of the reference\'s options the workload is solved with (default: RK4 + trapezoidal time, simulations/config.json)')
    ap.add_argument('--single-process', action='store_true',
                    help='one process drives all --gpus devices (one handle and stream per device), no torch.distributed')
    ap.add_argument('--process-group', action='store_true',
                    help='form the rank process group (RCCL) also at world size 1 and run the barrier and the reductions through it: a smoke test of the '
                         'multi-GPU code path on a one-GPU box (no scaling claim comes out of it)')
    args = ap.parse_args(argv)
    if args.steps is None:
        args.steps = dict(c1=500, c2=50, c3=100, c4=5)[args.workload]
    if args.warmup is None:
        args.warmup = 1 if args.workload == 'c4' else 10
    return args


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


PER_GPU_BATCH = dict(c1=1024, c2=8192, c3=8192, c4=512)


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

    listed0 = prob.follow_counts()[0]
    prob.time_first_pass(True)              # events around the first kernel of every launch too (the dominant kernel's own duration)
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
    measure.first_pass_ms = prob.first_pass_ms()[0]                                     # (side channel: the callers that want it read it right after the call)
    measure.listed_per_launch = (prob.follow_counts()[0] - listed0)/max(steps, 1)      # scenarios the first pass handed to the follow-up kernel
    prob.time_first_pass(False)
    for d in (d_scen, d_z, d_st, d_ov):
        if d is not None:
            prob.free(d)
    return elapsed, kernel_ms_total/steps, st


measure.first_pass_ms = None
measure.listed_per_launch = None


def build_workload(name, B, N, rank, start, device, transcription='rk'):
    "(solver, scenarios (B,4), overrides or None, description)"

    from mseetc import workloads as wl
    from mseetc.ocp import casadiSolver

    train, track, N0 = wl.config(name)
    N = N or N0
    extra, io = TRANSCRIPTIONS[transcription]
    opts = wl.options(N, **extra)
    if io is not None:
        opts['integrationOptions'] = dict(io)
    solver = casadiSolver(train, track, opts, device=device, startingPoint=start)
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

    if transcription != 'rk':
        text += ", transcription {} {}".format(extra, io)

    return solver, solver._scenarios(T, 0, 1, 1), overrides, text


def measure_single_process(solver, scen, overrides, steps, warmup, ndev):
    """
    The other dispatch style of SURVEY 8e: this process drives `ndev` devices, one handle and one stream each, the batch of
    every device resident in its HBM.  Returns (elapsed s, worst kernel ms per launch, stats of all devices stacked).
    """

    import numpy as np
    from mseetc._device import ST, OV

    first, others = solver._handles(list(range(ndev)))
    probs = [first] + list(others)
    B = scen.shape[0]
    bufs = []
    for k, prob in enumerate(probs):
        # weak scaling: every device solves a batch of the per-GPU size (its own seed would only change the numbers, not the work)
        d_scen, d_z, d_st = prob.alloc(scen.nbytes), prob.alloc(8*prob.nz*B), prob.alloc(8*ST['COUNT']*B)
        prob.to_device(d_scen, scen)
        d_ov = None
        if overrides is not None:
            ov = np.ascontiguousarray(overrides, dtype=np.float64).reshape(B, OV['COUNT'])
            d_ov = prob.alloc(ov.nbytes)
            prob.to_device(d_ov, ov)
        bufs.append((d_scen, d_z, d_st, d_ov))
    for _ in range(warmup):
        for prob, (d_scen, d_z, d_st, d_ov) in zip(probs, bufs):
            prob.solve_batch_device(B, d_scen, d_z, None, d_st, d_overrides=d_ov)
    for prob in probs:
        prob.synchronize()
    t0 = time.perf_counter()
    for prob in probs:
        prob.timer_begin()
    for _ in range(steps):
        for prob, (d_scen, d_z, d_st, d_ov) in zip(probs, bufs):      # launches return at once: the devices run side by side
            prob.solve_batch_device(B, d_scen, d_z, None, d_st, d_overrides=d_ov)
    ms = [prob.timer_end() for prob in probs]
    for prob in probs:
        prob.synchronize()
    elapsed = time.perf_counter() - t0
    st = np.zeros((len(probs), B, ST['COUNT']))
    for k, (prob, b) in enumerate(zip(probs, bufs)):
        prob.to_host(st[k], b[2])
        for d in b:
            if d is not None:
                prob.free(d)
    return elapsed, max(ms)/steps, st.reshape(-1, ST['COUNT'])


def measure_mpc(train, track, N, T, steps, warmup, device, warm=True, barrier=None, host_loop=False):
    """
    Config 4: `steps` shrinking-horizon loops (50 re-solves of every scenario each, stride 2 intervals, 1 % measurement noise) -- with the
    loop's bookkeeping on the device (mseetc.mpc.DeviceLoop: the 50 problem records are built and uploaded once, outside the timed region,
    since the grid sequence does not depend on the solutions; a timed step = one msd_mpc_run over the batch, including the download of the
    log), or `host_loop`: the host-side loop of mseetc.mpc.shrinkingHorizon (one launch per re-solve, bookkeeping in numpy).
    Returns (elapsed s, dict of loop statistics of the last loop).
    """

    import numpy as np
    from mseetc import workloads as wl
    from mseetc.mpc import shrinkingHorizon, DeviceLoop

    if host_loop:
        loop = None
        run = lambda: shrinkingHorizon(train, track, wl.options(N), T, numResolves=50, noise=0.01, seed=1, warmStart=warm, device=device)
        for _ in range(max(1, warmup) if warmup else 0):
            shrinkingHorizon(train, track, wl.options(N), T[:64], numResolves=2, noise=0.01, seed=1, warmStart=warm, device=device)
    else:
        loop = DeviceLoop(train, track, wl.options(N), 50, noise=0.01, warmStart=warm, device=device)
        run = lambda: loop.run(T, seed=1, keepZ=False)
        for _ in range(max(1, warmup) if warmup else 0):
            run()
    if barrier:
        barrier()
    t0 = time.perf_counter()
    good = failed = relaxed = 0
    for _ in range(steps):
        log = run()
        good += int(sum(int((l['status'] >= 0).sum()) for l in log))
        failed += int(sum(int((l['status'] < 0).sum()) for l in log))
        relaxed += int(sum(int(l['relaxed'].sum()) for l in log))
    if barrier:
        barrier()
    elapsed = time.perf_counter() - t0
    if loop is not None:
        loop.close()
    stage_iters = float(sum(l['numIntervals']*l['iterations'].sum() for l in log))
    info = {"resolves_per_loop": len(log), "scenarios": len(T), "successful": good, "failed": failed, "arrival_time_relaxed": relaxed,
            "ip_iterations_mean": float(np.mean([l['iterations'].mean() for l in log])),
            "kernel_ms_per_loop": float(sum(l['kernel_ms'] for l in log)), "stage_iterations_per_loop": stage_iters,
            "loop": "host" if host_loop else "device"}
    return elapsed, info


def summarize(B, N, steps, elapsed, launch_ms, st):
    import numpy as np
    from mseetc._device import ST
    iters = st[:, ST['ITERS']]
    return {"solves_per_s": B*steps/elapsed, "launch_ms": launch_ms, "batch": B, "num_intervals": N, "converged": int(np.sum(st[:, ST['STATUS']] >= 0)),
            "ip_iterations_mean": float(np.mean(iters)), "ip_iterations_max": float(np.max(iters)),
            "kkt_fallbacks": int(np.sum(st[:, ST['N_FALLBACK']]))}


def hbm_traffic(entry, key):
    """
    Measured HBM bytes per launch of workload `key` ('c1', 'c2', 'c3', 'c1/integrate_losses' ...), if the committed measurement
    belongs to the library that is running: (record or None, note).
    """

    tf = ROOT / 'profiles' / 'hbm_traffic.json'
    if not tf.exists():
        return None, "no profiles/hbm_traffic.json"
    try:
        rec = json.loads(tf.read_text())
    except Exception:
        return None, "unreadable profiles/hbm_traffic.json"
    if rec.get('kernel_digest') != entry.hip_digest():
        return None, "profiles/hbm_traffic.json was measured on another build of the kernels (digest mismatch): re-run tools/profile_round.sh"
    w = rec.get('workloads', {}).get(key)
    if w is None:
        return None, "profiles/hbm_traffic.json has no entry for workload " + key
    return w, rec.get('source')


VALU_ISSUE_PEAK_GINST = 1024*2.4/4      # wave-level double-rate VALU issue: 256 CUs x 4 SIMDs, 2.4 GHz, one wave64 instruction per 4 cycles


def roofline_block(entry, key, B, N, nz, stage_iters, launch_ms, geo, kernel_ms=None):
    """
    The roofline object of the bench line for one launch of the solver over B scenarios (stage_iters = N x sum of IP iterations).
    kernel_ms: duration of the dominant kernel alone (the first pass of a split launch; HIP events around it, msd_problem_first_pass_ms),
    launch_ms: first pass + follow-up kernel.

    The top-level fields always mean the same thing (round 5; rounds 3-4 switched them between two roofs depending on whether counters were on
    file): the bench contract's pricing -- ALGORITHMIC bytes per launch (SURVEY 8(d)'s streaming model S: 904 B per stage-iteration x the
    stage-iterations of the launch) over the dominant kernel's duration against the HBM peak.  The kernel does not generate that traffic (the
    iterate is register/LDS resident); what limits it is instruction issue of a lone wave per SIMD, reported in `valu_issue` when the SQ
    counters of the running library are on file (profiles/hbm_traffic.json, digest-checked).  `frac_useful` = SURVEY 8(d)'s flop model (400
    flop per stage-iteration) against the FP64 vector peak: efficiency, where `valu_issue.frac` is utilisation.
    """

    kms = kernel_ms or launch_ms
    achieved = BYTES_PER_STAGE_ITER*stage_iters/(kms*1e-3)/1e9
    compulsory = float(B*(8*nz + 168))
    rec, note = hbm_traffic(entry, key)
    traffic = rec.get('bytes_per_launch') if rec else None
    useful = (400.0*stage_iters/(kms*1e-3)/1e12)/FP64_VALU_PEAK_TFLOPS
    out = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved/HBM_PEAK_GBS,
           "model": "S = 904 B x N x sum(IP iterations) per launch (SURVEY 8d streaming model) / duration of the dominant kernel.  The iterate is "
                    "register/LDS resident: the kernel never generates that traffic -- the contract's pricing, not the limiter (see limiter, valu_issue)",
           "limiter": "valu-issue: one wave per SIMD (the register file allows no more) issuing FP64 VALU instructions with nothing to hide LDS/scratch latency behind",
           "frac_useful": useful}
    issue = rec.get('issue') if rec else None
    valu_issue = None
    if issue and issue.get('valu_instructions_per_launch'):
        valu = issue['valu_instructions_per_launch']*stage_iters/max(rec.get('stage_iterations_per_launch', stage_iters), 1.0)      # (scaled to this run's iteration count)
        ginst = valu/(kms*1e-3)/1e9
        valu_issue = {"bound": "valu-issue", "achieved": ginst, "peak": VALU_ISSUE_PEAK_GINST, "unit": "G wave-instructions/s", "frac": ginst/VALU_ISSUE_PEAK_GINST,
                      "model": "VALU instructions of the dominant kernel (SQ_INSTS_VALU of the profiling pass on this library, scaled by the stage-iterations of this run) / its "
                               "duration, against one double-rate wave64 instruction per 4 cycles on each of 1024 SIMDs at 2.4 GHz: utilisation of the issue slots, every emitted "
                               "instruction counted (frac_useful prices the algorithm's flops instead)",
                      "valu_instructions_per_stage_iteration": issue.get('valu_instructions_per_launch', 0)/max(rec.get('stage_iterations_per_launch', stage_iters), 1.0)}
    out.update({"traffic": traffic, "traffic_source": note, "valu_issue": valu_issue, "frac_model_S": out["frac"],
                "frac_compulsory": compulsory/(kms*1e-3)/1e9/HBM_PEAK_GBS, "compulsory_bytes_per_launch": compulsory,
                "frac_measured": (traffic/(launch_ms*1e-3)/1e9/HBM_PEAK_GBS) if traffic else None,
                "kernel": "msd::solve_kernel<{},{}> (one workgroup of {} threads per scenario, {} shooting nodes per lane); split launch: first pass (fused iteration) + follow-up kernel".format(geo[0], geo[1], geo[0], geo[1]),
                "kernel_ms": kms, "launch_ms": launch_ms, "stage_iterations_per_launch": stage_iters,
                "issue": issue, "fp64_valu_peak_tflops": FP64_VALU_PEAK_TFLOPS, "fp64_valu_frac_model": useful})
    if valu_issue:
        out["valu_instructions_per_stage_iteration"] = valu_issue["valu_instructions_per_stage_iteration"]
    return out


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
        return 2      # not a success: no JSON line was produced

    if args.gpus > 1 and not child and not args.single_process:
        return launch_ranks(args)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))

    if args.single_process:
        if world != 1:
            raise SystemExit("--single-process runs as one process (no launcher)")
    elif world != args.gpus:
        raise SystemExit("--gpus {} but WORLD_SIZE {}".format(args.gpus, world))

    import numpy as np

    if share:
        local_rank = local_rank % ndev
    torch.cuda.set_device(local_rank)
    red_dev = 'cpu' if share else 'cuda'

    backend = None      # what carries the barrier and the reductions between the ranks (the data path has no collective)
    grouped = world > 1 or (args.process_group and not args.single_process)      # (--process-group: the same code at world size 1)
    if grouped:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if world == 1 and 'MASTER_PORT' not in os.environ:
            os.environ['MASTER_PORT'] = str(free_port())
        if share:
            dist.init_process_group('gloo', rank=rank, world_size=world)
            backend = 'gloo (MSD_BENCH_SHARE_DEVICES=1: ranks share devices, RCCL refuses that)'
        else:
            try:
                dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))
                probe = torch.ones(1, dtype=torch.float64, device='cuda')
                dist.all_reduce(probe)      # the communicator is created by the first collective: a failure shows here, not in the timed region
                torch.cuda.synchronize()
                if int(probe.item()) != world:
                    raise RuntimeError("all_reduce over RCCL saw {} ranks, expected {}".format(int(probe.item()), world))
                backend = 'nccl (RCCL)'
            except Exception as exc:
                # Round 5: no silent fallback.  The data path has no collective -- gloo would carry the barrier and the two reductions just as well --
                # but a line measured that way must say so, and only on request: MSD_BENCH_GLOO_FALLBACK=1
                msg = "bench.py: RCCL process group failed on rank {} ({}: {})".format(rank, type(exc).__name__, exc)
                if os.environ.get('MSD_BENCH_GLOO_FALLBACK') != '1':
                    print(msg + "; set MSD_BENCH_GLOO_FALLBACK=1 to let gloo carry the barrier and the reductions instead", file=sys.stderr)
                    return 3
                print(msg + "; MSD_BENCH_GLOO_FALLBACK=1: falling back to gloo", file=sys.stderr)
                if dist.is_initialized():
                    dist.destroy_process_group()
                dist.init_process_group('gloo', rank=rank, world_size=world)
                red_dev = 'cpu'
                backend = 'gloo (fallback: RCCL failed with {})'.format(type(exc).__name__)

    def group_fields():
        "process-group facts of a multi-rank line: backend, ranks seen by a reduction, the device of every rank"
        if not grouped:
            return {"process_group_backend": None, "world_size_seen": 1, "device_of_rank": [local_rank]}
        seen = torch.ones(1, dtype=torch.int64, device=red_dev)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        devs = torch.zeros(world, dtype=torch.int64, device=red_dev)
        devs[rank] = local_rank
        dist.all_reduce(devs, op=dist.ReduceOp.SUM)
        return {"process_group_backend": backend, "world_size_seen": int(seen.item()), "device_of_rank": [int(v) for v in devs.tolist()]}

    def barrier(sync_only=False):
        torch.cuda.synchronize()
        if grouped and not sync_only:
            dist.barrier()
            torch.cuda.synchronize()

    from mseetc._device import ST

    B = args.batch or PER_GPU_BATCH[args.workload]
    ndrive = args.gpus if args.single_process else 1      # devices this process drives

    if args.workload == 'c4':
        # the shrinking-horizon loop: a step = 50 re-solves of every scenario (warm-started on the device), value = successful re-solves/s
        from mseetc import workloads as wl
        train, track, N = wl.config('c4')
        N = args.intervals or N
        T = wl.c1_times(B, seed=20260615 + rank)
        elapsed, info = measure_mpc(train, track, N, T, args.steps, args.warmup, local_rank, warm=True, barrier=barrier)
        if grouped:
            t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            cnt = torch.tensor([info['successful'], info['failed']], dtype=torch.int64, device=red_dev)
            dist.all_reduce(cnt, op=dist.ReduceOp.SUM)
            good_all, failed_all = int(cnt[0].item()), int(cnt[1].item())
        else:
            good_all, failed_all = info['successful'], info['failed']
        gf = group_fields()
        if rank == 0:
            from mseetc.ocp import casadiSolver
            probe = casadiSolver(train, track, wl.options(N), device=local_rank)
            geo, nz = probe.problem.geometry(), probe.problem.nz
            probe.close()
            launch_ms = info['kernel_ms_per_loop']
            line = {
                "metric": "OCP solves/sec (N=100, VIRM6, var-speed-limit track)",
                "value": good_all/elapsed, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                "ms_per_step": 1e3*elapsed/args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": "config 4: {} scenarios per GPU x 50 shrinking-horizon re-solves (stride 2 intervals from N={}, 1 % measurement noise on t and v, "
                                       "seed 20260615+rank), warm-started from the previous solution and multipliers on the device, loop bookkeeping on the device (msd_mpc_run; the 50 problem "
                                       "records are uploaded once, outside the timed region: the grid sequence does not depend on the solutions); value counts successful re-solves only; "
                                       "wall time per loop including the download of the log".format(B, N),
                           "batch_per_gpu": B, "num_intervals": N, "resolves_failed": failed_all, "resolves_successful": good_all,
                           "arrival_time_relaxed": info['arrival_time_relaxed'], "ip_iterations_mean": info['ip_iterations_mean'],
                           "parallelism": "scenarios sharded, no collective"},
                "roofline": roofline_block(entry, 'c4', B*info['resolves_per_loop'], N, nz, info['stage_iterations_per_loop'], launch_ms, geo),
            }
            line["config"].update(gf)
            line["roofline"]["launch_ms_note"] = "device time of one whole loop (events around its 50 re-solves: solver launches and bookkeeping kernels, shrinking horizons)"
            print(json.dumps(line), flush=True)
        if grouped:
            dist.destroy_process_group()
        return 0

    solver, scen, overrides, text = build_workload(args.workload, B, args.intervals, rank, args.start, local_rank, args.transcription)
    N = solver.numIntervals

    first_ms, listed = None, None
    if args.single_process:
        elapsed, launch_ms, st = measure_single_process(solver, scen, overrides, args.steps, args.warmup, ndrive)
    else:
        elapsed, launch_ms, st = measure(solver, scen, overrides, args.steps, args.warmup, barrier)
        first_ms, listed = (measure.first_pass_ms or None), measure.listed_per_launch

    if grouped:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    n_ok = int(np.sum(st[:, ST['STATUS']] >= 0))
    iters = st[:, ST['ITERS']]

    by_rank = None
    if grouped:
        ok = torch.tensor([n_ok], dtype=torch.int64, device=red_dev)
        dist.all_reduce(ok, op=dist.ReduceOp.SUM)
        n_ok_all = int(ok.item())
        # every rank's own scenarios (seed + rank): mean iteration count and first running time per rank, so that a line shows the ranks solved different batches
        slot = torch.zeros(2*world, dtype=torch.float64, device=red_dev)
        slot[2*rank], slot[2*rank + 1] = float(np.mean(iters)), float(scen[0, 1])
        dist.all_reduce(slot, op=dist.ReduceOp.SUM)
        by_rank = {"ip_iterations_mean": [float(v) for v in slot[0::2].tolist()], "first_running_time": [float(v) for v in slot[1::2].tolist()]}
    else:
        n_ok_all = n_ok
    gf = group_fields()

    if rank == 0:

        ngpu = world*ndrive
        total_solves = B*ngpu*args.steps
        value = total_solves/elapsed

        stage_iters = float(N*np.sum(iters))/ndrive          # units one launch processes (one device)
        geo = solver.problem.geometry()
        key = args.workload + ('' if args.transcription == 'rk' else '/' + args.transcription) + ('' if args.start == 'profile' else '/reference_start')

        start_text = ("every solve starts from the device-built speed profile (no information from earlier solves; same optimum as the reference's cold start)"
                      if args.start == 'profile' else "every solve cold-starts from the reference's point (ocp.py:325-339)")

        line = {
            "metric": "OCP solves/sec (N=100, VIRM6, var-speed-limit track)",
            "value": value, "unit": "solves/s", "n_gpus": ngpu, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3*elapsed/args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": text + ", " + start_text + ", KKT<=1e-8",
                       "batch_per_gpu": B, "num_intervals": N, "start": args.start, "converged": n_ok_all, "scenarios": B*ngpu,
                       "kkt_fallbacks": int(np.sum(st[:, ST['N_FALLBACK']])), "cycles_per_solve_mean": float(np.mean(st[:, ST['CYC_TOTAL']])),
                       "ip_iterations_mean": float(np.mean(iters)), "ip_iterations_max": float(np.max(iters)),
                       "dispatch": "one process, {} device handle(s)".format(ndrive) if args.single_process else "one process per GPU",
                       "parallelism": "scenarios sharded, no collective"},
            "roofline": roofline_block(entry, key, B, N, solver.problem.nz, stage_iters, launch_ms, geo, kernel_ms=first_ms),
        }
        line["config"]["handed_to_follow_up_kernel_per_launch"] = listed
        line["config"].update(gf)
        if by_rank is not None:
            line["config"]["by_rank"] = by_rank

        solver.close()

        if world == 1 and not args.single_process and not args.no_alt and args.workload == 'c1' and args.transcription == 'rk':
            line["alt"] = alt_workloads(args, local_rank)

        if not args.no_cpu_baseline and world == 1 and not args.single_process and args.transcription == 'rk':
            line["cpu_baseline"] = cpu_baseline(solver, scen, args.start, overrides is not None)

        print(json.dumps(line), flush=True)

    if grouped:
        dist.destroy_process_group()

    return 0


def two_streams(device, k):
    """
    What the tail of a 1024-scenario launch costs (never the headline): config 1 on two handles -- two streams -- whose launches alternate without waiting
    for each other, so that the workgroups of the next launch fill the SIMDs the previous one has left while its slowest scenarios finish (a launch of 1024
    scenarios on 1024 SIMDs lasts 27-28 iterations, the scenarios take 20.6 on average).  Each handle solves its own batch (seed + 0 / + 1), resident in HBM.
    """

    import numpy as np
    from mseetc._device import ST

    B = PER_GPU_BATCH['c1']
    sides = []
    for r in range(2):
        solver, scen, ov, text = build_workload('c1', B, 0, r, 'profile', device, 'rk')
        pr = solver.problem
        d_scen, d_z, d_st = pr.alloc(scen.nbytes), pr.alloc(8*pr.nz*B), pr.alloc(8*ST['COUNT']*B)
        pr.to_device(d_scen, scen)
        sides.append((solver, pr, d_scen, d_z, d_st))
    for _ in range(2):
        for solver, pr, d_scen, d_z, d_st in sides:
            pr.solve_batch_device(B, d_scen, d_z, None, d_st)
    for _, pr, *_r in sides:
        pr.synchronize()
    launches = 4*k
    t0 = time.perf_counter()
    for _ in range(launches):
        for solver, pr, d_scen, d_z, d_st in sides:
            pr.solve_batch_device(B, d_scen, d_z, None, d_st)
    for _, pr, *_r in sides:
        pr.synchronize()
    dt = time.perf_counter() - t0
    ok = 0
    for solver, pr, d_scen, d_z, d_st in sides:
        st = np.zeros((B, ST['COUNT']))
        pr.to_host(st, d_st)
        ok += int(np.sum(st[:, ST['STATUS']] >= 0))
        for d in (d_scen, d_z, d_st):
            pr.free(d)
        solver.close()
    return {"solves_per_s": 2*B*launches/dt, "ms_per_pair_of_launches": 1e3*dt/launches, "launches_per_stream": launches, "converged": ok, "scenarios": 2*B,
            "workload": "config 1 on two handles / two streams, launches of 1024 scenarios alternating without waiting for each other (wall clock over both streams): "
                        "what the tail of a 1024-scenario launch costs; never the headline"}


def alt_workloads(args, device):
    "The other workloads of SURVEY 8(d) at their per-GPU sizes, ten timed launches each after two warm-ups (N = 1 runs only)."

    import numpy as np
    from mseetc import workloads as wl

    from mseetc._device import ST as _ST
    ST_STATUS = _ST['STATUS']
    alt = {}
    k, w = 10, 2

    def one(name, workload, B, start='profile', transcription='rk', note=''):
        solver, scen, ov, text = build_workload(workload, B, 0, 0, start, device, transcription)
        e, ms, st = measure(solver, scen, ov, k, w)
        alt[name] = dict(summarize(scen.shape[0], solver.numIntervals, k, e, ms, st), workload=text + note, first_pass_ms=measure.first_pass_ms,
                         handed_to_follow_up_kernel_per_launch=measure.listed_per_launch)
        solver.close()

    one("reference_start", 'c1', PER_GPU_BATCH['c1'], start='reference', note=", cold start of ocp.py:325-339")
    alt["c1_two_streams"] = two_streams(device, k)
    one("c1_batch8192", 'c1', 8192)
    one("c2", 'c2', PER_GPU_BATCH['c2'])
    one("c3", 'c3', PER_GPU_BATCH['c3'])
    # the other transcriptions of the reference's options on the config-1 batch (their own kernel instantiations)
    for name in ('integrate_losses', 'irk_radau2', 'cvodes_tolerances'):
        one(name, 'c1', PER_GPU_BATCH['c1'], transcription=name)

    # the dynamic loss model (efficiency.py) on the configuration of simulations/figure5.py (fun2): 8.5 km crop, v0 = 1 m/s, vN = 100 km/h, forceMinPn = 0, limits
    # after the side effects of totalLossesFunction; 1024 running times of 1.05 ... 1.30 times the reference's minimum of 272.4726 s (figure5.py:96; the script's
    # own reserves are 1.0 ... 1.3)
    from mseetc.train import Train
    from mseetc.efficiency import totalLossesFunction
    from mseetc.ocp import casadiSolver as _cs
    for Nd in (100, 300):
        tr = Train(config={'id': 'NL_Intercity_VIRM6'})
        tr.forceMinPn = 0
        tr.powerLosses = totalLossesFunction(tr, auxiliaries=27000, etaGear=0.96)
        sv = _cs(tr, wl.track_00(8500), wl.options(Nd), device=device)
        Td = 272.4726*(1.05 + 0.25*np.random.default_rng(20260616).random(PER_GPU_BATCH['c1']))
        sc = sv._scenarios(Td, 0, 100/3.6, 1)
        # a first launch finds the running times that do not converge: at N = 300 narrow bands (around 1.213 and 1.285 times the minimum) end at the
        # iteration limit after hundreds of backtracking steps -- on the device and on the CPU oracle alike (tools/dyn_probe.py, profiles/r04): the optimum
        # sits on a kink of the tabulated loss model there.  A launch lasts as long as its slowest scenario, so the timed launches hold the others
        first = sv.problem.solve_batch(sc)
        good = first['stats'][:, _ST['STATUS']] >= 0
        e, ms, st = measure(sv, np.ascontiguousarray(sc[good]), None, k, w)
        alt["dynamic_losses_N%d" % Nd] = dict(summarize(int(good.sum()), Nd, k, e, ms, st), frac_model_S=BYTES_PER_STAGE_ITER*Nd*float(np.sum(st[:, _ST['ITERS']]))/(ms*1e-3)/1e9/HBM_PEAK_GBS,
                                               kernel="msd::solve_kernel<{},{}> with the loss table (DYN = 1)".format(*sv.problem.geometry()),
                                               running_times=len(Td), running_times_not_converging=int((~good).sum()), launch_ms_with_them=float(first['kernel_ms']),
                                               workload="simulations/figure5.py configuration with the dynamic loss model of efficiency.py (motor/converter table, gear, auxiliaries, "
                                                        "transformer), N = {}, 1024 running times 272.4726 s x (1.05 ... 1.30), v0 = 1 m/s, vN = 100 km/h; timed on the running times "
                                                        "that converge (the others end at the iteration limit on the device and on the CPU oracle alike)".format(Nd))
        sv.close()

        if Nd == 100:
            # ... and the same loss model integrated over the running time inside the NLP (integrateLosses=True: the switch of figure6.py:178; round 6, DYN = 3)
            import copy
            so = copy.deepcopy(wl.options(Nd)); so['integrateLosses'] = True
            sv = _cs(tr, wl.track_00(8500), so, device=device)
            sc = sv._scenarios(Td[:256], 0, 100/3.6, 1)
            first = sv.problem.solve_batch(sc)
            good = first['stats'][:, _ST['STATUS']] >= 0
            e, ms, st = measure(sv, np.ascontiguousarray(sc[good]), None, 3, 1)
            alt["dynamic_losses_integrated_N100"] = dict(summarize(int(good.sum()), Nd, 3, e, ms, st), kernel="msd::solve_kernel<{},{}> with the loss table integrated over the running time (DYN = 3)".format(*sv.problem.geometry()),
                                                         running_times=256, running_times_not_converging=int((~good).sum()),
                                                         workload="the figure-5 configuration with the dynamic loss model and integrateLosses=True (ocp.py:231-241 -> train.py:367-413; figure6.py:178), N = 100, 256 running times")
            sv.close()

    # one solve at a time -- how every script of the reference drives the API (table3.py:53-64: the minimum of five runs of casadiSolver.solve per horizon): wall time of
    # solve() with the DataFrame and its post-processing, wall time of the batched entry for the one scenario, the kernels' own time, and the CPU oracle on one thread
    # beside them.  figure-10 train (forceMinPn = 0, table3.py:18) on the full track, T = 1541 s.
    from oracle import oracle as _orc                      # the checker, timed here as the CPU baseline of the row (kind: port)
    import io, contextlib
    single = {}
    tr10 = wl.train_default()
    tr10.forceMinPn = 0; tr10.forceMin = -tr10.forceMax; tr10.powerMax = 3129277; tr10.powerMin = -tr10.powerMax; tr10.etaTraction = tr10.etaRgBrake = 0.73
    for Ns in (100, 300, 1000, 5000):
        sv = _cs(tr10, wl.track_00(), dict(numIntervals=Ns, maxIterations=1000, integrationOptions=dict(numSteps=1, numApproxSteps=1)), device=device)
        t_solve, t_batch, t_kernel = [], [], []
        with contextlib.redirect_stdout(io.StringIO()):
            sv.solve(1541.0)      # (first call: library and handle warm)
            for _ in range(5):
                t0 = time.perf_counter(); df, stt = sv.solve(1541.0); t_solve.append(time.perf_counter() - t0)
                t0 = time.perf_counter(); r = sv.solveBatch([1541.0]); t_batch.append(time.perf_counter() - t0)
                t_kernel.append(float(r['kernel_ms']))
        from mseetc.track import computeDiscretizationPoints as _cdp
        op = _orc.pack_problem(tr10, _cdp(wl.track_00(), Ns), dict(numIntervals=Ns, maxIterations=1000, energyOptimal=True, minimumVelocity=1, numSteps=1, numApproxSteps=1), 1, 0.27/0.73, 0.27, wl.track_00().length)
        t0 = time.perf_counter(); ro = _orc.solve(op, op.scenario(1541.0), start='profile'); t_cpu = time.perf_counter() - t0
        single["N%d" % Ns] = {"solve_ms": 1e3*min(t_solve), "solve_batch_of_one_ms": 1e3*min(t_batch), "kernel_ms": min(t_kernel), "ip_iterations": int(r['iterations'][0]),
                              "converged": bool(r['status'][0] >= 0 and df is not None), "cost_kWh": float(r['cost'][0]),
                              "cpu_oracle_one_thread_ms": 1e3*t_cpu, "cpu_oracle_iterations": int(ro['stats']['ITERS'])}
        sv.close()
    single["workload"] = ("casadiSolver(train, track, opts).solve(1541) one scenario at a time, figure-10 train on 00_var_speed_limit_100, profile start; min of 5 like table3.py:53-64: "
                          "solve_ms = wall time of solve() including the DataFrame and postProcessDataFrame (its CVODES re-simulation on the device), "
                          "solve_batch_of_one_ms = the batched entry for one scenario (upload, kernels, download), kernel_ms = device time of the kernels; "
                          "cpu_oracle_one_thread_ms = the C oracle on one host thread, same starting point.  The reference publishes 4.96 / 6.99 s for GPOPS on this problem (BASELINE.md)")
    alt["single_solve"] = single

    # the streamed kernels (N >= 640: stage blocks in device memory) at the batch size of the other rows, the CPU oracle on all host cores beside them
    # ("no CPU number stands beside the long-horizon rows")
    from mseetc.track import computeDiscretizationPoints as _cdp2
    streamed = {}
    ncores = usable_cores()
    for Ns in (700, 1000):
        sv = _cs(wl.train_default(), wl.track_00(), dict(numIntervals=Ns, maxIterations=1000, integrationOptions=dict(numSteps=1, numApproxSteps=1)), device=device)
        Ts = 1541*(1 + 0.15*np.random.default_rng(Ns).random(PER_GPU_BATCH['c1']))
        sc = sv._scenarios(Ts, 0, 1, 1)
        e, ms, st = measure(sv, sc, None, 2, 1)
        trn = wl.train_default()
        op = _orc.pack_problem(trn, _cdp2(wl.track_00(), Ns), dict(numIntervals=Ns, maxIterations=1000, energyOptimal=True, minimumVelocity=1, numSteps=1, numApproxSteps=1), 1,
                               (1 - trn.etaTraction)/trn.etaTraction, 1 - trn.etaRgBrake, wl.track_00().length)
        m = 4*ncores
        t0 = time.perf_counter(); _, stc, nfail = _orc.solve_batch(op, sc[:m], nthreads=ncores, start='profile'); dtc = time.perf_counter() - t0
        streamed["N%d" % Ns] = {"solves_per_s": len(Ts)/(ms*1e-3), "launch_ms": ms, "batch": len(Ts), "geometry": list(sv.problem.geometry()), "converged": int(np.sum(st[:, _ST['STATUS']] >= 0)),
                                "ip_iterations_mean": float(np.mean(st[:, _ST['ITERS']])), "cpu_oracle_solves_per_s": m/dtc, "cpu_cores": ncores,
                                "cpu_sample": "{} solves of the same batch, {:.1f} s, {} failed".format(m, dtc, int(nfail))}
        sv.close()
    streamed["workload"] = "config 1 problem (JSON-default train, full track) on 700 / 1000 intervals, 1024 running times per launch, profile start: the streamed kernels; CPU oracle (OpenMP over scenarios) beside them"
    alt["streamed_horizons"] = streamed

    # the host-buffer entry point (msd_solve_batch: scenarios from and results into host memory): the PCIe-inclusive rate of the same workload,
    # wall clock over ten calls -- upload of the scenario records, launch, download of z* and the statistics.  Never the headline value.
    hb = {}
    for B in (PER_GPU_BATCH['c1'], 8192):
        solver, scen, ov, text = build_workload('c1', B, 0, 0, 'profile', device, 'rk')
        for direct in (True, False):
            solver.problem.direct_results(direct)
            # warm-up like the timed loop, the previous call's results still held: the wrapper page-locks its two alternating result buffers here (hipHostMalloc of
            # 34 MB costs more than a call; rounds 3-4 had two of them inside the ten timed calls: 10.2 ms per call of 8192 against 7.4 ms in steady state)
            for _ in range(w + 2):
                r = solver.problem.solve_batch(scen)
            kk = 2*k
            t0 = time.perf_counter()
            for _ in range(kk):
                r = solver.problem.solve_batch(scen)
            dt = time.perf_counter() - t0
            hb[("batch_%d" if direct else "batch_%d_copied") % B] = {
                "solves_per_s": B*kk/dt, "ms_per_call": 1e3*dt/kk, "kernel_ms": float(r['kernel_ms']), "converged": int(np.sum(r['stats'][:, ST_STATUS] >= 0)),
                "bytes_to_device": int(scen.nbytes), "bytes_to_host": int(r['z'].nbytes + r['stats'].nbytes), "calls": kk,
                "results": "z* stored in the page-locked arrays by the kernels (msd_problem_direct_results)" if direct else "copied behind the launch"}
        solver.close()
    hb["workload"] = ("config 1 through msd_solve_batch: scenario records from a numpy array, results into the wrapper's page-locked arrays (two alternating buffers, allocated "
                      "before the timed calls); wall time per call in steady state including both transfers.  batch_N: the wrapper's default, the kernels store every "
                      "scenario's z* in the host array when the scenario is done; batch_N_copied: one device-to-host copy behind the launch (the C entry point's default)")
    alt["host_buffers"] = hb

    # config 4: shrinking-horizon MPC, 512 scenarios per GPU (4096 over 8), 50 re-solves each: wall time of the whole loop
    train, track, N = wl.config('c4')
    T = wl.c1_times(PER_GPU_BATCH['c4'], seed=20260615)
    c4 = {}
    for name, warm, host in (("cold", False, False), ("warm", True, False), ("warm_host_loop", True, True)):
        loops = 3
        wall, info = measure_mpc(train, track, N, T, loops, 1, device, warm=warm, host_loop=host)
        c4[name] = {"resolves_per_s": info['successful']/wall, "wall_s": wall/loops, "loops": loops, "resolves": info['resolves_per_loop'],
                                          "scenarios": len(T), "ip_iterations_mean": info['ip_iterations_mean'], "successful": info['successful']//loops,
                                          "failed": info['failed']//loops, "arrival_time_relaxed": info['arrival_time_relaxed']//loops,
                                          "kernel_ms_per_loop": info['kernel_ms_per_loop'], "loop": info['loop']}
    c4["workload"] = ("config 4: 512 scenarios per GPU x 50 shrinking-horizon re-solves (stride 2 intervals, 1 % measurement noise); cold / warm: the loop with its "
                      "bookkeeping on the device (msd_mpc_run: one call per loop, wall time including the download of the log; kernel_ms_per_loop = device time "
                      "of the whole loop between two events); warm_host_loop: the host-side loop (one launch per re-solve, transfers and numpy in between; "
                      "kernel_ms_per_loop = sum of the launches' kernel times); resolves_per_s counts successful re-solves only; a re-solve whose measured "
                      "state no longer allows the arrival time is repeated with the arrival time moved to its certified minimum (arrival_time_relaxed)")
    alt["c4"] = c4

    # the restoration phase (cold path): schedules 8 to 13 times the minimum running time from the reference's starting point, where its
    # filter line search breaks down on the way -- one launch, not timed against anything
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    train, track, N = wl.config('c1')
    Tl = np.linspace(11000.0, 20000.0, 64)
    out = {}
    for resto in (True, False):
        s = casadiSolver(train, track, wl.options(N), device=device, startingPoint='reference', restoration=resto)
        r = s.solveBatch(Tl)
        s.close()
        out["restoration" if resto else "restart_only"] = {"converged": int(np.sum(r['status'] >= 0)), "scenarios": len(Tl), "restoration_phases": int(np.sum(r['stats'][:, ST['N_RESTO']])),
                                                           "ip_iterations_mean": float(np.mean(r['iterations'])), "kernel_ms": float(r['kernel_ms']),
                                                           "cost_sum": float(np.sum(r['cost'][r['status'] >= 0]))}
    out["workload"] = ("config 1 problem, 64 running times 11000 ... 20000 s (8 to 13 times the minimum) from the reference's starting point: with the feasibility "
                       "restoration phase (IPOPT's behaviour, default) and with the restart from the other starting point only")
    alt["loose_schedules_reference_start"] = out
    # the same kind of schedule on horizons that the streamed kernels solve (700 intervals; 600: five-wave first pass + streamed follow-up kernel): restoration phases
    # and IPOPT's watchdog procedure (ten shortened iterations in a row) on the device -- profiles/r04/watchdog_survey.txt has the CPU oracle's counts
    long_out = {}
    for Nl in (600, 700):
        s = casadiSolver(train, track, dict(numIntervals=Nl, maxIterations=800, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='reference', device=device)
        s.solveBatch(np.linspace(8000.0, 20000.0, 16))
        r = s.solveBatch(np.linspace(8000.0, 20000.0, 16))
        s.close()
        long_out["N%d" % Nl] = {"converged": int(np.sum(r['status'] >= 0)), "scenarios": 16, "restoration_phases": int(np.sum(r['stats'][:, ST['N_RESTO']])),
                                "watchdog_procedures": int(np.sum(r['stats'][:, ST['N_WATCHDOG']])), "ip_iterations_mean": float(np.mean(r['iterations'])),
                                "kernel_ms": float(r['kernel_ms']), "cost_sum": float(np.sum(r['cost'][r['status'] >= 0]))}
    long_out["workload"] = ("config 1 track and train on 600 / 700 intervals, 16 running times 8000 ... 20000 s from the reference's starting point: the five-wave kernel followed up by "
                            "the streamed kernel / the streamed kernel itself, through restoration phases and watchdog procedures")
    alt["loose_schedules_long_horizons"] = long_out
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
