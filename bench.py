"""
Benchmark of the hot path: full cold-start solves of a batch of independent train-control OCPs on MI355X.

    python bench.py --gpus N --steps K --warmup W

A "step" is one pass of the HIP solver over one batch of synthetic scenarios that is already resident in HBM.
Workload = BASELINE.json configs[1]: B = 1024 scenarios per GPU, N = 100 shooting
intervals, track 00_var_speed_limit_100, train NL_Intercity_VIRM6 with the JSON defaults (both brakes), RK4 with
numSteps = 1 and trapezoidal time (numApproxSteps = 1), v0 = vN = 1 m/s, T_i = 1541 (1 + 0.15 u_i),
u = default_rng(20260612 + rank).random(B).  For N > 1 the driver launches one rank per GPU with torch.distributed.run;
scenarios are independent, so ranks share nothing but the barrier (weak scaling, no collective on the data path).

Prints ONE JSON line on rank 0 (see the field list in the task contract); `roofline` prices the solve kernel with the
streaming model S of SURVEY.md section 8d (904 B per stage-iteration); `cpu_baseline` times the CPU oracle (a port, not the
reference's CasADi/IPOPT, which cannot run here) on the host cores.
"""

import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent

for p in (str(ROOT / 'ms-eetc_amd'), str(ROOT), str(ROOT / 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

BYTES_PER_STAGE_ITER = 904.0   # SURVEY.md section 8d, streaming model S (nu = 2): 113 doubles
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.3 TB/s achievable)


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


def main():

    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=1024, help='scenarios per GPU')
    ap.add_argument('--intervals', type=int, default=100)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-build', action='store_true', help='never compile (profiled runs): exit non-zero when the library is stale')
    ap.add_argument('--start', default='profile', choices=['profile', 'reference'],
                    help="starting point of every solve: 'profile' (library default, built on the device from the scenario) or 'reference' (cold start of ocp.py:325-339)")
    ap.add_argument('--workload', default='c1', choices=['c1', 'c2'], help='c1: BASELINE configs[1] (the metric); c2: N=200 on CH_StGallen_Wil (extra measurement)')
    args = ap.parse_args()

    # build (or check) the native library before anything touches the GPU: hipcc must never run in a process that has initialised it
    import __graft_entry__ as entry
    if args.no_build:
        if entry.stale() and not os.environ.get('MSD_LIB'):       # MSD_LIB: a tuning build (tools/build_variant.py) is being measured
            raise SystemExit("bench.py --no-build: ms-eetc_amd/lib/libmseetc_hip.so is missing or stale; run `python3 __graft_entry__.py` first")
    else:
        entry.build()

    import torch

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))

    if world != args.gpus:
        raise SystemExit("--gpus {} but WORLD_SIZE {}: launch with torch.distributed.run --nproc-per-node {}".format(args.gpus, world, args.gpus))

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")

    torch.cuda.set_device(local_rank)

    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank))

    import cases
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST

    B, N = args.batch, args.intervals

    if args.workload == 'c2':
        N = 200 if args.intervals == 100 else args.intervals

    train, track = cases.train_default(), (cases.track_00() if args.workload == 'c1' else cases.track_CH())
    opts = dict(numIntervals=N, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1))
    solver = casadiSolver(train, track, opts, device=local_rank, startingPoint=args.start)
    prob = solver.problem

    T = cases.c1_times(B, seed=20260612 + rank) if args.workload == 'c1' else cases.c2_times(B, seed=20260613 + rank)
    scen = solver._scenarios(T, 0, 1, 1)

    # inputs resident in HBM before the timed region
    nz = prob.nz
    d_scen = prob.alloc(scen.nbytes)
    d_z = prob.alloc(8*nz*B)
    d_st = prob.alloc(8*ST['COUNT']*B)
    prob.to_device(d_scen, scen)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        prob.solve_batch_device(B, d_scen, d_z, None, d_st)
    prob.synchronize()

    barrier()
    t0 = time.perf_counter()
    prob.timer_begin()                      # HIP events on the stream the kernel is launched on
    for _ in range(args.steps):
        prob.solve_batch_device(B, d_scen, d_z, None, d_st)
    kernel_ms_total = prob.timer_end()      # waits for the last kernel
    prob.synchronize()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    barrier()

    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    st = np.zeros((B, ST['COUNT']))
    prob.to_host(st, d_st)
    n_ok = int(np.sum(st[:, ST['STATUS']] >= 0))
    iters = st[:, ST['ITERS']]

    if world > 1:
        ok = torch.tensor([n_ok], dtype=torch.int64, device='cuda')
        dist.all_reduce(ok, op=dist.ReduceOp.SUM)
        n_ok_all = int(ok.item())
    else:
        n_ok_all = n_ok

    if rank == 0:

        total_solves = B*world*args.steps
        value = total_solves/elapsed

        launch_ms = kernel_ms_total/args.steps
        stage_iters = float(N*np.sum(iters))                 # units one launch processes (this rank)
        achieved = BYTES_PER_STAGE_ITER*stage_iters/(launch_ms*1e-3)/1e9

        traffic = None
        tf = ROOT / 'profiles' / 'hbm_traffic.json'
        if tf.exists():
            try:
                traffic = json.loads(tf.read_text()).get('bytes_per_launch')
            except Exception:
                traffic = None

        start_text = ("every solve starts from the device-built speed profile (no information from earlier solves; same optimum as the reference's cold start)"
                      if args.start == 'profile' else "every solve cold-starts from the reference's point (ocp.py:325-339)")

        line = {
            "metric": "OCP solves/sec (N=100, VIRM6, var-speed-limit track)",
            "value": value, "unit": "solves/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3*elapsed/args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("config 1: B={} scenarios per GPU, N={}, track 00_var_speed_limit_100, train NL_Intercity_VIRM6 (JSON defaults, both brakes), "
                                    "RK4 numSteps=1 numApproxSteps=1, v0=vN=1, T_i=1541(1+0.15u_i) seed 20260612+rank, {}, KKT<=1e-8" if args.workload == 'c1' else
                                    "config 2 (extra): B={} scenarios per GPU, N={}, track CH_StGallen_Wil, train NL_Intercity_VIRM6, RK4 numSteps=1 numApproxSteps=1, "
                                    "v0=vN=1, T_i=1242(1+0.15u_i) seed 20260613+rank, {}, KKT<=1e-8").format(B, N, start_text),
                       "batch_per_gpu": B, "num_intervals": N, "start": args.start, "converged": n_ok_all, "scenarios": B*world,
                       "kkt_cycle_share": float(np.sum(st[:, ST['CYC_KKT']])/max(1.0, np.sum(st[:, ST['CYC_TOTAL']]))), "cycles_per_solve_mean": float(np.mean(st[:, ST['CYC_TOTAL']])),
                       "ip_iterations_mean": float(np.mean(iters)), "ip_iterations_max": float(np.max(iters)), "parallelism": "scenarios sharded, no collective"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved/HBM_PEAK_GBS, "traffic": traffic,
                         "model": "S = 904 B x N x sum(IP iterations) per launch (SURVEY 8d); iterate is LDS/register resident, so real HBM traffic is far below S",
                         "kernel": "msd::solve_kernel<64,2,1> (one wave per scenario, two shooting nodes per lane)", "launch_ms": launch_ms, "stage_iterations_per_launch": stage_iters},
        }

        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle
            oprob = cases.oracle_problem(train, track, N)
            ncores = usable_cores()
            # bounded sample of the same workload, sized from a first pass to about 12 s of wall time on all host cores
            t1 = time.perf_counter()
            oracle.solve_batch(oprob, scen, nthreads=ncores, start=args.start)
            first = time.perf_counter() - t1
            reps = int(min(64, max(1, np.ceil(12.0/max(first, 1e-3)))))
            sc = np.tile(scen, (reps, 1))
            sample = sc.shape[0]
            t1 = time.perf_counter()
            zc, stc, nfail = oracle.solve_batch(oprob, sc, nthreads=ncores, start=args.start)
            dt = time.perf_counter() - t1
            line["cpu_baseline"] = {"value": sample/dt, "unit": "solves/s", "cores": ncores, "kind": "port",
                                    "sample": "{} solves (the same batch, repeated), CPU oracle (oracle/ms_oracle.c, same algorithm and starting point, gcc -O2, OpenMP over scenarios), {:.1f} s wall, {} failed".format(sample, dt, nfail)}

        print(json.dumps(line), flush=True)

    prob.free(d_scen); prob.free(d_z); prob.free(d_st)

    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
