"""
Where does IPOPT's watchdog start on these problems?  (CPU only; checker-side tool.)

IPOPT starts its watchdog procedure after `watchdog_shortened_iter_trigger` = 10 successive iterations whose step the backtracking line
search shortened (IpBacktrackingLineSearch.cpp).  Round 4 restates the procedure in the oracle (oracle/ms_oracle.c: solve_core) and in the
general iteration of the kernels (msd_kernel.hpp: Solver::run); this survey runs the oracle over the benchmark workloads, loose schedules and
random problems from both starting points and reports, per group: the longest run of successive iterations with a backtracking step (round 3's
telemetry), the procedures started, those ended by an accepted trial point, and the trial points taken without the filter's consent.
The device side of the same rows: tests/test_watchdog.py (GPU tests), tests/tools/random_sweep.py.

usage: watchdog_survey.py [scenarios per workload = 256] [random problems = 40]
"""
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'ms-eetc_amd')); sys.path.insert(0, str(ROOT / 'tests'))
import cases                                   # noqa: E402
from oracle import oracle                      # noqa: E402
from mseetc import workloads                   # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
NRANDOM = int(sys.argv[2]) if len(sys.argv) > 2 else 40

worst = 0


def group(label, prob, scen):
    global worst
    for start in ('profile', 'reference'):
        oracle.max_shortened_run(reset=True); oracle.watchdog_counts(True); oracle.watchdog_forced_steps(True)
        z, st, nfail = oracle.solve_batch(prob, scen, start=start)
        run = oracle.max_shortened_run(reset=True)
        started, succeeded = oracle.watchdog_counts(True)
        forced = oracle.watchdog_forced_steps(True)
        worst = max(worst, run)
        it = st[:, oracle.ST['ITERS']]
        print('{:<34s} {:<9s} solves {:5d}  failed {:3d}  iterations {:5.1f} (max {:3.0f})  backtracking steps / solve {:5.2f}  longest shortened run {:2d}  watchdog started {:3d} succeeded {:3d} forced steps {:3d}  restoration phases {:3d}'.format(
            label, start, len(scen), int(nfail), it.mean(), it.max(), st[:, oracle.ST['N_BACKTRACK']].mean(), run, started, succeeded, forced, int(st[:, oracle.ST['N_RESTO']].sum())), flush=True)


def scenarios(times, v0=1.0, vN=1.0):
    return np.array([[0.0, T, v0*v0, vN*vN] for T in times])


train, track, N = workloads.config('c1')
group('config 1 (N = 100, track 00)', cases.oracle_problem(train, track, N), scenarios(workloads.c1_times(B)))
train, track, N = workloads.config('c2')
group('config 2 (N = 200, CH_StGallen_Wil)', cases.oracle_problem(train, track, N), scenarios(workloads.c2_times(max(B//4, 16))))
group('figure 10 train, N = 100', cases.oracle_problem(cases.train_fig10(), workloads.track_00(), 100), scenarios(workloads.c1_times(B, seed=7)))
group('config 1, loose schedules', cases.oracle_problem(*workloads.config('c1')), scenarios(np.linspace(3000, 20000, 64)))
for Nl in (300, 600, 700, 1200):      # the same track on longer horizons (192 x 2, 320 x 2 + streamed follow-up, streamed kernels)
    group('loose schedules, N = {}'.format(Nl), cases.oracle_problem(workloads.train_default(), workloads.track_00(), Nl, maxIterations=800), scenarios(np.linspace(3000, 20000, 16)))

# config 3: perturbed rolling stock, one oracle problem per scenario
from mseetc.track import computeDiscretizationPoints      # noqa: E402
train, track, N = workloads.config('c3')
T3, pert = workloads.c3_scenarios(max(B//2, 32), train)
pts = computeDiscretizationPoints(track, N)
opts = dict(numIntervals=N, maxIterations=500, energyOptimal=True, minimumVelocity=1, numSteps=1, numApproxSteps=1)
for start in ('profile', 'reference'):
    oracle.max_shortened_run(reset=True)
    its = []
    for k in range(len(T3)):
        tr = workloads.train_default()
        tr.mass, tr.r0, tr.r1, tr.r2 = pert['mass'][k], pert['r0'][k], pert['r1'][k], pert['r2'][k]
        prob = oracle.pack_problem(tr, pts, opts, 1, (1 - tr.etaTraction)/tr.etaTraction, 1 - tr.etaRgBrake, track.length)
        r = oracle.solve(prob, prob.scenario(float(T3[k])), start=start)
        its.append(r['stats']['ITERS'])
    run = oracle.max_shortened_run(reset=True)
    worst = max(worst, run)
    print('{:<34s} {:<9s} solves {:5d}  iterations {:5.1f} (max {:3.0f})  longest shortened run {:2d}'.format('config 3 (perturbed rolling stock)', start, len(T3), np.mean(its), np.max(its), run), flush=True)

# config 4 in miniature: shrinking-horizon re-solves of 16 scenarios from the state the previous solution reaches 16 intervals on (cold starts)
train, track, N = workloads.config('c4')
runs4 = []
for T in workloads.c1_times(16, seed=20260615):
    cur = workloads.track_00()
    t_now, v_now, pos, prev = 0.0, 1.0, 0.0, None
    for k in range(0, 40, 8):
        Nk = N - 2*k
        prob = cases.oracle_problem(train, cur, Nk)
        oracle.max_shortened_run(reset=True)
        r = oracle.solve_dual(prob, prob.scenario(float(T), t_now, 1.0, v_now), start='profile')
        runs4.append(oracle.max_shortened_run(reset=True))
        if r['stats']['STATUS'] != 0:
            break
        z = r['z']; stp = 5; j = 16
        t_now, v_now = float(z[stp*j + 3]), float(np.sqrt(z[stp*j + 4]))
        cur = workloads.track_00(); pos += float(prob.positions[j]); cur.updateLimits(positionStart=pos)
worst = max([worst] + runs4)
print('config 4 in miniature (16 scenarios x 5 cold re-solves on the shrinking horizon): longest shortened run {}'.format(max(runs4)), flush=True)

if NRANDOM:
    from test_gpu_parity import _random_problem      # noqa: E402  (the generator only: no GPU call)
    runs = []
    for seed in range(NRANDOM):
        with tempfile.TemporaryDirectory() as tmp:
            train, track, N, rng = _random_problem(seed, Path(tmp))
            v0, vN = float(rng.uniform(1, 15)), float(rng.uniform(1, 15))
            # running times from the minimum-time solve of the oracle
            pt = cases.oracle_problem(train, track, N, energyOptimal=False)
            rt = oracle.solve(pt, pt.scenario(3*track.length/train.velocityMax, 0.0, vN, v0), start='profile')
            if rt['stats']['STATUS'] != 0:
                continue
            tmin = float(rt['z'][-2])
            pe = cases.oracle_problem(train, track, N)
            for start in ('profile', 'reference'):
                oracle.max_shortened_run(reset=True)
                oracle.solve_batch(pe, scenarios(tmin*np.array([1.05, 1.1, 1.2, 1.45, 2.0]), v0, vN), start=start)
                runs.append(oracle.max_shortened_run(reset=True))
    worst = max([worst] + runs)
    print('{} random problems x 5 running times x 2 starts: longest shortened run {} (histogram of the per-batch maxima: {})'.format(
        NRANDOM, max(runs), np.bincount(runs).tolist()))

print('longest run of successive iterations with a backtracking step over everything:', worst, '(the watchdog counts iterations with MORE than one backtracking step and starts at 10 of them)')
