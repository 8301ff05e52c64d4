"""
How closely does the device follow the oracle iteration for iteration?  (GPU box; checker-side tool.)  Config 1 / config 2 / figure-10 scenarios from
both starting points: histogram of (GPU iterations - oracle iterations) per scenario and the largest objective deviation.
usage: iteration_parity.py [scenarios = 256]
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'ms-eetc_amd')); sys.path.insert(0, str(ROOT / 'tests'))
import cases                                   # noqa: E402
from oracle import oracle                      # noqa: E402
from mseetc import workloads                   # noqa: E402
from mseetc.ocp import casadiSolver            # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for label, train, track, N, T in (('config 1', *workloads.config('c1'), workloads.c1_times(B)),
                                  ('config 2', *workloads.config('c2'), workloads.c2_times(max(B//4, 16))),
                                  ('figure 10 train', cases.train_fig10(), workloads.track_00(), 100, workloads.c1_times(B, seed=7))):
    prob = cases.oracle_problem(train, track, N)
    scen = np.array([[0.0, t, 1.0, 1.0] for t in T])
    for start in ('profile', 'reference'):
        s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint=start)
        res = s.solveBatch(T)
        s.close()
        z, st, nfail = oracle.solve_batch(prob, scen, start=start)
        diff = (res['iterations'] - st[:, oracle.ST['ITERS']]).astype(int)
        vals, counts = np.unique(diff, return_counts=True)
        dev = np.max(np.abs(res['cost'] - st[:, oracle.ST['OBJ']])/np.abs(st[:, oracle.ST['OBJ']]))
        print('{:<16s} {:<9s} scenarios {:4d}  GPU - oracle iterations: {}  largest objective deviation {:.1e}  failed {} / {}'.format(
            label, start, len(T), dict(zip(vals.tolist(), counts.tolist())), dev, int(np.sum(res['status'] != 0)), int(nfail)), flush=True)
