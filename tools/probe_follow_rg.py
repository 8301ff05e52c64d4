"""Probe: the follow-up kernels of the one-brake family on the 64 x 1 geometry (N <= 63): loose schedules from the reference's start (restoration phase)."""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT/'tests'), str(ROOT/'ms-eetc_amd'), str(ROOT)]
import numpy as np, cases
from mseetc.ocp import casadiSolver
from mseetc._device import ST
which = sys.argv[1] if len(sys.argv) > 1 else 'rg'
train = cases.train_fig10() if which == 'rg' else cases.train_default()
for N, crop in ((40, 16000), (63, 30000)):
    track = cases.track_00(crop)
    for start in ('profile', 'reference'):
        s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=800, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint=start)
        T = np.array([700.0, 2500.0, 6000.0, 9000.0, 14000.0])*(crop/16000.0)
        print(which, N, start, 'launching', flush=True)
        r = s.solveBatch(T, classifyFailures=False)
        print(which, N, start, 'status', r['status'], 'iters', r['iterations'], 'resto', r['stats'][:, ST['N_RESTO']], 'follow', s.problem.follow_counts(), flush=True)
        s.close()
