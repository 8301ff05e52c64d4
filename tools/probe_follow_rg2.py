import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT/'tests'), str(ROOT/'ms-eetc_amd'), str(ROOT)]
import numpy as np, cases
from mseetc.ocp import casadiSolver
from mseetc._device import ST
train = cases.train_fig10()
N, crop = 40, 16000
track = cases.track_00(crop)
mode = sys.argv[1]
T = np.array([float(a) for a in sys.argv[2:]])
s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=800, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='reference', restoration=(mode == 'resto'))
print(mode, T, 'launching', flush=True)
r = s.solveBatch(T, classifyFailures=False)
print(mode, 'status', r['status'], 'iters', r['iterations'], 'resto', r['stats'][:, ST['N_RESTO']], 'follow', s.problem.follow_counts(), flush=True)
