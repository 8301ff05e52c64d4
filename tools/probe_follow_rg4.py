import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT/'tests'), str(ROOT/'ms-eetc_amd'), str(ROOT)]
import numpy as np, cases
from mseetc.ocp import casadiSolver
from mseetc._device import ST
train = cases.train_fig10()
N, crop = int(sys.argv[1]), 16000
track = cases.track_00(crop)
T = np.array([float(a) for a in sys.argv[2:]])
s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=800, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='reference', restoration=False)
r = s.solveBatch(T, classifyFailures=False)
print('N', N, 'geometry', s.problem.geometry(), 'status', r['status'], 'iters', r['iterations'], 'nreg', r['stats'][:, ST['N_REG']], 'follow', s.problem.follow_counts(), flush=True)
