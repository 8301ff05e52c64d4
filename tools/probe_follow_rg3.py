import sys, ctypes
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT/'tests'), str(ROOT/'ms-eetc_amd'), str(ROOT)]
import numpy as np, cases
from mseetc.ocp import casadiSolver
from mseetc._device import ST, lib, _check
train = cases.train_fig10() if sys.argv[1] == 'rg' else cases.train_default()
N, crop = 40, 16000
track = cases.track_00(crop)
T = np.array([float(a) for a in sys.argv[2:]])
s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=800, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='reference', restoration=False)
r = s.solveBatch(T, classifyFailures=False)
out = (ctypes.c_int*40)()
_check(lib().msd_problem_follow_counts(s.problem._h, out, 40))
print('status', r['status'], 'iters', r['iterations'], 'counts', list(out)[:7], 'hdr+entries (from [3])', list(out)[7:30], flush=True)
