"Probe (GPU box): loose schedules on other rolling stock / horizons, both starting points, with and without the restoration phase."
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'ms-eetc_amd')); sys.path.insert(0, str(ROOT / 'tests'))
import numpy as np
import cases
from mseetc.ocp import casadiSolver
from mseetc._device import ST
T = [6000.0, 9000.0, 14000.0]
for N, variant in ((63, 'both'), (127, 'rg'), (255, 'both'), (300, 'rg'), (450, 'both')):
    train = cases.train_default() if variant == 'both' else cases.train_fig10()
    track = cases.track_00()
    for start in ('profile', 'reference'):
        for resto in (True, False):
            s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=800, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint=start, restoration=resto)
            r = s.solveBatch(T, classifyFailures=False)
            s.close()
            print(N, variant, start, 'restoration', resto, 'status', r['status'], 'iters', r['iterations'], 'n_resto', r['stats'][:, ST['N_RESTO']].astype(int), 'cost', np.round(r['cost'], 6))
