"Probe (GPU box): very loose schedules of config 1 from both starting points, with and without the restoration phase, against the oracle."
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'ms-eetc_amd')); sys.path.insert(0, str(ROOT / 'tests'))
import numpy as np
import cases
from oracle import oracle
from mseetc.ocp import casadiSolver
from mseetc._device import ST
train, track = cases.train_default(), cases.track_00()
T = [8000.0, 12000.0, 20000.0, 900.0, 1455.0]
prob = cases.oracle_problem(train, track, 100)
for start in ('profile', 'reference'):
    for resto in (True, False):
        s = casadiSolver(train, track, dict(numIntervals=100, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint=start, restoration=resto)
        r = s.solveBatch(T, classifyFailures=False)
        s.close()
        print(start, 'restoration', resto, 'status', r['status'], 'iters', r['iterations'], 'n_resto', r['stats'][:, ST['N_RESTO']].astype(int), 'cost', np.round(r['cost'], 6), 'ms', r['kernel_ms'])
    oracle.lib().oracle_set_restoration(1)
    for t in T:
        o = oracle.solve(prob, prob.scenario(t), start=start)['stats']
        print('   oracle', start, t, int(o['STATUS']), int(o['ITERS']), int(o['N_RESTO']), round(o['OBJ'], 6))
