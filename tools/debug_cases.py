"""Diagnostic (GPU): a few problem shapes through the product path, with status / iterations / fallback counters."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('ms-eetc_amd', '', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, cases
from mseetc.ocp import casadiSolver
from mseetc._device import ST

def run(name, train, track, N, T, **kw):
    start = kw.pop('start', 'profile')
    s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1), **kw), startingPoint=start)
    r = s.solveBatch(np.atleast_1d(T), **({'initialVelocity': 1.0, 'terminalVelocity': 100/3.6} if name.startswith('mintime') else {}))
    st = r['stats']
    print(name, 'N', N, 'status', st[:, ST['STATUS']].astype(int).tolist(), 'iters', st[:, ST['ITERS']].astype(int).tolist(), 'fallback', st[:, ST['N_FALLBACK']].astype(int).tolist(),
          'nreg', st[:, ST['N_REG']].astype(int).tolist(), 'obj', np.round(st[:, ST['OBJ']], 6).tolist(), flush=True)

for N in (60, 100, 127, 150, 200, 300):
    run('fig10', cases.train_fig10(), cases.track_00(), N, [1541.0, 1600.0])
for N in (100, 300):
    run('mintime', cases.train_fig5(), cases.track_00(8500), N, [400.0], energyOptimal=False)
for N in (100, 200, 300):
    run('default', cases.train_default(), cases.track_00(), N, [1600.0])
