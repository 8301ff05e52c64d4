"""Diagnostic (GPU): iteration log of one fig10 / reference-start solve (columns: iter obj primal dual log10(mu) |d| alpha_du alpha_pr)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('ms-eetc_amd', '', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, cases
from mseetc.ocp import casadiSolver
from mseetc._device import ST
np.set_printoptions(linewidth=200, precision=6, suppress=False)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
s = casadiSolver(cases.train_fig10(), cases.track_00(), dict(numIntervals=N, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='reference')
scen = s._scenarios([1541.0], 0, 1, 1)
out = s.problem.solve_batch(scen, history=64)
st = out['stats'][0]
print('status', st[ST['STATUS']], 'iters', st[ST['ITERS']], 'fallbacks', st[ST['N_FALLBACK']], 'nreg', st[ST['N_REG']], 'nsoc', st[ST['N_SOC']], 'nback', st[ST['N_BACKTRACK']])
h = out['hist']
for r in h:
    print(' '.join('%13.6e' % v for v in r))
