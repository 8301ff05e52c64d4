"""Diagnostic (GPU): kernel time of the config-1 batch (median of 20 launches) with the library MSD_LIB names, the spread of the iteration counts over
the batch, and -- with a -DMSD_TELEMETRY=1 build -- the shader cycles per scenario.  Does not go through bench.py (which builds the product library)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('ms-eetc_amd', '', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, cases
from mseetc.ocp import casadiSolver
from mseetc._device import ST
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
solver = casadiSolver(cases.train_default(), cases.track_00(), dict(numIntervals=N, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1)))
scen = solver._scenarios(cases.c1_times(B), 0, 1, 1)
solver.problem.direct_results(False)      # (kernel time: results copied behind the launch, not stored across the link by the kernels)
ms = []
for k in range(25):
    out = solver.problem.solve_batch(scen)
    ms.append(out['kernel_ms'])
st = out['stats']; it = st[:, ST['ITERS']]; cy = st[:, ST['CYC_TOTAL']]
print(os.environ.get('MSD_LIB', 'product'), 'N', N, 'batch', B, 'kernel ms median %.4f min %.4f' % (np.median(ms[5:]), np.min(ms[5:])), 'solves/s %.0f' % (B/np.median(ms[5:])*1e3),
      'converged', int((st[:, 0] == 0).sum()), 'objective sum %.9f' % st[:, ST['OBJ']].sum())
print('iterations: mean %.2f min %d max %d; histogram' % (it.mean(), it.min(), it.max()), np.bincount(it.astype(int))[int(it.min()):].tolist())
if cy.max() > 0:
    print('cycles per scenario: mean %.0f max %.0f; per iteration mean %.0f; scenario of max cycles has %d iterations' % (cy.mean(), cy.max(), (cy/it).mean(), it[np.argmax(cy)]))
