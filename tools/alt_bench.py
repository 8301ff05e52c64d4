"""
The other transcriptions of the reference's options on the config-1 batch (bench.py's alt entries), a few launches each:
    python tools/alt_bench.py [batch]
"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/ms-eetc_amd', R]
import numpy as np
from mseetc import workloads as wl
from mseetc.ocp import casadiSolver
from mseetc._device import lib
# (the pickers' tuning switches go through the ABI -- msd_tuning of include/mseetc_aux.h; this script keeps its two environment knobs for A/B runs)
lib().msd_tuning(b'two_nodes_per_lane', int(os.environ.get('MSD_GEOMETRY2') == '64x2'))
lib().msd_tuning(b'no_full', int(os.environ.get('MSD_NO_FULL') == '1'))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
train, track, N = wl.config('c1')
T = wl.c1_times(B, seed=20260612)
for name, extra, io in (("static", dict(), dict(numSteps=1, numApproxSteps=1)),
                        ("integrate_losses", dict(integrateLosses=True), dict(numSteps=1, numApproxSteps=1)),
                        ("irk_radau2", dict(integrationMethod='IRK'), dict(order=2, numSteps=1, numApproxSteps=1)),
                        ("cvodes_tolerances", dict(integrationMethod='CVODES'), dict())):
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationOptions=io, **extra), startingPoint='profile')
    scen = solver._scenarios(T, 0, 1, 1)
    solver.problem.direct_results(False)      # (kernel time)
    solver.problem.solve_batch(scen)
    ms = []
    for _ in range(5):
        out = solver.problem.solve_batch(scen)
        ms.append(out['kernel_ms'])
    st = out['stats']
    print('%-18s geometry %s  kernel %.3f ms (min %.3f)  %.0f solves/s  iters %.2f  converged %d/%d' % (name, os.environ.get('MSD_GEOMETRY2', 'default') + ('/nofull' if os.environ.get('MSD_NO_FULL') == '1' else ''),
          np.mean(ms), np.min(ms), B/np.mean(ms)*1e3, st[:, 1].mean(), int((st[:, 0] >= 0).sum()), B))
    solver.close()
