"""
Kernel time per launch over a range of horizons (the figure-10 train of simulations/table3.py and the JSON-default train):
    python tools/horizon_timing.py [batch]          (MSD_LIB selects a tuning build)
"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/tests', R, R + '/ms-eetc_amd']
import numpy as np, cases
from mseetc.ocp import casadiSolver
B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1024
ONLY = [int(v) for a in sys.argv if a.startswith('--only=') for v in a[7:].split(',')]      # e.g. --only=700,1000
for variant in ('both', 'fig10'):
    train = cases.train_default() if variant == 'both' else cases.train_fig10()
    track = cases.track_00()
    for N, b in ((100, B), (200, B), (300, B), (383, B), (450, B), (511, B), (560, B), (600, B), (639, B), (700, max(1, B//16)), (700, B), (1000, max(1, B//16)), (1000, B), (2000, max(1, B//16))):
        if ONLY and N not in ONLY:
            continue
        solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=1000, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='profile')
        solver.problem.direct_results(False)      # (kernel time)
        T = 1541*(1 + 0.15*np.random.default_rng(N).random(b))
        scen = solver._scenarios(T, 0, 1, 1)
        solver.problem.solve_batch(scen)
        ms = [solver.problem.solve_batch(scen)['kernel_ms'] for _ in range(3)]
        out = solver.problem.solve_batch(scen)
        st = out['stats']
        print('%-6s N %4d geometry %-10s batch %5d  kernel %9.3f ms  %9.0f solves/s  iters %.1f  converged %d' % (variant, N, solver.problem.geometry(), b, np.mean(ms), b/np.mean(ms)*1e3, st[:, 1].mean(), (st[:, 0] >= 0).sum()), flush=True)
        solver.close()
