"""
Every launch geometry against the oracle on batches whose horizon is not a multiple of the slot count (the last wave has idle node
slots): tests/tools/geometry_sweep.py [scenarios per horizon]      (MSD_LIB selects a tuning build)
"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R + '/tests', R, R + '/ms-eetc_amd']
import numpy as np, cases
from mseetc.ocp import casadiSolver
from oracle import oracle
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
bad = 0
for variant in ('both', 'fig10'):
    train = cases.train_default() if variant == 'both' else cases.train_fig10()
    track = cases.track_00()
    for N in (60, 100, 127, 209, 255, 300, 383, 450, 500, 511, 530, 560):
        solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='profile')
        T = 1541*(1 + 0.15*np.random.default_rng(N).random(B))
        res = solver.solveBatch(T)
        prob = cases.oracle_problem(train, track, N)
        scen = np.stack([prob.scenario(float(t))[[oracle.DP['T0'], oracle.DP['TEND'], oracle.DP['V0SQ'], oracle.DP['VNSQ']]] for t in T])
        z, st, nfail = oracle.solve_batch(prob, scen, nthreads=0, start='profile')
        dobj = np.abs(res['cost'] - st[:, 2])/np.abs(st[:, 2])
        dit = np.abs(res['iterations'] - st[:, 1])
        same_mu = np.abs(res['stats'][:, 4] - st[:, 4]) <= 1e-3*st[:, 4]
        worst = float(np.max(np.where(same_mu, dobj, 0))) if same_mu.any() else 0.0
        flag = (res['status'] != 0).any() or worst > 1e-8 or float(np.max(dobj)) > 1e-7 or dit.max() > 2
        bad += int(flag)
        print('%-6s N %3d geometry %s  converged %d/%d  max |dobj| %.1e (same final mu: %.1e)  max |diters| %d  fallbacks %d %s' % (variant, N, solver.problem.geometry(), (res['status'] == 0).sum(), B, dobj.max(), worst, dit.max(),
              int(res['stats'][:, 13].sum()), 'MISMATCH' if flag else ''))
        solver.close()
print('mismatching horizons:', bad)
