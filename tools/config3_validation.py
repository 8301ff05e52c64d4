"""
BASELINE config 3 at full size on one GPU: 65 536 scenarios with perturbed running times and rolling stock (mass, r0, r1, r2),
success rate, iteration statistics, and the objective of a random sample against the CPU oracle solving the same NLPs.
"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/tests', R, R + '/ms-eetc_amd']
import cases
from mseetc.ocp import casadiSolver
from oracle import oracle
from oracle.oracle import DP

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
train, track = cases.train_default(), cases.track_00()
opts = dict(numIntervals=100, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1))
solver = casadiSolver(train, track, opts)
rng = np.random.default_rng(20260614)
T = 1541*(1 + 0.15*rng.random(B))
n = np.clip(rng.standard_normal((4, B)), -2, 2)
mass, r0, r1, r2 = train.mass*(1 + 0.05*n[0]), train.r0*(1 + 0.05*n[1]), train.r1*(1 + 0.05*n[2]), train.r2*(1 + 0.05*n[3])
t0 = time.perf_counter()
res = solver.solveBatch(T, mass=mass, r0=r0, r1=r1, r2=r2)
wall = time.perf_counter() - t0
ok = res['status'] >= 0
print('%d scenarios: %d converged (%.4f %%), kernel %.1f ms (%.0f solves/s), wall %.2f s, iterations mean %.1f max %d'
      % (B, ok.sum(), 100*ok.mean(), res['kernel_ms'], B/(res['kernel_ms']*1e-3), wall, res['iterations'].mean(), res['iterations'].max()))
sample = rng.choice(B, 256, replace=False)
worst = 0.0
for k in sample:
    tr = cases.train_default()
    tr.mass, tr.r0, tr.r1, tr.r2 = mass[k], r0[k], r1[k], r2[k]
    prob = cases.oracle_problem(tr, track, 100)
    ref = oracle.solve(prob, prob.scenario(T[k]), start='profile')
    assert ref['stats']['STATUS'] == 0
    worst = max(worst, abs(res['cost'][k] - ref['stats']['OBJ'])/abs(ref['stats']['OBJ']))
print('objective vs oracle on 256 random scenarios: max relative difference %.2e' % worst)
