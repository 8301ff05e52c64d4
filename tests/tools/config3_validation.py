"""
BASELINE config 3 at full size on one GPU: 65 536 scenarios with perturbed running times and rolling stock (mass, r0, r1, r2),
success rate, iteration statistics, scan fallbacks, and the objective of a random sample against the CPU oracle solving the same NLPs.
"""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R, R + '/ms-eetc_amd']
from mseetc import workloads as wl
from mseetc.ocp import casadiSolver
from mseetc.track import computeDiscretizationPoints
from mseetc._device import ST
from oracle import oracle

B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
train, track, N = wl.config('c3')
solver = casadiSolver(train, track, wl.options(N))
T, pert = wl.c3_scenarios(B, train)
t0 = time.perf_counter()
res = solver.solveBatch(T, **pert)
wall = time.perf_counter() - t0
ok = res['status'] >= 0
print('%d scenarios: %d converged (%.4f %%), kernel %.1f ms (%.0f solves/s), wall %.2f s, iterations mean %.1f max %d, scan fallbacks %d, inertia corrections %d'
      % (B, ok.sum(), 100*ok.mean(), res['kernel_ms'], B/(res['kernel_ms']*1e-3), wall, res['iterations'].mean(), res['iterations'].max(),
         int(res['stats'][:, ST['N_FALLBACK']].sum()), int(res['stats'][:, ST['N_REG']].sum())))
rng = np.random.default_rng(7)
sample = rng.choice(B, 256, replace=False)
pts = computeDiscretizationPoints(track, N)
opts = dict(numIntervals=N, maxIterations=500, energyOptimal=True, minimumVelocity=1, numSteps=1, numApproxSteps=1)
worst = 0.0
for k in sample:
    tr = wl.train_default()
    tr.mass, tr.r0, tr.r1, tr.r2 = pert['mass'][k], pert['r0'][k], pert['r1'][k], pert['r2'][k]
    prob = oracle.pack_problem(tr, pts, opts, 1, (1 - tr.etaTraction)/tr.etaTraction, 1 - tr.etaRgBrake, track.length)
    ref = oracle.solve(prob, prob.scenario(T[k]), start='profile')
    assert ref['stats']['STATUS'] == 0
    worst = max(worst, abs(res['cost'][k] - ref['stats']['OBJ'])/abs(ref['stats']['OBJ']))
print('objective vs oracle on 256 random scenarios: max relative difference %.2e' % worst)
