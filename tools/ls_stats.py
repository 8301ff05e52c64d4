"""Diagnostic (GPU): how the line search behaves over the config-1 batch -- iterations, backtracking steps, second-order corrections and regularised
iterations per solve, and how many scenarios the first pass hands to the follow-up kernel.  Sizes what a change to the trial-point evaluation can win."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('ms-eetc_amd', '', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, cases
from mseetc.ocp import casadiSolver
from mseetc._device import ST
for N in (100, 200):
    solver = casadiSolver(cases.train_default(), cases.track_00(), dict(numIntervals=N, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1)))
    T = cases.c1_times(1024)
    out = solver.problem.solve_batch(solver._scenarios(T, 0, 1, 1))
    st = out['stats']
    it, nb, ns, nr, nf = (st[:, ST[k]] for k in ('ITERS', 'N_BACKTRACK', 'N_SOC', 'N_REG', 'N_FALLBACK'))
    print("N", N, "iters mean %.2f" % it.mean(), "backtracking steps per solve %.3f" % nb.mean(), "solves with any %.1f %%" % (100*(nb > 0).mean()),
          "soc %.3f" % ns.mean(), "regularised %.3f" % nr.mean(), "follow-up %.1f %%" % (100*(nf > 0).mean()), "share of iterations with a shortened step <= %.2f %%" % (100*nb.sum()/it.sum()))
