"""Where the time of a shrinking-horizon run goes (BASELINE config 4 per GPU: 4096 scenarios, 50 re-solves).  Needs a GPU."""
import os, sys, time, cProfile, pstats
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/tests', R, R + '/ms-eetc_amd']
import cases
from mseetc.mpc import shrinkingHorizon

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
train, track = cases.train_default(), cases.track_00()
opts = dict(numIntervals=100, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1))
T = cases.c1_times(B)
shrinkingHorizon(train, track, opts, T[:64], numResolves=2, noise=0.01, seed=1)      # warm up
for warm in (False, True):
    t0 = time.time()
    pr = cProfile.Profile(); pr.enable()
    log = shrinkingHorizon(train, track, opts, T, numResolves=K, noise=0.01, seed=1, warmStart=warm)
    pr.disable()
    wall = time.time() - t0
    n = sum(len(l['status']) for l in log)
    print('warmStart=%s: %d re-solves x %d scenarios in %.3f s -> %.0f re-solves/s; iterations per re-solve %.1f; failures %d'
          % (warm, len(log), B, wall, n/wall, np.mean([l['iterations'].mean() for l in log]), sum(int((l['status'] < 0).sum()) for l in log)))
    pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
