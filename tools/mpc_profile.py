"""Where the time of a shrinking-horizon run goes (BASELINE config 4: 4096 scenarios x 50 re-solves on one GPU, 512 per GPU on eight).  Needs a GPU."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/ms-eetc_amd']
from mseetc import workloads as wl
from mseetc import mpc
from mseetc.mpc import shrinkingHorizon

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = int(sys.argv[2]) if len(sys.argv) > 2 else 50
train, track, N = wl.config('c4')
opts = wl.options(N)
T = wl.c1_times(B, seed=20260615)
kernel = []
orig = mpc.casadiSolver.solveBatch
def timed(self, *a, **kw):
    r = orig(self, *a, **kw)
    kernel.append(r['kernel_ms'])
    return r
mpc.casadiSolver.solveBatch = timed
shrinkingHorizon(train, track, opts, T[:64], numResolves=2, noise=0.01, seed=1)      # warm up
for warm in (False, True):
    kernel.clear()
    t0 = time.perf_counter()
    log = shrinkingHorizon(train, track, opts, T, numResolves=K, noise=0.01, seed=1, warmStart=warm)
    wall = time.perf_counter() - t0
    n = sum(len(l['status']) for l in log)
    it = np.array([l['iterations'].mean() for l in log]); mx = np.array([l['iterations'].max() for l in log])
    print('warmStart=%s: %d re-solves x %d scenarios in %.3f s -> %.0f re-solves/s; kernels %.1f ms in %d launches (%.0f %% of the wall); iterations per re-solve mean %.1f, '
          'mean of the per-launch maximum %.1f; failures %d' % (warm, len(log), B, wall, n/wall, sum(kernel), len(kernel), 100*sum(kernel)*1e-3/wall, it.mean(), mx.mean(),
                                                             sum(int((l['status'] < 0).sum()) for l in log)))
