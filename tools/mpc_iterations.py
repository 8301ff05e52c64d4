"""Iteration statistics of the warm-started config-4 loop per re-solve: mean, 90 %, 99 %, max, and how many scenarios exceed twice the mean."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / 'ms-eetc_amd'))
import numpy as np
from mseetc import workloads as wl
from mseetc.mpc import shrinkingHorizon
train, track, N = wl.config('c4')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = wl.c1_times(B, seed=20260615)
log = shrinkingHorizon(train, track, wl.options(N), T, numResolves=50, noise=0.01, seed=1, warmStart=True)
for k, l in enumerate(log):
    it = l['iterations']
    print('%2d N %3d mean %5.1f  p90 %3d  p99 %3d  max %3d  >2*mean %3d  failed %3d' % (k, l['numIntervals'], it.mean(), np.percentile(it, 90), np.percentile(it, 99), it.max(),
                                                                                   int((it > 2*it.mean()).sum()), int((l['status'] < 0).sum())))
