"""Which re-solves of the config-4 loop fail, and how (status codes per re-solve index)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent / 'ms-eetc_amd'))
import numpy as np
from mseetc import workloads as wl
from mseetc.mpc import shrinkingHorizon
train, track, N = wl.config('c4')
T = wl.c1_times(512, seed=20260615)
log = shrinkingHorizon(train, track, wl.options(N), T, numResolves=50, noise=0.01, seed=1, warmStart=True, device=0)
print('keys', list(log[0].keys()))
for k,l in enumerate(log):
    st=l['status']; bad=st[st<0]
    if len(bad): print(k, 'N', l.get('numIntervals'), 'failed', len(bad), 'codes', dict(zip(*np.unique(bad, return_counts=True))))
