"""Diagnostic (GPU): the host shrinking-horizon loop of config 4 (512 x 50), cold and warm; for every re-solve that ends with failed scenarios: which, their
measured state and arrival time, relaxed or not.   usage: debug_c4_step.py [out.npz]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('ms-eetc_amd', '', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
from mseetc import workloads as wl
from mseetc.mpc import shrinkingHorizon
train, track, N = wl.config('c4')
T = wl.c1_times(512, seed=20260615)
out = {}
for warm in (False, True):
    log = shrinkingHorizon(train, track, wl.options(N), T, numResolves=50, noise=0.01, seed=1, warmStart=warm)
    for k, r in enumerate(log):
        bad = np.flatnonzero(r['status'] < 0)
        if bad.size:
            print('warm' if warm else 'cold', 'k', k, 'N', r['numIntervals'], 'position', r['position'], 'failed', bad.tolist(), 'status', r['status'][bad].tolist(),
                  't0', r['t0'][bad].tolist(), 'v0', r['v0'][bad].tolist(), 'T', r['T'][bad].tolist(), 'relaxed', r['relaxed'][bad].tolist(), 'iters', r['iterations'][bad].tolist())
            out['%s_%d' % ('warm' if warm else 'cold', k)] = np.array([[i, r['position'], r['numIntervals'], r['t0'][i], r['v0'][i], r['T'][i], r['status'][i]] for i in bad])
if len(sys.argv) > 1:
    np.savez(sys.argv[1], **out)
