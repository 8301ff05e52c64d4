"""Diagnostic (GPU): one re-solve of the config-4 loop -- re-solve K of the shrinking horizon from the measured state (t0, v0) with arrival time T -- from both
starting points with the iteration log; the time-optimal twin too.   usage: debug_c4_resolve.py K t0 v0 T [rows]"""
import os, sys, copy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('ms-eetc_amd', '', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
from mseetc import workloads as wl
from mseetc.track import computeDiscretizationPoints
from mseetc.ocp import casadiSolver
from mseetc._device import ST
np.set_printoptions(linewidth=200)
kt = int(sys.argv[1]); t0 = float(sys.argv[2]); v0 = float(sys.argv[3]); T = float(sys.argv[4]); rows = int(sys.argv[5]) if len(sys.argv) > 5 else 60
train, track, N = wl.config('c4')
cur = copy.deepcopy(track)
for k in range(kt):
    pts = computeDiscretizationPoints(cur, N - 2*k); nxt = copy.deepcopy(cur); nxt.updateLimits(positionStart=float(pts.index.values[2])); cur = nxt
Nk = N - 2*kt
for eo in (True, False):
    for start in ('profile', 'reference'):
        opts = wl.options(Nk); opts['energyOptimal'] = eo
        s = casadiSolver(train, cur, opts, startingPoint=start, restoration=False, watchdogTrigger=-1)
        Tl = T if eo else t0 + max(3*cur.length/train.velocityMax, 3*(T - t0))
        scen = s._scenarios([Tl], t0, 1.0, v0)
        batch = np.repeat(scen, int(os.environ.get('BATCH', '1')), axis=0)
        out = s.problem.solve_batch(batch, history=200)
        st = out['stats'][0]
        print('energy' if eo else 'twin', start, 'status', st[ST['STATUS']], 'iters', st[ST['ITERS']], 'obj', st[2], 'kkt', st[3], 'nreg', st[ST['N_REG']], 'nsoc', st[ST['N_SOC']], 'nback', st[ST['N_BACKTRACK']],
              'tN', out['z'][0][-2] - t0, 'all statuses', np.unique(out['stats'][:, 0]), flush=True)
        h = out['hist']
        for r in h[:min(rows, int(st[ST['ITERS']]) + 1)]:
            print('   ' + ' '.join('%12.5e' % v for v in r))
        s.close()
