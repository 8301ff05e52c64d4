"""
The dynamic-loss bench workload (figure-5 configuration, bench.py: alt.dynamic_losses_N*): which running times do not converge, and how.
    python tools/dyn_probe.py [N ...]
"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT/'ms-eetc_amd'), str(ROOT)]
import numpy as np
from mseetc import workloads as wl
from mseetc.train import Train
from mseetc.efficiency import totalLossesFunction
from mseetc.ocp import casadiSolver
from mseetc._device import ST
for N in [int(a) for a in sys.argv[1:]] or [100, 300]:
    tr = Train(config={'id': 'NL_Intercity_VIRM6'}); tr.forceMinPn = 0
    tr.powerLosses = totalLossesFunction(tr, auxiliaries=27000, etaGear=0.96)
    for start in ('profile', 'reference'):
        sv = casadiSolver(tr, wl.track_00(8500), wl.options(N), startingPoint=start)
        u = np.random.default_rng(20260616).random(1024)
        T = 272.4726*(1.05 + 0.25*u)
        r = sv.solveBatch(T, terminalVelocity=100/3.6, initialVelocity=1, classifyFailures=False)
        bad = np.flatnonzero(r['status'] < 0)
        print('N', N, start, 'kernel_ms %.1f' % r['kernel_ms'], 'failed', len(bad), 'iters mean %.1f max %d' % (r['iterations'].mean(), r['iterations'].max()),
              [(round(float(T[k]/272.4726), 5), int(r['status'][k]), int(r['iterations'][k]), int(r['stats'][k, ST['N_BACKTRACK']])) for k in bad[:16]])
        sv.close()
