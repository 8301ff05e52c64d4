"""Diagnostic (GPU): kernel time of the figure-5 configuration with the dynamic loss model (bench.py: alt.dynamic_losses_N100 / N300) with the library MSD_LIB names.
usage: dyn_time.py [N] [batch]   (MSD_GEOMETRY2=64x2: two nodes per lane where the picker would take 128 x 1)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('ms-eetc_amd', '', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
from mseetc import workloads as wl
from mseetc.train import Train
from mseetc.efficiency import totalLossesFunction
from mseetc.ocp import casadiSolver
from mseetc._device import ST, lib
lib().msd_tuning(b'two_nodes_per_lane', int(os.environ.get('MSD_GEOMETRY2') == '64x2'))
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
tr = Train(config={'id': 'NL_Intercity_VIRM6'}); tr.forceMinPn = 0
tr.powerLosses = totalLossesFunction(tr, auxiliaries=27000, etaGear=0.96)
sv = casadiSolver(tr, wl.track_00(8500), wl.options(N))
Td = 272.4726*(1.05 + 0.25*np.random.default_rng(20260616).random(B))
sc = sv._scenarios(Td, 0, 100/3.6, 1)
sv.problem.direct_results(False)      # (kernel time)
first = sv.problem.solve_batch(sc)
good = first['stats'][:, ST['STATUS']] >= 0
scg = np.ascontiguousarray(sc[good])
ms = [sv.problem.solve_batch(scg)['kernel_ms'] for _ in range(8)]
out = sv.problem.solve_batch(scg); st = out['stats']
print(os.path.basename(os.environ.get('MSD_LIB', 'product')), os.environ.get('MSD_GEOMETRY2', ''), 'N', N, 'geometry', sv.problem.geometry(), 'batch', int(good.sum()), 'of', B,
      'kernel ms median %.4f min %.4f' % (np.median(ms[2:]), np.min(ms[2:])), 'solves/s %.0f' % (good.sum()/np.median(ms[2:])*1e3), 'iters %.2f' % st[:, ST['ITERS']].mean(),
      'objective sum %.9f' % st[:, ST['OBJ']].sum(), 'first launch ms %.2f' % first['kernel_ms'])
sv.close()
