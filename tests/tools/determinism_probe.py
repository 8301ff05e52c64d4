"""Diagnostic (GPU; MSD_LIB selects the library): one random problem of the sweep from the reference's starting point, six launches of the same five running times --
status, iterations and residuals of every launch (a library is deterministic when the six lines agree), multipliers of a failed solve.   usage: determinism_probe.py SEED 0"""
import os, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
for p in ('ms-eetc_amd', '', 'tests'):
    sys.path.insert(0, str(ROOT / p))
import numpy as np
import cases
from oracle import oracle
from mseetc._device import ST
from test_gpu_parity import _random_problem, _solver
np.set_printoptions(linewidth=220, precision=4)
seed, k = int(sys.argv[1]), int(sys.argv[2])
factors = [1.05, 1.1, 1.2, 1.45, 2.0] if len(sys.argv) < 4 else [float(x) for x in sys.argv[3].split(',')]      # (third argument: the running times, as multiples of the minimum)
with tempfile.TemporaryDirectory() as tmp:
    train, track, N, rng = _random_problem(seed, Path(tmp))
    v0, vN = float(rng.uniform(1, 15)), float(rng.uniform(1, 15))
    po = cases.oracle_problem(train, track, N, energyOptimal=False, losses='none')
    tmin = float(oracle.solve(po, po.scenario(3*track.length/train.velocityMax, 0.0, vN, v0), start='profile')['z'][-2])
    s = _solver(train, track, N, start='reference')
    for rep in range(6):
        res = s.solveBatch(tmin*np.array(factors), initialVelocity=v0, terminalVelocity=vN, multipliers=True)
        st = res['stats']
        print('rep', rep, 'status', res['status'], 'iters', res['iterations'], 'kkt', st[:, ST['KKT']], 'dual', st[:, ST['DUAL_INF']], 'viol', st[:, ST['CONSTR_VIOL']])
        bad = np.flatnonzero(res['status'] < 0)
        for b in bad[:1]:
            lam = res['lam_g'][b]; z = res['z'][b]
            print('   scenario', b, 'N', N, 'pn', train.forceMinPn, 'max |lam_g| %.3e at %d of %d' % (np.nanmax(np.abs(lam)), int(np.nanargmax(np.abs(lam))), lam.size), 'nan in lam', int(np.isnan(lam).sum()),
                  'max |z| %.3e' % np.nanmax(np.abs(z)), 'nan in z', int(np.isnan(z).sum()))
            rows = lam.reshape(N, -1) if lam.size % N == 0 else None
            if rows is not None:
                print('   max |lam_g| per row kind', np.nanmax(np.abs(rows), axis=0), ' interval of the maximum per kind', np.nanargmax(np.abs(rows), axis=0))
    s.close()
