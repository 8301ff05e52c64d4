"""Watchdog / restoration counts on loose schedules of long horizons: device against the oracle (GPU box)."""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'ms-eetc_amd')); sys.path.insert(0, str(ROOT / 'tests'))
import cases
from oracle import oracle
from mseetc.ocp import casadiSolver
from mseetc._device import ST

train, track = cases.train_default(), cases.track_00()
for N in (300, 600, 700, 1200):
    Ts = list(np.linspace(3000, 20000, 16))
    s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=800, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='reference')
    res = s.solveBatch(Ts)
    ms = res['kernel_ms']
    s.close()
    prob = cases.oracle_problem(train, track, N, maxIterations=800)
    z, st, nfail = oracle.solve_batch(prob, np.array([[0.0, T, 1.0, 1.0] for T in Ts]), start='reference')
    print('N', N, 'device: converged', int(np.sum(res['status'] >= 0)), 'iterations', res['iterations'].tolist(), 'watchdog', res['stats'][:, ST['N_WATCHDOG']].astype(int).tolist(),
          'restoration', res['stats'][:, ST['N_RESTO']].astype(int).tolist(), 'kernel ms', round(float(ms), 1))
    print('N', N, 'oracle: failed', int(nfail), 'iterations', st[:, oracle.ST['ITERS']].astype(int).tolist(), 'watchdog', st[:, oracle.ST['N_WATCHDOG']].astype(int).tolist(),
          'restoration', st[:, oracle.ST['N_RESTO']].astype(int).tolist(), 'max cost difference %.2e' % np.max(np.abs(res['cost'] - st[:, oracle.ST['OBJ']])/np.abs(st[:, oracle.ST['OBJ']])), flush=True)
