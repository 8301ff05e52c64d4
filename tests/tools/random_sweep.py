"""
One-off robustness sweep on a GPU box: the random problems of tests/test_gpu_parity.py::test_randomized_problems_vs_oracle for many more
seeds than the test suite carries; prints every seed whose GPU solves fail or disagree with the oracle.  usage: random_sweep.py FIRST LAST
"""
import os
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'ms-eetc_amd')); sys.path.insert(0, str(ROOT / 'tests'))
import cases                                   # noqa: E402
from oracle import oracle                      # noqa: E402
from test_gpu_parity import _random_problem, _solver      # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, last):
    with tempfile.TemporaryDirectory() as tmp:
        train, track, N, rng = _random_problem(seed, Path(tmp))
        if os.environ.get('SWEEP_VERBOSE'):
            print('seed', seed, 'N', N, 'pn', train.forceMinPn, 'rg', train.forceMin, flush=True)
        v0, vN = float(rng.uniform(1, 15)), float(rng.uniform(1, 15))
        fast = _solver(train, track, N, energyOptimal=False, start='profile')
        rt = fast.solveBatch([3*track.length/train.velocityMax], initialVelocity=v0, terminalVelocity=vN)
        fast.close()
        if rt['status'][0] != 0:
            print('seed', seed, 'N', N, 'time-optimal twin failed', rt['status'][0]); bad += 1
            continue
        tmin = float(rt['z'][0][-2])
        T = tmin*np.array([float(x) for x in os.environ.get('SWEEP_FACTORS', '1.05,1.1,1.2,1.45,2.0').split(',')])      # (SWEEP_FACTORS: other multiples of the minimum running time)
        for start in ('profile', 'reference'):
            s = _solver(train, track, N, start=start)
            res = s.solveBatch(T, initialVelocity=v0, terminalVelocity=vN)
            s.close()
            if not np.all(res['status'] == 0):
                print('seed', seed, 'N', N, start, 'status', res['status'], 'iters', res['iterations']); bad += 1
                continue
            if start == 'profile':
                pe = cases.oracle_problem(train, track, N)
                ref = oracle.solve(pe, pe.scenario(float(T[1]), 0.0, vN, v0), start='profile')
                dev = abs(res['cost'][1] - ref['stats']['OBJ'])/max(abs(ref['stats']['OBJ']), 1.0)      # (kWh; loose schedules on downhill tracks cost nothing)
                # two solves that end on different barrier parameters (the last barrier test looks at rounding noise: tests/test_gpu_parity.py)
                # differ by about mu times the number of active bounds
                same_mu = abs(res['stats'][1, 4] - ref['stats']['MU']) <= 1e-3*ref['stats']['MU']
                if ref['stats']['STATUS'] != 0 or dev > (1e-7 if same_mu else 1e-6):
                    print('seed', seed, 'N', N, 'oracle status', ref['stats']['STATUS'], 'objective deviation', dev); bad += 1
                cost_p, mu_p = res['cost'], res['stats'][:, 4]
            else:
                rel = np.abs(res['cost'] - cost_p)/np.maximum(np.abs(cost_p), 1.0)
                same = np.abs(res['stats'][:, 4] - mu_p) <= 1e-3*mu_p
                dev = float(np.max(np.where(same, rel, rel/10)))      # 1e-6 on the same final barrier parameter, 1e-5 across
                if dev > 1e-6:
                    print('seed', seed, 'N', N, 'starts disagree', dev); bad += 1
print('seeds', first, '...', last - 1, ':', bad, 'findings')
