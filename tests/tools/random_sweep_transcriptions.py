"""
Robustness sweep for integrateLosses and the collocation / adaptive shooting integrators: the random problems of
tests/test_shooting_integrators.py::test_gpu_randomized_problems_with_other_transcriptions for many more seeds.
usage: random_sweep_transcriptions.py FIRST LAST
"""
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'ms-eetc_amd')); sys.path.insert(0, str(ROOT / 'tests'))
import cases                                   # noqa: E402
from oracle import oracle                      # noqa: E402
from mseetc.ocp import casadiSolver            # noqa: E402
from test_gpu_parity import _random_problem    # noqa: E402
from test_shooting_integrators import TRANSCRIPTIONS      # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
nsolves = 0
for seed in range(first, last):
    with tempfile.TemporaryDirectory() as tmp:
        train, track, N, rng = _random_problem(seed, Path(tmp))
        v0, vN = float(rng.uniform(2, 15)), float(rng.uniform(2, 15))
        fast = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, energyOptimal=False, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='profile')
        rt = fast.solveBatch([3*track.length/train.velocityMax], initialVelocity=v0, terminalVelocity=vN)
        fast.close()
        if rt['status'][0] != 0:
            print('seed', seed, 'time-optimal twin failed'); bad += 1
            continue
        T = float(rt['z'][0][-2])*np.array([1.06, 1.15, 1.4, 1.9])
        for which in sorted(TRANSCRIPTIONS):
            extra, io, integration = TRANSCRIPTIONS[which]
            costs, mus = {}, {}
            for start in ('profile', 'reference'):
                s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationOptions=io, **extra), startingPoint=start)
                res = s.solveBatch(T, initialVelocity=v0, terminalVelocity=vN)
                s.close()
                nsolves += len(T)
                if not np.all(res['status'] == 0):
                    print('seed', seed, 'N', N, which, start, 'status', res['status'], 'iters', res['iterations']); bad += 1
                costs[start] = res['cost']
                mus[start] = res['stats'][:, 4]
            dev = np.max(np.abs(costs['profile'] - costs['reference'])/np.maximum(np.abs(costs['reference']), 1.0))
            if dev > 1e-5:
                print('seed', seed, 'N', N, which, 'starts disagree', dev); bad += 1
            prob = cases.oracle_problem(train, track, N, numSteps=io.get('numSteps', 1), numApproxSteps=io.get('numApproxSteps', 0), integration=integration)
            ref = oracle.solve(prob, prob.scenario(float(T[1]), 0.0, vN, v0), start='profile')
            dev = abs(costs['profile'][1] - ref['stats']['OBJ'])/max(abs(ref['stats']['OBJ']), 1.0)
            # two solves that end on different barrier parameters (9.1e-10 against 2.5e-9: the last barrier test looks at rounding noise) differ by
            # about the difference times the number of active bounds -- 1.6e-6 kWh at N = 165, which shows on a journey that costs 1 kWh
            same_mu = abs(mus['profile'][1] - ref['stats']['MU']) <= 1e-3*ref['stats']['MU']
            if ref['stats']['STATUS'] != 0 or dev > (1e-7 if same_mu else 1e-5):
                print('seed', seed, 'N', N, which, 'oracle status', ref['stats']['STATUS'], 'objective deviation', dev); bad += 1
print('seeds', first, '...', last - 1, ':', nsolves, 'solves,', bad, 'findings')
