"""
Robustness sweep for user-supplied loss functions (efficiency.TabulatedLosses): random trains / tracks / horizons of
tests/test_gpu_parity.py::_random_problem, each with a random loss function L(F, v) (copper ~ F^2, iron / friction ~ v and v^2, converter
share of the power, a smooth non-polynomial part; cheaper in braking) -- GPU solves from both starting points at four running times against
each other, one of them against the oracle on the same table, and the table against the function.
usage: random_sweep_loss_functions.py FIRST LAST
"""
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / 'ms-eetc_amd')); sys.path.insert(0, str(ROOT / 'tests'))
from oracle import oracle                      # noqa: E402
from mseetc.ocp import casadiSolver            # noqa: E402
from mseetc.track import computeDiscretizationPoints      # noqa: E402
from test_gpu_parity import _random_problem    # noqa: E402

first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
nsolves = 0
worst_table = 0.0
for seed in range(first, last):
    with tempfile.TemporaryDirectory() as tmp:
        train, track, N, rng = _random_problem(seed, Path(tmp))
        v0, vN = float(rng.uniform(2, 15)), float(rng.uniform(2, 15))
        Fm, Vm = train.forceMax, train.velocityMax
        c = rng.uniform(0.3, 1.0, 6)
        # shares of the maximum power Fm*Vm: copper 6 %, converter 5 / 8 %, iron + friction 2 %, a saturating part 2 %
        def fun(f, v, c=c, Fm=Fm, Vm=Vm):
            P = Fm*Vm
            base = 0.02*c[0]*P*(0.5*v/Vm + 0.5*(v/Vm)**2) + 0.02*c[1]*P*(np.sqrt(1 + (3*f/Fm)**2) - 1)/3
            return base + (0.06*c[2]*P*(f/Fm)**2 + 0.05*c[3]*f*v)*(f >= 0) + (0.04*c[4]*P*(f/Fm)**2 - 0.08*c[5]*f*v)*(f < 0)
        train.powerLosses = fun
        table = train.lossesCallable()
        worst_table = max(worst_table, table.maxDeviation)
        fast = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, energyOptimal=False, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='profile')
        rt = fast.solveBatch([3*track.length/train.velocityMax], initialVelocity=v0, terminalVelocity=vN)
        fast.close()
        if rt['status'][0] != 0:
            print('seed', seed, 'time-optimal twin failed'); bad += 1
            continue
        T = float(rt['z'][0][-2])*np.array([1.06, 1.15, 1.4, 1.9])
        costs = {}
        for start in ('profile', 'reference'):
            s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint=start)
            res = s.solveBatch(T, initialVelocity=v0, terminalVelocity=vN)
            s.close()
            nsolves += len(T)
            if not np.all(res['status'] == 0):
                print('seed', seed, 'N', N, start, 'status', res['status'], 'iters', res['iterations']); bad += 1
            costs[start] = res['cost']
        dev = np.max(np.abs(costs['profile'] - costs['reference'])/np.maximum(np.abs(costs['reference']), 1.0))
        if dev > 1e-5:
            print('seed', seed, 'N', N, 'starts disagree', dev); bad += 1
        oracle.set_loss_table(table.parameters(train.mass*train.rho))
        prob = oracle.pack_problem(train, computeDiscretizationPoints(track, N), dict(numIntervals=N, maxIterations=500, energyOptimal=True, minimumVelocity=1, numSteps=1, numApproxSteps=1),
                                   2, 0.0, 0.0, track.length)
        ref = oracle.solve(prob, prob.scenario(float(T[1]), 0.0, vN, v0), start='profile')
        dev = abs(costs['profile'][1] - ref['stats']['OBJ'])/max(abs(ref['stats']['OBJ']), 1.0)
        if ref['stats']['STATUS'] != 0 or dev > 1e-7:
            print('seed', seed, 'N', N, 'oracle status', ref['stats']['STATUS'], 'objective deviation', dev); bad += 1
print('seeds', first, '...', last - 1, ':', nsolves, 'solves,', bad, 'findings; largest table deviation', '%.1e' % worst_table)
