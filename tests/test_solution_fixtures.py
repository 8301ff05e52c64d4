"""
Frozen solutions (tests/golden/solutions.json, produced by the oracle -- see make_solution_fixtures.py):
the oracle must reproduce them on CPU, the HIP path on GPU.
"""

import json
import sys
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).resolve().parent / 'golden'
sys.path.insert(0, str(GOLD))

import make_solution_fixtures as mk  # noqa: E402

with open(GOLD / 'solutions.json') as fh:
    SOL = json.load(fh)


@pytest.mark.parametrize('name', sorted(SOL))
def test_oracle_reproduces_fixture(name):
    from oracle import oracle
    c = SOL[name]['config']
    prob = mk.build(c)
    res = oracle.solve(prob, prob.scenario(**c['kw']))
    assert res['stats']['STATUS'] == 0 and int(res['stats']['ITERS']) == SOL[name]['iters']
    assert abs(res['stats']['OBJ'] - SOL[name]['obj']) <= 1e-10*abs(SOL[name]['obj'])
    ref = np.array(SOL[name]['z'])
    assert np.max(np.abs(res['z'] - ref)/np.maximum(1, np.abs(ref))) < 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize('start', ['reference', 'profile'])
@pytest.mark.parametrize('name', sorted(SOL))
def test_gpu_reproduces_fixture(name, start):
    # the fixtures were generated from the reference's starting point: that run must reproduce them iterate for iterate;
    # the profile start must land on the same optimum with fewer iterations
    import cases
    from mseetc.ocp import casadiSolver
    c = SOL[name]['config']
    train = dict(default=cases.train_default, fig10=cases.train_fig10, fig5=cases.train_fig5)[c['train']]()
    if c['losses'] == 'none':
        train.powerLosses = lambda f, v: 0
    track = cases.track_CH() if c['track'] == 'CH' else cases.track_00(c['crop'])
    solver = casadiSolver(train, track, dict(numIntervals=c['N'], maxIterations=500, energyOptimal=c['eo'],
                                              integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint=start)
    kw = dict(c['kw'])
    res = solver.solveBatch(kw.pop('terminalTime'), **kw)
    assert res['status'][0] == 0
    assert abs(res['stats'][0, 2] - SOL[name]['obj']) <= 1e-8*abs(SOL[name]['obj'])     # north_star bar: 1e-4
    ref = np.array(SOL[name]['z'])
    if start == 'reference':
        assert np.max(np.abs(res['z'][0] - ref)/np.maximum(1, np.abs(ref))) < 1e-6
        assert abs(int(res['iterations'][0]) - SOL[name]['iters']) <= 2
    else:
        assert np.max(np.abs(res['z'][0] - ref)/np.maximum(1, np.abs(ref))) < 1e-5
        assert int(res['iterations'][0]) < SOL[name]['iters']
