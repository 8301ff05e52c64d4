"""
Generates tests/golden/solutions.json: regression vectors of full solves produced by the CPU ORACLE (oracle/ms_oracle.c)
-- NOT outputs of the reference (CasADi/IPOPT cannot run in the build container).  They freeze the
solutions the oracle and the HIP path agree on, so that later changes to either are caught on CPU and on GPU.

    python tests/golden/make_solution_fixtures.py
"""

import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
for p in (HERE.parent.parent / 'ms-eetc_amd', HERE.parent.parent, HERE.parent):
    sys.path.insert(0, str(p))

import cases  # noqa: E402
from oracle import oracle  # noqa: E402

CASES = {
    'c1_T1541': dict(train='default', track='00', crop=None, N=100, eo=True, losses='static', kw=dict(terminalTime=1541.0)),
    'c1_T1700': dict(train='default', track='00', crop=None, N=100, eo=True, losses='static', kw=dict(terminalTime=1700.0)),
    'c2_T1242': dict(train='default', track='CH', crop=None, N=200, eo=True, losses='static', kw=dict(terminalTime=1242.0)),
    'fig10_N100': dict(train='fig10', track='00', crop=None, N=100, eo=True, losses='static', kw=dict(terminalTime=1541.0)),
    'mintime_N100': dict(train='fig5', track='00', crop=8500, N=100, eo=False, losses='none',
                         kw=dict(terminalTime=400.0, terminalVelocity=100/3.6, initialVelocity=1.0)),
    'mpc_like': dict(train='default', track='00', crop=20000, N=40, eo=True, losses='static',
                     kw=dict(terminalTime=900.0, initialTime=100.0, initialVelocity=20.0, terminalVelocity=5.0)),
}


def build(c):
    train = dict(default=cases.train_default, fig10=cases.train_fig10, fig5=cases.train_fig5)[c['train']]()
    track = cases.track_CH() if c['track'] == 'CH' else cases.track_00(c['crop'])
    return cases.oracle_problem(train, track, c['N'], energyOptimal=c['eo'], losses=c['losses'])


if __name__ == '__main__':
    out = {}
    for name, c in CASES.items():
        prob = build(c)
        res = oracle.solve(prob, prob.scenario(**c['kw']))
        assert res['stats']['STATUS'] == 0, name
        out[name] = dict(config=c, z=[float(x) for x in res['z']], obj=float(res['stats']['OBJ']), iters=int(res['stats']['ITERS']))
        print(name, out[name]['obj'], out[name]['iters'])
    with open(HERE / 'solutions.json', 'w') as fh:
        json.dump(out, fh)
