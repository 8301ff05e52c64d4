"""Diagnostic (GPU): one random problem of tests/test_gpu_parity.py::_random_problem at multiples of its minimum running time.
usage: probe_seed.py SEED START(profile|reference) RESTORATION(0|1) WATCHDOG(trigger, -1 = off) [FACTORS=1.05,1.1,1.2,1.45,2.0]"""
import os, sys, tempfile
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
for p in ('ms-eetc_amd', '', 'tests'):
    sys.path.insert(0, str(ROOT / p))
import numpy as np
import cases
from oracle import oracle
from mseetc._device import ST
from test_gpu_parity import _random_problem, _solver
seed, start, resto, wd = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
factors = [float(x) for x in (sys.argv[5] if len(sys.argv) > 5 else '1.05,1.1,1.2,1.45,2.0').split(',')]
with tempfile.TemporaryDirectory() as tmp:
    train, track, N, rng = _random_problem(seed, Path(tmp))
    v0, vN = float(rng.uniform(1, 15)), float(rng.uniform(1, 15))
    po = cases.oracle_problem(train, track, N, energyOptimal=False, losses='none')
    tmin = float(oracle.solve(po, po.scenario(3*track.length/train.velocityMax, 0.0, vN, v0), start='profile')['z'][-2])
    print('seed', seed, 'N', N, 'pn', train.forceMinPn, 'rg', train.forceMin, 'tmin', tmin, flush=True)
    s = _solver(train, track, N, start=start, restoration=bool(resto), watchdogTrigger=wd)
    res = s.solveBatch(tmin*np.array(factors), initialVelocity=v0, terminalVelocity=vN)
    print('status', res['status'], 'iters', res['iterations'], 'nreg', res['stats'][:, ST['N_REG']], 'nresto', res['stats'][:, ST['N_RESTO']], 'follow', s.problem.follow_counts(), flush=True)
    s.close()
