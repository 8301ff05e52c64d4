"""Diagnostic (GPU): one random problem of the transcription sweep (tests/tools/random_sweep_transcriptions.py) -- GPU against the oracle per running time.
usage: debug_seed_transcription.py SEED WHICH   (WHICH: a key of tests/test_shooting_integrators.py: TRANSCRIPTIONS)"""
import os, sys, tempfile
from pathlib import Path
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('ms-eetc_amd', '', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
import cases
from oracle import oracle
from mseetc.ocp import casadiSolver
from mseetc._device import ST
from test_gpu_parity import _random_problem
from test_shooting_integrators import TRANSCRIPTIONS
seed, which = int(sys.argv[1]), sys.argv[2]
with tempfile.TemporaryDirectory() as tmp:
    train, track, N, rng = _random_problem(seed, Path(tmp))
    v0, vN = float(rng.uniform(2, 15)), float(rng.uniform(2, 15))
    fast = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, energyOptimal=False, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='profile')
    tmin = float(fast.solveBatch([3*track.length/train.velocityMax], initialVelocity=v0, terminalVelocity=vN)['z'][0][-2])
    fast.close()
    T = tmin*np.array([1.06, 1.15, 1.4, 1.9])
    extra, io, integration = TRANSCRIPTIONS[which]
    prob = cases.oracle_problem(train, track, N, numSteps=io.get('numSteps', 1), numApproxSteps=io.get('numApproxSteps', 0), integration=integration)
    for start in ('profile', 'reference'):
        s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationOptions=io, **extra), startingPoint=start)
        res = s.solveBatch(T, initialVelocity=v0, terminalVelocity=vN)
        s.close()
        for k, t in enumerate(T):
            ref = oracle.solve(prob, prob.scenario(float(t), 0.0, vN, v0), start=start)
            print(start, 'T/tmin %.2f' % (t/tmin), 'gpu', res['status'][k], res['iterations'][k], '%.12e' % res['cost'][k], 'mu %.3e' % res['stats'][k, ST['MU']], 'kkt %.2e' % res['stats'][k, 3],
                  '| oracle', ref['stats']['STATUS'], ref['stats']['ITERS'], '%.12e' % ref['stats']['OBJ'], 'mu %.3e' % ref['stats']['MU'], 'kkt %.2e' % ref['stats']['KKT'],
                  '| dz %.2e' % np.max(np.abs(res['z'][k] - ref['z'])/np.maximum(1.0, np.abs(ref['z']))))
