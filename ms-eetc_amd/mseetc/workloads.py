"""
The benchmark workloads of SURVEY.md section 8(d) (BASELINE.json configs 1-4): trains, tracks, solver options and the
synthetic scenario batches with their seeds.  Used by bench.py, the tools and the tests, so that all of them measure and
check the same problems.

Reference mechanisms behind the scenario parameters: `casadiSolver.solve(terminalTime, initialTime, terminalVelocity,
initialVelocity)` (mseetc/ocp.py:310) for the running times, `Train(config={...})` overrides (mseetc/train.py:44-62) for the
rolling stock of config 3, `Track.updateLimits(positionStart)` (mseetc/track.py:420-450) for the re-solves of config 4.
"""

import numpy as np

from .track import Track
from .train import Train

# simulations/config.json of the reference: RK4, one step, trapezoidal time
INTEGRATION = dict(order=4, numSteps=1, numApproxSteps=1)


def train_default():
    return Train(config={'id': 'NL_Intercity_VIRM6'})


def track_00(crop=None):
    track = Track(config={'id': '00_var_speed_limit_100'})
    if crop is not None:
        track.updateLimits(positionEnd=crop)
    return track


def track_CH():
    return Track(config={'id': 'CH_StGallen_Wil'})


def options(numIntervals, maxIterations=500, **kw):
    return dict(numIntervals=numIntervals, maxIterations=maxIterations, integrationOptions=dict(INTEGRATION), **kw)


def c1_times(B, seed=20260612):
    "Config 1 running times: T_i = 1541 (1 + 0.15 u_i), u = default_rng(seed).random(B)"
    return 1541*(1 + 0.15*np.random.default_rng(seed).random(B))


def c2_times(B, seed=20260613):
    "Config 2 running times: T_i = 1242 (1 + 0.15 u_i)"
    return 1242*(1 + 0.15*np.random.default_rng(seed).random(B))


def c3_scenarios(B, train, seed=20260614):
    """
    Config 3: running times as config 1 (own seed) and rolling stock perturbed per scenario: mass (1 + 0.05 n1),
    r0, r1, r2 (1 + 0.05 n2..4), n ~ N(0, 1) clipped to +-2.  Returns (T, dict(mass=, r0=, r1=, r2=)).
    """
    rng = np.random.default_rng(seed)
    T = 1541*(1 + 0.15*rng.random(B))
    n = np.clip(rng.standard_normal((4, B)), -2, 2)
    return T, dict(mass=train.mass*(1 + 0.05*n[0]), r0=train.r0*(1 + 0.05*n[1]), r1=train.r1*(1 + 0.05*n[2]), r2=train.r2*(1 + 0.05*n[3]))


def config(name):
    "(train, track, numIntervals) of a named workload: 'c1', 'c2', 'c3', 'c4'"
    if name in ('c1', 'c3', 'c4'):
        return train_default(), track_00(), 100
    if name == 'c2':
        return train_default(), track_CH(), 200
    raise ValueError("unknown workload {!r}".format(name))
