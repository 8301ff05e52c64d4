"""
Capture script for the IPOPT boundary ("if a machine with casadi 3.6.3 ever becomes available, capture (z*, cost, iters) for the
BASELINE problem shapes as fixtures").

The build container has no CasADi/IPOPT, so parity at that boundary is pinned by reference-held constants only.  This script is the
harness for a machine that HAS them: it imports the reference package itself (REFERENCE=/path/to/ms-eetc checkout, default /root/reference), drives
`casadiSolver(train, track, opts).solve(...)` the way the reference's scripts do (simulations/figure10.py:36-46, table3.py:48-64, figure5.py:98-146),
records what IPOPT returned -- the raw z*, the multipliers of g, cost, iteration count, return status -- and writes one
tests/golden/ipopt_<case>.json per case.  tests/test_ipopt_fixtures.py compares the oracle (CPU) and the HIP path (GPU) with every such file that exists
and skips when there is none.  Own code; nothing of the reference is copied, and the script never travels to the GPU box's run (it needs the reference).

    REFERENCE=/path/to/ms-eetc python tests/golden/make_ipopt_fixtures.py [case ...]

Cases = the problem shapes of BASELINE.json's configs 0-3 (config 4's re-solves are config 1's shape from a later node):
  c0_*   figure5.py: VIRM6 without the pneumatic brake after totalLossesFunction's side effects, track cropped to 8.5 km, T = 272.4726 x {1.0 ... 1.3},
         N = 100 and the file's 300, loss models fun0 / fun1 / fun2
  c1_*   N = 100, 00_var_speed_limit_100, JSON-default train, three running times of the benchmark's range
  c2_*   N = 200, CH_StGallen_Wil
  c3_*   config 1 with a perturbed train (mass, Davis coefficients set on the Train object, like figure10.py:17-22 sets its attributes)
  fig10_* figure10.py's configuration (the one GPOPS-II solved), N = 100 / 300
"""

import json
import os
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REFERENCE = Path(os.environ.get('REFERENCE', '/root/reference'))

INTEGRATION = {'order': 4, 'numSteps': 1, 'numApproxSteps': 1}      # simulations/config.json


def _opts(N, **kw):
    o = {'maxIterations': 500, 'numIntervals': N, 'integrationMethod': 'RK', 'integrationOptions': dict(INTEGRATION)}
    o.update(kw)
    return o


def cases():
    "name -> (train builder, track builder, options, solve arguments); builders take the reference's modules"

    def virm6(mods, **attrs):
        train = mods['Train'](config={'id': 'NL_Intercity_VIRM6'})
        for k, v in attrs.items():
            setattr(train, k, v)
        return train

    def track00(mods, end=None):
        track = mods['Track'](config={'id': '00_var_speed_limit_100'})
        if end is not None:
            track.updateLimits(positionEnd=end)
        return track

    out = {}
    # config 0 (figure5.py:84-146)
    for N in (100, 300):
        for tp in (0, 10, 20, 30):
            for lm in ('fun0', 'fun1', 'fun2'):
                def train_c0(mods, lm=lm):
                    train = virm6(mods, forceMinPn=0)
                    fun2 = mods['totalLossesFunction'](train, auxiliaries=27000, etaGear=0.96)      # (mutates the train: efficiency.py:64-71)
                    eta = 0.73
                    train.powerLosses = {'fun0': (lambda f, v: 0), 'fun1': (lambda f, v: f*v*(f > 0)*(1 - eta)/eta - (1 - eta)*f*v*(f < 0)), 'fun2': fun2}[lm]
                    return train
                out['c0_N%d_T%d_%s' % (N, tp, lm)] = (train_c0, lambda mods: track00(mods, 8500), _opts(N, minimumVelocity=1),
                                                      dict(terminalTime=272.4726*(1 + tp/100), terminalVelocity=100/3.6, initialVelocity=1))
    # config 1
    for T in (1541.0, 1620.0, 1772.0):
        out['c1_T%d' % T] = (lambda mods: virm6(mods), lambda mods: track00(mods), _opts(100), dict(terminalTime=T, terminalVelocity=1, initialVelocity=1))
    # config 2
    for T in (1242.0, 1330.0, 1428.0):
        out['c2_T%d' % T] = (lambda mods: virm6(mods), lambda mods: mods['Track'](config={'id': 'CH_StGallen_Wil'}), _opts(200),
                             dict(terminalTime=T, terminalVelocity=1, initialVelocity=1))
    # config 3: perturbed rolling stock (the reference's scripts set train attributes between constructions: figure10.py:17-22)
    def train_c3(mods):
        train = virm6(mods)
        train.mass *= 1.04; train.r0 *= 0.97; train.r1 *= 1.05; train.r2 *= 0.96
        return train
    out['c3_T1600'] = (train_c3, lambda mods: track00(mods), _opts(100), dict(terminalTime=1600.0, terminalVelocity=1, initialVelocity=1))
    # figure 10 (figure10.py:16-42)
    for N in (100, 300):
        def train_f10(mods):
            return virm6(mods, forceMinPn=0, powerMax=3129277, powerMin=-3129277, etaTraction=0.73, etaRgBrake=0.73, forceMin=-virm6(mods).forceMax)
        out['fig10_N%d' % N] = (train_f10, lambda mods: track00(mods), _opts(N, minimumVelocity=1), dict(terminalTime=1541.0, terminalVelocity=1, initialVelocity=1))
    return out


def capture(name, spec, mods):
    "one solve of the reference; the raw NLP solution is read off the nlpsol call inside casadiSolver.solve (ocp.py:359)"
    make_train, make_track, opts, kw = spec
    train, track = make_train(mods), make_track(mods)
    solver = mods['casadiSolver'](train, track, opts)
    raw = {}
    inner = solver.solver

    def recording(**args):
        sol = inner(**args)
        raw['x'] = np.array(sol['x']).flatten().tolist()
        raw['lam_g'] = np.array(sol['lam_g']).flatten().tolist()
        raw['f'] = float(sol['f'])
        return sol
    recording.stats = inner.stats
    solver.solver = recording
    df, stats = solver.solve(kw['terminalTime'], terminalVelocity=kw['terminalVelocity'], initialVelocity=kw['initialVelocity'])
    return dict(case=name, options=opts, solve=kw, z=raw.get('x'), lam_g=raw.get('lam_g'), f_scaled=raw.get('f'), cost=stats['Cost'],
                iters=int(stats['IP iterations']), status=stats['Solver status'], converged=df is not None,
                energy_kWh=None if df is None else float(df['Energy [kWh]'].sum()),
                train=dict(mass=train.mass, rho=train.rho, forceMax=train.forceMax, forceMin=train.forceMin, forceMinPn=train.forceMinPn,
                           powerMax=train.powerMax, powerMin=train.powerMin, velocityMax=train.velocityMax, r0=train.r0, r1=train.r1, r2=train.r2,
                           etaTraction=getattr(train, 'etaTraction', None), etaRgBrake=getattr(train, 'etaRgBrake', None)),
                points=[float(p) for p in solver.points.index.values])


def main(argv):
    try:
        import casadi
    except ImportError:
        raise SystemExit("casadi is not importable here: this script is for a machine with casadi 3.6.3 (setup.py:12 of the reference) and IPOPT")
    sys.path.insert(0, str(REFERENCE))
    from mseetc.ocp import casadiSolver
    from mseetc.train import Train
    from mseetc.track import Track
    from mseetc.efficiency import totalLossesFunction
    mods = dict(casadiSolver=casadiSolver, Train=Train, Track=Track, totalLossesFunction=totalLossesFunction)
    todo = cases()
    names = argv or sorted(todo)
    os.chdir(str(REFERENCE / 'simulations'))      # (the reference's loaders use paths relative to its scripts)
    for name in names:
        rec = capture(name, todo[name], mods)
        rec['casadi'] = casadi.__version__
        with open(HERE / ('ipopt_%s.json' % name), 'w') as fh:
            json.dump(rec, fh)
        print(name, rec['status'], rec['iters'], rec['cost'])


if __name__ == '__main__':
    main(sys.argv[1:])
