"""
Bit-for-bit comparison of two builds of the HIP library on a GPU box -- the device half of the hunt for reads before writes (task 1 iii):
a diagnostic build with -ftrivial-auto-var-init=pattern (tools/build_variant.py pattern --flags "-ftrivial-auto-var-init=pattern": every local that the
compiler cannot prove written starts as a NaN pattern instead of whatever the register held) must return exactly the bits of the product build.  A local
read before its first write (round 5: Solver::evs) shows as a difference or as a failed solve; identical results say the pattern never reached a value
that decides anything.

    python tools/compare_libraries.py <library A> <library B>        (each library runs in a process of its own: MSD_LIB)

Cases: every kernel family at small sizes from both starting points -- fused first pass, first pass with the least-squares estimate, LDS-resident and
streamed follow-up kernels (restoration phases, second-order corrections), one-brake and both-brakes structure, the general kernels, dynamic loss table,
integrateLosses, collocation and adaptive shooting, per-scenario rolling stock, the shrinking-horizon loop.
"""
import os
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def run_cases(out):
    for p in ('ms-eetc_amd', '', 'tests'):
        sys.path.insert(0, str(ROOT / p))
    import numpy as np
    import cases
    from mseetc.ocp import casadiSolver
    from test_gpu_parity import _random_problem
    res = {}

    def put(name, r):
        st = r['stats'].copy()
        st[:, 11] = st[:, 12] = 0      # CYC_TOTAL, CYC_KKT: time stamps
        res[name + '/z'] = r['z']; res[name + '/stats'] = st

    io = dict(numSteps=1, numApproxSteps=1)
    both, rg = cases.train_default(), cases.train_fig10()
    for start in ('profile', 'reference'):
        for tag, train in (('both', both), ('rg', rg)):
            for N, crop in ((40, 16000), (100, None), (150, None)):
                track = cases.track_00(crop) if crop else cases.track_00()
                s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=800, integrationOptions=io), startingPoint=start)
                scale = (crop or 48531)/48531.0
                T = np.concatenate([1541.0*scale*(1 + 0.15*np.random.default_rng(N).random(48)), np.array([4000.0, 9000.0, 14000.0])*scale])      # (the loose ones: restoration phases)
                put('%s/%s/N%d' % (start, tag, N), s.solveBatch(T))
                s.close()
    # no structure compiled in: pneumatic brake only
    pn = cases.train_default(); pn.forceMin = 0
    for start in ('profile', 'reference'):
        s = casadiSolver(pn, cases.track_00(), dict(numIntervals=100, maxIterations=800, integrationOptions=io), startingPoint=start)
        put('%s/pn_only/N100' % start, s.solveBatch(cases.c1_times(32)))
        s.close()
    # random problems of the sweep that went through the follow-up kernels' cold paths in round 5
    with tempfile.TemporaryDirectory() as tmp:
        for seed in (15, 105, 125, 176):
            train, track, N, rng = _random_problem(seed, Path(tmp))
            v0, vN = float(rng.uniform(1, 15)), float(rng.uniform(1, 15))
            tw = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, energyOptimal=False, integrationOptions=io), startingPoint='profile')
            tmin = float(tw.solveBatch([3*track.length/train.velocityMax], initialVelocity=v0, terminalVelocity=vN)['z'][0][-2])
            tw.close()
            for start in ('profile', 'reference'):
                s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=800, integrationOptions=io), startingPoint=start)
                put('seed%d/%s' % (seed, start), s.solveBatch(tmin*np.array([1.05, 1.1, 1.2, 1.45, 2.0, 3.0]), initialVelocity=v0, terminalVelocity=vN))
                s.close()
    # the other transcriptions and loss models
    T = cases.c1_times(32)
    for name, extra, opt in (('intloss', dict(integrateLosses=True), io), ('irk', dict(integrationMethod='IRK'), dict(order=2, numSteps=1, numApproxSteps=1)),
                             ('cvodes', dict(integrationMethod='CVODES'), dict()), ('joint_rk4', dict(), dict(numSteps=2, numApproxSteps=0))):
        s = casadiSolver(both, cases.track_00(), dict(numIntervals=100, maxIterations=500, integrationOptions=opt, **extra), startingPoint='profile')
        put(name, s.solveBatch(T))
        s.close()
    from mseetc.train import Train
    from mseetc.efficiency import totalLossesFunction
    dyn = Train(config={'id': 'NL_Intercity_VIRM6'}); dyn.forceMinPn = 0
    dyn.powerLosses = totalLossesFunction(dyn, auxiliaries=27000, etaGear=0.96)
    for N in (60, 100):
        s = casadiSolver(dyn, cases.track_00(8500), dict(numIntervals=N, maxIterations=500, integrationOptions=io), startingPoint='profile')
        put('dynamic/N%d' % N, s.solveBatch(272.4726*np.linspace(1.05, 1.3, 16), terminalVelocity=100/3.6))
        s.close()
    # per-scenario rolling stock, shrinking-horizon loop
    s = casadiSolver(both, cases.track_00(), dict(numIntervals=100, maxIterations=500, integrationOptions=io), startingPoint='profile')
    rng = np.random.default_rng(3)
    put('config3', s.solveBatch(cases.c1_times(64), mass=both.mass*(1 + 0.05*np.clip(rng.standard_normal(64), -2, 2)), r0=both.r0*(1 + 0.05*np.clip(rng.standard_normal(64), -2, 2))))
    s.close()
    from mseetc.mpc import shrinkingHorizon
    log = shrinkingHorizon(both, cases.track_00(), dict(numIntervals=100, maxIterations=500, integrationOptions=io), cases.c1_times(16), 8, stride=10, noise=0.01, seed=1, warmStart=True)
    res['mpc/t0'] = np.stack([l['t0'] for l in log]); res['mpc/cost'] = np.stack([l['cost'] for l in log]); res['mpc/iters'] = np.stack([l['iterations'] for l in log])
    np.savez(out, **res)


def main():
    if len(sys.argv) == 3 and sys.argv[1] == '--child':
        return run_cases(sys.argv[2])
    import numpy as np
    a, b = sys.argv[1], sys.argv[2]
    outs = []
    with tempfile.TemporaryDirectory() as tmp:
        for k, lib in enumerate((a, b)):
            out = os.path.join(tmp, 'r%d.npz' % k)
            env = dict(os.environ, MSD_LIB=str(Path(lib).resolve()))
            r = subprocess.run([sys.executable, __file__, '--child', out], env=env, capture_output=True, text=True)
            if r.returncode != 0:
                print('library', lib, 'failed:\n', r.stderr[-3000:])
                return 2
            outs.append(dict(np.load(out)))
    ra, rb = outs
    bad = 0
    for key in sorted(ra):
        same = ra[key].shape == rb[key].shape and np.array_equal(ra[key], rb[key], equal_nan=True)
        if not same:
            bad += 1
            d = np.abs(ra[key] - rb[key]) if ra[key].shape == rb[key].shape else None
            print('DIFFERENT', key, 'max abs difference', None if d is None else float(np.nanmax(d)), 'entries', None if d is None else int((d > 0).sum()))
            if key.endswith('/stats') and d is not None:
                rows = np.flatnonzero((d > 0).any(axis=1))
                print('    scenarios', rows[:10], 'status A', ra[key][rows[:10], 0], 'B', rb[key][rows[:10], 0], 'iterations A', ra[key][rows[:10], 1], 'B', rb[key][rows[:10], 1])
    nsolve = sum(v.shape[0] for k, v in ra.items() if k.endswith('/stats'))
    nfail = sum(int((v[:, 0] < 0).sum()) for k, v in ra.items() if k.endswith('/stats'))
    print('%d result arrays, %d solves (%d failed in A): %d arrays differ between\n  A %s\n  B %s' % (len(ra), nsolve, nfail, bad, a, b))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
