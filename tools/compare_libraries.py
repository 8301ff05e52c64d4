"""
Comparison of two builds of the HIP library on a GPU box -- the device half of the hunt for reads before writes (task 1 iii): a diagnostic
build with -ftrivial-auto-var-init=pattern (tools/build_variant.py pattern --flags "-ftrivial-auto-var-init=pattern": every local that the compiler cannot
prove written starts as a NaN pattern instead of whatever the register held) against the product build.  A local read before its first write (round 5:
Solver::evs) that decides anything shows as a NaN, a failed solve or another path through the iteration.

What "equal" can mean: the two builds are not bit-identical in their arithmetic -- the initialising stores that survive change the shape of the code and with it
where the compiler contracts a*b + c into an FMA (HIP's default -ffp-contract=fast): measured on the fused benchmark path, 7e-12 in z.  So the rule is: same
status for every solve; the objective to 1e-7 relative (1e-9 absolute); iteration counts within 2 for solves of at most 100 iterations (a loose schedule that
crawls for hundreds of iterations or ends at the iteration limit follows rounding noise in either build: its status and objective are compared, not its path);
every entry finite where the other build's is.  Bit-identical arrays are counted as well.

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
    same_bits = sum(1 for k in ra if ra[k].shape == rb[k].shape and np.array_equal(ra[k], rb[k], equal_nan=True))
    bad, worst_obj, worst_z, nsolve, nfail, nloose = 0, 0.0, 0.0, 0, 0, 0
    for key in sorted(k for k in ra if k.endswith('/stats')):
        A, B = ra[key], rb[key]
        nsolve += len(A); nfail += int((A[:, 0] < 0).sum())
        finds = []
        if not np.array_equal(A[:, 0], B[:, 0]):
            finds.append('status A %s B %s' % (A[:, 0].astype(int).tolist(), B[:, 0].astype(int).tolist()))
        if not np.array_equal(np.isfinite(A), np.isfinite(B)):
            finds.append('non-finite entries differ')
        ok = (A[:, 0] >= 0) & (B[:, 0] >= 0)
        dobj = np.abs(A[ok, 2] - B[ok, 2])/np.maximum(np.abs(A[ok, 2]), 1e-2)
        worst_obj = max(worst_obj, float(dobj.max()) if dobj.size else 0.0)
        if dobj.size and dobj.max() > 1e-7:
            finds.append('objective differs by %.2e (scenarios %s)' % (dobj.max(), np.flatnonzero(ok)[dobj > 1e-7].tolist()))
        short = ok & (A[:, 1] <= 100) & (B[:, 1] <= 100)
        nloose += int((ok & ~short).sum())
        dit = np.abs(A[short, 1] - B[short, 1])
        if dit.size and dit.max() > 2:
            finds.append('iterations A %s B %s' % (A[short, 1][dit > 2].astype(int).tolist(), B[short, 1][dit > 2].astype(int).tolist()))
        zk = key[:-len('stats')] + 'z'
        if zk in ra and short.any():
            dz = np.abs(ra[zk][short] - rb[zk][short])/np.maximum(1.0, np.abs(ra[zk][short]))
            worst_z = max(worst_z, float(np.nanmax(dz)))
        if finds:
            bad += 1
            print('FINDING', key[:-len('/stats')], '; '.join(finds))
    for key in ('mpc/t0', 'mpc/cost'):
        if key in ra and not np.allclose(ra[key], rb[key], rtol=1e-6, atol=1e-6, equal_nan=True):
            bad += 1
            print('FINDING', key, 'max difference', float(np.nanmax(np.abs(ra[key] - rb[key]))))
    print('%d result arrays (%d bit-identical), %d solves (%d failed in A and B alike unless listed, %d long or crawling ones compared by status and objective only): %d findings; '
          'largest objective difference %.1e relative, largest difference of a variable on the ordinary solves %.1e\n  A %s\n  B %s'
          % (len(ra), same_bits, nsolve, nfail, nloose, bad, worst_obj, worst_z, a, b))
    return 1 if bad else 0


if __name__ == '__main__':
    sys.exit(main())
