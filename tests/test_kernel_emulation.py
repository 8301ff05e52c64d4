"""
The device header ms-eetc_amd/csrc/msd_kernel.hpp compiled for the host (tests/hip_emu: one OS thread per GPU thread,
pthread barriers) and compared with the oracle.  This checks the kernel's arithmetic and control flow on CPU -- and is what
the sanitizer build (SAN=1 tests/hip_emu/build.sh) runs -- but not wave-level behaviour: barrier placement in divergent code
can only be seen on the GPU (tests/test_gpu_parity.py).  Emulation is test tooling; the product never uses it.
"""

import ctypes
import subprocess
from pathlib import Path

import numpy as np
import pytest

import cases

EMU = Path(__file__).resolve().parent / 'hip_emu'


def load_emulation():
    "Build (when out of date) and load the host emulation of the kernels."
    src = sorted(EMU.glob('*.cpp')) + sorted(EMU.glob('*.h')) + [EMU / 'build.sh', EMU / 'hip' / 'hip_runtime.h'] \
        + sorted((EMU.parent.parent / 'ms-eetc_amd' / 'csrc').glob('*.hpp')) + [EMU.parent.parent / 'include' / 'mseetc_hip.h']
    import os
    if os.environ.get('MSD_EMU_LIB'):      # a sanitizer build made by tests/hip_emu/run_sanitizers.sh
        so = Path(os.environ['MSD_EMU_LIB'])
        lib = ctypes.CDLL(str(so))
        from mseetc._device import ProblemDesc
        dp = ctypes.POINTER(ctypes.c_double)
        lib.emu_solve_batch.argtypes = [ctypes.POINTER(ProblemDesc), ctypes.c_int, dp, dp, dp, dp, dp, ctypes.c_int]
        return lib
    so = EMU / 'libmsd_emu.so'
    if not so.exists() or so.stat().st_mtime < max(f.stat().st_mtime for f in src):
        subprocess.run([str(EMU / 'build.sh')], check=True)
    lib = ctypes.CDLL(str(EMU / 'libmsd_emu.so'))
    from mseetc._device import ProblemDesc
    dp = ctypes.POINTER(ctypes.c_double)
    lib.emu_solve_batch.argtypes = [ctypes.POINTER(ProblemDesc), ctypes.c_int, dp, dp, dp, dp, dp, ctypes.c_int]
    return lib


@pytest.fixture(scope='module')
def emu():
    return load_emulation()


@pytest.mark.parametrize('N,crop,T,start', [(30, 12000, 520.0, 'reference'), (70, 30000, 1100.0, 'reference'), (70, 30000, 1100.0, 'profile'),
                                            # two solves with a second-order correction on the way (oracle: N_SOC = 1): the cold block inside the fused iteration
                                            (40, 16000, 804.9041795334854, 'profile'), (60, 30000, 1140.8291957305269, 'profile')])
def test_emulated_kernel_matches_oracle(emu, N, crop, T, start):
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from oracle import oracle
    train, track = cases.train_default(), cases.track_00(crop)
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=300, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint=start)
    scen = solver._scenarios(T, 0, 1, 1)
    nz = (4 + int(solver.withPnBrake))*N + 2
    z, lam, st, hist = np.zeros((1, nz)), np.zeros((1, 7*N)), np.zeros((1, ST['COUNT'])), np.zeros((8, 8))
    d = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert emu.emu_solve_batch(ctypes.byref(solver._desc), 1, d(scen), d(z), d(lam), d(st), d(hist), 8) == 0
    prob = cases.oracle_problem(train, track, N)
    ref = oracle.solve(prob, prob.scenario(T), start=start)
    assert st[0, ST['STATUS']] == 0 and ref['stats']['STATUS'] == 0
    assert int(st[0, ST['ITERS']]) == int(ref['stats']['ITERS'])
    assert int(st[0, ST['N_SOC']]) == int(ref['stats']['N_SOC']) == (1 if T in (804.9041795334854, 1140.8291957305269) else 0)
    assert np.max(np.abs(z[0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-8
    # the multipliers of a converged solve are determined to the solver tolerance (1e-8 on the scaled problem): relative bound
    assert np.max(np.abs(lam[0] - ref['lam_g'])/np.maximum(1, np.abs(ref['lam_g']))) < 1e-7


@pytest.mark.parametrize('variant,start', [('both', 'reference'), ('rg', 'reference'), ('both', 'profile')])
def test_emulated_kernels_read_no_shared_memory_before_writing_it(emu, variant, start, monkeypatch):
    """
    EMU_POISON=1: the emulated workgroup's LDS and work area start as NaN instead of zero.  On the device a read of shared memory before its first write
    sees what the kernel before left there -- the kind of fault that shows as a non-deterministic failure on the GPU and never in a zero-initialised
    emulation.  First pass + follow-up kernel of both rolling-stock structures from both starting points: bit for bit the results of the zero-initialised
    run, history included.
    """
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    train, track, N, T = (cases.train_fig10() if variant == 'rg' else cases.train_default()), cases.track_00(30000), 70, 1100.0
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=300, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint=start)
    scen = solver._scenarios(T, 0, 1, 1)
    nz = (4 + int(solver.withPnBrake))*N + 2
    d = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    runs = {}
    for poison in ('0', '1'):
        monkeypatch.setenv('EMU_POISON', poison)
        z, lam, st, hist = np.zeros((1, nz)), np.zeros((1, 7*N)), np.zeros((1, ST['COUNT'])), np.zeros((40, 8))
        assert emu.emu_solve_batch(ctypes.byref(solver._desc), 1, d(scen), d(z), d(lam), d(st), d(hist), 40) == 0
        assert st[0, ST['STATUS']] == 0
        st[0, ST['CYC_TOTAL']] = st[0, ST['CYC_KKT']] = 0      # (time stamps)
        runs[poison] = (z, lam, st, hist)
    for a, b in zip(runs['0'], runs['1']):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('N,start', [(70, 'profile'), (70, 'reference'), (40, 'profile')])
def test_emulated_one_brake_kernels_match_oracle(emu, N, start):
    """
    The kernels with the structure of the reference's scripts compiled in (FULL_RG: forceMinPn = 0, figure10.py:17): first pass (fused
    iteration, behind the least-squares multiplier estimate when the start is the reference's) + follow-up kernel, as host threads.
    """
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from oracle import oracle
    train, track = cases.train_fig10(), cases.track_00(30000)
    T = 1100.0
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=300, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint=start)
    assert not solver.withPnBrake
    scen = solver._scenarios(T, 0, 1, 1)
    nz = 4*N + 2
    z, lam, st, hist = np.zeros((1, nz)), np.zeros((1, 7*N)), np.zeros((1, ST['COUNT'])), np.zeros((8, 8))
    d = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert emu.emu_solve_batch(ctypes.byref(solver._desc), 1, d(scen), d(z), d(lam), d(st), d(hist), 8) == 0
    prob = cases.oracle_problem(train, track, N)
    ref = oracle.solve(prob, prob.scenario(T), start=start)
    assert st[0, ST['STATUS']] == 0 and ref['stats']['STATUS'] == 0
    # (the last convergence test looks at a dual infeasibility that is rounding noise by then: one of the two may take a barrier reduction more;
    #  tests/test_gpu_parity.py allows the same two iterations)
    assert abs(int(st[0, ST['ITERS']]) - int(ref['stats']['ITERS'])) <= 2
    assert np.max(np.abs(z[0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-6      # (1e-7 where the two end at the same barrier parameter)


@pytest.mark.parametrize('N,variant', [(300, 'fig10'), (300, 'both'), (530, 'fig10')])
def test_emulated_multiwave_geometries_match_oracle(emu, N, variant):
    """
    The multi-wave geometries of the round-2 fence question -- 192 x 2 (three waves, N = 300: the last wave has idle node slots) and
    320 x 2 (five waves) -- as host threads: same iterates as the oracle.  Also what the sanitizer run (tests/hip_emu/run_sanitizers.sh)
    sees of them: uninitialised or out-of-bounds reads of a node slot nobody owns would show here.
    """
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from oracle import oracle
    train = cases.train_default() if variant == 'both' else cases.train_fig10()
    track = cases.track_00()
    T = 1600.0
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=300, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='profile')
    scen = solver._scenarios(T, 0, 1, 1)
    nz = (4 + int(solver.withPnBrake))*N + 2
    z, lam, st, hist = np.zeros((1, nz)), np.zeros((1, 7*N)), np.zeros((1, ST['COUNT'])), np.zeros((8, 8))
    d = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert emu.emu_solve_batch(ctypes.byref(solver._desc), 1, d(scen), d(z), d(lam), d(st), d(hist), 8) == 0
    prob = cases.oracle_problem(train, track, N)
    ref = oracle.solve(prob, prob.scenario(T), start='profile')
    assert st[0, ST['STATUS']] == 0 and ref['stats']['STATUS'] == 0
    assert abs(int(st[0, ST['ITERS']]) - int(ref['stats']['ITERS'])) <= 1
    assert int(st[0, ST['N_FALLBACK']]) == 0
    assert np.max(np.abs(z[0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-7


def test_emulated_streamed_kernel_matches_oracle(emu, monkeypatch):
    "The long-horizon variant (node fields, stage blocks and exchange arrays in device memory, serial sweeps) at a thread count the emulation can afford."
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from oracle import oracle
    monkeypatch.setenv('EMU_GEOMETRY', 'stream')
    N, T = 150, 1541.0
    train, track = cases.train_default(), cases.track_00()
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=300, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='profile')
    scen = solver._scenarios(T, 0, 1, 1)
    nz = (4 + int(solver.withPnBrake))*N + 2
    z, lam, st, hist = np.zeros((1, nz)), np.zeros((1, 7*N)), np.zeros((1, ST['COUNT'])), np.zeros((8, 8))
    d = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert emu.emu_solve_batch(ctypes.byref(solver._desc), 1, d(scen), d(z), d(lam), d(st), d(hist), 8) == 0
    prob = cases.oracle_problem(train, track, N)
    ref = oracle.solve(prob, prob.scenario(T), start='profile')
    assert st[0, ST['STATUS']] == 0 and int(st[0, ST['ITERS']]) == int(ref['stats']['ITERS'])
    assert np.max(np.abs(z[0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-8


def test_emulated_warm_start_matches_oracle(emu):
    "msd_solve_batch_warm semantics: same iterates as the oracle's warm start, same optimum as a cold solve, fewer iterations."
    from mseetc.ocp import casadiSolver
    from mseetc._device import ProblemDesc, ST
    from oracle import oracle
    N, crop, T = 40, 16000, 700.0
    train, track = cases.train_default(), cases.track_00(crop)
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=300, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='reference')
    prob = cases.oracle_problem(train, track, N)
    first = oracle.solve(prob, prob.scenario(T))
    assert first['stats']['STATUS'] == 0
    T2 = T*1.01
    cold = oracle.solve(prob, prob.scenario(T2))
    warm = oracle.solve(prob, prob.scenario(T2), guess=first['z'], mu0=1e-2, push=1e-3)
    assert warm['stats']['STATUS'] == 0 and warm['stats']['ITERS'] < cold['stats']['ITERS']
    assert abs(warm['stats']['OBJ'] - cold['stats']['OBJ']) < 1e-7*abs(cold['stats']['OBJ'])
    dp = ctypes.POINTER(ctypes.c_double)
    emu.emu_solve_batch_warm.argtypes = [ctypes.POINTER(ProblemDesc), ctypes.c_int, dp, dp, dp, ctypes.c_double, ctypes.c_double, dp, dp, dp, dp,
                                         ctypes.c_int]
    scen = solver._scenarios(T2, 0, 1, 1)
    nz = (4 + int(solver.withPnBrake))*N + 2
    z, lam, st, hist = np.zeros((1, nz)), np.zeros((1, 7*N)), np.zeros((1, ST['COUNT'])), np.zeros((8, 8))
    d = lambda a: a.ctypes.data_as(dp)
    guess = np.ascontiguousarray(first['z'])
    assert emu.emu_solve_batch_warm(ctypes.byref(solver._desc), 1, d(scen), None, d(guess), 1e-2, 1e-3, d(z), d(lam), d(st), d(hist), 8) == 0
    assert st[0, ST['STATUS']] == 0
    assert int(st[0, ST['ITERS']]) == int(warm['stats']['ITERS'])
    assert np.max(np.abs(z[0] - warm['z'])/np.maximum(1, np.abs(warm['z']))) < 1e-8


def test_emulated_primal_dual_warm_start_matches_oracle(emu):
    """
    A solve records its multipliers; the re-solve of the horizon shortened by two intervals starts from the tail of its solution
    and of its multipliers (barrier parameter 1e-4, no least-squares estimate): same iterates as the oracle's primal-dual warm
    start, a third of the iterations of a cold solve.
    """
    import copy
    from mseetc.ocp import casadiSolver
    from mseetc.track import computeDiscretizationPoints
    from mseetc._device import ProblemDesc, ST
    from oracle import oracle
    dp = ctypes.POINTER(ctypes.c_double)
    emu.emu_solve_batch_warm.argtypes = [ctypes.POINTER(ProblemDesc), ctypes.c_int, dp, dp, dp, ctypes.c_double, ctypes.c_double, dp, dp, dp, dp, ctypes.c_int]
    emu.emu_set_duals.argtypes = [dp, ctypes.c_longlong, dp]
    emu.emu_set_duals.restype = None
    d = lambda a: a.ctypes.data_as(dp)
    N, crop, T = 40, 16000, 700.0
    train, track = cases.train_default(), cases.track_00(crop)
    opts = lambda n: dict(numIntervals=n, maxIterations=300, integrationOptions=dict(numSteps=1, numApproxSteps=1))
    s1 = casadiSolver(train, track, opts(N), startingPoint='profile')
    stp = 4 + int(s1.withPnBrake)
    z1, lam1, st1, hist = np.zeros((1, stp*N + 2)), np.zeros((1, 7*N)), np.zeros((1, ST['COUNT'])), np.zeros((8, 8))
    duals1 = np.zeros((N + 1, oracle.DUAL_STRIDE))
    emu.emu_set_duals(None, 0, d(duals1))
    assert emu.emu_solve_batch(ctypes.byref(s1._desc), 1, d(s1._scenarios(T, 0, 1, 1)), d(z1), d(lam1), d(st1), d(hist), 8) == 0
    prob1 = cases.oracle_problem(train, track, N)
    ref1 = oracle.solve_dual(prob1, prob1.scenario(T), start='profile')
    assert st1[0, ST['STATUS']] == 0 and abs(int(st1[0, ST['ITERS']]) - int(ref1['stats']['ITERS'])) <= 1
    assert np.allclose(duals1, ref1['duals'], rtol=1e-3, atol=1e-6)      # converged multipliers (the two sides may stop one iteration apart)
    # two intervals further: cropped track, measured state slightly off the plan
    pos = computeDiscretizationPoints(track, N).index.values
    track2 = copy.deepcopy(track); track2.updateLimits(positionStart=float(pos[2]))
    s2 = casadiSolver(train, track2, opts(N - 2), startingPoint='profile')
    t_now, v_now = ref1['z'][stp*2 + stp - 2]*1.004, np.sqrt(ref1['z'][stp*2 + stp - 1])*0.996
    scen2 = s2._scenarios(T, t_now, 1, v_now)
    guess = np.ascontiguousarray(ref1['z'][stp*2:])
    z2, lam2, st2 = np.zeros((1, stp*(N - 2) + 2)), np.zeros((1, 7*(N - 2))), np.zeros((1, ST['COUNT']))
    duals2 = np.zeros((N - 1, oracle.DUAL_STRIDE))
    tail = np.ascontiguousarray(ref1['duals'][2:])      # both sides start from the same numbers
    emu.emu_set_duals(d(tail), 0, d(duals2))
    assert emu.emu_solve_batch_warm(ctypes.byref(s2._desc), 1, d(scen2), None, d(guess), 1e-4, 1e-3, d(z2), d(lam2), d(st2), d(hist), 8) == 0
    emu.emu_set_duals(None, 0, None)
    prob2 = cases.oracle_problem(train, track2, N - 2)
    sc2 = prob2.scenario(T, t_now, 1.0, v_now)
    warm = oracle.solve_dual(prob2, sc2, guess=guess, duals=ref1['duals'][2:], mu0=1e-4, push=1e-3)
    cold = oracle.solve(prob2, sc2, start='profile')
    assert warm['stats']['STATUS'] == 0 and st2[0, ST['STATUS']] == 0
    assert abs(int(st2[0, ST['ITERS']]) - int(warm['stats']['ITERS'])) <= 1
    assert int(st2[0, ST['ITERS']]) <= 0.6*int(cold['stats']['ITERS'])      # (9 or 10 against 18: the last convergence test can fall either way)
    assert np.max(np.abs(z2[0] - warm['z'])/np.maximum(1, np.abs(warm['z']))) < 1e-7
    assert abs(st2[0, ST['OBJ']] - cold['stats']['OBJ']) <= 1e-7*abs(cold['stats']['OBJ'])


@pytest.mark.parametrize('variant', ['dynamic_losses', 'no_pneumatic_brake_time_optimal'])
def test_emulated_kernel_other_stage_systems(emu, variant):
    "The two stage-system variants the static both-brakes cases do not reach: dynamic loss rows (couplings folded in assemble) and no Fpb."
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from mseetc.train import Train
    from mseetc.track import computeDiscretizationPoints
    from oracle import oracle
    N = 30
    track = cases.track_00(8500)
    if variant == 'dynamic_losses':
        from mseetc.efficiency import totalLossesFunction
        train = Train(config={'id': 'NL_Intercity_VIRM6'})
        train.forceMinPn = 0
        train.powerLosses = totalLossesFunction(train, auxiliaries=27000, etaGear=0.96)
        eo, T = True, 272.4726*1.2
        oracle.set_loss_table(train.powerLosses.parameters(train.mass*train.rho))
        pts = computeDiscretizationPoints(track, N)
        prob = oracle.pack_problem(train, pts, dict(numIntervals=N, maxIterations=300, energyOptimal=True, minimumVelocity=1, numSteps=1, numApproxSteps=1),
                                   2, 0.0, 0.0, track.length)
    else:
        train = cases.train_fig5()
        eo, T = False, 400.0
        prob = cases.oracle_problem(train, track, N, energyOptimal=False, losses='none', maxIterations=300)
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=300, energyOptimal=eo, integrationOptions=dict(numSteps=1, numApproxSteps=1)),
                          startingPoint='profile')
    scen = solver._scenarios(T, 0, 100/3.6, 1)
    nz = (4 + int(solver.withPnBrake))*N + 2
    rows = solver.problem_rows if hasattr(solver, 'problem_rows') else (2 + 3 + (2 if eo else 0))
    z, lam, st, hist = np.zeros((1, nz)), np.zeros((1, rows*N)), np.zeros((1, ST['COUNT'])), np.zeros((8, 8))
    d = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert emu.emu_solve_batch(ctypes.byref(solver._desc), 1, d(scen), d(z), d(lam), d(st), d(hist), 8) == 0
    ref = oracle.solve(prob, prob.scenario(T, 0.0, 100/3.6, 1.0), start='profile')
    assert st[0, ST['STATUS']] == 0 and ref['stats']['STATUS'] == 0
    assert abs(int(st[0, ST['ITERS']]) - int(ref['stats']['ITERS'])) <= 1
    assert abs(st[0, ST['OBJ']] - ref['stats']['OBJ']) <= 1e-9*abs(ref['stats']['OBJ'])
    assert np.max(np.abs(z[0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-7


@pytest.mark.skipif(not __import__('os').environ.get('RUN_SANITIZERS'), reason="opt-in (several minutes): RUN_SANITIZERS=1")
def test_emulation_under_sanitizers():
    "ASan + UBSan over the emulated kernel (all emulation tests, every geometry they use): tests/hip_emu/run_sanitizers.sh must pass."
    r = subprocess.run([str(EMU / 'run_sanitizers.sh')], capture_output=True, text=True, timeout=3600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-2000:])
