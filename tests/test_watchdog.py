"""
IPOPT's watchdog procedure (what IPOPT does behind ocp.py:290,359 -- the reference leaves watchdog_shortened_iter_trigger = 10 and
watchdog_trial_iter_max = 3 at their defaults).  oracle/ms_oracle.c: solve_core restates it (header there: restated from the published
implementation, IPOPT's sources are neither in the reference tree nor in this image -- parity unpinned), msd_kernel.hpp: Solver::run is the device
code: the general iteration runs the procedure, the fused iteration counts the shortened iterations and hands the scenario over when it is due.

Pins: (1) the benchmark schedules never reach the trigger -- on / off changes nothing there; (2) with the trigger lowered (an option of IPOPT the
tests use, `watchdogTrigger`) the procedure starts on ordinary problems: the emulated kernel (CPU) and the GPU kernel follow the oracle iterate for
iterate through start, success and stop; (3) on loose schedules of long horizons it starts at IPOPT's own trigger, in the oracle and on the device
alike (GPU test below, profiles/r04/watchdog_survey.txt); (4) the trial points a running procedure takes without the filter's consent: oracle only
-- they occur on degenerate zero-cost journeys, where the last bits decide the path and no two implementations stay together.
"""

import ctypes

import numpy as np
import pytest

import cases


def test_oracle_watchdog_is_dormant_on_the_benchmark_schedules():
    from oracle import oracle
    from mseetc import workloads
    prob = cases.oracle_problem(*workloads.config('c1'))
    off = cases.oracle_problem(*workloads.config('c1'), watchdogTrigger=-1)
    scen = np.array([[0.0, T, 1.0, 1.0] for T in workloads.c1_times(24)])
    for start in ('profile', 'reference'):
        oracle.watchdog_counts(True)
        z1, st1, nf1 = oracle.solve_batch(prob, scen, start=start)
        assert oracle.watchdog_counts(True) == (0, 0)
        z0, st0, nf0 = oracle.solve_batch(off, scen, start=start)
        assert nf0 == nf1 == 0 and np.array_equal(z0, z1) and np.array_equal(st0[:, :oracle.ST['N_WATCHDOG']], st1[:, :oracle.ST['N_WATCHDOG']])


@pytest.mark.parametrize('N,crop,T,trigger,expect', [(30, 12000, 3000.0, 1, (4, 2)), (30, 12000, 3000.0, 2, (1, 0)), (100, None, 12000.0, 1, (5, 1))])
def test_oracle_watchdog_with_a_lowered_trigger(N, crop, T, trigger, expect):
    """
    Start, success (the trial point passes against the stored reference values: the filter gets the reference point) and stop (the trial point
    cannot be evaluated: back to the stored point, line search from half the maximal step) on ordinary problems; same optimum as without.
    """
    from oracle import oracle
    train, track = cases.train_default(), (cases.track_00(crop) if crop else cases.track_00())
    on = cases.oracle_problem(train, track, N, watchdogTrigger=trigger)
    off = cases.oracle_problem(train, track, N, watchdogTrigger=-1)
    oracle.watchdog_counts(True); oracle.watchdog_forced_steps(True)
    r1 = oracle.solve(on, on.scenario(T), start='profile')
    counts = oracle.watchdog_counts(True)
    assert counts == expect and int(r1['stats']['N_WATCHDOG']) == expect[0] and oracle.watchdog_forced_steps(True) == 0
    r0 = oracle.solve(off, off.scenario(T), start='profile')
    assert oracle.watchdog_counts(True) == (0, 0) and r0['stats']['N_WATCHDOG'] == 0
    assert r1['stats']['STATUS'] == r0['stats']['STATUS'] == 0
    assert abs(r1['stats']['OBJ'] - r0['stats']['OBJ']) <= 1e-9*abs(r0['stats']['OBJ'])


def test_oracle_watchdog_takes_trial_points_without_the_filters_consent(tmp_path):
    """
    The part of the procedure the ordinary problems never reach: three times the minimum running time on a random track (a zero-cost journey: the
    energy optimum is degenerate, the iteration crawls with shortened steps).  The procedure starts at IPOPT's own trigger, takes a trial point
    the filter would not accept and ends with one it accepts against the stored reference: 79 instead of 139 iterations to the same optimum.
    (Round 5: seed 86 -- rounds 3-4 used seed 92, whose path to the optimum changed with the last interval's elimination: ms_oracle.c, compute_direction.)
    """
    from oracle import oracle
    from test_gpu_parity import _random_problem      # (the generator only: no GPU call)
    train, track, N, rng = _random_problem(86, tmp_path)
    v0, vN = float(rng.uniform(1, 15)), float(rng.uniform(1, 15))
    pt = cases.oracle_problem(train, track, N, energyOptimal=False)
    rt = oracle.solve(pt, pt.scenario(3*track.length/train.velocityMax, 0.0, vN, v0), start='profile')
    assert rt['stats']['STATUS'] == 0
    T = 3.0*float(rt['z'][-2])
    res = {}
    for trig in (0, -1):
        pe = cases.oracle_problem(train, track, N, watchdogTrigger=trig)
        oracle.watchdog_counts(True); oracle.watchdog_forced_steps(True)
        r = oracle.solve(pe, pe.scenario(T, 0.0, vN, v0), start='reference')
        res[trig] = (r, oracle.watchdog_counts(True), oracle.watchdog_forced_steps(True))
    on, off = res[0], res[-1]
    assert on[1] == (1, 1) and on[2] >= 1 and off[1] == (0, 0) and off[2] == 0
    assert on[0]['stats']['STATUS'] == off[0]['stats']['STATUS'] == 0
    assert on[0]['stats']['ITERS'] < off[0]['stats']['ITERS']
    assert abs(on[0]['stats']['OBJ']) < 1e-4 and abs(off[0]['stats']['OBJ']) < 1e-4      # kWh: a journey that costs nothing


@pytest.mark.parametrize('N,crop,T,trigger', [(30, 12000, 3000.0, 1), (30, 12000, 3000.0, 2)])
def test_emulated_kernel_follows_the_oracle_through_the_watchdog(N, crop, T, trigger):
    """
    The kernel code on host threads (tests/hip_emu): the first pass counts the shortened iterations and hands the scenario to the follow-up
    kernel, whose general iteration starts the procedure where the oracle does -- same number of procedures, same iteration count, same point.
    """
    from test_kernel_emulation import load_emulation
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from oracle import oracle
    emu = load_emulation()
    train, track = cases.train_default(), cases.track_00(crop)
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=300, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='profile',
                          watchdogTrigger=trigger)
    scen = solver._scenarios(T, 0, 1, 1)
    nz = (4 + int(solver.withPnBrake))*N + 2
    z, lam, st, hist = np.zeros((1, nz)), np.zeros((1, 7*N)), np.zeros((1, ST['COUNT'])), np.zeros((8, 8))
    d = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert emu.emu_solve_batch(ctypes.byref(solver._desc), 1, d(scen), d(z), d(lam), d(st), d(hist), 8) == 0
    prob = cases.oracle_problem(train, track, N, watchdogTrigger=trigger)
    ref = oracle.solve(prob, prob.scenario(T), start='profile')
    assert st[0, ST['STATUS']] == 0 and ref['stats']['STATUS'] == 0
    assert int(st[0, ST['N_WATCHDOG']]) == int(ref['stats']['N_WATCHDOG']) >= 1
    assert int(st[0, ST['ITERS']]) == int(ref['stats']['ITERS'])
    assert int(st[0, ST['N_BACKTRACK']]) == int(ref['stats']['N_BACKTRACK'])
    assert np.max(np.abs(z[0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-7


def _wd_family_case(family, trigger):
    "(train, track, N, T, solver options, oracle problem, solve keywords): a solve of the family on which the oracle's watchdog starts with the lowered trigger"
    from oracle.oracle import IP
    import test_restoration
    if family == 'time-optimal':      # the general static family (no structure compiled in): the time-optimal twin of the config-1 problem
        N, T = 30, 3000.0
        train, track, kw = cases.train_default(), cases.track_00(12000), {}
        opts = dict(numIntervals=N, maxIterations=300, energyOptimal=False, integrationOptions=dict(numSteps=1, numApproxSteps=1))
        prob = cases.oracle_problem(train, track, N, energyOptimal=False, losses='none', maxIterations=300, watchdogTrigger=trigger)
    else:
        N, T = (60, 310.0) if family == 'dynamic' else (30, 3000.0)
        train, track, opts, prob, kw = test_restoration._family_case(family, N, 300)
        prob.ip[IP['WATCHDOG_TRIGGER']] = trigger
    return train, track, N, T, opts, prob, kw


@pytest.mark.parametrize('family,trigger', [('time-optimal', 1), ('dynamic', 1), ('integrateLosses', 1), ('integrateLosses', 2)])
def test_emulated_watchdog_in_the_other_kernel_families(family, trigger):
    """
    Round 5: IPOPT's watchdog procedure for every kernel family.  The LDS-resident kernels without the structure of the NLP compiled in -- the general
    static family (here: a time-optimal problem), the dynamic loss table, integrateLosses -- are first-pass kernels: they count the shortened iterations and
    hand the scenario to the streamed follow-up kernel of their family when the procedure is due (rounds 1-4: they only counted).  Kernel code on host
    threads against the oracle: same number of procedures, same iteration count, same point.
    """
    from test_kernel_emulation import load_emulation
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from oracle import oracle
    emu = load_emulation()
    train, track, N, T, opts, prob, kw = _wd_family_case(family, trigger)
    solver = casadiSolver(train, track, opts, startingPoint='profile', watchdogTrigger=trigger)
    scen = solver._scenarios(T, 0, kw.get('terminalVelocity', 1), kw.get('initialVelocity', 1))
    nz = (4 + int(solver.withPnBrake))*N + 2
    z, lam, st, hist = np.zeros((1, nz)), np.zeros((1, 7*N)), np.zeros((1, ST['COUNT'])), np.zeros((8, 8))
    d = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert emu.emu_solve_batch(ctypes.byref(solver._desc), 1, d(scen), d(z), d(lam), d(st), d(hist), 8) == 0
    ref = oracle.solve(prob, prob.scenario(T, **kw), start='profile')
    assert st[0, ST['STATUS']] == 0 and ref['stats']['STATUS'] == 0
    assert int(st[0, ST['N_WATCHDOG']]) == int(ref['stats']['N_WATCHDOG']) >= 1
    assert int(st[0, ST['ITERS']]) == int(ref['stats']['ITERS'])
    assert np.max(np.abs(z[0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-6


# ------------------------------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu
RK11 = dict(numSteps=1, numApproxSteps=1)


@gpu
def test_gpu_watchdog_with_a_lowered_trigger_vs_oracle():
    "The HIP kernels through start / success / stop of the procedure, scenario for scenario against the oracle; the hand-over is counted (reason 6)."
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from oracle import oracle
    train, track, N = cases.train_default(), cases.track_00(), 100
    Ts = [1500.0, 3000.0, 6000.0, 9000.0, 12000.0]
    for trigger in (1, 2):
        s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationOptions=RK11), startingPoint='profile', watchdogTrigger=trigger)
        res = s.solveBatch(Ts)
        total, why = s.problem.follow_counts()
        s.close()
        prob = cases.oracle_problem(train, track, N, watchdogTrigger=trigger)
        started = 0
        for k, T in enumerate(Ts):
            ref = oracle.solve(prob, prob.scenario(T), start='profile')
            assert res['status'][k] == int(ref['stats']['STATUS']) == 0
            assert int(res['stats'][k, ST['N_WATCHDOG']]) == int(ref['stats']['N_WATCHDOG'])
            assert abs(int(res['iterations'][k]) - int(ref['stats']['ITERS'])) <= 1
            assert abs(res['cost'][k] - ref['stats']['OBJ']) <= 1e-8*abs(ref['stats']['OBJ'])
            assert np.max(np.abs(res['z'][k] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-5
            started += int(ref['stats']['N_WATCHDOG'])
        assert started >= 2 and why[6] >= 1      # (reason 6: the fused iteration reached the trigger and left the scenario to the general one)


@gpu
@pytest.mark.parametrize('N', [600, 700])
def test_gpu_watchdog_starts_at_ipopts_trigger_on_loose_schedules_of_long_horizons(N):
    """
    IPOPT's own trigger (10 shortened iterations in a row): loose schedules on 600 / 700 intervals from the reference's starting point run into it
    -- in the oracle and on the device (five-wave kernel followed up by the streamed one; the streamed kernel) -- between restoration phases.
    These solves crawl for hundreds of iterations with steps of 1e-4 and less; the two implementations do not stay together iterate for iterate
    there (tools/wd_probe.py, profiles/r04/watchdog_device.txt: the procedure starts in both, not always in the same scenarios), the optimum
    they reach is the same to the last digits.
    """
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from oracle import oracle
    train, track = cases.train_default(), cases.track_00()
    Ts = np.linspace(8000, 20000, 8)
    s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=800, integrationOptions=RK11), startingPoint='reference')
    res = s.solveBatch(Ts)
    s.close()
    prob = cases.oracle_problem(train, track, N, maxIterations=800)
    z, st, nfail = oracle.solve_batch(prob, np.array([[0.0, T, 1.0, 1.0] for T in Ts]), start='reference')
    assert nfail == 0 and np.all(res['status'] == 0)
    assert st[:, oracle.ST['N_WATCHDOG']].sum() >= 1 and res['stats'][:, ST['N_WATCHDOG']].sum() >= 1
    assert st[:, oracle.ST['N_RESTO']].sum() >= 1 and res['stats'][:, ST['N_RESTO']].sum() >= 1
    assert np.max(np.abs(res['cost'] - st[:, oracle.ST['OBJ']])/np.abs(st[:, oracle.ST['OBJ']])) <= 1e-7
    assert np.max(np.abs(res['z'] - z)/np.maximum(1, np.abs(z))) < 1e-4


@gpu
@pytest.mark.parametrize('family,trigger', [('time-optimal', 1), ('dynamic', 1), ('integrateLosses', 1), ('integrateLosses', 2)])
def test_gpu_watchdog_in_the_other_kernel_families(family, trigger):
    "The same on the device: first-pass kernel (hand-over, reason 6) + streamed follow-up kernel of the family, against the oracle."
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from oracle import oracle
    train, track, N, T, opts, prob, kw = _wd_family_case(family, trigger)
    s = casadiSolver(train, track, opts, startingPoint='profile', watchdogTrigger=trigger)
    res = s.solveBatch([T], **kw)
    total, why = s.problem.follow_counts()
    s.close()
    ref = oracle.solve(prob, prob.scenario(T, **kw), start='profile')
    assert res['status'][0] == int(ref['stats']['STATUS']) == 0
    assert int(res['stats'][0, ST['N_WATCHDOG']]) == int(ref['stats']['N_WATCHDOG']) >= 1 and why[6] >= 1
    assert abs(int(res['iterations'][0]) - int(ref['stats']['ITERS'])) <= 1
    assert abs(res['stats'][0, ST['OBJ']] - ref['stats']['OBJ']) <= 1e-8*max(1.0, abs(ref['stats']['OBJ']))      # (the NLP's objective; `cost` is seconds for a time-optimal problem)
    assert np.max(np.abs(res['z'][0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-5
