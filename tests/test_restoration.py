"""
Feasibility restoration phase (what IPOPT does behind ocp.py:290,359 when its filter line search breaks down, surfaced at
ocp.py:362-370): oracle/ms_oracle.c: restoration() restates IPOPT's MinC_1Nrm restoration phase, csrc/msd_resto.hpp is the device code.

No reference-held vector exists for it (CasADi/IPOPT cannot run here): the pins are (1) the Newton system of the restoration problem
checked equation by equation inside the oracle (ORACLE_DEBUG >= 2: direction_residual), (2) the optimum a restored solve reaches being the
optimum the other starting point reaches without restoration, (3) the kernel following the oracle iterate for iterate through the
restoration phases (emulation here, GPU below).
"""

import ctypes

import numpy as np
import pytest

import cases


def _loose_case():
    # 8 to 13 times the minimum running time from the reference's starting point: the power rows are violated by 60 % there and the
    # filter line search breaks down on the way (test_gpu_parity.py::test_loose_schedules_converge_from_both_starts)
    return cases.train_default(), cases.track_00(), 100, [12000.0, 20000.0]


def test_oracle_restoration_reaches_the_optimum_of_the_other_start():
    from oracle import oracle
    train, track, N, Ts = _loose_case()
    prob = cases.oracle_problem(train, track, N)
    for T in Ts:
        ref = oracle.solve(prob, prob.scenario(T), start='profile')
        assert ref['stats']['STATUS'] == 0 and ref['stats']['N_RESTO'] == 0
        res = oracle.solve(prob, prob.scenario(T), start='reference')
        # converged from the reference's point through the restoration phase -- not by the restart from the other starting point,
        # which would have added the failed attempt's iterations and left N_RESTO = 0 in the record of the second attempt
        assert res['stats']['STATUS'] == 0 and res['stats']['N_RESTO'] >= 1
        assert abs(res['stats']['OBJ'] - ref['stats']['OBJ']) <= 1e-8*abs(ref['stats']['OBJ'])
        assert np.max(np.abs(res['z'] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-6
    # without the phase the same solve ends where the line search breaks down (and the restart has to rescue it)
    oracle.lib().oracle_set_restoration(0)
    try:
        res = oracle.solve(prob, prob.scenario(Ts[0]), start='reference')
        assert res['stats']['STATUS'] == 0 and res['stats']['N_RESTO'] == 0
    finally:
        oracle.lib().oracle_set_restoration(1)


def test_oracle_restoration_on_an_infeasible_running_time():
    """
    A running time below the minimum: the restoration phases reduce the infeasibility to what the train cannot make up and the solve
    ends with a failure status -- Restoration_Failed (-2: the restoration problem's own line search breaks down; its 1-norm objective
    leaves the distribution of the missing time over the intervals undetermined, the Newton systems are nearly singular along it),
    Infeasible_Problem_Detected (-6) or the iteration limit.  Never a success, and the infeasibility that is left is the missing time.
    """
    from oracle import oracle
    train, track = cases.train_default(), cases.track_00(12000)
    prob = cases.oracle_problem(train, track, 30, maxIterations=200)
    res = oracle.solve(prob, prob.scenario(300.0), start='reference')
    assert res['stats']['STATUS'] in (-1, -2, -6) and res['stats']['N_RESTO'] >= 1
    fast = cases.oracle_problem(train, track, 30, energyOptimal=False)
    tmin = oracle.solve(fast, fast.scenario(2000.0), start='profile')
    assert tmin['stats']['STATUS'] == 0 and tmin['z'][-2] > 300.0


def test_emulated_restoration_phase_follows_the_oracle():
    """
    The kernel (host emulation: tests/hip_emu) through two restoration phases of an infeasible problem, history row by history row
    against the oracle: entry (iteration and point), the restoration iterations (original objective and infeasibility at the restoration
    iterates, dual infeasibility and barrier parameter of the restoration problem) and the return to the general iteration.
    """
    from test_kernel_emulation import load_emulation
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from oracle import oracle
    emu = load_emulation()
    N, crop, T, cap = 30, 12000, 300.0, 64
    train, track = cases.train_default(), cases.track_00(crop)
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=60, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='reference')
    scen = solver._scenarios(T, 0, 1, 1)
    nz = (4 + int(solver.withPnBrake))*N + 2
    z, lam, st, hist = np.zeros((1, nz)), np.zeros((1, 7*N)), np.zeros((1, ST['COUNT'])), np.zeros((cap, 8))
    d = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert emu.emu_solve_batch(ctypes.byref(solver._desc), 1, d(scen), d(z), d(lam), d(st), d(hist), cap) == 0
    prob = cases.oracle_problem(train, track, N, maxIterations=60)
    ref = oracle.solve(prob, prob.scenario(T), start='reference', history=True)
    assert int(st[0, ST['STATUS']]) == int(ref['stats']['STATUS']) == -1
    # the iteration limit after a restoration phase counts as a breakdown: both sides repeat the solve from the other starting point (60 + 60
    # iterations; the statistics and the history are those of the second attempt, which goes through restoration phases of its own)
    assert int(st[0, ST['ITERS']]) == int(ref['stats']['ITERS']) == 120
    assert int(st[0, ST['N_RESTO']]) == int(ref['stats']['N_RESTO']) >= 2
    h = ref['hist']
    for i in range(60):
        assert np.allclose(hist[i, 1:5], h[i, 1:5], rtol=1e-4, atol=1e-9), (i, hist[i], h[i])
    assert np.max(np.abs(z[0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-4


def _family_case(family, N, maxit, crop=12000):
    "(train, track, solver options, oracle problem, solve keywords) of the kernel families outside the LDS-resident Runge-Kutta kernels"
    from oracle import oracle
    from mseetc.track import computeDiscretizationPoints
    train, track, kw = cases.train_default(), cases.track_00(crop) if crop else cases.track_00(), {}
    if family in ('streamed', 'RK'):
        opts = dict(numIntervals=N, maxIterations=maxit, integrationOptions=dict(numSteps=1, numApproxSteps=1))
        prob = cases.oracle_problem(train, track, N, maxIterations=maxit)
    elif family in ('collocation', 'IRK'):
        opts = dict(numIntervals=N, maxIterations=maxit, integrationMethod='IRK', integrationOptions=dict(order=2, numSteps=1, numApproxSteps=0))
        prob = cases.oracle_problem(train, track, N, numSteps=1, numApproxSteps=0, maxIterations=maxit,
                                    integration=dict(integrationMethod='IRK', order=2, collMethod='radau', maxIter=10))
    elif family == 'CVODES':
        opts = dict(numIntervals=N, maxIterations=maxit, integrationMethod='CVODES', integrationOptions=dict())
        prob = cases.oracle_problem(train, track, N, numSteps=1, numApproxSteps=0, maxIterations=maxit, integration=dict(integrationMethod='CVODES', absTol=1e-8, relTol=1e-6))
    elif family == 'integrateLosses':
        opts = dict(numIntervals=N, maxIterations=maxit, integrateLosses=True, integrationOptions=dict(numSteps=1, numApproxSteps=1))
        prob = cases.oracle_problem(train, track, N, maxIterations=maxit, integration=dict(integrateLosses=True))
    else:
        # the dynamic loss model on the configuration of simulations/figure5.py (8.5 km, v0 = 1 m/s, vN = 100 km/h, forceMinPn = 0)
        from mseetc.train import Train
        from mseetc.efficiency import totalLossesFunction
        assert family == 'dynamic'
        train = Train(config={'id': 'NL_Intercity_VIRM6'})
        train.forceMinPn = 0
        train.powerLosses = totalLossesFunction(train, auxiliaries=27000, etaGear=0.96)
        track = cases.track_00(8500)
        oracle.set_loss_table(train.powerLosses.parameters(train.mass*train.rho))
        opts = dict(numIntervals=N, maxIterations=maxit, integrationOptions=dict(numSteps=1, numApproxSteps=1))
        prob = oracle.pack_problem(train, computeDiscretizationPoints(track, N), dict(numIntervals=N, maxIterations=maxit, energyOptimal=True, minimumVelocity=1,
                                                                                     numSteps=1, numApproxSteps=1), 2, 0.0, 0.0, track.length)
        kw = dict(terminalVelocity=100/3.6, initialVelocity=1)
    return train, track, opts, prob, kw


@pytest.mark.parametrize('family', ['streamed', 'collocation', 'dynamic', 'integrateLosses'])
def test_emulated_restoration_phase_in_the_other_kernel_families(family, monkeypatch):
    """
    The phase outside the LDS-resident Runge-Kutta kernels.  `streamed`: the long-horizon kernels (node fields and stage blocks in device memory): a first
    pass + the follow-up kernel of the same geometry, which holds the phase -- same history as the oracle, row by row.  `collocation`, `dynamic` (loss
    table of efficiency.py), `integrateLosses` (ocp.py:231-241): LDS-resident first-pass kernels; the scenario whose line search breaks down is listed and
    solved again, with the phase, by the streamed follow-up kernel of the family (msd_api.hip: make_plan; the emulation launches the two like the host code
    does).  Round 5 for the last two: their loss rows reach into the next node (b_{i+1}, resp. t_{i+1} through the running time), which the Newton system
    of the restoration problem carries as cross terms (riccati_resto; the oracle's compute_direction, whose Newton residuals ORACLE_DEBUG=2 prints).
    An infeasible running time: both sides go through restoration phases and end at the iteration limit of the second attempt.
    """
    from test_kernel_emulation import load_emulation
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from oracle import oracle
    emu = load_emulation()
    N, cap = 30, 64
    T = 200.0 if family == 'dynamic' else 300.0
    if family == 'streamed':
        monkeypatch.setenv('EMU_GEOMETRY', 'stream')
    maxit = 40      # (per attempt: the emulation runs the streamed kernel's 512 lanes as host threads -- a second per iteration)
    train, track, opts, prob, kw = _family_case(family, N, maxit)
    solver = casadiSolver(train, track, opts, startingPoint='reference')
    scen = solver._scenarios(T, 0, kw.get('terminalVelocity', 1), kw.get('initialVelocity', 1))
    nz = (4 + int(solver.withPnBrake))*N + 2
    z, lam, st, hist = np.zeros((1, nz)), np.zeros((1, 7*N)), np.zeros((1, ST['COUNT'])), np.zeros((cap, 8))
    d = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert emu.emu_solve_batch(ctypes.byref(solver._desc), 1, d(scen), d(z), d(lam), d(st), d(hist), cap) == 0
    ref = oracle.solve(prob, prob.scenario(T, **kw), start='reference', history=True)
    assert int(st[0, ST['STATUS']]) == int(ref['stats']['STATUS']) == -1
    assert int(st[0, ST['ITERS']]) == int(ref['stats']['ITERS']) == 2*maxit
    assert int(st[0, ST['N_RESTO']]) == int(ref['stats']['N_RESTO']) >= 2
    h = ref['hist']
    # (collocation: the Newton iteration inside the integrator -- OptionsIRK.maxIter steps, no convergence test, like casadi.simpleIRK -- lets the two drift
    #  apart: 1.2e-4 in the dual infeasibility of the restoration iterates from row 24 on, 1e-4 in everything after forty iterations)
    #  integrateLosses: the loss integrals come from an adaptive integrator (CVODES' tolerances, msd_lossint.hpp); the two implementations agree to 1e-10 up
    #  to the breakdown at row 20 -- dual infeasibility 3e9 -- and to 1e-5 ... 1e-4 on the restoration iterates behind it, which start from that point)
    for i in range(36 if family == 'collocation' else maxit):
        rtol = 3e-4 if family == 'collocation' else (1e-3 if i > 25 else 1e-4) if family == 'integrateLosses' else 1e-4
        assert np.allclose(hist[i, 1:5], h[i, 1:5], rtol=rtol, atol=1e-9), (i, hist[i], h[i])
    assert np.max(np.abs(z[0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < (1e-2 if family == 'collocation' else 1e-4)


# ------------------------------------------------------------------------------------------------------------------------------
gpu = pytest.mark.gpu
RK11 = dict(numSteps=1, numApproxSteps=1)      # the transcription cases.oracle_problem packs by default


@gpu
def test_gpu_loose_schedules_from_the_reference_start_through_restoration():
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from oracle import oracle
    train, track, N, Ts = _loose_case()
    opts = dict(numIntervals=N, maxIterations=500, integrationOptions=RK11)
    ref = casadiSolver(train, track, opts, startingPoint='profile')
    best = ref.solveBatch(Ts)
    ref.close()
    assert np.all(best['status'] == 0) and np.all(best['stats'][:, ST['N_RESTO']] == 0)
    s = casadiSolver(train, track, opts, startingPoint='reference')
    res = s.solveBatch(Ts)
    s.close()
    assert np.all(res['status'] == 0), res['status']
    assert np.all(res['stats'][:, ST['N_RESTO']] >= 1)          # through the restoration phase, not the restart
    assert np.max(np.abs(res['cost'] - best['cost'])/np.abs(best['cost'])) < 1e-7
    prob = cases.oracle_problem(train, track, N)
    for k, T in enumerate(Ts):
        chk = oracle.solve(prob, prob.scenario(T), start='reference')
        assert chk['stats']['STATUS'] == 0 and chk['stats']['N_RESTO'] >= 1
        assert abs(res['cost'][k] - chk['stats']['OBJ']) <= 1e-7*abs(chk['stats']['OBJ'])
    # restoration=False: ABI 4's behaviour -- the line search breaks down, the scenario is solved again from the other starting point
    s = casadiSolver(train, track, opts, startingPoint='reference', restoration=False)
    old = s.solveBatch(Ts)
    s.close()
    assert np.all(old['status'] == 0) and np.all(old['stats'][:, ST['N_RESTO']] == 0)
    assert np.max(np.abs(old['cost'] - best['cost'])/np.abs(best['cost'])) < 1e-7
    # the package's default transcription (joint RK4 for the time, numApproxSteps = 0): 12 000 s breaks down from BOTH starting points
    # without the restoration phase, and converges from both to the same optimum with it
    costs = []
    for start in ('profile', 'reference'):
        s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500), startingPoint=start)
        r = s.solveBatch([12000.0])
        s.close()
        assert r['status'][0] == 0 and r['stats'][0, ST['N_RESTO']] >= 1
        costs.append(r['cost'][0])
    assert abs(costs[0] - costs[1]) <= 1e-6*abs(costs[1])


@gpu
def test_gpu_restoration_phase_follows_the_oracle():
    "The GPU kernel through the restoration phases of an infeasible problem, against the oracle (same case as the emulation test, 120 iterations)."
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from oracle import oracle
    N, crop, T = 30, 12000, 300.0
    train, track = cases.train_default(), cases.track_00(crop)
    s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=60, integrationOptions=RK11), startingPoint='reference')
    res = s.solveBatch([T, T*1.0001], classifyFailures=False)
    s.close()
    prob = cases.oracle_problem(train, track, N, maxIterations=60)
    for k, Tk in enumerate([T, T*1.0001]):
        ref = oracle.solve(prob, prob.scenario(Tk), start='reference')
        assert res['status'][k] == int(ref['stats']['STATUS']) == -1
        assert res['iterations'][k] == int(ref['stats']['ITERS'])
        assert int(res['stats'][k, ST['N_RESTO']]) == int(ref['stats']['N_RESTO']) >= 2
        assert abs(res['cost'][k] - ref['stats']['OBJ']) <= 1e-4*abs(ref['stats']['OBJ'])
        assert np.max(np.abs(res['z'][k] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-4


@gpu
@pytest.mark.parametrize('N,variant', [(63, 'both'), (127, 'rg'), (255, 'both'), (300, 'rg'), (450, 'both')])
def test_gpu_restoration_on_every_geometry(N, variant):
    """
    One and several waves per scenario, one and two nodes per lane, general and structure-specialised kernels: a loose schedule from the
    reference's point (restoration phase) reaches the optimum of the profile start.
    """
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    train = cases.train_default() if variant == 'both' else cases.train_fig10()
    track = cases.track_00()
    T = [9000.0, 14000.0]
    ref = casadiSolver(train, track, dict(numIntervals=N, maxIterations=800, integrationOptions=RK11), startingPoint='profile')
    best = ref.solveBatch(T)
    ref.close()
    s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=800, integrationOptions=RK11), startingPoint='reference')
    res = s.solveBatch(T)
    s.close()
    assert np.all(best['status'] == 0) and np.all(res['status'] == 0), (best['status'], res['status'])
    assert np.max(np.abs(res['cost'] - best['cost'])/np.abs(best['cost'])) < 1e-6


@gpu
@pytest.mark.parametrize('family,N,T', [('CVODES', 100, 14000.0), ('CVODES', 100, 20000.0), ('IRK', 300, 9000.0), ('RK', 600, 9000.0), ('RK', 700, 9000.0), ('RK', 1200, 20000.0),
                                        ('integrateLosses', 100, 14000.0), ('dynamic', 100, 1500.0)])
def test_gpu_restoration_in_the_other_kernel_families(family, N, T):
    """
    The restoration phase for every kernel family.  Adaptive ('CVODES') and collocation ('IRK') shooting: first-pass kernels + the streamed follow-up kernel
    of the family; 600 intervals: the five-wave kernel + the streamed one; 700 and 1200 intervals: the streamed kernels themselves (first pass + follow-up
    kernel of the same geometry).  Round 5: integrateLosses (ocp.py:231-241) and the dynamic loss model (efficiency.py; figure-5 configuration) -- first-pass
    kernels + the streamed follow-up kernel of their family, the loss rows' couplings with the next node as cross terms of the restoration problem's Newton
    system.  A loose schedule from the reference's starting point goes through restoration phases in the oracle and on the device and reaches the oracle's
    optimum (dynamic loss table: to 1e-4 -- the table's kinks leave the two solvers on iterates 1e-5 apart in cost).
    """
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from oracle import oracle
    train, track, opts, prob, kw = _family_case(family, N, 800, crop=None)
    s = casadiSolver(train, track, opts, startingPoint='reference')
    res = s.solveBatch([T], **kw)
    s.close()
    ref = oracle.solve(prob, prob.scenario(T, **kw), start='reference')
    assert res['status'][0] == int(ref['stats']['STATUS']) == 0
    assert int(ref['stats']['N_RESTO']) >= 1 and int(res['stats'][0, ST['N_RESTO']]) >= 1
    assert abs(res['cost'][0] - ref['stats']['OBJ']) <= (1e-4 if family == 'dynamic' else 1e-7)*abs(ref['stats']['OBJ'])
    if family != 'dynamic':
        assert np.max(np.abs(res['z'][0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-4
