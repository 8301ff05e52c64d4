"""
Transcription options beyond the default (f4 of SURVEY.md section 8): the other two shooting integrators and integrateLosses.

The other two shooting integrators of the NLP (reference: OptionsCasadiSolver.integrationMethod = 'IRK' / 'CVODES', ocp.py:26,92 ->
TrainIntegrator, train.py:303-322): collocation (casadi.simpleIRK) and integration to tolerances (CVODES' role, played by an adaptive
Dormand-Prince pair).  CPU part: the oracle's restatement against checkers that share nothing with it (the numpy collocation equations
solved by scipy, a DOP853 reference solution, finite differences of the oracle's own values) and the emulated kernel against the
oracle.  GPU part: the HIP kernels against the oracle through the C ABI.

integrateLosses (ocp.py:28,231-241): the loss slacks bound the loss power integrated over the running time of the interval; oracle against the
closed-form solution of the time-domain speed equation and finite differences, kernels against the oracle.
"""

import ctypes

import numpy as np
import pytest
from scipy.integrate import solve_ivp

import cases
from oracle import oracle
from oracle.oracle import DP
from nlp_numpy import kkt_certificate
from test_integrators import _irk_numpy

IRK2 = dict(integrationMethod='IRK', order=2, collMethod='radau', maxIter=10)
IRK3L = dict(integrationMethod='IRK', order=3, collMethod='legendre', maxIter=10)
ADAPT = dict(integrationMethod='CVODES', absTol=1e-8, relTol=1e-6)


def _problem(N=30, crop=12000, numSteps=1, numApprox=0, integration=None, train=None, **kw):
    return cases.oracle_problem(train or cases.train_default(), cases.track_00(crop), N, numSteps=numSteps, numApproxSteps=numApprox, integration=integration, **kw)


# ---- the oracle's integrators ----------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('order,scheme,numSteps,numApprox', [(2, 'radau', 1, 0), (3, 'legendre', 2, 0), (1, 'radau', 3, 0), (2, 'radau', 1, 2),
                                                             (4, 'radau', 2, 3), (9, 'radau', 1, 0), (5, 'legendre', 1, 1)])
def test_oracle_collocation_values_vs_numpy(order, scheme, numSteps, numApprox):
    from mseetc.train import collocationTables
    prob = _problem(numSteps=numSteps, numApprox=numApprox, integration=dict(integrationMethod='IRK', order=order, collMethod=scheme, maxIter=30))
    model = cases.train_default().exportModel()
    C, D = collocationTables(order, scheme)
    rng = np.random.default_rng(11)
    for _ in range(12):
        b0, w, ds, grad, curv = rng.uniform(100, 1500), rng.uniform(-0.3, 0.4), rng.uniform(10, 400), rng.uniform(-0.012, 0.012), rng.uniform(-1/320, 1/320)
        out = oracle.stage_eval(prob, b0, w, ds, grad, curv)
        G = model.resistance(grad, curv)
        if numApprox == 0:
            t, b = _irk_numpy(model, C, D, numSteps, 0.0, b0, ds, w, G, True)
        else:
            bs = [b0] + [_irk_numpy(model, C, D, numSteps, 0.0, b0, ds, w, G, False, h=j/numApprox)[1] for j in range(1, numApprox + 1)]
            t = sum(2*ds/numApprox/(np.sqrt(bs[j]) + np.sqrt(bs[j + 1])) for j in range(numApprox))
            b = bs[-1]
        assert abs(out[0] - t) <= 1e-11*abs(t) and abs(out[1] - b) <= 1e-11*abs(b)


@pytest.mark.parametrize('integration,numSteps,numApprox', [(IRK2, 1, 0), (IRK3L, 2, 2), (dict(IRK2, order=4), 3, 1), (ADAPT, 1, 0)])
def test_oracle_integrator_derivatives_by_finite_differences(integration, numSteps, numApprox):
    prob = _problem(numSteps=numSteps, numApprox=numApprox, integration=integration)
    if integration is ADAPT:
        # the derivatives are those of the accepted steps: differencing needs a fixed step sequence, which tolerances this loose
        # do not give; tight tolerances make the steps' influence on the value vanish instead
        prob.dp[DP['INT_ATOL']], prob.dp[DP['INT_RTOL']] = 1e-13, 1e-12
    ev = lambda b, w, ds: np.array(oracle.stage_eval(prob, b, w, ds, 0.003, 0.001))
    rng = np.random.default_rng(5)
    for _ in range(6):
        b, w, ds = rng.uniform(150, 1400), rng.uniform(-0.3, 0.4), rng.uniform(20, 350)
        h = 1e-5
        f0 = ev(b, w, ds)
        gb = (ev(b*(1 + h), w, ds) - ev(b*(1 - h), w, ds))/(2*h*b)
        gw = (ev(b, w + h, ds) - ev(b, w - h, ds))/(2*h)
        # layout: tau, b+, dtau/db, dtau/dw, db+/db, db+/dw, then the Hessians (bb, bw, ww) of tau and of b+
        fd = np.array([gb[0], gw[0], gb[1], gw[1], gb[2], gw[2], gw[3], gb[4], gw[4], gw[5]])
        # (central differences of step 1e-5 carry about 1e-6 of noise on the smallest second derivatives)
        scale = np.maximum(np.abs(fd), 1e-5*np.max(np.abs(fd)))
        assert np.max(np.abs(f0[2:] - fd)/scale) < 1e-5, (integration, f0[2:], fd)


def test_oracle_adaptive_integrator_vs_dop853():
    prob = _problem(integration=ADAPT)
    dp = prob.dp
    rng = np.random.default_rng(9)
    for _ in range(10):
        b0, w, ds, grad = rng.uniform(100, 1500), rng.uniform(-0.3, 0.4), rng.uniform(10, 400), rng.uniform(-0.012, 0.012)
        out = oracle.stage_eval(prob, b0, w, ds, grad, 0.0)
        G = dp[DP['G']]*grad/dp[DP['RHO']]
        rhs = lambda s, y: [1/np.sqrt(y[1]), 2*(w - (dp[DP['SR0']] + dp[DP['SR1']]*np.sqrt(y[1]) + dp[DP['SR2']]*y[1]) - G)]
        sol = solve_ivp(rhs, [0, ds], [0.0, b0], rtol=1e-13, atol=1e-13, method='DOP853')
        # relTol = 1e-6 per step: the global error stays well below 1e-5
        assert abs(out[0] - sol.y[0, -1]) <= 1e-5*abs(sol.y[0, -1]) and abs(out[1] - sol.y[1, -1]) <= 1e-5*abs(sol.y[1, -1])


def test_adaptive_pair_against_a_bdf_code_at_the_reference_tolerances():
    """
    Pins the difference the substitution behind 'CVODES' makes (reference: train.py:312-322 integrates an interval with SUNDIALS CVODES --
    variable-order BDF -- at absTol 1e-8 / relTol 1e-6; here an adaptive Dormand-Prince pair at the same tolerances plays that role).  A BDF code
    of the same family (scipy's variable-order BDF, Newton iteration, same tolerances) and the oracle's pair integrate 40 random intervals; both are
    compared with a DOP853 solution at 1e-13.  Measured: pair 5.2e-8, BDF code 1.4e-6 relative from the exact map, 1.4e-6 between the two -- the
    size of the difference a user switching from the reference sees per interval is the BDF code's own error; the asserts leave a margin.
    """
    prob = _problem(integration=ADAPT)
    dp = prob.dp
    rng = np.random.default_rng(31)
    worst_pair, worst_bdf, worst_between = 0.0, 0.0, 0.0
    for _ in range(40):
        b0, w, ds, grad = rng.uniform(100, 1500), rng.uniform(-0.3, 0.4), rng.uniform(10, 400), rng.uniform(-0.012, 0.012)
        out = oracle.stage_eval(prob, b0, w, ds, grad, 0.0)[:2]
        G = dp[DP['G']]*grad/dp[DP['RHO']]
        rhs = lambda s, y: [1/np.sqrt(y[1]), 2*(w - (dp[DP['SR0']] + dp[DP['SR1']]*np.sqrt(y[1]) + dp[DP['SR2']]*y[1]) - G)]
        exact = solve_ivp(rhs, [0, ds], [0.0, b0], rtol=1e-13, atol=1e-13, method='DOP853').y[:, -1]
        bdf = solve_ivp(rhs, [0, ds], [0.0, b0], rtol=ADAPT['relTol'], atol=ADAPT['absTol'], method='BDF').y[:, -1]
        worst_pair = max(worst_pair, float(np.max(np.abs(out - exact)/np.abs(exact))))
        worst_bdf = max(worst_bdf, float(np.max(np.abs(bdf - exact)/np.abs(exact))))
        worst_between = max(worst_between, float(np.max(np.abs(out - bdf)/np.abs(exact))))
    assert worst_pair <= 1e-6, worst_pair
    assert worst_bdf <= 1e-4, worst_bdf
    assert worst_between <= max(2*worst_bdf, 2e-6), (worst_between, worst_bdf)
    assert worst_pair <= worst_bdf, (worst_pair, worst_bdf)      # the stand-in is the more accurate of the two at these tolerances


def test_adaptive_transcription_optimum_is_insensitive_to_the_integrator_tolerance():
    """
    NLP-level side of the same pin: the optimum of the 'CVODES' transcription at the reference's default tolerances differs from the optimum at
    tolerances a thousand times tighter by 1.5e-8 relative in energy (bound here: 1e-6, two orders below north_star's 1e-4), so an integrator
    of the same accuracy class (the reference's BDF code, 30 times less accurate per interval above) stays below that bound too.
    """
    objs = []
    for tol in (dict(absTol=1e-8, relTol=1e-6), dict(absTol=1e-11, relTol=1e-9)):
        prob = _problem(integration=dict(ADAPT, **tol))
        r = oracle.solve(prob, prob.scenario(520.0, 0.0, 8.0, 8.0), start='profile')
        assert r['stats']['STATUS'] == 0
        objs.append(r['stats']['OBJ'])
    assert abs(objs[0] - objs[1]) <= 1e-6*abs(objs[1]), objs


def test_oracle_nlp_with_accurate_integrators_agree():
    """
    Three transcriptions whose integrators are all accurate on this grid (RK4 with 32 steps, 5-point Radau with 4 steps, the adaptive
    pair at tight tolerances; the train enters and leaves at 8 m/s, away from the 1/v singularity of the time equation) are the same NLP
    up to the integration error: their optima agree to 1e-7, a thousand times closer than the one-step RK4 transcription is to them,
    whichever integrator code produced them.
    """
    T, v = 520.0, 8.0
    settings = [dict(numSteps=32, numApprox=0), dict(numSteps=4, numApprox=0, integration=dict(IRK2, order=5)),
                dict(integration=dict(ADAPT, absTol=1e-11, relTol=1e-10)), dict(numSteps=1, numApprox=0)]
    objs = []
    for kw in settings:
        prob = _problem(**kw)
        r = oracle.solve(prob, prob.scenario(T, 0.0, v, v), start='profile')
        assert r['stats']['STATUS'] == 0
        objs.append(r['stats']['OBJ'])
    assert abs(objs[1] - objs[0]) < 1e-7*objs[0] and abs(objs[2] - objs[0]) < 1e-7*objs[0], objs
    assert abs(objs[3] - objs[0]) > 1e-3*objs[0], objs


# ---- the emulated kernel ----------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize('method,io,integration,extra', [('IRK', dict(order=2, numSteps=1, numApproxSteps=0), IRK2, {}),
                                                         ('IRK', dict(order=3, collMethod='legendre', numSteps=2, numApproxSteps=2), IRK3L, {}),
                                                         ('CVODES', dict(), ADAPT, {}),
                                                         ('IRK', dict(order=2, numSteps=1, numApproxSteps=1), IRK2, dict(integrateLosses=True)),      # both options at once
                                                         ('CVODES', dict(), ADAPT, dict(integrateLosses=True))])
def test_emulated_kernel_with_other_integrators_matches_oracle(method, io, integration, extra):
    from test_kernel_emulation import load_emulation
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    lib = load_emulation()
    N, T = 30, 520.0
    train, track = cases.train_default(), cases.track_00(12000)
    integration = dict(integration, **extra)
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=300, integrationMethod=method, integrationOptions=io, **extra), startingPoint='profile')
    scen = solver._scenarios(T, 0, 1, 1)
    nz = (4 + int(solver.withPnBrake))*N + 2
    z, lam, st, hist = np.zeros((1, nz)), np.zeros((1, 7*N)), np.zeros((1, ST['COUNT'])), np.zeros((8, 8))
    d = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert lib.emu_solve_batch(ctypes.byref(solver._desc), 1, d(scen), d(z), d(lam), d(st), d(hist), 8) == 0
    prob = _problem(N, 12000, io.get('numSteps', 1), io.get('numApproxSteps', 0), integration)
    ref = oracle.solve(prob, prob.scenario(T), start='profile')
    assert st[0, ST['STATUS']] == 0 and ref['stats']['STATUS'] == 0
    if not extra:
        assert int(st[0, ST['ITERS']]) == int(ref['stats']['ITERS'])
        assert np.max(np.abs(z[0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-8
    else:
        # with the integrated loss rows the kernel's Newton step is the oracle's up to the folding of the running time into (b, f, p): the two logs
        # agree to ten digits through the last barrier problem; after the final full step the dual infeasibility is 1e-8 in the kernel against
        # 2e-11 in the oracle (same with 'RK' shooting on this case), so the test against 1e-8 can fall one barrier update later
        assert 0 <= int(st[0, ST['ITERS']]) - int(ref['stats']['ITERS']) <= 2
        assert np.max(np.abs(z[0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-4      # (another final barrier parameter: 9e-10 against 2.5e-9)
        assert abs(st[0, 2] - ref['stats']['OBJ']) <= 3e-9*abs(ref['stats']['OBJ'])


# ---- the HIP kernels ----------------------------------------------------------------------------------------------------------------

CASES = [('IRK', dict(order=2, numSteps=1, numApproxSteps=0), IRK2, 30, 12000, 520.0),
         ('IRK', dict(order=2, numSteps=1, numApproxSteps=1), IRK2, 100, None, 1541.0),
         ('IRK', dict(order=3, collMethod='legendre', numSteps=2, numApproxSteps=2), IRK3L, 60, 30000, 1100.0),
         ('IRK', dict(order=9, numSteps=1, numApproxSteps=1), dict(IRK2, order=9), 40, 12000, 520.0),
         ('CVODES', dict(), ADAPT, 100, None, 1541.0),
         ('CVODES', dict(absTol=1e-10, relTol=1e-9), dict(ADAPT, absTol=1e-10, relTol=1e-9), 150, None, 1600.0)]


@pytest.mark.gpu
@pytest.mark.parametrize('method,io,integration,N,crop,T', CASES)
def test_gpu_other_integrators_vs_oracle(method, io, integration, N, crop, T):
    from mseetc.ocp import casadiSolver
    train, track = cases.train_default(), cases.track_00(crop) if crop else cases.track_00()
    for start in ('profile', 'reference'):
        solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=400, integrationMethod=method, integrationOptions=io), startingPoint=start)
        Ts = T*np.array([1.0, 1.04, 1.11])
        res = solver.solveBatch(Ts)
        prob = cases.oracle_problem(train, track, N, numSteps=io.get('numSteps', 1), numApproxSteps=io.get('numApproxSteps', 0), integration=integration)
        for k, Tk in enumerate(Ts):
            ref = oracle.solve(prob, prob.scenario(Tk), start=start)
            assert res['status'][k] == 0 and ref['stats']['STATUS'] == 0
            assert abs(res['cost'][k] - ref['stats']['OBJ']) <= 1e-8*abs(ref['stats']['OBJ'])
            assert np.max(np.abs(res['z'][k] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-5
            assert abs(int(res['iterations'][k]) - int(ref['stats']['ITERS'])) <= 6


@pytest.mark.gpu
def test_gpu_adaptive_integrator_shared_evaluation_equals_the_sequential_one(monkeypatch):
    """
    The 'CVODES' kernels with one node per lane evaluate the jets of a long interval -- the first and the last of a journey from and to
    standstill: 25-28 accepted steps -- with the lanes of the wave together (msd_kernel.hpp: Solver::coop_adaptive: values first, the local jet
    of step k on lane k, chain-rule composition by the owner); the two-nodes-per-lane geometry (a tuning switch of the pickers) runs every interval
    the sequential way (msd_integ.hpp: dopri_tb_jet).  Same discrete map, same derivatives: the two solves take the same iterations and agree to
    rounding.  v0 = vN = 1 m/s (both ends long) and a journey that starts at speed (one long interval, in the second wave).
    """
    from mseetc.ocp import casadiSolver
    train, track = cases.train_default(), cases.track_00()
    Ts = 1541.0*np.array([1.0, 1.03, 1.08, 1.15])
    for v0 in (1.0, 20.0):
        out = {}
        for geometry in ('default', '64x2'):
            from mseetc._device import lib
            assert lib().msd_tuning(b'two_nodes_per_lane', int(geometry == '64x2')) == 0
            solver = casadiSolver(train, track, dict(numIntervals=100, maxIterations=400, integrationMethod='CVODES'), startingPoint='profile')
            scen = solver._scenarios(Ts, 0, 1, v0)
            out[geometry] = solver.problem.solve_batch(scen)
            assert tuple(solver.problem.geometry()) == ((128, 1) if geometry == 'default' else (64, 2))
            solver.close()
            lib().msd_tuning(b'two_nodes_per_lane', 0)
        a, b = out['default'], out['64x2']
        assert np.all(a['stats'][:, 0] == 0) and np.all(b['stats'][:, 0] == 0)
        assert np.array_equal(a['stats'][:, 1], b['stats'][:, 1])                                     # iterations
        assert np.max(np.abs(a['stats'][:, 2] - b['stats'][:, 2])/np.abs(b['stats'][:, 2])) < 1e-11      # objective
        assert np.max(np.abs(a['z'] - b['z'])/np.maximum(1, np.abs(b['z']))) < 1e-8


def _dynamic_train():
    "figure5.py's train with the dynamic loss model of efficiency.py (no pneumatic brake)"
    from mseetc.train import Train
    from mseetc.efficiency import totalLossesFunction
    train = Train(config={'id': 'NL_Intercity_VIRM6'})
    train.forceMinPn = 0
    train.powerLosses = totalLossesFunction(train, auxiliaries=27000, etaGear=0.96)
    return train


@pytest.mark.gpu
@pytest.mark.parametrize('method,io,integration', [('IRK', dict(order=2, numSteps=1, numApproxSteps=1), IRK2), ('CVODES', dict(), ADAPT)])
def test_gpu_other_integrators_with_dynamic_losses_vs_oracle(method, io, integration):
    """
    The reference builds any integrationMethod with any loss model (ocp.py:92 next to efficiency.py:101-141): collocation and
    tolerance-controlled shooting together with the dynamic loss table, figure5.py's problem (8.5 km, v0 = 1, vN = 100 km/h).
    """
    from mseetc.ocp import casadiSolver
    from mseetc.track import computeDiscretizationPoints
    from mseetc.train import collocationTables
    train, track, N = _dynamic_train(), cases.track_00(8500), 60
    oracle.set_loss_table(train.powerLosses.parameters(train.mass*train.rho))
    if method == 'IRK':
        oracle.set_collocation(*collocationTables(integration['order'], integration['collMethod']))
    pts = computeDiscretizationPoints(track, N)
    opts = dict(numIntervals=N, maxIterations=500, energyOptimal=True, minimumVelocity=1, numSteps=io.get('numSteps', 1), numApproxSteps=io.get('numApproxSteps', 0))
    opts.update(integration)
    prob = oracle.pack_problem(train, pts, opts, 2, 0.0, 0.0, track.length)
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationMethod=method, integrationOptions=io), startingPoint='profile')
    T = [272.4726*r for r in (1.1, 1.2, 1.3)]
    res = solver.solveBatch(T, terminalVelocity=100/3.6, initialVelocity=1)
    assert np.all(res['status'] == 0), res['status']
    for k, t in enumerate(T):
        ref = oracle.solve(prob, prob.scenario(t, terminalVelocity=100/3.6, initialVelocity=1), start='profile')
        assert ref['stats']['STATUS'] == 0
        assert abs(res['cost'][k] - ref['stats']['OBJ']) <= 1e-8*abs(ref['stats']['OBJ'])
        assert np.max(np.abs(res['z'][k] - ref['z'])/np.maximum(1.0, np.abs(ref['z']))) <= 1e-5
        assert abs(int(res['iterations'][k]) - int(ref['stats']['ITERS'])) <= 6


@pytest.mark.gpu
@pytest.mark.parametrize('method,io,integration,combo', [('IRK', dict(order=2, numSteps=1, numApproxSteps=1), IRK2, 'dynamic'), ('CVODES', dict(), ADAPT, 'dynamic'),
                                                          ('IRK', dict(order=2, numSteps=1, numApproxSteps=1), IRK2, 'integrate_losses')])
def test_gpu_combined_options_beyond_255_intervals_on_the_streamed_kernel(method, io, integration, combo):
    """
    Round 4: the collocation / adaptive shooting integrators together with the dynamic loss model or with integrateLosses no longer stop at the 255
    intervals of their LDS-resident kernels (msd_kernels_compose.hip): N = 300 runs on the streamed kernels of msd_kernels_stream4.hip (the
    reference builds any option set at any N, simulations/table3.py:34).  Against the oracle, figure5.py's problem resp. the figure-10 train.
    """
    from mseetc.ocp import casadiSolver
    from mseetc.track import computeDiscretizationPoints
    from mseetc.train import collocationTables
    N = 300
    if method == 'IRK':
        oracle.set_collocation(*collocationTables(integration['order'], integration['collMethod']))
    if combo == 'dynamic':
        train, track = _dynamic_train(), cases.track_00(8500)
        oracle.set_loss_table(train.powerLosses.parameters(train.mass*train.rho))
        pts = computeDiscretizationPoints(track, N)
        opts = dict(numIntervals=N, maxIterations=500, energyOptimal=True, minimumVelocity=1, numSteps=io.get('numSteps', 1), numApproxSteps=io.get('numApproxSteps', 0))
        opts.update(integration)
        prob = oracle.pack_problem(train, pts, opts, 2, 0.0, 0.0, track.length)
        solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationMethod=method, integrationOptions=io), startingPoint='profile')
        T, kw = [272.4726*1.15, 272.4726*1.25], dict(terminalVelocity=100/3.6, initialVelocity=1)
    else:
        train, track = cases.train_fig10(), cases.track_00()
        both = dict(integration); both['integrateLosses'] = True
        prob = cases.oracle_problem(train, track, N, numSteps=io.get('numSteps', 1), numApproxSteps=io.get('numApproxSteps', 0), integration=both)
        solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationMethod=method, integrationOptions=io, integrateLosses=True), startingPoint='profile')
        T, kw = [1600.0], {}
    assert solver.problem.geometry() == (512, 2)
    res = solver.solveBatch(T, **kw)
    assert np.all(res['status'] == 0), res['status']
    for k, t in enumerate(T):
        ref = oracle.solve(prob, prob.scenario(t, **kw), start='profile')
        assert ref['stats']['STATUS'] == 0
        assert abs(res['cost'][k] - ref['stats']['OBJ']) <= 1e-8*abs(ref['stats']['OBJ'])
        assert np.max(np.abs(res['z'][k] - ref['z'])/np.maximum(1.0, np.abs(ref['z']))) <= 1e-5
        assert abs(int(res['iterations'][k]) - int(ref['stats']['ITERS'])) <= 6
    solver.close()


@pytest.mark.gpu
@pytest.mark.parametrize('what', ['irk', 'cvodes', 'integrate_losses', 'dynamic'])
def test_gpu_other_transcriptions_on_the_streamed_kernel(what):
    """
    Beyond 560 intervals the stage blocks leave LDS (simulations/table3.py:34 sweeps numIntervals up to 5000 with 'RK'; the reference
    builds the other transcriptions at any N too): the streamed kernels of the other shooting integrators, of integrateLosses and of
    the dynamic loss model (up to 1023 intervals) against the oracle at N = 700.
    """
    from mseetc.ocp import casadiSolver
    from mseetc.track import computeDiscretizationPoints
    N = 700
    if what == 'dynamic':
        train, track = _dynamic_train(), cases.track_00(8500)
        oracle.set_loss_table(train.powerLosses.parameters(train.mass*train.rho))
        pts = computeDiscretizationPoints(track, N)
        prob = oracle.pack_problem(train, pts, dict(numIntervals=N, maxIterations=1000, energyOptimal=True, minimumVelocity=1, numSteps=1, numApproxSteps=1), 2, 0.0, 0.0, track.length)
        solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=1000, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='profile')
        T, kw = [272.4726*1.2], dict(terminalVelocity=100/3.6, initialVelocity=1)
    else:
        train, track = cases.train_fig10(), cases.track_00()
        extra, io, integration = {'irk': (dict(integrationMethod='IRK'), dict(order=2, numSteps=1, numApproxSteps=1), IRK2),
                                  'cvodes': (dict(integrationMethod='CVODES'), dict(), ADAPT),
                                  'integrate_losses': (dict(integrateLosses=True), dict(numSteps=1, numApproxSteps=1), dict(integrateLosses=True))}[what]
        prob = cases.oracle_problem(train, track, N, numSteps=io.get('numSteps', 1), numApproxSteps=io.get('numApproxSteps', 0), maxIterations=1000, integration=integration)
        solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=1000, integrationOptions=io, **extra), startingPoint='profile')
        T, kw = [1600.0], {}
    assert solver.problem.geometry() == (512, 2)
    res = solver.solveBatch(T, **kw)
    assert np.all(res['status'] == 0), res['status']
    ref = oracle.solve(prob, prob.scenario(T[0], **kw), start='profile')
    assert ref['stats']['STATUS'] == 0
    assert abs(res['cost'][0] - ref['stats']['OBJ']) <= 1e-8*abs(ref['stats']['OBJ'])
    assert np.max(np.abs(res['z'][0] - ref['z'])/np.maximum(1.0, np.abs(ref['z']))) <= 1e-5
    assert abs(int(res['iterations'][0]) - int(ref['stats']['ITERS'])) <= 6
    solver.close()


@pytest.mark.gpu
@pytest.mark.parametrize('method,io,integration,N,crop,T', [('IRK', dict(order=2, numSteps=1, numApproxSteps=1), IRK2, 100, None, (1541.0, 1650.0)),
                                                            ('IRK', dict(order=3, collMethod='legendre', numSteps=2, numApproxSteps=0), IRK3L, 40, 12000, (640.0,)),
                                                            ('CVODES', dict(), ADAPT, 100, None, (1541.0, 1700.0)),
                                                            ('CVODES', dict(), ADAPT, 200, None, (1580.0,))])
def test_gpu_integrated_losses_with_other_shooting_integrators_vs_oracle(method, io, integration, N, crop, T):
    """
    ocp.py:92 with ocp.py:231-241: the loss rows integrate the loss power with an integrator of their own (train.py:367-413), whatever integrates the
    shooting intervals -- kernels with both options (msd_kernels_compose.hip) against the oracle from both starting points.
    """
    from mseetc.ocp import casadiSolver
    train, track = cases.train_default(), cases.track_00(crop)
    prob = cases.oracle_problem(train, track, N, numSteps=io.get('numSteps', 1), numApproxSteps=io.get('numApproxSteps', 0), integration=dict(integration, integrateLosses=True))
    for start in ('profile', 'reference'):
        solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrateLosses=True, integrationMethod=method, integrationOptions=io), startingPoint=start)
        res = solver.solveBatch(list(T))
        solver.close()
        assert np.all(res['status'] == 0), (method, start, res['status'])
        for k, t in enumerate(T):
            ref = oracle.solve(prob, prob.scenario(float(t)), start=start)
            assert ref['stats']['STATUS'] == 0
            assert abs(res['cost'][k] - ref['stats']['OBJ']) <= 1e-7*abs(ref['stats']['OBJ']), (method, start, k)
            assert np.max(np.abs(res['z'][k] - ref['z'])/np.maximum(1.0, np.abs(ref['z']))) <= 1e-4, (method, start, k)
            assert abs(int(res['iterations'][k]) - int(ref['stats']['ITERS'])) <= 6


@pytest.mark.gpu
def test_gpu_stage_function_of_other_integrators_vs_oracle_and_standalone_kernels():
    """
    msd_stage_eval of an 'IRK' / 'CVODES' problem: values against the stand-alone interval integrators (msd_interval_integrate,
    csrc/msd_integrators.hip -- other code), values and both derivative orders against the oracle.
    """
    from mseetc.ocp import casadiSolver
    from mseetc.train import TrainIntegrator
    train, track = cases.train_default(), cases.track_00()
    model = train.exportModel()
    rng = np.random.default_rng(21)
    n = 200
    b, w, ds = rng.uniform(100, 1500, n), rng.uniform(-0.3, 0.4, n), rng.uniform(10, 400, n)
    grad, curv = rng.uniform(-0.012, 0.012, n), rng.uniform(-1/320, 1/320, n)
    w = np.maximum(w, (16.0 - b)/(2*ds) + 0.25)      # stay clear of standstill within the interval (resistances + gradient < 0.25 N/kg)
    for method, io, integration in [('IRK', dict(order=3, numSteps=2, numApproxSteps=0), dict(IRK2, order=3)), ('IRK', dict(order=2, numSteps=1, numApproxSteps=2), IRK2),
                                    ('CVODES', dict(), ADAPT)]:
        solver = casadiSolver(train, track, dict(numIntervals=50, integrationMethod=method, integrationOptions=io))
        out = solver.problem.stage_eval(b, w, ds, grad, curv)
        alone = TrainIntegrator(model, method, io).solveMany(np.zeros(n), b, ds, w, grad, curv)
        assert np.max(np.abs(out[:, 0] - alone['time'])/np.abs(alone['time'])) < 1e-9
        assert np.max(np.abs(out[:, 1] - alone['velSquared'])/np.abs(alone['velSquared'])) < 1e-9
        prob = cases.oracle_problem(train, track, 50, numSteps=io.get('numSteps', 1), numApproxSteps=io.get('numApproxSteps', 0), integration=integration)
        for k in range(0, n, 4):
            ref = np.array(oracle.stage_eval(prob, b[k], w[k], ds[k], grad[k], curv[k]))
            assert np.max(np.abs(out[k] - ref)/np.maximum(np.abs(ref), 1e-9*np.max(np.abs(ref)))) < 1e-8, (method, k)


@pytest.mark.gpu
def test_gpu_other_integrators_surface_and_limits():
    from mseetc.ocp import casadiSolver
    from mseetc._device import DeviceError
    train, track = cases.train_default(), cases.track_00()
    solver = casadiSolver(train, track, dict(numIntervals=100, integrationMethod='IRK', integrationOptions=dict(order=2, numApproxSteps=1)))
    df, stats = solver.solve(1541)
    assert stats['IP iterations'] > 5 and abs(df.index[-1] - 1541) < 1e-4      # (the bound of t_N is relaxed by 1e-8 relative, like IPOPT does)
    rk = casadiSolver(train, track, dict(numIntervals=100, integrationOptions=dict(numApproxSteps=1)))
    assert abs(stats['Cost'] - rk.solve(1541)[1]['Cost']) < 1e-3*stats['Cost']      # both integrate b accurately on this grid
    with pytest.raises(DeviceError):
        casadiSolver(train, track, dict(numIntervals=1100, integrationMethod='CVODES')).solve(1541)      # beyond the streamed kernels of the other integrators
    with pytest.raises(DeviceError):
        casadiSolver(train, track, dict(numIntervals=1100, integrateLosses=True, integrationMethod='IRK')).solve(1541)      # that combination: up to 1023 intervals too (round 4; 255 before)
    with pytest.raises(NotImplementedError):
        from mseetc.efficiency import totalLossesFunction
        dyn = cases.train_default(); dyn.forceMinPn = 0; dyn.powerLosses = totalLossesFunction(dyn)
        casadiSolver(dyn, track, dict(numIntervals=50, integrateLosses=True, integrationMethod='IRK'))      # the loss table under the integral (round 6: tests/test_integrated_loss_table.py) runs with 'RK' shooting
    with pytest.raises(DeviceError):
        casadiSolver(train, track, dict(numIntervals=1100, integrateLosses=True)).solve(1541)
    # integrateLosses through the reference's surface: same optimum as the mid-point rows to about 1e-4 (X = ds up to the RK4 error)
    il = casadiSolver(train, track, dict(numIntervals=100, integrateLosses=True, integrationOptions=dict(numApproxSteps=1)))
    dfi, sti = il.solve(1541)
    assert abs(sti['Cost'] - rk.solve(1541)[1]['Cost']) < 3e-4*sti['Cost']
    assert il.solve(1541, initialVelocity=5, terminalVelocity=3)[1]['Solver status'] == 'Solve_Succeeded'


# ---- integrateLosses ------------------------------------------------------------------------------------------------------------------

def _distance_closed_form(a, sr1, sr2, v0, dt):
    """
    X(dt) for dv/dt = a - sr1 v - sr2 v^2 (a = w - sr0 - G), v(0) = v0: the Riccati equation has the roots v+- of sr2 v^2 + sr1 v - a = 0;
    with k = sr2 (v+ - v-) and C = (v0 - v+)/(v0 - v-):  X = v+ dt + ln((1 - C exp(-k dt))/(1 - C))/sr2.  Complex arithmetic covers the
    oscillatory case (a below -sr1^2/(4 sr2): strong braking), where the roots are a conjugate pair and X stays real.
    """
    disc = np.sqrt(complex(sr1*sr1 + 4*sr2*a))
    vp, vm = (-sr1 + disc)/(2*sr2), (-sr1 - disc)/(2*sr2)
    k, C = sr2*(vp - vm), (v0 - vp)/(v0 - vm)
    X = vp*dt + np.log((1 - C*np.exp(-k*dt))/(1 - C))/sr2
    assert abs(X.imag) < 1e-9*max(1.0, abs(X.real))
    return X.real


def _loss_distance(prob, v0, dt, w, grad=0.0, curv=0.0):
    L = oracle.lib()
    dptr, iptr = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
    L.oracle_loss_distance.restype = None
    L.oracle_loss_distance.argtypes = [iptr, dptr] + [ctypes.c_double]*5 + [dptr]
    out = np.zeros(10)
    L.oracle_loss_distance(oracle._i(prob.ip), oracle._d(prob.dp), v0, dt, w, grad, curv, oracle._d(out))
    return out


def test_oracle_loss_distance_vs_closed_form_and_finite_differences():
    prob = _problem(numApprox=1, integration=dict(integrateLosses=True))
    dp = prob.dp
    rng = np.random.default_rng(13)
    for _ in range(20):
        v0, dt, w, grad = rng.uniform(2, 40), rng.uniform(1, 30), rng.uniform(-0.6, 0.5), rng.uniform(-0.012, 0.012)
        if v0 + dt*(w - 0.15) < 1.0:
            continue       # the train would stop within the interval
        out = _loss_distance(prob, v0, dt, w, grad)
        a = w - dp[DP['SR0']] - dp[DP['G']]*grad/dp[DP['RHO']]
        X = _distance_closed_form(a, dp[DP['SR1']], dp[DP['SR2']], v0, dt)
        assert abs(out[0] - X) <= 2e-6*abs(X)                    # reltol 1e-6 per step (train.py:396)
        h = 1e-4
        g = [(_loss_distance(prob, v0 + h, dt, w, grad) - _loss_distance(prob, v0 - h, dt, w, grad))/(2*h),
             (_loss_distance(prob, v0, dt + h, w, grad) - _loss_distance(prob, v0, dt - h, w, grad))/(2*h),
             (_loss_distance(prob, v0, dt, w + h, grad) - _loss_distance(prob, v0, dt, w - h, grad))/(2*h)]
        fd = np.array([g[0][0], g[1][0], g[2][0], g[0][1], g[0][2], g[0][3], g[1][2], g[1][3], g[2][3]])
        # the differences see the step-size controller's decisions (1e-6 relative on X): loose bound on the first, looser on the second
        # derivatives; the closed form pins the first derivatives independently below
        scale = np.maximum(np.abs(fd), 1e-3*np.max(np.abs(fd[:3])))
        assert np.max(np.abs(out[1:4] - fd[:3])/scale[:3]) < 1e-3
        assert np.max(np.abs(out[4:] - fd[3:])/np.maximum(np.abs(fd[3:]), 1e-2*np.max(np.abs(fd[3:])))) < 5e-2
        hh = 1e-5
        gX = [(_distance_closed_form(a, dp[DP['SR1']], dp[DP['SR2']], v0 + hh, dt) - _distance_closed_form(a, dp[DP['SR1']], dp[DP['SR2']], v0 - hh, dt))/(2*hh),
              (_distance_closed_form(a, dp[DP['SR1']], dp[DP['SR2']], v0, dt + hh) - _distance_closed_form(a, dp[DP['SR1']], dp[DP['SR2']], v0, dt - hh))/(2*hh),
              (_distance_closed_form(a + hh, dp[DP['SR1']], dp[DP['SR2']], v0, dt) - _distance_closed_form(a - hh, dp[DP['SR1']], dp[DP['SR2']], v0, dt))/(2*hh)]
        assert np.max(np.abs(out[1:4] - np.array(gX))/np.abs(gX)) < 2e-5


def test_oracle_nlp_with_integrated_losses():
    """
    At a solution the running time, the speeds and the forces of an interval are consistent, so the distance X covered in the running
    time is the interval length up to the integration error of the one-step RK4 map: the optimum is that of the mid-point transcription
    to about 1e-4, and the loss rows hold with X from the closed form.  Both starts reach it.
    """
    train, track = cases.train_default(), cases.track_00()
    mid = cases.oracle_problem(train, track, 100)
    il = cases.oracle_problem(train, track, 100, integration=dict(integrateLosses=True))
    r0 = oracle.solve(mid, mid.scenario(1541.0), start='profile')
    objs = []
    for start in ('profile', 'reference'):
        r = oracle.solve(il, il.scenario(1541.0), start=start)
        assert r['stats']['STATUS'] == 0
        objs.append(r['stats']['OBJ'])
    assert abs(objs[0] - objs[1]) < 1e-7*objs[0]
    assert abs(objs[0] - r0['stats']['OBJ']) < 3e-4*objs[0]
    # rows at the solution, X from the closed form
    z = r['z']; dp = il.dp; N = 100; stp = 5
    ct, cr = dp[DP['LOSS_CT']], dp[DP['LOSS_CR']]
    worst = 0.0
    for i in range(N):
        f, p, s, t, b = z[stp*i:stp*i + 5]
        t1 = z[stp*(i + 1) + 3] if i + 1 < N else z[stp*N]
        G = dp[DP['G']]*il.grad[i]/dp[DP['RHO']]
        X = _distance_closed_form(f + p - dp[DP['SR0']] - G, dp[DP['SR1']], dp[DP['SR2']], np.sqrt(b), t1 - t)
        worst = min(worst, (s - ct*f*X)/max(1.0, abs(s)), (s + cr*f*X)/max(1.0, abs(s)))
        assert abs(X - il.ds[i]) < 2e-2*il.ds[i]
    assert worst > -1e-5


@pytest.mark.parametrize('N,crop,T,start', [(30, 12000, 520.0, 'reference'), (70, 30000, 1100.0, 'profile')])
def test_emulated_kernel_with_integrated_losses_matches_oracle(N, crop, T, start):
    from test_kernel_emulation import load_emulation
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    lib = load_emulation()
    train, track = cases.train_default(), cases.track_00(crop)
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=300, integrateLosses=True, integrationOptions=dict(numSteps=1, numApproxSteps=1)),
                          startingPoint=start)
    scen = solver._scenarios(T, 0, 1, 1)
    nz = (4 + int(solver.withPnBrake))*N + 2
    z, lam, st, hist = np.zeros((1, nz)), np.zeros((1, 7*N)), np.zeros((1, ST['COUNT'])), np.zeros((8, 8))
    d = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert lib.emu_solve_batch(ctypes.byref(solver._desc), 1, d(scen), d(z), d(lam), d(st), d(hist), 8) == 0
    prob = cases.oracle_problem(train, track, N, integration=dict(integrateLosses=True))
    ref = oracle.solve(prob, prob.scenario(T), start=start)
    assert st[0, ST['STATUS']] == 0 and ref['stats']['STATUS'] == 0
    assert int(st[0, ST['ITERS']]) == int(ref['stats']['ITERS'])
    assert np.max(np.abs(z[0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-7
    # (the folded stage blocks of the kernel against the oracle's unfolded system: the multipliers of the time equation agree)
    assert np.max(np.abs(lam[0] - ref['lam_g'])/np.maximum(1, np.abs(ref['lam_g']))) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('N,crop,T,track_name', [(100, None, 1541.0, '00'), (60, 30000, 1100.0, '00'), (200, None, 1900.0, 'CH'), (30, 12000, 520.0, '00')])
def test_gpu_integrated_losses_vs_oracle(N, crop, T, track_name):
    from mseetc.ocp import casadiSolver
    train = cases.train_default()
    track = cases.track_CH() if track_name == 'CH' else (cases.track_00(crop) if crop else cases.track_00())
    for start in ('profile', 'reference'):
        solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=400, integrateLosses=True, integrationOptions=dict(numApproxSteps=1)), startingPoint=start)
        Ts = T*np.array([1.0, 1.05, 1.12])
        res = solver.solveBatch(Ts, multipliers=True)
        prob = cases.oracle_problem(train, track, N, integration=dict(integrateLosses=True))
        for k, Tk in enumerate(Ts):
            ref = oracle.solve(prob, prob.scenario(Tk), start=start)
            assert res['status'][k] == 0 and ref['stats']['STATUS'] == 0
            assert abs(res['cost'][k] - ref['stats']['OBJ']) <= 1e-8*abs(ref['stats']['OBJ'])
            assert np.max(np.abs(res['z'][k] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-5
            assert abs(int(res['iterations'][k]) - int(ref['stats']['ITERS'])) <= 6
            # multipliers in the reference's row order: the time-equation multipliers come out of the folded stage blocks (msd_kernel.hpp: direction)
            # (multipliers of a solve converged to 1e-8 are less sharp than its primal point when the iterate paths differ in the last bits;
            #  on one path -- the emulation test above -- they agree to 1e-6)
            assert np.max(np.abs(res['lam_g'][k] - ref['lam_g'])/np.maximum(1, np.abs(ref['lam_g']))) < 5e-3


def _certify(nlp, z, lam_g, dp, loss_rtol):
    """
    First-order conditions of the integrateLosses NLP checked with the numpy restatement (tests/nlp_numpy.py), whose loss rows use the
    CLOSED-FORM distance: the rows of the solver (1e-6 integration tolerance, train.py:396) and of the checker differ by that much.
    """
    cert = kkt_certificate(nlp, z, lam_g, dp[DP['T0']], dp[DP['TEND']], dp[DP['V0SQ']], dp[DP['VNSQ']])
    assert cert['feas_g'] < loss_rtol and cert['feas_z'] < 1.5e-8
    assert cert['stat'] < 2e-5 and cert['sign_g'] < 2e-5
    return cert


@pytest.mark.parametrize('track_name,N,T', [('00', 100, 1600.0), ('CH', 200, 1300.0)])
def test_kkt_certificate_of_oracle_solution_with_integrated_losses(track_name, N, T):
    train = cases.train_default()
    track = cases.track_CH() if track_name == 'CH' else cases.track_00()
    prob = cases.oracle_problem(train, track, N, integration=dict(integrateLosses=True))
    dp = prob.scenario(T)
    out = oracle.solve(prob, dp, start='profile')
    assert out['stats']['STATUS'] == 0
    nlp = cases.numpy_nlp(prob)
    assert nlp.integrateLosses
    z = out['z']
    obj, g = oracle.nlp_eval(prob, dp, z)
    assert abs(obj - nlp.obj(z)) <= 1e-12*abs(obj)
    assert np.max(np.abs(g - nlp.cons(z))/np.maximum(1.0, np.abs(g))) < 5e-6       # integrated against closed-form loss rows
    _certify(nlp, z, out['lam_g'], dp, 2e-5)


@pytest.mark.gpu
def test_kkt_certificate_of_gpu_solution_with_integrated_losses():
    from mseetc.ocp import casadiSolver
    train, track = cases.train_default(), cases.track_00()
    solver = casadiSolver(train, track, dict(numIntervals=100, integrateLosses=True, integrationOptions=dict(numApproxSteps=1)))
    res = solver.solveBatch([1541.0, 1700.0], multipliers=True)
    prob = cases.oracle_problem(train, track, 100, integration=dict(integrateLosses=True))
    nlp = cases.numpy_nlp(prob)
    for k, T in enumerate((1541.0, 1700.0)):
        assert res['status'][k] == 0
        _certify(nlp, res['z'][k], res['lam_g'][k], prob.scenario(T), 2e-5)


# ---- random problems with the other transcriptions ---------------------------------------------------------------------------------

TRANSCRIPTIONS = {
    'intloss': (dict(integrateLosses=True), dict(numSteps=1, numApproxSteps=1), dict(integrateLosses=True)),
    'irk': (dict(integrationMethod='IRK'), dict(order=2, numSteps=1, numApproxSteps=1), IRK2),
    'irk3x2': (dict(integrationMethod='IRK'), dict(order=3, collMethod='legendre', numSteps=2, numApproxSteps=2), IRK3L),
    'cvodes': (dict(integrationMethod='CVODES'), dict(), ADAPT),
}


@pytest.mark.gpu
@pytest.mark.parametrize('seed', range(8))
@pytest.mark.parametrize('which', sorted(TRANSCRIPTIONS))
def test_gpu_randomized_problems_with_other_transcriptions(which, seed, tmp_path):
    """
    The random tracks / trains / horizons of test_gpu_parity.test_randomized_problems_vs_oracle (up to 255 intervals: one or two waves per
    scenario) under integrateLosses, the collocation and the adaptive shooting integrator: running times 10 ... 40 % above the minimum
    the time-optimal twin finds, energy and trajectory against the oracle.
    """
    from test_gpu_parity import _random_problem
    from mseetc.ocp import casadiSolver
    train, track, N, rng = _random_problem(seed, tmp_path)
    v0, vN = float(rng.uniform(2, 15)), float(rng.uniform(2, 15))
    extra, io, integration = TRANSCRIPTIONS[which]
    fast = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, energyOptimal=False, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='profile')
    rt = fast.solveBatch([3*track.length/train.velocityMax], initialVelocity=v0, terminalVelocity=vN)
    assert rt['status'][0] == 0
    T = float(rt['z'][0][-2])*np.array([1.1, 1.4])
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationOptions=io, **extra), startingPoint='profile')
    res = solver.solveBatch(T, initialVelocity=v0, terminalVelocity=vN)
    assert np.all(res['status'] == 0), (which, seed, N, res['status'])
    prob = cases.oracle_problem(train, track, N, numSteps=io.get('numSteps', 1), numApproxSteps=io.get('numApproxSteps', 0), integration=integration)
    for k in range(2):
        ref = oracle.solve(prob, prob.scenario(float(T[k]), 0.0, vN, v0), start='profile')
        assert ref['stats']['STATUS'] == 0
        assert abs(res['cost'][k] - ref['stats']['OBJ']) <= 1e-7*abs(ref['stats']['OBJ']), (which, seed, N, k)
        assert np.max(np.abs(res['z'][k] - ref['z'])/np.maximum(1.0, np.abs(ref['z']))) <= 1e-4, (which, seed, N, k)
    solver.close(); fast.close()


def test_integrated_losses_converge_to_the_gpops_energy():
    """
    gpops/00_var_speed_limit_100_GPOPS{I,II}.csv hold 440.1415 / 440.1406 kWh for the figure10.py configuration -- the continuous problem,
    which the mid-point loss rows and the integrated ones both discretise: the integrateLosses transcription must extrapolate to the same
    number (tests/test_oracle_pins.py does this for the mid-point rows).
    """
    import pandas as pd
    from pathlib import Path
    gold = Path(__file__).resolve().parent / 'golden'
    e = {}
    for N in (100, 300):
        prob = cases.oracle_problem(cases.train_fig10(), cases.track_00(), N, integration=dict(integrateLosses=True))
        r = oracle.solve(prob, prob.scenario(1541.0), start='profile')
        assert r['stats']['STATUS'] == 0
        e[N] = r['stats']['OBJ']
    richardson = (9*e[300] - e[100])/8
    g1 = pd.read_csv(gold / '00_var_speed_limit_100_GPOPSI.csv')['Energy [kWh]'].iloc[0]
    g2 = pd.read_csv(gold / '00_var_speed_limit_100_GPOPSII.csv')['Energy [kWh]'].iloc[0]
    assert abs(richardson - g1) < 0.02 and abs(richardson - g2) < 0.02, (e, richardson)


@pytest.mark.parametrize('integration,numApprox', [(IRK2, 1), (ADAPT, 0)])
def test_other_shooting_integrators_converge_to_the_gpops_energy(integration, numApprox):
    "The same pin for the 'IRK' and 'CVODES' transcriptions: N = 100 / 300 on the figure10.py configuration extrapolate to GPOPS-II's 440.14 kWh."
    import pandas as pd
    from pathlib import Path
    gold = Path(__file__).resolve().parent / 'golden'
    e = {}
    for N in (100, 300):
        prob = cases.oracle_problem(cases.train_fig10(), cases.track_00(), N, numApproxSteps=numApprox, integration=integration)
        r = oracle.solve(prob, prob.scenario(1541.0), start='profile')
        assert r['stats']['STATUS'] == 0
        e[N] = r['stats']['OBJ']
    richardson = (9*e[300] - e[100])/8
    g1 = pd.read_csv(gold / '00_var_speed_limit_100_GPOPSI.csv')['Energy [kWh]'].iloc[0]
    g2 = pd.read_csv(gold / '00_var_speed_limit_100_GPOPSII.csv')['Energy [kWh]'].iloc[0]
    assert abs(richardson - g1) < 0.02 and abs(richardson - g2) < 0.02, (e, richardson)


@pytest.mark.gpu
@pytest.mark.parametrize('numSteps,numApprox', [(2, 0), (1, 2), (3, 1)])
def test_gpu_integrated_losses_with_other_rk_settings(numSteps, numApprox):
    "integrateLosses on top of the joint (t, b) RK4 map, several RK steps and several trapezoid pieces: the folding uses whatever d tau the map has."
    from mseetc.ocp import casadiSolver
    train, track = cases.train_default(), cases.track_00(30000)
    N, T = 80, 1150.0
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=400, integrateLosses=True, integrationOptions=dict(numSteps=numSteps, numApproxSteps=numApprox)),
                          startingPoint='profile')
    res = solver.solveBatch([T, 1.1*T], initialVelocity=8, terminalVelocity=6, multipliers=True)
    prob = cases.oracle_problem(train, track, N, numSteps=numSteps, numApproxSteps=numApprox, integration=dict(integrateLosses=True))
    for k, Tk in enumerate((T, 1.1*T)):
        ref = oracle.solve(prob, prob.scenario(Tk, 0.0, 6.0, 8.0), start='profile')
        assert res['status'][k] == 0 and ref['stats']['STATUS'] == 0
        assert abs(res['cost'][k] - ref['stats']['OBJ']) <= 1e-8*abs(ref['stats']['OBJ'])
        assert np.max(np.abs(res['z'][k] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-5
        assert np.max(np.abs(res['lam_g'][k] - ref['lam_g'])/np.maximum(1, np.abs(ref['lam_g']))) < 5e-3
