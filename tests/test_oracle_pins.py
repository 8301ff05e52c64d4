"""
Pins of the CPU oracle against numbers the reference stores and
against an independent numpy restatement of the NLP.  CPU only.
"""

import numpy as np
import pandas as pd
import pytest
from pathlib import Path
from scipy.integrate import solve_ivp

import cases
from nlp_numpy import kkt_certificate
from oracle import oracle
from oracle.oracle import DP, IP

GOLD = Path(__file__).resolve().parent / 'golden'


def _ipdp(numSteps=1, numApprox=1):
    prob = cases.oracle_problem(cases.train_default(), cases.track_00(), 100, numSteps=numSteps, numApproxSteps=numApprox)
    return prob


# ---- integrator -----------------------------------------------------------------

def _truth(prob, b0, w, ds, grad=0.0):
    dp = prob.dp
    G = dp[DP['G']]*grad/dp[DP['RHO']]
    rhs = lambda s, y: [1/np.sqrt(y[1]), 2*(w - (dp[DP['SR0']] + dp[DP['SR1']]*np.sqrt(y[1]) + dp[DP['SR2']]*y[1]) - G)]
    sol = solve_ivp(rhs, [0, ds], [0.0, b0], rtol=1e-13, atol=1e-13, method='DOP853')
    return sol.y[0, -1], sol.y[1, -1]


@pytest.mark.parametrize('v0kmh,f,expect', [(36.61894, -0.5, 1.0), (37.95880, -0.5, 10.0)])
def test_figure4_braking_constants(v0kmh, f, expect):
    # simulations/figure4.py:22-23: braking with f = -0.5 N/kg over 100 m ends at 1 resp. 10 km/h
    prob = _ipdp(numSteps=50, numApprox=0)
    b = (v0kmh/3.6)**2
    t = 0.0
    for _ in range(100):                    # 1 m pieces like the reference's high-resolution loop
        out = oracle.stage_eval(prob, b, f, 1.0)
        t, b = t + out[0], out[1]
    assert abs(np.sqrt(b)*3.6 - expect) < 2e-3
    tt, bt = _truth(prob, (v0kmh/3.6)**2, f, 100.0)
    assert abs(b - bt) < 1e-8*max(1, bt)
    assert abs(t - tt) < 1e-6*tt


@pytest.mark.parametrize('numSteps,numApprox', [(1, 1), (1, 0), (3, 2), (2, 0)])
def test_interval_map_matches_numpy_and_converges(numSteps, numApprox):
    prob = _ipdp(numSteps, numApprox)
    nlp = cases.numpy_nlp(prob)
    rng = np.random.default_rng(3)
    for _ in range(20):
        b0, w, ds, grad = rng.uniform(300, 1500), rng.uniform(-0.4, 0.5), rng.uniform(5, 300), rng.uniform(-0.015, 0.015)
        out = oracle.stage_eval(prob, b0, w, ds, grad)
        # same formulas in numpy (vector of length 1)
        nlp.ds = np.array([ds]); nlp.G = np.array([prob.dp[DP['G']]*grad/prob.dp[DP['RHO']]])
        tau, bp = nlp.interval(np.array([b0]), np.array([w]))
        assert abs(out[0] - tau[0]) <= 1e-13*abs(tau[0]) and abs(out[1] - bp[0]) <= 1e-13*abs(bp[0])


def test_interval_map_derivatives_by_complex_step():
    prob = _ipdp(2, 2)
    nlp = cases.numpy_nlp(prob)
    rng = np.random.default_rng(5)
    for _ in range(10):
        b0, w, ds = rng.uniform(300, 1500), rng.uniform(-0.4, 0.5), rng.uniform(5, 300)
        out = oracle.stage_eval(prob, b0, w, ds)
        nlp.ds = np.array([ds]); nlp.G = np.array([0.0])
        h = 1e-30
        tb, bb = nlp.interval(np.array([b0 + 1j*h]), np.array([w + 0j]))
        tw, bw = nlp.interval(np.array([b0 + 0j]), np.array([w + 1j*h]))
        for got, ref in [(out[2], tb.imag[0]/h), (out[3], tw.imag[0]/h), (out[4], bb.imag[0]/h), (out[5], bw.imag[0]/h)]:
            assert abs(got - ref) <= 1e-11*max(1e-6, abs(ref))
        # second derivatives: central differences of the first ones
        e = 1e-5
        def first(b, w_):
            o = oracle.stage_eval(prob, b, w_, ds)
            return o[2:6]
        db = (first(b0*(1 + e), w) - first(b0*(1 - e), w))/(2*e*b0)
        dw = (first(b0, w + e) - first(b0, w - e))/(2*e)
        for got, ref in [(out[6], db[0]), (out[7], dw[0]), (out[7], db[1]), (out[8], dw[1]), (out[9], db[2]), (out[10], dw[2]), (out[10], db[3]), (out[11], dw[3])]:
            assert abs(got - ref) <= 1e-4*max(1e-9, abs(ref)) + 1e-12


def test_trapezoid_time_formula():
    # numSteps = numApproxSteps = 1  ->  t+ = t + 2 ds/(sqrt(b) + sqrt(b+))
    prob = _ipdp(1, 1)
    out = oracle.stage_eval(prob, 300.0, 0.2, 250.0)
    assert abs(out[0] - 2*250.0/(np.sqrt(300.0) + np.sqrt(out[1]))) < 1e-13


# ---- full solves ------------------------------------------------------------------------

def _solve(prob, **kw):
    dp = prob.scenario(**kw)
    out = oracle.solve(prob, dp)
    assert out['stats']['STATUS'] == 0
    return dp, out


def test_gpops_energy_is_the_discretisation_limit():
    # gpops/00_var_speed_limit_100_GPOPS{I,II}.csv: 440.1415 / 440.1406 kWh for the figure10.py configuration.
    # The multiple-shooting value converges to it as O(1/N^2).
    e = {}
    for N in (100, 300):
        prob = cases.oracle_problem(cases.train_fig10(), cases.track_00(), N)
        dp, out = _solve(prob, terminalTime=1541.0)
        e[N] = out['stats']['OBJ']
    assert abs(e[100] - 448.6395) < 1e-3       # regression of the session probe value 448.640
    assert abs(e[300] - 441.0838) < 1e-3
    richardson = (9*e[300] - e[100])/8
    g1 = pd.read_csv(GOLD / '00_var_speed_limit_100_GPOPSI.csv')['Energy [kWh]'].iloc[0]
    g2 = pd.read_csv(GOLD / '00_var_speed_limit_100_GPOPSII.csv')['Energy [kWh]'].iloc[0]
    assert abs(richardson - g1) < 0.005 and abs(richardson - g2) < 0.005      # (measured: 0.0025 / 0.0016 kWh)


def test_gpops_trajectory_is_the_limit_of_the_shooting_solutions():
    """
    gpops/00_var_speed_limit_100_GPOPSII.csv holds the trajectory too (t, s, v: what figure10.py:50-55,81-85 overlays on the DMS solution).  The oracle's
    solutions of the same configuration approach it with N: max |dv| 2.15 / 0.64 / 0.21 m/s, max |dt| 10.2 / 5.0 / 2.3 s at N = 100 / 300 / 1000.
    """
    dev = {}
    for N in (100, 300, 1000):
        prob = cases.oracle_problem(cases.train_fig10(), cases.track_00(), N, maxIterations=1000)
        out = oracle.solve(prob, prob.scenario(terminalTime=1541.0), start='profile')
        assert out['stats']['STATUS'] == 0
        dev[N] = cases.gpops_profile_deviation(out['z'], prob.ds)
    assert dev[100][0] < 2.3 and dev[300][0] < 0.7 and dev[1000][0] < 0.25, dev
    assert dev[100][2] < 11 and dev[300][2] < 5.5 and dev[1000][2] < 2.6, dev
    assert dev[1000][1] < 0.065


def test_minimum_time_constant_of_figure5():
    # simulations/figure5.py:96: minimumTime = 272.4726 is the result of a time-optimal casadiSolver run
    # (8.5 km crop, v0 = 1, vN = 100 km/h, config.json -> N = 300) -- an output of the reference's IPOPT path.
    prob = cases.oracle_problem(cases.train_fig5(), cases.track_00(8500), 300, energyOptimal=False, losses='none')
    dp, out = _solve(prob, terminalTime=400.0, terminalVelocity=100/3.6, initialVelocity=1)
    assert abs(out['z'][-2] - 272.4726) < 1.5e-4


@pytest.mark.parametrize('case', ['c1', 'c1_tight', 'c2', 'fig10', 'mintime'])
def test_kkt_certificate_of_oracle_solutions(case):
    if case == 'c1':
        prob, kw = cases.oracle_problem(cases.train_default(), cases.track_00(), 100), dict(terminalTime=1600.0)
    elif case == 'c1_tight':
        prob, kw = cases.oracle_problem(cases.train_default(), cases.track_00(), 60), dict(terminalTime=1541.0)
    elif case == 'c2':
        prob, kw = cases.oracle_problem(cases.train_default(), cases.track_CH(), 200), dict(terminalTime=1242.0)
    elif case == 'fig10':
        prob, kw = cases.oracle_problem(cases.train_fig10(), cases.track_00(), 100), dict(terminalTime=1541.0)
    else:
        prob = cases.oracle_problem(cases.train_fig5(), cases.track_00(8500), 100, energyOptimal=False, losses='none')
        kw = dict(terminalTime=400.0, terminalVelocity=100/3.6, initialVelocity=1)
    dp, out = _solve(prob, **kw)
    nlp = cases.numpy_nlp(prob)
    z = out['z']
    obj, g = oracle.nlp_eval(prob, dp, z)
    assert abs(obj - nlp.obj(z)) <= 1e-12*abs(obj)
    assert np.allclose(g, nlp.cons(z), rtol=1e-12, atol=1e-12)
    cert = kkt_certificate(nlp, z, out['lam_g'], dp[DP['T0']], dp[DP['TEND']], dp[DP['V0SQ']], dp[DP['VNSQ']])
    assert cert['feas_g'] < 1.5e-8 and cert['feas_z'] < 1.5e-8   # bound_relax_factor = 1e-8
    assert cert['stat'] < 1e-6
    assert cert['sign_g'] < 1e-6


def test_solves_are_deterministic():
    # the reference's scripts assert identical iteration counts over repeated solves (figure6.py:191-193)
    prob = cases.oracle_problem(cases.train_default(), cases.track_00(), 100)
    dp = prob.scenario(1541.0)
    a, b = oracle.solve(prob, dp), oracle.solve(prob, dp)
    assert a['stats']['ITERS'] == b['stats']['ITERS'] and np.array_equal(a['z'], b['z'])


def test_batch_equals_single():
    prob = cases.oracle_problem(cases.train_default(), cases.track_00(), 100)
    T = cases.c1_times(6)
    scen = np.array([[0.0, t, 1.0, 1.0] for t in T])
    zb, st, nfail = oracle.solve_batch(prob, scen, nthreads=2)
    assert nfail == 0
    for k in range(len(T)):
        one = oracle.solve(prob, prob.scenario(float(T[k])))
        assert np.array_equal(one['z'], zb[k])
