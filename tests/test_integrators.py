"""
TrainIntegrator with the 'IRK' (collocation) and 'CVODES' (adaptive) integrators -- reference: mseetc/train.py:303-322,
used by simulations/figure4.py.  CPU part: the collocation tables (casadi.collocation_points / collocation_interpolators restated
on the host).  GPU part: the device integrators against numpy/scipy restatements, and the reference's own consistency check of
figure4.py:109-140 (space-domain integration over 100 m against the time-domain one, 1e-8).
"""

import warnings

import numpy as np
import pytest

import cases


def test_collocation_points_known_values():
    from mseetc.train import collocationPoints
    assert np.allclose(collocationPoints(1, 'radau'), [1.0])
    assert np.allclose(collocationPoints(2, 'radau'), [1/3, 1.0], atol=1e-15)
    assert np.allclose(collocationPoints(3, 'radau'), [(4 - np.sqrt(6))/10, (4 + np.sqrt(6))/10, 1.0], atol=1e-15)
    assert np.allclose(collocationPoints(1, 'legendre'), [0.5])
    assert np.allclose(collocationPoints(2, 'legendre'), [0.5 - np.sqrt(3)/6, 0.5 + np.sqrt(3)/6], atol=1e-15)
    assert np.allclose(collocationPoints(3, 'legendre'), [0.5 - np.sqrt(15)/10, 0.5, 0.5 + np.sqrt(15)/10], atol=1e-15)
    for bad in (0, 10, 2.5):
        with pytest.raises(ValueError):
            collocationPoints(bad, 'radau')
    with pytest.raises(ValueError):
        collocationPoints(3, 'lobatto')


@pytest.mark.parametrize('scheme', ['radau', 'legendre'])
@pytest.mark.parametrize('order', range(1, 10))
def test_collocation_tables_differentiate_polynomials_exactly(order, scheme):
    from mseetc.train import collocationPoints, collocationTables
    tau = np.array([0.0] + collocationPoints(order, scheme))
    assert np.all(np.diff(tau) > 0) and tau[-1] <= 1.0
    C, D = collocationTables(order, scheme)
    rng = np.random.default_rng(order)
    poly = np.poly1d(rng.normal(size=order + 1))          # degree <= order: interpolated exactly by order + 1 points
    x = poly(tau)
    assert np.allclose(x @ C, poly.deriv()(tau), rtol=0, atol=1e-7*np.max(np.abs(C)))
    assert abs(x @ D - poly(1.0)) <= 1e-9*max(1.0, np.max(np.abs(x)))


def test_integrator_option_validation():
    from mseetc.train import TrainIntegrator, OptionsIRK, OptionsCVODES
    model = cases.train_default().exportModel()
    with pytest.raises(ValueError):
        TrainIntegrator(model, 'EULER')
    for bad in ({'order': 0}, {'order': 10}, {'numSteps': 0}, {'collMethod': 'lobatto'}, {'maxIter': 0}, {'jit': 1}, {'nope': 1}):
        with pytest.raises(ValueError):
            OptionsIRK(bad)
    for bad in ({'absTol': 1.0}, {'relTol': 0.0}, {'nope': 1}):
        with pytest.raises(ValueError):
            OptionsCVODES(bad)
    integ = TrainIntegrator(model, 'IRK', {'order': 3})
    assert integ.opts.collMethod == 'radau' and len(integ._params) == 4 + 16 + 4
    nobrake = cases.train_default(); nobrake.forceMinPn = 0
    with pytest.raises(ValueError):
        TrainIntegrator(nobrake.exportModel(), 'CVODES').solve(0, 100.0, 50.0, traction=0.1, pnBrake=-0.1)


# ---- numpy restatements used as checkers on the GPU box -------------------------------------------------------------------------

def _rhs(model, ds, w, G, b):
    v = np.sqrt(b)
    return ds/v, 2*ds*(w - (model.sr0 + model.sr1*v + model.sr2*b) - G)


def _irk_numpy(model, C, D, numSteps, t, b, ds, w, G, joint, h=1.0):
    from scipy.optimize import fsolve
    d = len(D) - 1
    dt = h/numSteps
    for _ in range(numSteps):
        def eqs(v):
            vt, vb = (v[:d], v[d:]) if joint else (None, v)
            xs_b = np.concatenate([[b], vb])
            ft, fb = _rhs(model, ds, w, G, vb)
            rb = dt*fb - xs_b @ C[:, 1:]
            if not joint:
                return rb
            xs_t = np.concatenate([[t], vt])
            return np.concatenate([dt*ft - xs_t @ C[:, 1:], rb])
        v0 = np.concatenate([np.full(d, t), np.full(d, b)]) if joint else np.full(d, b)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            v = fsolve(eqs, v0, xtol=1e-14)
        if joint:
            t = D[0]*t + v[:d] @ D[1:]
            b = D[0]*b + v[d:] @ D[1:]
        else:
            b = D[0]*b + v @ D[1:]
    return t, b


@pytest.mark.gpu
@pytest.mark.parametrize('order,scheme,numSteps,numApprox', [(3, 'radau', 1, 0), (2, 'legendre', 2, 0), (1, 'radau', 3, 0), (5, 'radau', 1, 0),
                                                             (3, 'radau', 1, 1), (4, 'legendre', 2, 3), (9, 'radau', 1, 0)])
def test_collocation_integrator_vs_numpy(order, scheme, numSteps, numApprox):
    from mseetc.train import TrainIntegrator, collocationTables
    model = cases.train_default().exportModel()
    integ = TrainIntegrator(model, 'IRK', {'order': order, 'collMethod': scheme, 'numSteps': numSteps, 'numApproxSteps': numApprox, 'maxIter': 50})
    C, D = collocationTables(order, scheme)
    rng = np.random.default_rng(7)
    n = 40
    t0, b0 = rng.uniform(0, 500, n), rng.uniform(4, 1600, n)
    ds, w = rng.uniform(5, 400, n), rng.uniform(-0.3, 0.5, n)
    grad, curv = rng.uniform(-0.015, 0.015, n), rng.uniform(-1/320, 1/320, n)
    keep = b0 + 2*ds*(w - 0.05) > 4.0      # stay clear of standstill within the interval
    out = integ.solveMany(t0, b0, ds, w, grad, curv)
    for k in np.flatnonzero(keep):
        G = model.resistance(grad[k], curv[k])
        if numApprox == 0:
            t, b = _irk_numpy(model, C, D, numSteps, t0[k], b0[k], ds[k], w[k], G, True)
        else:
            bs = [b0[k]] + [_irk_numpy(model, C, D, numSteps, 0.0, b0[k], ds[k], w[k], G, False, h=j/numApprox)[1] for j in range(1, numApprox + 1)]
            t = t0[k] + sum(2*ds[k]/numApprox/(np.sqrt(bs[j]) + np.sqrt(bs[j + 1])) for j in range(numApprox))
            b = bs[-1]
        assert abs(out['velSquared'][k] - b) <= 1e-9*max(1.0, abs(b)), (k, out['velSquared'][k], b)
        assert abs(out['time'][k] - t) <= 1e-9*max(1.0, abs(t)), (k, out['time'][k], t)


@pytest.mark.gpu
def test_adaptive_integrator_vs_scipy_and_order_of_collocation():
    from scipy.integrate import solve_ivp
    from mseetc.train import TrainIntegrator
    model = cases.train_default().exportModel()
    tight = TrainIntegrator(model, 'CVODES', {'absTol': 1e-12, 'relTol': 1e-14})
    rng = np.random.default_rng(3)
    n = 24
    b0, ds, w = rng.uniform(25, 1600, n), rng.uniform(20, 500, n), rng.uniform(-0.2, 0.5, n)
    grad = rng.uniform(-0.01, 0.01, n)
    out = tight.solveMany(np.zeros(n), b0, ds, w, grad, 0.0)
    for k in range(n):
        G = model.resistance(grad[k], 0.0)
        if b0[k] + 2*ds[k]*(w[k] - 0.05 - abs(G)) < 9.0:
            continue
        ref = solve_ivp(lambda s, y: _rhs(model, ds[k], w[k], G, y[1]), (0, 1), [0.0, b0[k]], method='DOP853', rtol=1e-13, atol=1e-13)
        assert abs(out['time'][k] - ref.y[0, -1]) <= 1e-9*max(1.0, ref.y[0, -1])
        assert abs(out['velSquared'][k] - ref.y[1, -1]) <= 1e-9*ref.y[1, -1]
    # the collocation error falls with the order (2d - 1 for Radau): one 300 m interval, accelerating from 20 km/h
    exact = tight.solve(0.0, (20/3.6)**2, 300.0, traction=0.4)
    errs = []
    for order in (1, 2, 3, 4):
        o = TrainIntegrator(model, 'IRK', {'order': order, 'maxIter': 50}).solve(0.0, (20/3.6)**2, 300.0, traction=0.4)
        errs.append(abs(o['velSquared'] - exact['velSquared']) + abs(o['time'] - exact['time']))
    assert errs[0] > errs[1] > errs[2] > errs[3] and errs[3] < 1e-2*errs[0]


@pytest.mark.gpu
def test_figure4_space_versus_time_consistency():
    # simulations/figure4.py:16-140: 100 steps of 1 m with the tight adaptive integrator; cases c and d (braking with -0.5 N/kg from
    # 36.61894 and 37.95880 km/h) end at 1 and 10 km/h; integrating the same motion in the time domain over the elapsed time must
    # return the 100 m and the same speed to 1e-8 (figure4.py:109-140)
    from mseetc.train import TrainIntegrator
    from mseetc import _device
    train = cases.train_default()
    model = train.exportModel()
    integ = TrainIntegrator(model, 'CVODES', {'absTol': 1e-12, 'relTol': 1e-14})
    ends = {}
    for name, f, v0 in (('a', 0.5, 1.0), ('b', 0.5, 10.0), ('c', -0.5, 36.61894), ('d', -0.5, 37.95880)):
        t, b = 0.0, (v0/3.6)**2
        for _ in range(100):
            out = integ.solve(time=t, velocitySquared=b, ds=1.0, traction=f)
            t, b = out['time'], out['velSquared']
        ends[name] = (t, np.sqrt(b))
        pos, vel = _device.resimulate(model, [[f]], [[t]], [0.0], [0.0], [0.0], [v0/3.6])
        assert abs(pos[0, -1] - 100.0) <= 1e-8 and abs(vel[0, -1] - np.sqrt(b)) <= 1e-8
    assert abs(ends['c'][1]*3.6 - 1.0) < 2e-3 and abs(ends['d'][1]*3.6 - 10.0) < 1e-3


@pytest.mark.gpu
def test_rolling_resistance_integration_vs_scipy_and_post_processing():
    # TrainIntegrator.calcRollingResistance (train.py:416-454) and the 'Rolling resistance [kWh]' column of
    # postProcessDataFrame(integrateRollingResistance=True) (utils.py:296-320)
    from scipy.integrate import solve_ivp
    from mseetc.train import TrainIntegrator
    from mseetc.ocp import casadiSolver
    from mseetc.utils import postProcessDataFrame
    train = cases.train_default()
    model = train.exportModel()
    integ = TrainIntegrator(model, 'RK')
    with pytest.raises(ValueError):
        integ.calcRollingResistance(10.0, 100.0)
    with pytest.raises(ValueError):
        integ.initRollingResistance(solver='EULER')
    integ.initRollingResistance(solver='CVODES')
    rng = np.random.default_rng(5)
    n = 16
    v0, ds, f = rng.uniform(5, 40, n), rng.uniform(20, 500, n), rng.uniform(0.0, 0.4, n)
    grad = rng.uniform(-0.01, 0.01, n)
    loss, vEnd = integ.calcRollingResistance(v0, ds, f, 0.0, grad, 0.0)
    for k in range(n):
        G = model.resistance(grad[k], 0.0)
        def rhs(s, y):
            v = np.sqrt(y[0]); rr = model.sr0 + model.sr1*v + model.sr2*y[0]
            return [2*ds[k]*(f[k] - rr - G), ds[k]*rr]
        if v0[k]**2 + 2*ds[k]*(f[k] - 0.05 - abs(G)) < 9.0:
            continue
        ref = solve_ivp(rhs, (0, 1), [v0[k]**2, 0.0], method='DOP853', rtol=1e-12, atol=1e-12)
        assert abs(loss[k] - ref.y[1, -1]) <= 2e-6*ref.y[1, -1] + 1e-7      # integrated at abstol 1e-8 / reltol 1e-6 (train.py:436)
        assert abs(vEnd[k] - np.sqrt(ref.y[0, -1])) <= 1e-5*vEnd[k]
    one = integ.calcRollingResistance(float(v0[0]), float(ds[0]), float(f[0]), 0.0, float(grad[0]), 0.0)
    assert one[0] == loss[0] and one[1] == vEnd[0]
    # solver='RK' (train.py:428-432: casadi.simpleRK(fun, 2, 4)): two steps of classic RK4 on (b, e) over the unit interval, restated in numpy
    rk = TrainIntegrator(model, 'RK')
    rk.initRollingResistance(solver='RK')
    lossRK, vEndRK = rk.calcRollingResistance(v0, ds, f, 0.0, grad, 0.0)
    for k in range(n):
        G = model.resistance(grad[k], 0.0)
        fun = lambda y: np.array([2*ds[k]*(f[k] - (model.sr0 + model.sr1*np.sqrt(y[0]) + model.sr2*y[0]) - G), ds[k]*(model.sr0 + model.sr1*np.sqrt(y[0]) + model.sr2*y[0])])
        y = np.array([v0[k]**2, 0.0])
        if v0[k]**2 + 2*ds[k]*(f[k] - 0.05 - abs(G)) < 9.0:
            continue
        for _ in range(2):
            k1 = fun(y); k2 = fun(y + 0.25*k1); k3 = fun(y + 0.25*k2); k4 = fun(y + 0.5*k3)
            y = y + (0.5/6)*(k1 + 2*k2 + 2*k3 + k4)
        assert abs(lossRK[k] - y[1]) <= 1e-12*abs(y[1]) and abs(vEndRK[k] - np.sqrt(y[0])) <= 1e-12*vEndRK[k]
        assert abs(lossRK[k] - loss[k]) <= 1e-3*loss[k]      # two RK4 steps against the adaptive pair
    # post-processing column: present, NaN in the last row, close to the mid-point estimate on a solved trajectory
    solver = casadiSolver(train, cases.track_00(), dict(numIntervals=100, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1)))
    df, stats = solver.solve(1600.0)
    full = postProcessDataFrame(df[['Position [m]', 'Velocity [m/s]', 'Force (el) [N]', 'Force (pnb) [N]', 'Slacks']], solver.points, train,
                                CVODES=False, integrateRollingResistance=True)
    col = full['Rolling resistance [kWh]'].values
    assert np.isnan(col[-1]) and np.all(col[:-1] > 0)
    accel = full['Force (el) [N]'].values[:-1] >= 0      # where the reference's force convention equals the real one
    vel, pos = full['Velocity [m/s]'].values, full['Position [m]'].values
    vm = 0.5*(vel[:-1] + vel[1:])
    mid = (1e-6/3.6)*(train.r0 + train.r1*vm + train.r2*vm**2)*np.diff(pos)
    coast = accel & (np.abs(full['Force (pnb) [N]'].values[:-1]) < 1.0) & (vel[:-1] > 10)
    assert coast.sum() > 20 and np.allclose(col[:-1][coast], mid[coast], rtol=0.05)
