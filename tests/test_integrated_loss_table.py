"""
integrateLosses=True together with a loss table -- the dynamic loss model of mseetc/efficiency.py (or any tabulated loss function) integrated over the running
time of every interval inside the NLP (reference: mseetc/ocp.py:118-120,231-241 -> TrainIntegrator.initLosses / calcLosses, mseetc/train.py:367-413; the
reference's switch for it sits at simulations/figure6.py:178).  Round 6: oracle (ms_oracle.c: loss_energy), kernels (csrc/msd_lossint_table.hpp, DYN = 3).

  * the two loss integrals E_tr, E_rgb and their derivatives against scipy and differences (oracle, CPU);
  * the NLP: both starting points reach one optimum, the loss slacks equal the independently integrated losses of the solution, the optimum is next to
    the mid-point transcription's (oracle, CPU);
  * the emulated kernels follow the oracle iterate for iterate (CPU);
  * the HIP path against the oracle on the figure-6 configuration, and the NLP's slack sum against the post-processing integration (GPU).
"""

import ctypes

import numpy as np
import pytest

import cases
from oracle import oracle
from oracle.oracle import DP

RK11 = dict(numSteps=1, numApproxSteps=1)


def _dynamic_train():
    "figure5.py / figure6.py: VIRM6 without the pneumatic brake, the dynamic loss model of efficiency.py"
    from mseetc.train import Train
    from mseetc.efficiency import totalLossesFunction
    train = Train(config={'id': 'NL_Intercity_VIRM6'})
    train.forceMinPn = 0
    train.powerLosses = totalLossesFunction(train, auxiliaries=27000, etaGear=0.96)
    return train


def _problem(train, track, N, integrateLosses=True, maxIterations=500):
    from mseetc.track import computeDiscretizationPoints
    oracle.set_loss_table(train.powerLosses.parameters(train.mass*train.rho))
    opts = dict(numIntervals=N, maxIterations=maxIterations, energyOptimal=True, minimumVelocity=1, numSteps=1, numApproxSteps=1, integrateLosses=integrateLosses)
    return oracle.pack_problem(train, computeDiscretizationPoints(track, N), opts, 2, 0.0, 0.0, track.length)


def _loss_energy(prob, v0, dt, w, f, grad=0.0, tol=(0.0, 0.0)):
    L = oracle.lib()
    iptr, dptr = ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)
    L.oracle_loss_energy.restype = None
    L.oracle_loss_energy.argtypes = [iptr, dptr] + [ctypes.c_double]*8 + [dptr]
    out = np.zeros(30)
    L.oracle_loss_energy(oracle._i(prob.ip), oracle._d(prob.dp), v0, dt, w, f, grad, 0.0, tol[0], tol[1], oracle._d(out))
    return out.reshape(2, 15)


def _scipy_losses(train, prob, v0, dt, w, f, grad):
    "the two split loss integrals by an independent route: the host's loss function (efficiency.py restated in mseetc/efficiency.py), split like utils.py:197-220, scipy's RK45"
    from scipy.integrate import solve_ivp
    M = train.mass*train.rho
    sr0, sr1, sr2, g, rho = [prob.dp[DP[k]] for k in ('SR0', 'SR1', 'SR2', 'G', 'RHO')]
    G = g*grad/rho
    spec = lambda ff, v: train.powerLosses(ff*M, v)/M
    tol = 1e-6      # (slope of the linear extension by a difference: 1e-10 like utils.py:207 drowns in the table's rounding)
    Ltr = lambda ff, v: spec(ff, v) if ff >= 0 else (spec(tol, v) - spec(0.0, v))/tol*ff + spec(0.0, v)
    Lrg = lambda ff, v: spec(ff, v) if ff < 0 else (spec(0.0, v) - spec(-tol, v))/tol*ff + spec(0.0, v)
    sol = solve_ivp(lambda t, y: [w - (sr0 + sr1*y[0] + sr2*y[0]**2) - G, Ltr(f, y[0]), Lrg(f, y[0])], [0, dt], [v0, 0.0, 0.0], rtol=1e-9, atol=1e-11)
    return sol.y[1:, -1]


def test_oracle_loss_integrals_vs_scipy_and_differences():
    train = _dynamic_train()
    prob = _problem(train, cases.track_00(8500), 30)
    idx = [(0, 0), (0, 1), (0, 2), (0, 3), (1, 1), (1, 2), (1, 3), (2, 2), (2, 3), (3, 3)]
    # (operating points whose speed range stays inside one smooth piece of the model: no crossing of the turning speed 14.6 m/s or of the table's 20 km/h edge,
    #  where the differences straddle a kink of efficiency.py:7-12,40)
    for (v0, dt, w, f, grad) in [(20.0, 20.0, 0.25, 0.25, 0.002), (8.0, 6.0, 0.30, 0.30, 0.0), (25.0, 12.0, -0.5, -0.2, 0.0), (30.0, 10.0, -0.15, -0.15, -0.004)]:
        E = _loss_energy(prob, v0, dt, w, f, grad)                       # CVODES' tolerances (train.py:396)
        Et = _loss_energy(prob, v0, dt, w, f, grad, tol=(1e-13, 1e-11))   # tight: what the differences below are taken of
        ref = _scipy_losses(train, prob, v0, dt, w, f, grad)
        truth = 0 if f >= 0 else 1                                       # the row whose loss function is the table itself (the other one is its linear extension)
        assert abs(Et[truth, 0] - ref[truth]) <= 1e-7*abs(ref[truth])
        assert np.all(np.abs(Et[:, 0] - ref) <= 5e-5*np.maximum(1e-3, np.abs(ref)))      # (the extension row: the reference takes its slope by a difference)
        assert np.all(np.abs(E[:, 0] - Et[:, 0]) <= 1e-4*np.maximum(1e-3, np.abs(Et[:, 0])))      # reltol 1e-6 per step
        h, x0 = 1e-4, np.array([v0, dt, w, f])
        g, H = np.zeros((2, 4)), np.zeros((2, 4, 4))
        for a in range(4):
            xp, xm = x0.copy(), x0.copy()
            xp[a] += h; xm[a] -= h
            ep, em = _loss_energy(prob, *xp, grad, tol=(1e-13, 1e-11)), _loss_energy(prob, *xm, grad, tol=(1e-13, 1e-11))
            g[:, a] = (ep[:, 0] - em[:, 0])/(2*h)
            H[:, a, :] = (ep[:, 1:5] - em[:, 1:5])/(2*h)
        assert np.max(np.abs(Et[:, 1:5] - g)/np.maximum(np.abs(g), 1e-3*np.max(np.abs(g)))) < 2e-4
        fd = np.array([0.5*(H[truth, a, b] + H[truth, b, a]) for a, b in idx])
        # second derivatives of the row that is the table itself (the linear extension drops d3L/df dv2 by convention: ms_oracle.c: loss_rows)
        assert np.max(np.abs(Et[truth, 5:] - fd)/np.maximum(np.abs(fd), 2e-2*np.max(np.abs(fd)))) < 2e-2, (v0, dt, w, f)


def test_oracle_nlp_with_integrated_loss_table():
    """
    figure5.py's configuration (8.5 km, v0 = 1 m/s, vN = 100 km/h) with the dynamic loss model and the switch of figure6.py:178 on.  Both starting points reach
    one optimum; at it every slack equals the larger of the two loss integrals of its interval, evaluated by an independent route (scipy over the host's loss
    function) at the solution's (v_i, t_{i+1} - t_i, f_i); and the energy is next to the mid-point transcription's (the two bound the same losses).
    """
    train, track, N, T = _dynamic_train(), cases.track_00(8500), 40, 272.4726*1.2
    prob = _problem(train, track, N)
    kw = dict(terminalVelocity=100/3.6, initialVelocity=1.0)
    out = {s: oracle.solve(prob, prob.scenario(T, **kw), start=s) for s in ('profile', 'reference')}
    assert out['profile']['stats']['STATUS'] == 0 and out['reference']['stats']['STATUS'] == 0
    assert abs(out['profile']['stats']['OBJ'] - out['reference']['stats']['OBJ']) <= 1e-6*abs(out['reference']['stats']['OBJ'])
    z = out['profile']['z']
    f, s, t, b = z[0:4*N:4], z[1:4*N:4], np.r_[z[2:4*N:4], z[-2]], np.r_[z[3:4*N:4], z[-1]]
    total = 0.0
    for i in range(0, N, 3):
        ref = _scipy_losses(train, prob, np.sqrt(b[i]), t[i + 1] - t[i], f[i], f[i], prob.grad[i])
        assert abs(s[i] - max(ref)) <= 2e-5*max(1e-2, abs(max(ref))), (i, s[i], ref)
        total += s[i]
    assert total > 0
    mid = _problem(train, track, N, integrateLosses=False)
    rm = oracle.solve(mid, mid.scenario(T, **kw), start='profile')
    assert rm['stats']['STATUS'] == 0
    assert abs(out['profile']['stats']['OBJ'] - rm['stats']['OBJ']) <= 2e-2*abs(rm['stats']['OBJ'])


@pytest.mark.parametrize('N,start', [(30, 'profile'), (30, 'reference'), (70, 'profile')])
def test_emulated_kernels_with_integrated_loss_table_match_oracle(N, start):
    "The kernels of the family (first pass + streamed follow-up kernel; 64 x 1 and 128 x 1) as host threads: same iterates as the oracle."
    from test_kernel_emulation import load_emulation
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    emu = load_emulation()
    train, track, T = _dynamic_train(), cases.track_00(8500), 272.4726*1.2
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=300, integrateLosses=True, integrationOptions=RK11), startingPoint=start)
    assert solver._desc.integrate_losses == 1 and solver._desc.loss_kind == 2
    scen = solver._scenarios(T, 0, 100/3.6, 1)
    nz = 4*N + 2
    z, lam, st, hist = np.zeros((1, nz)), np.zeros((1, 7*N)), np.zeros((1, ST['COUNT'])), np.zeros((8, 8))
    d = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert emu.emu_solve_batch(ctypes.byref(solver._desc), 1, d(scen), d(z), d(lam), d(st), d(hist), 8) == 0
    prob = _problem(train, track, N, maxIterations=300)
    ref = oracle.solve(prob, prob.scenario(T, 0.0, 100/3.6, 1.0), start=start)
    assert st[0, ST['STATUS']] == 0 and ref['stats']['STATUS'] == 0
    assert abs(int(st[0, ST['ITERS']]) - int(ref['stats']['ITERS'])) <= 1
    assert abs(st[0, ST['OBJ']] - ref['stats']['OBJ']) <= 1e-9*abs(ref['stats']['OBJ'])
    assert np.max(np.abs(z[0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-7


@pytest.mark.gpu
def test_gpu_integrated_loss_table_vs_oracle_and_post_processing():
    """
    The HIP path on the figure-6 configuration with the switch of figure6.py:178 on (dynamic loss model, integrateLosses=True), N = 100 and N = 60: against
    the oracle (objective 1e-6, variables 1e-4: the table's kinks leave two correct solvers on iterates a little apart, like the mid-point dynamic rows);
    and the NLP's own loss accounting -- the slack sum -- against the losses the post-processing integrates along the same trajectory
    (utils.py:261-289 -> msd_integrate_losses): 1e-6 of the total.
    """
    from mseetc.ocp import casadiSolver
    from mseetc.utils import postProcessDataFrame
    train, track = _dynamic_train(), cases.track_00(8500)
    for N in (100, 60):
        Ts = 272.4726*np.array([1.1, 1.2, 1.3])
        solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrateLosses=True, integrationOptions=RK11), startingPoint='profile')
        res = solver.solveBatch(Ts, terminalVelocity=100/3.6, initialVelocity=1)
        assert np.all(res['status'] == 0), res['status']
        prob = _problem(train, track, N)
        for k, T in enumerate(Ts):
            ref = oracle.solve(prob, prob.scenario(float(T), 0.0, 100/3.6, 1.0), start='profile')
            assert ref['stats']['STATUS'] == 0
            assert abs(res['cost'][k] - ref['stats']['OBJ']) <= 1e-6*abs(ref['stats']['OBJ'])
            assert np.max(np.abs(res['z'][k] - ref['z'])/np.maximum(1.0, np.abs(ref['z']))) <= 1e-4
            assert abs(int(res['iterations'][k]) - int(ref['stats']['ITERS'])) <= 5
        df, stats = solver.solve(float(Ts[1]), terminalVelocity=100/3.6, initialVelocity=1)
        assert df is not None
        dfi = postProcessDataFrame(df.copy(), solver.points, train, integrateLosses=True)
        unit = 1e-6/3.6
        nlp_losses = unit*float(np.nansum(df['Slacks'].values))            # slacks [J] (ocp.py:405: s * totalMass), the NLP's integrated losses
        post_losses = float(np.nansum(dfi['Losses [kWh]'].values.astype(float)))
        assert abs(nlp_losses - post_losses) <= 1e-6*abs(post_losses), (N, nlp_losses, post_losses)
        solver.close()
