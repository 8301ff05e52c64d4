"""
Host logic of the shrinking-horizon driver (mseetc/mpc.py) without a device: moving a solution between grids, and the
closed loop with the oracle standing in for the solver (cold vs warm start reach the same optima).
"""

import numpy as np

import cases


def _layout(N, pn, t, b, f, p, s):
    stp = 4 + pn
    z = np.zeros(stp*N + 2)
    body = z[:stp*N].reshape(N, stp)
    body[:, 0] = f
    if pn:
        body[:, 1] = p
    body[:, 1 + pn] = s
    body[:, 2 + pn] = t[:N]
    body[:, 3 + pn] = b[:N]
    z[stp*N], z[stp*N + 1] = t[N], b[N]
    return z


def test_transfer_onto_a_sub_grid_is_exact():
    from mseetc.mpc import transferSolution
    rng = np.random.default_rng(0)
    for pn in (0, 1):
        N = 12
        pos = np.cumsum(np.concatenate([[0.0], rng.uniform(50, 400, N)]))
        t, b = np.cumsum(rng.uniform(5, 20, N + 1)), rng.uniform(4, 900, N + 1)
        f, p, s = rng.normal(size=N), -rng.random(N), rng.random(N)
        z = _layout(N, pn, t, b, f, p, s)
        out = transferSolution(np.stack([z, 2*z]), pos, pos[3:], pn)
        want = _layout(N - 3, pn, t[3:], b[3:], f[3:], p[3:], s[3:])
        assert out.shape == (2, want.size)
        assert np.allclose(out[0], want, rtol=0, atol=1e-12) and np.allclose(out[1], 2*want, rtol=0, atol=1e-12)


def test_transfer_interpolates_states_and_holds_controls():
    from mseetc.mpc import transferSolution
    pos = np.array([0.0, 100.0, 300.0, 600.0])
    t, b = np.array([0.0, 10.0, 25.0, 40.0]), np.array([1.0, 100.0, 400.0, 1.0])
    z = _layout(3, 1, t, b, np.array([0.3, 0.1, -0.2]), np.array([0.0, 0.0, -0.1]), np.array([0.05, 0.01, 0.02]))
    new = np.array([200.0, 350.0, 600.0])
    out = transferSolution(z, pos, new, True)[0]
    body = out[:10].reshape(2, 5)
    assert np.allclose(body[:, 3], [17.5, 27.5]) and np.allclose(body[:, 4], [250.0, 400 - 399/6])
    assert np.allclose(body[:, 0], [0.1, -0.2]) and np.allclose(body[:, 1], [0.0, -0.1]) and np.allclose(body[:, 2], [0.01, 0.02])
    assert out[10] == 40.0 and out[11] == 1.0


class OracleSolver():
    "casadiSolver look-alike backed by the oracle (checker only; the packing front end never touches the device)"

    def __init__(self, train, track, opts):
        from mseetc.ocp import casadiSolver
        self._front = casadiSolver(train, track, opts)
        self.points, self.withPnBrake = self._front.points, self._front.withPnBrake
        io = opts.get('integrationOptions', {})
        self._prob = cases.oracle_problem(train, track, opts['numIntervals'], numSteps=io.get('numSteps', 1), numApproxSteps=io.get('numApproxSteps', 0))

    def solveBatch(self, T, initialTime=0, terminalVelocity=1, initialVelocity=1, guess=None, warmMu=1e-2, warmPush=1e-3, classifyFailures=False):
        from oracle import oracle
        scen = self._front._scenarios(T, initialTime, terminalVelocity, initialVelocity)
        rows = []
        for k in range(scen.shape[0]):
            dp = self._prob.scenario(scen[k, 1], scen[k, 0], np.sqrt(scen[k, 3]), np.sqrt(scen[k, 2]))
            rows.append(oracle.solve(self._prob, dp, guess=None if guess is None else guess[k], mu0=warmMu, push=warmPush))
        st = lambda key: np.array([r['stats'][key] for r in rows])
        return dict(z=np.stack([r['z'] for r in rows]), status=st('STATUS').astype(int), iterations=st('ITERS').astype(int), cost=st('OBJ'))


def test_closed_loop_warm_equals_cold_on_the_oracle():
    from mseetc.mpc import shrinkingHorizon
    train, track = cases.train_default(), cases.track_00(crop=20000)
    opts = dict(numIntervals=40, maxIterations=300, integrationOptions=dict(numSteps=1, numApproxSteps=1))
    T = np.array([820.0, 900.0])
    make = lambda a, b, c: OracleSolver(a, b, c)
    cold = shrinkingHorizon(train, track, opts, T, numResolves=4, noise=0.01, seed=3, solverFactory=make)
    warm = shrinkingHorizon(train, track, opts, T, numResolves=4, noise=0.01, seed=3, solverFactory=make, warmStart=True)
    assert len(cold) == len(warm) == 4
    for k, (c, w) in enumerate(zip(cold, warm)):
        assert np.all(c['status'] == 0) and np.all(w['status'] == 0)
        assert np.allclose(c['cost'], w['cost'], rtol=1e-6) and np.allclose(c['t0'], w['t0'], rtol=1e-6)
        if k > 0:
            assert np.all(w['iterations'] < c['iterations'])
