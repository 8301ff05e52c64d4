"""
ctypes front end of the CPU oracle (oracle/ms_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() as the
checker and by bench.py's cpu_baseline leg; never by the product package.

`pack_problem` restates, independently of the product's host code, how the reference
turns (train, track, options) into NLP data: mseetc/ocp.py:96-125 (specific bounds,
accInf = 10, power bounds, grid) and :266-269 (interior speed bound).
"""

import ctypes
import os
import subprocess
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent

IP = dict(N=0, WITH_PN=1, HAS_POWER=2, ENERGY_OPT=3, NUM_STEPS=4, NUM_APPROX=5, LOSS_KIND=6, MAX_ITER=7, INTEGRATOR=8, COLL_DEGREE=9, NEWTON_ITERS=10, INTEGRATE_LOSSES=11, WATCHDOG_TRIGGER=12, COUNT=13)
DP = dict(SR0=0, SR1=1, SR2=2, G=3, RHO=4, FMAX=5, FMIN=6, FMIN_PN=7, PW_UPPER=8, PW_LOWER=9, ACC_MIN=10, ACC_MAX=11,
          LOSS_CT=12, LOSS_CR=13, VMIN_SQ=14, OBJ_DEN=15, TOL=16, T0=17, TEND=18, V0SQ=19, VNSQ=20, INT_ATOL=21, INT_RTOL=22, COUNT=23)
ST = dict(STATUS=0, ITERS=1, OBJ=2, KKT=3, MU=4, DUAL_INF=5, CONSTR_VIOL=6, COMPL=7, N_REG=8, N_SOC=9, N_BACKTRACK=10, N_RESTO=11, N_WATCHDOG=12, COUNT=13)

_lib = None


def build(force=False):
    "Compile oracle/libms_oracle.so with gcc (no-op when up to date)."

    so = HERE / 'libms_oracle.so'
    src = [HERE / 'ms_oracle.c', HERE / 'ms_oracle.h']

    # content hash, not file times: those do not survive the copy to the GPU box in any useful order
    import hashlib
    digest = hashlib.sha256(b''.join(s.read_bytes() for s in src + [HERE / 'Makefile'] if s.exists())).hexdigest()
    stamp = HERE / 'libms_oracle.so.stamp'

    if force or not so.exists() or not stamp.exists() or stamp.read_text().strip() != digest:
        subprocess.run(['make', '-B', '-C', str(HERE), 'libms_oracle.so'], check=True, capture_output=True)
        stamp.write_text(digest)

    return so


def lib():

    global _lib

    if _lib is None:

        so = HERE / 'libms_oracle.so'

        if os.environ.get('MS_ORACLE_LIB'):      # an instrumented build of the same sources (debugging sessions)
            so = Path(os.environ['MS_ORACLE_LIB'])
        elif not so.exists():
            build()

        _lib = ctypes.CDLL(str(so))

        dptr = ctypes.POINTER(ctypes.c_double)
        iptr = ctypes.POINTER(ctypes.c_int)

        _lib.oracle_solve.restype = ctypes.c_int
        _lib.oracle_solve.argtypes = [iptr, dptr, dptr, dptr, dptr, dptr, dptr, dptr, dptr, dptr, ctypes.c_int]
        _lib.oracle_solve_batch.restype = ctypes.c_int
        _lib.oracle_solve_warm.argtypes = [iptr, dptr, dptr, dptr, dptr, dptr, dptr, ctypes.c_double, ctypes.c_double, dptr, dptr, dptr, dptr,
                                           ctypes.c_int]
        _lib.oracle_solve_warm.restype = ctypes.c_int
        _lib.oracle_solve_dual.argtypes = [iptr, dptr, dptr, dptr, dptr, dptr, ctypes.c_int, dptr, dptr, ctypes.c_double, ctypes.c_double, dptr, dptr, dptr, dptr]
        _lib.oracle_solve_dual.restype = ctypes.c_int
        _lib.oracle_solve_start.argtypes = [iptr, dptr, dptr, dptr, dptr, dptr, ctypes.c_int, dptr, dptr, dptr, dptr, ctypes.c_int]
        _lib.oracle_solve_start.restype = ctypes.c_int
        _lib.oracle_solve_batch_start.argtypes = [iptr, dptr, dptr, dptr, dptr, dptr, ctypes.c_int, ctypes.c_int, dptr, dptr, dptr, ctypes.c_int]
        _lib.oracle_solve_batch_start.restype = ctypes.c_int
        _lib.oracle_solve_batch.argtypes = [iptr, dptr, dptr, dptr, dptr, dptr, ctypes.c_int, dptr, dptr, dptr, ctypes.c_int]
        _lib.oracle_stage_eval.restype = None
        _lib.oracle_stage_eval.argtypes = [iptr, dptr] + [ctypes.c_double]*5 + [dptr]
        _lib.oracle_set_loss_table.restype = None
        _lib.oracle_set_loss_table.argtypes = [dptr]
        _lib.oracle_set_collocation.restype = None
        _lib.oracle_set_collocation.argtypes = [ctypes.c_int, dptr, dptr]
        _lib.oracle_loss_rows.restype = None
        _lib.oracle_loss_rows.argtypes = [dptr, ctypes.c_double, ctypes.c_double, dptr]
        _lib.oracle_nlp_eval.restype = None
        _lib.oracle_nlp_eval.argtypes = [iptr, dptr, dptr, dptr, dptr, dptr, dptr, dptr]

    return _lib


def _d(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _i(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int))


_loss_keepalive = None


def set_loss_table(block):
    "Install the parameter block of the dynamic loss model (kept alive here; one model at a time)."

    global _loss_keepalive
    _loss_keepalive = np.ascontiguousarray(block, dtype=np.float64)
    lib().oracle_set_loss_table(_d(_loss_keepalive))


def set_collocation(C, D):
    "Install the tables of the collocation integrator: C (d+1, d+1), D (d+1) on the points {0} + collocation points (copied)."

    C = np.ascontiguousarray(C, dtype=np.float64)
    D = np.ascontiguousarray(D, dtype=np.float64)
    lib().oracle_set_collocation(len(D) - 1, _d(C), _d(D))


INTEGRATOR = dict(RK=0, IRK=1, CVODES=2)


def loss_rows(block, f, v):
    "(2, 6): g, g_f, g_v, g_ff, g_fv, g_vv of the traction and the regenerative-brake loss row at (f [N/kg], v [m/s])"

    block = np.ascontiguousarray(block, dtype=np.float64)
    out = np.zeros(12)
    lib().oracle_loss_rows(_d(block), float(f), float(v), _d(out))
    return out.reshape(2, 6)


class Problem():
    "Flat NLP data of one (train, track, options) triple."

    def __init__(self, ip, dp, ds, grad, curv, bmax, vlim0, vlimN, totalMass, positions):
        self.ip = np.ascontiguousarray(ip, dtype=np.int32)
        self.dp = np.ascontiguousarray(dp, dtype=np.float64)
        self.ds = np.ascontiguousarray(ds, dtype=np.float64)
        self.grad = np.ascontiguousarray(grad, dtype=np.float64)
        self.curv = np.ascontiguousarray(curv, dtype=np.float64)
        self.bmax = np.ascontiguousarray(bmax, dtype=np.float64)
        self.vlim0, self.vlimN = vlim0, vlimN
        self.totalMass = totalMass
        self.positions = positions

    @property
    def N(self):
        return int(self.ip[IP['N']])

    @property
    def withPn(self):
        return int(self.ip[IP['WITH_PN']])

    @property
    def nz(self):
        return (4 + self.withPn)*self.N + 2

    @property
    def rowsPerInterval(self):
        return (2 if self.ip[IP['HAS_POWER']] else 0) + 3 + (2 if self.ip[IP['ENERGY_OPT']] else 0)

    def scenario(self, terminalTime, initialTime=0.0, terminalVelocity=1.0, initialVelocity=1.0):
        "dp with the four real-time parameters substituted (ocp.py:343-355)."

        vmin = np.sqrt(self.dp[DP['VMIN_SQ']])
        v0 = min(max(initialVelocity, vmin), self.vlim0)
        vN = min(max(terminalVelocity, vmin), self.vlimN)

        dp = self.dp.copy()
        dp[DP['T0']], dp[DP['TEND']], dp[DP['V0SQ']], dp[DP['VNSQ']] = initialTime, terminalTime, v0**2, vN**2

        return dp


def pack_problem(train, points, opts, lossKind, ct, cr, trackLength, tol=1e-8):
    """
    (train attribute bag, grid frame, option dict) -> Problem.
    `points`: DataFrame of computeDiscretizationPoints (index = positions).
    `opts`: dict with numIntervals, maxIterations, energyOptimal, minimumVelocity, numSteps, numApproxSteps and, for the other two
    integrators of train.py:303-322, integrationMethod ('RK', 'IRK', 'CVODES'), order, maxIter, absTol, relTol (the collocation
    tables go in through set_collocation).
    """

    N = int(opts['numIntervals'])
    pos = points.index.values.astype(float)

    assert len(pos) == N + 1

    totalMass = train.mass*train.rho   # ocp.py:97

    withRg = train.forceMin != 0       # ocp.py:101-102
    withPn = train.forceMinPn != 0

    accInf = 10.0                      # ocp.py:104

    fmax = train.forceMax/totalMass if train.forceMax is not None else accInf
    fminRg = train.forceMin/totalMass if train.forceMin is not None else -accInf
    fminPn = train.forceMinPn/totalMass if train.forceMinPn is not None else -accInf

    pmax = train.powerMax/totalMass if train.powerMax is not None else None
    pmin = train.powerMin/totalMass if train.powerMin is not None else None

    accMax = min(accInf, train.accMax if train.accMax is not None else accInf)          # ocp.py:113
    accMin = max(-accInf, -abs(train.accMin) if train.accMin is not None else -accInf)  # ocp.py:114

    hasPower = pmax is not None or pmin is not None   # ocp.py:184

    if hasPower:
        upper = pmax if pmax is not None else fmax*train.velocityMax                                        # ocp.py:186
        lower = 0 if not withRg else (pmin if pmin is not None else fminRg*train.velocityMax)               # ocp.py:187
    else:
        upper = lower = 0.0

    vlim = points['Speed limit [m/s]'].values.astype(float)

    bmax = np.zeros(N + 1)

    for i in range(1, N):
        bmax[i] = min(vlim[i], train.velocityMax, vlim[i - 1])**2   # ocp.py:266-269

    energyOptimal = bool(opts['energyOptimal'])

    objDen = 3.6/(1e-6*totalMass) if energyOptimal else trackLength/train.velocityMax   # ocp.py:278,282

    ip = np.zeros(IP['COUNT'], dtype=np.int32)
    ip[IP['N']] = N
    ip[IP['WITH_PN']] = int(withPn)
    ip[IP['HAS_POWER']] = int(hasPower)
    ip[IP['ENERGY_OPT']] = int(energyOptimal)
    ip[IP['NUM_STEPS']] = int(opts.get('numSteps', 1))
    ip[IP['NUM_APPROX']] = int(opts.get('numApproxSteps', 0))
    ip[IP['LOSS_KIND']] = int(lossKind)
    ip[IP['MAX_ITER']] = int(opts.get('maxIterations', 1000))
    ip[IP['INTEGRATOR']] = INTEGRATOR[opts.get('integrationMethod', 'RK')]
    ip[IP['COLL_DEGREE']] = int(opts.get('order', 0)) if ip[IP['INTEGRATOR']] == 1 else 0
    ip[IP['NEWTON_ITERS']] = int(opts.get('maxIter', 10))
    ip[IP['INTEGRATE_LOSSES']] = int(bool(opts.get('integrateLosses', False)))
    ip[IP['WATCHDOG_TRIGGER']] = int(opts.get('watchdogTrigger', 0))      # IPOPT's watchdog_shortened_iter_trigger (0: its default, 10)
    if ip[IP['INTEGRATOR']] == 2:
        ip[IP['NUM_APPROX']] = 0     # train.py:314

    dp = np.zeros(DP['COUNT'])
    dp[DP['SR0']], dp[DP['SR1']], dp[DP['SR2']] = train.r0/totalMass, train.r1/totalMass, train.r2/totalMass
    dp[DP['G']], dp[DP['RHO']] = train.g, train.rho
    dp[DP['FMAX']] = fmax
    dp[DP['FMIN']] = fminRg if withRg else 0.0     # ocp.py:175
    dp[DP['FMIN_PN']] = fminPn
    dp[DP['PW_UPPER']], dp[DP['PW_LOWER']] = abs(upper), abs(lower)
    dp[DP['ACC_MIN']], dp[DP['ACC_MAX']] = accMin, accMax
    dp[DP['LOSS_CT']], dp[DP['LOSS_CR']] = ct, cr
    dp[DP['VMIN_SQ']] = float(opts.get('minimumVelocity', 1))**2
    dp[DP['OBJ_DEN']] = objDen
    dp[DP['TOL']] = tol
    dp[DP['INT_ATOL']], dp[DP['INT_RTOL']] = float(opts.get('absTol', 1e-8)), float(opts.get('relTol', 1e-6))

    grad = points['Gradient [permil]'].values[:N].astype(float)/1e3   # ocp.py:195
    curv = points['Curvature [1/m]'].values[:N].astype(float)         # ocp.py:196

    return Problem(ip, dp, np.diff(pos), grad, curv, bmax, float(vlim[0]), float(vlim[-1]), totalMass, pos)


START = dict(reference=0, profile=1)


def solve(prob, dp, history=False, guess=None, mu0=1e-3, push=1e-3, start='reference'):
    """
    One solve -> dict(z, lam_g, stats[, hist]).  guess (nz,): primal warm start with barrier parameter mu0 and interior
    push; otherwise start = 'reference' (cold start of ocp.py:325-339) or 'profile' (see ms_oracle.c).
    """

    L = lib()
    z = np.zeros(prob.nz)
    lam = np.zeros(prob.rowsPerInterval*prob.N)
    st = np.zeros(ST['COUNT'])
    cap = int(prob.ip[IP['MAX_ITER']]) + 2 if history else 0
    hist = np.zeros((max(cap, 1), 8))
    dp = np.ascontiguousarray(dp, dtype=np.float64)

    if guess is not None:
        guess = np.ascontiguousarray(guess, dtype=np.float64)
        assert guess.shape == (prob.nz,)

    if guess is None:
        L.oracle_solve_start(_i(prob.ip), _d(dp), _d(prob.ds), _d(prob.grad), _d(prob.curv), _d(prob.bmax), START[start], _d(z), _d(lam),
                             _d(st), _d(hist) if history else None, cap)
    else:
        L.oracle_solve_warm(_i(prob.ip), _d(dp), _d(prob.ds), _d(prob.grad), _d(prob.curv), _d(prob.bmax), _d(guess), float(mu0),
                            float(push), _d(z), _d(lam), _d(st), _d(hist) if history else None, cap)

    out = dict(z=z, lam_g=lam, stats={k: st[v] for k, v in ST.items() if k != 'COUNT'})

    if history:
        out['hist'] = hist[:int(st[ST['ITERS']]) + 1]

    return out


DUAL_STRIDE = 27     # OR_DUAL_STRIDE


def solve_dual(prob, dp, guess=None, duals=None, mu0=1e-4, push=1e-3, start='reference'):
    """
    One solve that records its multipliers ('duals': (N + 1, 27)) and, with `guess` (nz,) and `duals`, starts from both
    (primal-dual warm start).  Without a guess it is a plain solve from `start`.
    """

    L = lib()
    z = np.zeros(prob.nz)
    lam = np.zeros(prob.rowsPerInterval*prob.N)
    st = np.zeros(ST['COUNT'])
    out = np.zeros((prob.N + 1, DUAL_STRIDE))
    dp = np.ascontiguousarray(dp, dtype=np.float64)
    g = None if guess is None else np.ascontiguousarray(guess, dtype=np.float64)
    d = None if duals is None else np.ascontiguousarray(duals, dtype=np.float64)
    assert g is None or g.shape == (prob.nz,)
    assert d is None or d.shape == (prob.N + 1, DUAL_STRIDE)
    L.oracle_solve_dual(_i(prob.ip), _d(dp), _d(prob.ds), _d(prob.grad), _d(prob.curv), _d(prob.bmax), START[start],
                        _d(g) if g is not None else None, _d(d) if d is not None else None, float(mu0), float(push), _d(z), _d(lam), _d(out), _d(st))
    return dict(z=z, lam_g=lam, duals=out, stats={k: st[v] for k, v in ST.items() if k != 'COUNT'})


def solve_batch(prob, scen, nthreads=0, start='reference'):
    "scen: (B,4) array of (t0, T, v0sq, vNsq).  Returns (z (B,nz), stats (B,ST_COUNT), nfail)."

    L = lib()
    scen = np.ascontiguousarray(scen, dtype=np.float64)
    B = scen.shape[0]
    z = np.zeros((B, prob.nz))
    st = np.zeros((B, ST['COUNT']))

    nfail = L.oracle_solve_batch_start(_i(prob.ip), _d(prob.dp), _d(prob.ds), _d(prob.grad), _d(prob.curv), _d(prob.bmax), START[start], B,
                                       _d(scen), _d(z), _d(st), int(nthreads))

    return z, st, nfail


def max_shortened_run(reset=True):
    "Longest run of successive shortened steps of the solves since the last reset (what IPOPT's watchdog trigger of 10 looks at)."

    L = lib()
    L.oracle_max_shortened_run.restype = ctypes.c_int
    L.oracle_max_shortened_run.argtypes = [ctypes.c_int]

    return int(L.oracle_max_shortened_run(1 if reset else 0))


def set_watchdog(on):
    "IPOPT's watchdog procedure on (default) / off."

    lib().oracle_set_watchdog(1 if on else 0)


def watchdog_counts(reset=True):
    "(started, ended by an accepted trial point) over the solves since the last reset."

    a, b = ctypes.c_int(0), ctypes.c_int(0)
    lib().oracle_watchdog_counts(ctypes.byref(a), ctypes.byref(b), 1 if reset else 0)

    return a.value, b.value


def watchdog_forced_steps(reset=True):
    "Trial points a running watchdog procedure took without the filter's consent, over the solves since the last reset."

    return int(lib().oracle_watchdog_forced_steps(1 if reset else 0))


def stage_eval(prob_or_ipdp, b, w, ds, grad=0.0, curv=0.0):
    "tau, bplus and their first/second derivatives wrt (b, w) for one interval."

    L = lib()
    ip, dp = (prob_or_ipdp.ip, prob_or_ipdp.dp) if isinstance(prob_or_ipdp, Problem) else prob_or_ipdp
    out = np.zeros(12)
    L.oracle_stage_eval(_i(np.ascontiguousarray(ip, dtype=np.int32)), _d(np.ascontiguousarray(dp, dtype=np.float64)),
                        float(b), float(w), float(ds), float(grad), float(curv), _d(out))

    return out


def nlp_eval(prob, dp, z):
    "Objective and the constraint rows (reference order) at z."

    L = lib()
    obj = np.zeros(1)
    g = np.zeros(prob.rowsPerInterval*prob.N)
    z = np.ascontiguousarray(z, dtype=np.float64)
    dp = np.ascontiguousarray(dp, dtype=np.float64)
    L.oracle_nlp_eval(_i(prob.ip), _d(dp), _d(prob.ds), _d(prob.grad), _d(prob.curv), _d(z), _d(obj), _d(g))

    return float(obj[0]), g
