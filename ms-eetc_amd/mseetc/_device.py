"""
ctypes binding of the HIP solver library (C ABI: include/mseetc_hip.h).

There is no CPU fallback: if the shared library is missing or no MI355X is visible,
every entry point raises.  The library is built in-tree by `__graft_entry__.build()`
(hipcc --offload-arch=gfx950) as ms-eetc_amd/lib/libmseetc_hip.so.
"""

import ctypes
import os
from pathlib import Path

import numpy as np

LIB_PATH = Path(os.environ.get('MSD_LIB', Path(__file__).resolve().parent.parent / 'lib' / 'libmseetc_hip.so'))

ABI_VERSION = 6
INTEGRATOR_ADAPTIVE, INTEGRATOR_COLLOCATION = 1, 2     # MSD_INTEGRATOR_* (also the methods of msd_interval_integrate)
ST = dict(STATUS=0, ITERS=1, OBJ=2, KKT=3, MU=4, DUAL_INF=5, CONSTR_VIOL=6, COMPL=7, N_REG=8, N_SOC=9, N_BACKTRACK=10, CYC_TOTAL=11, CYC_KKT=12, N_FALLBACK=13, N_RESTO=14, N_WATCHDOG=15, COUNT=16)
# (ITERS counts both attempts of a solve that was repeated from the other starting point: after a breakdown, or after the iteration limit behind a
#  restoration phase -- it can reach twice max_iterations then)
SC_COUNT = 4
OV = dict(SR0=0, SR1=1, SR2=2, F_MAX=3, F_MIN=4, F_MIN_PN=5, PW_UPPER=6, PW_LOWER=7, OBJ_DEN=8, TOTAL_MASS=9, COUNT=10)
HIST_COLS = 8

STATUS_MAXITER, STATUS_LINESEARCH, STATUS_INFEASIBLE = -1, -2, -6
# -2: the filter line search broke down and the feasibility restoration phase (csrc/msd_resto.hpp, IPOPT's MinC_1Nrm restoration) did not
# find an acceptable point either -- IPOPT's 'Restoration_Failed'.  -6: the restoration phase converged to a stationary point of the
# infeasibility (device), or the minimum-running-time certificate of a failed scenario says so (casadiSolver._classify_failures).
STATUS_TEXT = {0: 'Solve_Succeeded', 1: 'Solved_To_Acceptable_Level', -1: 'Maximum_Iterations_Exceeded',
               -2: 'Restoration_Failed', -3: 'Error_In_Step_Computation', -4: 'Invalid_Number_Detected',
               -5: 'Search_Direction_Becomes_Too_Small', -6: 'Infeasible_Problem_Detected'}

_dptr = ctypes.POINTER(ctypes.c_double)


class ProblemDesc(ctypes.Structure):
    "struct msd_problem_desc"

    _fields_ = [('abi_version', ctypes.c_int), ('num_intervals', ctypes.c_int), ('with_pn_brake', ctypes.c_int),
                ('has_power_rows', ctypes.c_int), ('energy_optimal', ctypes.c_int), ('num_steps', ctypes.c_int),
                ('num_approx_steps', ctypes.c_int), ('loss_kind', ctypes.c_int), ('max_iterations', ctypes.c_int),
                ('start_kind', ctypes.c_int), ('integrator', ctypes.c_int), ('coll_degree', ctypes.c_int), ('newton_iterations', ctypes.c_int),
                ('integrate_losses', ctypes.c_int), ('no_restoration', ctypes.c_int), ('watchdog_trigger', ctypes.c_int),
                ('sr0', ctypes.c_double), ('sr1', ctypes.c_double), ('sr2', ctypes.c_double), ('g', ctypes.c_double), ('rho', ctypes.c_double),
                ('f_max', ctypes.c_double), ('f_min', ctypes.c_double), ('f_min_pn', ctypes.c_double),
                ('pw_upper', ctypes.c_double), ('pw_lower', ctypes.c_double), ('acc_min', ctypes.c_double), ('acc_max', ctypes.c_double),
                ('loss_ct', ctypes.c_double), ('loss_cr', ctypes.c_double), ('vmin_sq', ctypes.c_double), ('obj_den', ctypes.c_double),
                ('tol', ctypes.c_double), ('int_abstol', ctypes.c_double), ('int_reltol', ctypes.c_double), ('reserved_d', ctypes.c_double*5),
                ('ds', _dptr), ('grad', _dptr), ('curv', _dptr), ('bmax', _dptr),
                ('loss_table', _dptr), ('loss_table_len', ctypes.c_int), ('reserved_tail', ctypes.c_int), ('coll_tables', _dptr)]


class MpcPlan(ctypes.Structure):
    "struct msd_mpc_plan (include/mseetc_mpc.h)"

    _fields_ = [('num_resolves', ctypes.c_int), ('stride', ctypes.c_int), ('problems', ctypes.POINTER(ProblemDesc)), ('twins', ctypes.POINTER(ProblemDesc)),
                ('vlim_first', _dptr), ('length', _dptr), ('tail', ctypes.POINTER(ctypes.c_ubyte)), ('vmin', ctypes.c_double), ('vmax_train', ctypes.c_double),
                ('terminal_velocity', ctypes.c_double), ('warm_start', ctypes.c_int), ('warm_mu', ctypes.c_double), ('warm_push', ctypes.c_double),
                ('noise', ctypes.c_double), ('relax_infeasible', ctypes.c_int), ('late_margin', ctypes.c_double)]


MPC = dict(T0=0, V0=1, T=2, STATUS=3, ITERS=4, OBJ=5, RELAXED=6, COUNT=7)      # MSD_MPC_*


class DeviceError(RuntimeError):
    pass


_lib = None


def lib():
    "Load the HIP library; raises DeviceError when it has not been built."

    global _lib

    if _lib is None:

        if not LIB_PATH.exists():
            raise DeviceError("HIP solver library {} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc, gfx950). There is no CPU fallback.".format(LIB_PATH))

        L = ctypes.CDLL(str(LIB_PATH))
        vp = ctypes.c_void_p

        L.msd_last_error.restype = ctypes.c_char_p
        L.msd_device_count.restype = ctypes.c_int
        L.msd_problem_create.argtypes = [ctypes.POINTER(ProblemDesc), ctypes.c_int, ctypes.POINTER(vp)]
        L.msd_problem_destroy.argtypes = [vp]
        L.msd_problem_reconfigure.argtypes = [vp, ctypes.POINTER(ProblemDesc)]
        L.msd_problem_nz.argtypes = [vp]
        L.msd_problem_rows_per_interval.argtypes = [vp]
        L.msd_solve_batch.argtypes = [vp, ctypes.c_int, _dptr, _dptr, _dptr, _dptr, ctypes.POINTER(ctypes.c_float)]
        L.msd_solve_batch_ex.argtypes = [vp, ctypes.c_int, _dptr, _dptr, _dptr, _dptr, _dptr, ctypes.POINTER(ctypes.c_float)]
        L.msd_solve_batch_warm.argtypes = [vp, ctypes.c_int, _dptr, _dptr, _dptr, ctypes.c_double, ctypes.c_double, _dptr, _dptr, _dptr,
                                           ctypes.POINTER(ctypes.c_float)]
        L.msd_interval_integrate.argtypes = [ctypes.c_int, ctypes.c_int, _dptr, ctypes.c_int, _dptr, ctypes.c_int] + [_dptr]*8 + [ctypes.POINTER(ctypes.c_int)]
        L.msd_interval_last_error.restype = ctypes.c_char_p
        L.msd_solve_batch_shifted.argtypes = [vp, ctypes.c_int, _dptr, _dptr, ctypes.c_int, ctypes.c_double, ctypes.c_double, _dptr, _dptr, _dptr,
                                              ctypes.POINTER(ctypes.c_float)]
        L.msd_problem_keep_duals.argtypes = [vp, ctypes.c_int]
        L.msd_problem_direct_results.argtypes = [vp, ctypes.c_int]
        L.msd_solve_batch_multi.argtypes = [ctypes.POINTER(vp), ctypes.c_int, ctypes.c_int, _dptr, _dptr, _dptr, ctypes.c_double, ctypes.c_double, _dptr, _dptr,
                                            _dptr, ctypes.POINTER(ctypes.c_float)]
        L.msd_solve_batch_device.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp]
        L.msd_solve_batch_device_ex.argtypes = [vp, ctypes.c_int, vp, vp, vp, vp, vp]
        L.msd_problem_geometry.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.msd_synchronize.argtypes = [vp]
        L.msd_problem_follow_counts.argtypes = [vp, ctypes.POINTER(ctypes.c_int), ctypes.c_int]
        L.msd_problem_time_first_pass.argtypes = [vp, ctypes.c_int]
        L.msd_problem_first_pass_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
        L.msd_device_alloc.argtypes = [vp, ctypes.c_ulonglong, ctypes.POINTER(vp)]
        L.msd_device_free.argtypes = [vp, vp]
        L.msd_copy_to_device.argtypes = [vp, vp, vp, ctypes.c_ulonglong]
        L.msd_copy_to_host.argtypes = [vp, vp, vp, ctypes.c_ulonglong]
        L.msd_timer_begin.argtypes = [vp]
        L.msd_timer_end.argtypes = [vp, ctypes.POINTER(ctypes.c_float)]
        L.msd_stage_eval.argtypes = [vp, ctypes.c_int, _dptr, _dptr, _dptr, _dptr, _dptr, _dptr]
        L.msd_set_history.argtypes = [vp, _dptr, ctypes.c_int]
        L.msd_post_last_error.restype = ctypes.c_char_p
        L.msd_resimulate.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, _dptr, _dptr, _dptr, _dptr, _dptr, _dptr, _dptr,
                                     ctypes.c_double, ctypes.c_double, _dptr, _dptr]
        L.msd_integrate_losses.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, _dptr, ctypes.c_int, ctypes.c_double, ctypes.c_double, _dptr, ctypes.c_int,
                                           _dptr, _dptr, _dptr, _dptr, _dptr, _dptr, ctypes.c_double, ctypes.c_double, _dptr, _dptr]

        L.msd_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
        L.msd_fastmath_probe.argtypes = [ctypes.c_int, ctypes.c_int] + [_dptr]*4
        L.msd_host_alloc.argtypes = [ctypes.c_ulonglong, ctypes.POINTER(vp)]
        L.msd_host_free.argtypes = [vp]
        L.msd_mpc_create.argtypes = [vp, vp, ctypes.POINTER(MpcPlan), ctypes.POINTER(vp)]
        L.msd_mpc_destroy.argtypes = [vp]
        L.msd_mpc_run.argtypes = [vp, ctypes.c_int, _dptr, ctypes.c_double, ctypes.c_double, _dptr, _dptr, _dptr, _dptr, ctypes.POINTER(ctypes.c_float)]
        L.msd_mpc_nz.argtypes = [vp, ctypes.c_int]

        _lib = L

    return _lib


def _check(rc):

    if rc != 0:
        msg = lib().msd_last_error().decode()
        if rc == -1:
            raise ValueError(msg)
        raise DeviceError("msd error {}: {}".format(rc, msg))


def _d(a):
    return a.ctypes.data_as(_dptr)


START = dict(reference=0, profile=1)   # MSD_START_*


def make_desc(N, withPn, hasPower, energyOptimal, numSteps, numApproxSteps, lossKind, maxIterations, sr, g, rho, fmax, fmin, fminPn,
              pwUpper, pwLower, accMin, accMax, ct, cr, vminSq, objDen, tol, ds, grad, curv, bmax, lossTable=None, start='reference',
              integrator=None, integrateLosses=False, restoration=True, watchdogTrigger=0):
    """
    Fill a ProblemDesc; the numpy arrays are kept alive on the returned object.  integrator: None ('RK'), ('CVODES', absTol, relTol) or
    ('IRK', order, maxIter, C, D) with the tables of mseetc.train.collocationTables.  integrateLosses: ocp.py:28,231-241.
    """

    d = ProblemDesc()
    d.abi_version = ABI_VERSION
    d.num_intervals, d.with_pn_brake, d.has_power_rows, d.energy_optimal = int(N), int(withPn), int(hasPower), int(energyOptimal)
    d.num_steps, d.num_approx_steps, d.loss_kind, d.max_iterations = int(numSteps), int(numApproxSteps), int(lossKind), int(maxIterations)
    d.start_kind = START[start]
    d.no_restoration = 0 if restoration else 1
    d.watchdog_trigger = int(watchdogTrigger)      # 0: IPOPT's default (10 shortened iterations in a row)
    d.sr0, d.sr1, d.sr2 = sr
    d.g, d.rho = g, rho
    d.f_max, d.f_min, d.f_min_pn = fmax, fmin, fminPn
    d.pw_upper, d.pw_lower, d.acc_min, d.acc_max = pwUpper, pwLower, accMin, accMax
    d.loss_ct, d.loss_cr, d.vmin_sq, d.obj_den, d.tol = ct, cr, vminSq, objDen, tol
    keep = [np.ascontiguousarray(a, dtype=np.float64) for a in (ds, grad, curv, bmax)]
    d.ds, d.grad, d.curv, d.bmax = [_d(a) for a in keep]
    if lossTable is not None:
        keep.append(np.ascontiguousarray(lossTable, dtype=np.float64))
        d.loss_table, d.loss_table_len = _d(keep[-1]), len(keep[-1])
    if integrator is not None and integrator[0] == 'CVODES':
        d.integrator, d.int_abstol, d.int_reltol = INTEGRATOR_ADAPTIVE, float(integrator[1]), float(integrator[2])
    elif integrator is not None and integrator[0] == 'IRK':
        d.integrator, d.coll_degree, d.newton_iterations = INTEGRATOR_COLLOCATION, int(integrator[1]), int(integrator[2])
        keep.append(np.concatenate([np.asarray(integrator[3], dtype=np.float64).ravel(), np.asarray(integrator[4], dtype=np.float64).ravel()]))
        d.coll_tables = _d(keep[-1])
    elif integrator is not None:
        raise ValueError("Unknown integration method!")
    d.integrate_losses = int(bool(integrateLosses))
    d._keep = keep

    return d


class _Pinned():
    "Page-locked host memory (msd_host_alloc), freed with the object."

    def __init__(self, nbytes):
        self.ptr, self.nbytes = ctypes.c_void_p(), int(nbytes)
        _check(lib().msd_host_alloc(self.nbytes, ctypes.byref(self.ptr)))

    def __del__(self):
        try:
            if self.ptr:
                lib().msd_host_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


class _ResultArrays():
    """
    Result arrays of the host-buffer entry points in page-locked memory: a device-to-host copy into them runs at the link's rate without
    staging or page faults.  A buffer is handed out again once no array or view of its previous use is alive (reference count of the
    ctypes object every such array has as its base).  Up to POOL buffers per role are kept, so that a caller who still holds the results of
    the previous call -- `r = solve_batch(...)` in a loop -- alternates between two buffers instead of page-locking a new one per call
    (hipHostMalloc of the 34 MB of an 8192-scenario batch costs more than the copy it speeds up).
    """

    POOL = 3

    def __init__(self):
        self._cache = {}

    def empty(self, role, shape):
        "An array of `shape` in page-locked memory, NOT initialised: every element is written by the device-to-host copy that follows."
        import sys
        count = int(np.prod(shape))
        if count == 0:
            return np.zeros(shape)
        pool = self._cache.setdefault(role, [])
        pool[:] = [e for e in pool if len(e) == count]      # (another batch size: the old buffers go)
        carr = None
        for k in range(len(pool)):
            if sys.getrefcount(pool[k]) == 2:      # (the pool and getrefcount's own argument: no array or view of an earlier use is alive)
                carr = pool[k]
                break
        if carr is None:
            buf = _Pinned(8*count)
            carr = (ctypes.c_double*count).from_address(buf.ptr.value)
            carr._owner = buf
            if len(pool) >= self.POOL:
                pool.pop(0)      # (still referenced by the caller's arrays: freed when those go)
            pool.append(carr)
        return np.frombuffer(carr, dtype=np.float64, count=count).reshape(shape)      # (every element is written by the copy that follows)


class DeviceProblem():
    "Owner of an msd_handle."

    def __init__(self, desc, device=0):

        L = lib()
        self._h = ctypes.c_void_p()
        self.desc = desc
        _check(L.msd_problem_create(ctypes.byref(desc), int(device), ctypes.byref(self._h)))
        self.N = desc.num_intervals
        self.nz = L.msd_problem_nz(self._h)
        self.rowsPerInterval = L.msd_problem_rows_per_interval(self._h)
        self.device = device
        self._results = _ResultArrays()
        self.direct_results(True)      # (the result arrays below are page-locked: the kernels store z* there themselves)

    def reconfigure(self, desc):
        "Load another problem into this handle (stream and device buffers are kept): msd_problem_reconfigure."

        L = lib()
        _check(L.msd_problem_reconfigure(self._h, ctypes.byref(desc)))
        self.desc = desc
        self.N = desc.num_intervals
        self.nz = L.msd_problem_nz(self._h)
        self.rowsPerInterval = L.msd_problem_rows_per_interval(self._h)

        return self

    def close(self):

        if getattr(self, '_h', None):
            # (msd_problem_destroy refuses while a receding-horizon loop -- msd_mpc_create -- still runs on the handle's stream: the handle is kept then,
            #  close the loop first; silently dropping it would leak the device buffers and the stream)
            rc = lib().msd_problem_destroy(self._h)
            if rc != 0:
                raise DeviceError("msd_problem_destroy: " + (lib().msd_last_error() or b'').decode())
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def solve_batch(self, scen, want_multipliers=False, history=0, overrides=None, guess=None, warmMu=1e-2, warmPush=1e-3, shift=None):
        """
        scen: (B,4) host array (t0, T, v0sq, vNsq), overrides: optional (B, OV['COUNT']), guess: optional (B, nz) primal
        warm start (barrier parameter warmMu, interior push warmPush) -> dict(z, stats, lam_g, kernel_ms[, hist])
        """

        L = lib()
        scen = np.ascontiguousarray(scen, dtype=np.float64).reshape(-1, SC_COUNT)
        B = scen.shape[0]
        z = self._results.empty('z', (B, self.nz))
        st = self._results.empty('st', (B, ST['COUNT']))
        lam = self._results.empty('lam', (B, self.rowsPerInterval*self.N)) if want_multipliers else None
        ms = ctypes.c_float(0)
        hist = None

        if history:
            hist = np.zeros((int(history), HIST_COLS))
            _check(L.msd_set_history(self._h, _d(hist), int(history)))

        if overrides is not None:
            overrides = np.ascontiguousarray(overrides, dtype=np.float64).reshape(B, OV['COUNT'])

        if guess is not None:
            guess = np.ascontiguousarray(guess, dtype=np.float64).reshape(B, self.nz)

        if shift is not None:
            # warm start from this handle's previous solve, `shift` intervals down the horizon, without leaving the device
            if guess is not None:
                raise ValueError("Give either a guess or a shift!")
            _check(L.msd_solve_batch_shifted(self._h, B, _d(scen), _d(overrides) if overrides is not None else None, int(shift), float(warmMu),
                                             float(warmPush), _d(z), _d(lam) if lam is not None else None, _d(st), ctypes.byref(ms)))
        else:
            _check(L.msd_solve_batch_warm(self._h, B, _d(scen), _d(overrides) if overrides is not None else None,
                                          _d(guess) if guess is not None else None, float(warmMu), float(warmPush), _d(z),
                                          _d(lam) if lam is not None else None, _d(st), ctypes.byref(ms)))

        out = dict(z=z, stats=st, lam_g=lam, kernel_ms=float(ms.value))

        if history:
            L.msd_set_history(self._h, None, 0)
            out['hist'] = hist[:int(st[0, ST['ITERS']]) + 1]

        return out

    def solve_batch_multi(self, others, scen, want_multipliers=False, overrides=None, guess=None, warmMu=1e-2, warmPush=1e-3):
        """
        solve_batch over this handle and `others` (DeviceProblems of the same problem, normally on other devices): contiguous
        slices of the batch, one per handle, all in flight at once (msd_solve_batch_multi).
        """

        L = lib()
        scen = np.ascontiguousarray(scen, dtype=np.float64).reshape(-1, SC_COUNT)
        B = scen.shape[0]
        z = np.zeros((B, self.nz))
        st = np.zeros((B, ST['COUNT']))
        lam = np.zeros((B, self.rowsPerInterval*self.N)) if want_multipliers else None
        ms = ctypes.c_float(0)

        if overrides is not None:
            overrides = np.ascontiguousarray(overrides, dtype=np.float64).reshape(B, OV['COUNT'])

        if guess is not None:
            guess = np.ascontiguousarray(guess, dtype=np.float64).reshape(B, self.nz)

        handles = (ctypes.c_void_p*(1 + len(others)))(self._h, *[o._h for o in others])
        _check(L.msd_solve_batch_multi(handles, len(handles), B, _d(scen), _d(overrides) if overrides is not None else None,
                                       _d(guess) if guess is not None else None, float(warmMu), float(warmPush), _d(z),
                                       _d(lam) if lam is not None else None, _d(st), ctypes.byref(ms)))

        return dict(z=z, stats=st, lam_g=lam, kernel_ms=float(ms.value))

    # ---- device-resident path (benchmark) ------------------------------------------------

    def alloc(self, nbytes):
        p = ctypes.c_void_p()
        _check(lib().msd_device_alloc(self._h, int(nbytes), ctypes.byref(p)))
        return p

    def free(self, p):
        _check(lib().msd_device_free(self._h, p))

    def to_device(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        _check(lib().msd_copy_to_device(self._h, dptr, arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes))

    def to_host(self, arr, dptr):
        _check(lib().msd_copy_to_host(self._h, arr.ctypes.data_as(ctypes.c_void_p), dptr, arr.nbytes))

    def solve_batch_device(self, B, d_scen, d_z, d_lam, d_stats, d_overrides=None):
        _check(lib().msd_solve_batch_device_ex(self._h, int(B), d_scen, d_overrides, d_z, d_lam, d_stats))

    def direct_results(self, on=True):
        """
        Let the kernels store z* (and the multipliers) of solve_batch in the page-locked result arrays themselves instead of copying them behind
        the launch (msd_problem_direct_results).  The handle then keeps no device copy: solve_batch(shift=...) right after such a solve fails.
        """

        _check(lib().msd_problem_direct_results(self._h, int(bool(on))))
        return self

    def keep_duals(self, on=True):
        "Record the multipliers of every solve on the device; shifted warm starts then start from them too (msd_problem_keep_duals)."
        _check(lib().msd_problem_keep_duals(self._h, int(bool(on))))
        return self

    def geometry(self):
        "(threads per scenario, shooting nodes per thread) of the launch"
        nt, spt = ctypes.c_int(0), ctypes.c_int(0)
        _check(lib().msd_problem_geometry(self._h, ctypes.byref(nt), ctypes.byref(spt)))
        return nt.value, spt.value

    def synchronize(self):
        _check(lib().msd_synchronize(self._h))

    def time_first_pass(self, on=True):
        "Record HIP events around the first kernel of every launch (msd_problem_time_first_pass)."
        _check(lib().msd_problem_time_first_pass(self._h, int(bool(on))))

    def first_pass_ms(self):
        "(mean duration of the first kernel over the last launches [ms], launches averaged) -- msd_problem_first_pass_ms"
        ms, n = ctypes.c_float(0), ctypes.c_int(0)
        _check(lib().msd_problem_first_pass_ms(self._h, ctypes.byref(ms), ctypes.byref(n)))
        return float(ms.value), int(n.value)

    def follow_counts(self):
        "Scenarios the first-pass kernel handed to the follow-up kernel so far: (total, by reason[7]) -- msd_problem_follow_counts."
        out = (ctypes.c_int*8)()
        _check(lib().msd_problem_follow_counts(self._h, out, 8))
        return int(out[0]), [int(v) for v in out[1:]]

    def timer_begin(self):
        _check(lib().msd_timer_begin(self._h))

    def timer_end(self):
        ms = ctypes.c_float(0)
        _check(lib().msd_timer_end(self._h, ctypes.byref(ms)))
        return float(ms.value)

    def stage_eval(self, b, w, ds, grad, curv):
        arrs = [np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64) for a in (b, w, ds, grad, curv)]
        n = len(arrs[0])
        out = np.zeros((n, 12))
        _check(lib().msd_stage_eval(self._h, n, *[_d(a) for a in arrs], _d(out)))
        return out


def stage_eval(model, optsRK, time, velocitySquared, ds, force, gradient, curvature):
    "TrainIntegrator.solve on the device for arrays of intervals (train.py:347-364)."

    n = len(np.atleast_1d(velocitySquared))
    one = np.ones(1)
    desc = make_desc(1, model.withPnBrake, False, True, optsRK.numSteps, optsRK.numApproxSteps, 0, 1, (model.sr0, model.sr1, model.sr2),
                     model.g, model.rho, 1.0, -1.0, -1.0, 0.0, 0.0, -1.0, 1.0, 0.0, 0.0, 1.0, 1.0, 1e-8, one, 0*one, 0*one, np.ones(2))
    prob = DeviceProblem(desc)

    try:
        out = prob.stage_eval(velocitySquared, force, ds, gradient, curvature)
    finally:
        prob.close()

    return {'time': np.atleast_1d(np.asarray(time, dtype=float)) + out[:, 0], 'velSquared': out[:, 1], 'sens': out}


def interval_integrate(model, method, params, time, velocitySquared, ds, force, gradient, curvature, device=0):
    """
    TrainIntegrator.solve with the adaptive ('CVODES') or the collocation ('IRK') integrator for arrays of intervals.
    method: 1 adaptive, params (abstol, reltol); 2 collocation, params (order, numSteps, numApproxSteps, maxIter, C.ravel(), D).
    Returns dict(time, velSquared, status).
    """

    arrs = [np.ascontiguousarray(np.atleast_1d(a), dtype=np.float64) for a in (time, velocitySquared, ds, force, gradient, curvature)]
    n = max(len(a) for a in arrs)
    arrs = [np.ascontiguousarray(np.broadcast_to(a, (n,))) for a in arrs]
    params = _c(params)
    train5 = _c([model.sr0, model.sr1, model.sr2, model.g, model.rho])
    t, b, st = np.zeros(n), np.zeros(n), np.zeros(n, dtype=np.int32)
    L = lib()
    rc = L.msd_interval_integrate(int(device), n, _d(train5), int(method), _d(params), len(params), *[_d(a) for a in arrs], _d(t), _d(b),
                                  st.ctypes.data_as(ctypes.POINTER(ctypes.c_int)))
    if rc != 0:
        msg = L.msd_interval_last_error().decode()
        if rc == -1:
            raise ValueError(msg)
        raise DeviceError("msd integrator error {}: {}".format(rc, msg))

    return {'time': t, 'velSquared': b, 'status': st}


def fastmath_probe(x, device=0):
    """frcp, fsqrt and the reciprocal square root of csrc/msd_fastmath.hpp on the operands x (test hook: msd_fastmath_probe)."""
    L = lib()
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = [np.empty_like(x) for _ in range(3)]
    if L.msd_fastmath_probe(int(device), x.size, _d(x), *[_d(a) for a in out]) != 0:
        raise DeviceError(L.msd_interval_last_error().decode())
    return out


def _check_post(rc):

    if rc != 0:
        msg = lib().msd_post_last_error().decode()
        if rc == -1:
            raise ValueError(msg)
        raise DeviceError("msd post-processing error {}: {}".format(rc, msg))


def _c(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a if shape is None else a.reshape(shape)


def resimulate(model, force, dts, grad, curv, s0, v0, abstol=1e-12, reltol=1e-14, device=0):
    """
    Time-domain re-simulation with accumulated errors (utils.py:164-194).  force, dts: (B, N) specific total force and interval
    durations; grad, curv: (N,); s0, v0: (B,).  Returns positions and velocities (B, N+1).
    """

    force, dts = _c(force), _c(dts)
    B, N = force.shape
    pos, vel = np.zeros((B, N + 1)), np.zeros((B, N + 1))
    train5 = _c([model.sr0, model.sr1, model.sr2, model.g, model.rho])
    _check_post(lib().msd_resimulate(int(device), B, N, _d(train5), _d(force), _d(dts), _d(_c(grad)), _d(_c(curv)), _d(_c(s0)), _d(_c(v0)),
                                     float(abstol), float(reltol), _d(pos), _d(vel)))
    return pos, vel


def integrate_losses(model, lossKind, ct, cr, lossTable, forceEl, forcePn, dts, grad, curv, vstart, abstol=1e-8, reltol=1e-6, device=0):
    "Energy [J/kg] lost in traction / regenerative braking over every interval (train.py:367-413); arrays (B, N)."

    forceEl, forcePn, dts, vstart = _c(forceEl), _c(forcePn), _c(dts), _c(vstart)
    B, N = forceEl.shape
    etr, ebr = np.zeros((B, N)), np.zeros((B, N))
    train5 = _c([model.sr0, model.sr1, model.sr2, model.g, model.rho])
    tab = _c(lossTable) if lossTable is not None else None
    _check_post(lib().msd_integrate_losses(int(device), B, N, _d(train5), int(lossKind), float(ct), float(cr), _d(tab) if tab is not None else None,
                                           len(tab) if tab is not None else 0, _d(forceEl), _d(forcePn), _d(dts), _d(_c(grad)), _d(_c(curv)), _d(vstart),
                                           float(abstol), float(reltol), _d(etr), _d(ebr)))
    return etr, ebr


class DeviceLoop():
    """
    Owner of an msd_mpc_handle: the shrinking-horizon loop with its bookkeeping on the device (include/mseetc_mpc.h).  `problem` / `twin`:
    DeviceProblem of the first re-solve's energy problem / time-optimal twin (twin and twinDescs None: failed re-solves are left as they are).
    """

    def __init__(self, problem, twin, descs, twinDescs, stride, vlimFirst, lengths, tail, vmin, vmaxTrain, terminalVelocity, warmStart, warmMu, warmPush,
                 noise, relaxInfeasible, lateMargin):

        K = len(descs)
        self.K, self.problem, self.twin = K, problem, twin
        self._descs = (ProblemDesc*K)(*descs)
        self._twins = (ProblemDesc*K)(*twinDescs) if twinDescs is not None else None
        self._keep = [descs, twinDescs, np.ascontiguousarray(vlimFirst, dtype=np.float64), np.ascontiguousarray(lengths, dtype=np.float64),
                      np.ascontiguousarray(tail, dtype=np.uint8)]
        plan = MpcPlan()
        plan.num_resolves, plan.stride = K, int(stride)
        plan.problems = ctypes.cast(self._descs, ctypes.POINTER(ProblemDesc))
        plan.twins = ctypes.cast(self._twins, ctypes.POINTER(ProblemDesc)) if self._twins is not None else None
        plan.vlim_first, plan.length = _d(self._keep[2]), _d(self._keep[3])
        plan.tail = self._keep[4].ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte))
        plan.vmin, plan.vmax_train, plan.terminal_velocity = float(vmin), float(vmaxTrain), float(terminalVelocity)
        plan.warm_start, plan.warm_mu, plan.warm_push = int(bool(warmStart)), float(warmMu), float(warmPush)
        plan.noise, plan.relax_infeasible, plan.late_margin = float(noise), int(bool(relaxInfeasible)), float(lateMargin)
        self._m = ctypes.c_void_p()
        _check(lib().msd_mpc_create(problem._h, twin._h if twin is not None else None, ctypes.byref(plan), ctypes.byref(self._m)))
        self.nz = [lib().msd_mpc_nz(self._m, k) for k in range(K)]

    def run(self, T, initialTime, initialVelocity, n1, n2, keepZ=True):
        "One loop over the scenarios with arrival times T.  Returns (log (K, B, MPC['COUNT']), list of z (B, nz_k) or None, device ms of the loop)."

        T = np.ascontiguousarray(T, dtype=np.float64)
        B = T.shape[0]
        n1 = np.ascontiguousarray(n1, dtype=np.float64) if n1 is not None else None
        n2 = np.ascontiguousarray(n2, dtype=np.float64) if n2 is not None else None
        if self.K > 1 and n1 is not None and (n1.shape != (self.K - 1, B) or n2.shape != (self.K - 1, B)):
            raise ValueError("noise draws must have shape ({}, {})".format(self.K - 1, B))
        log = np.zeros((self.K, B, MPC['COUNT']))
        zflat = np.zeros(B*sum(self.nz)) if keepZ else None
        ms = ctypes.c_float(0)
        _check(lib().msd_mpc_run(self._m, B, _d(T), float(initialTime), float(initialVelocity), _d(n1) if n1 is not None else None,
                                 _d(n2) if n2 is not None else None, _d(log), _d(zflat) if keepZ else None, ctypes.byref(ms)))
        zs = None
        if keepZ:
            zs, off = [], 0
            for nz in self.nz:
                zs.append(zflat[off:off + B*nz].reshape(B, nz)); off += B*nz
        return log, zs, float(ms.value)

    def close(self):

        if getattr(self, '_m', None):
            lib().msd_mpc_destroy(self._m)
            self._m = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
