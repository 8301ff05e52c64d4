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


class MsanProxy:
    """
    Stands in for the emulation library when MSD_EMU_MSAN names the MemorySanitizer build of the emulation (tests/hip_emu/build_msan.sh: an executable --
    MSan needs an instrumented main program, the interpreter is not one).  Every emu_solve_batch* call is written to a case file, run by that program
    (tests/hip_emu/emu_msan_main.cpp) and read back into the caller's arrays; a report of MemorySanitizer (exit code 77) fails the calling test with the
    report as the message.  The emulation tests run unchanged (tests/hip_emu/run_msan.sh).
    """

    DUAL_STRIDE = 27      # MSD_DUAL_STRIDE

    class _Fn:
        def __init__(self, f):
            self.f = f

        def __call__(self, *a):
            return self.f(*a)

    def __init__(self, exe):
        self.exe = str(exe)
        self.dual_in = self.dual_out = None
        self.dual_stride = 0
        self.emu_solve_batch = self._Fn(lambda desc, nscen, scen, z, lam, st, hist, cap: self._run(desc, nscen, scen, None, None, 0.0, 0.0, z, lam, st, hist, cap))
        self.emu_solve_batch_warm = self._Fn(self._run)
        self.emu_set_duals = self._Fn(self._set_duals)

    @staticmethod
    def _addr(p):
        return None if p is None else ctypes.cast(p, ctypes.c_void_p).value

    @classmethod
    def _view(cls, p, n):
        a = cls._addr(p)
        return None if not a else np.ctypeslib.as_array((ctypes.c_double*n).from_address(a))

    def _set_duals(self, dual_in, stride, dual_out):
        self.dual_in, self.dual_stride, self.dual_out = dual_in, int(stride), dual_out

    def _run(self, desc, nscen, scen, ovr, guess, mu0, push, z, lam, st, hist, cap):
        import os
        import tempfile
        d = desc._obj
        N = d.num_intervals
        nz = (4 + d.with_pn_brake)*N + 2
        rpi = (2 if d.has_power_rows else 0) + 3 + (2 if d.energy_optimal else 0)
        rec = (N + 1)*self.DUAL_STRIDE
        loss = self._view(d.loss_table, d.loss_table_len) if d.loss_kind == 2 else None
        coll = self._view(d.coll_tables, (d.coll_degree + 1)**2 + d.coll_degree + 1) if d.integrator == 2 else None
        g = self._view(guess, nscen*nz)
        o = self._view(ovr, nscen*10)
        nrec_in = 0 if (self._addr(self.dual_in) is None or g is None) else (1 if self.dual_stride == 0 else nscen)
        din = None
        if nrec_in:
            span = rec if nrec_in == 1 else (nscen - 1)*self.dual_stride + rec
            raw = self._view(self.dual_in, span)
            din = np.concatenate([raw[k*self.dual_stride:k*self.dual_stride + rec] for k in range(nrec_in)])
        has_out = self._addr(self.dual_out) is not None
        hdr = np.array([0x4d53414e, ctypes.sizeof(d), nscen, cap, o is not None, g is not None, nrec_in, has_out, 0 if nrec_in <= 1 else rec,
                        0 if loss is None else loss.size, 0 if coll is None else coll.size, rec], dtype=np.int64)
        with tempfile.TemporaryDirectory() as tmp:
            case, res = os.path.join(tmp, 'case.bin'), os.path.join(tmp, 'result.bin')
            with open(case, 'wb') as f:
                f.write(hdr.tobytes()); f.write(np.array([mu0, push]).tobytes()); f.write(bytes(d))
                for a in (self._view(d.ds, N), self._view(d.grad, N), self._view(d.curv, N), self._view(d.bmax, N + 1), loss, coll,
                          self._view(scen, nscen*4), o, g, din):
                    if a is not None:
                        f.write(np.ascontiguousarray(a, dtype=np.float64).tobytes())
            env = dict(os.environ, MSAN_OPTIONS='exit_code=77:halt_on_error=1')
            r = subprocess.run([self.exe, case, res], capture_output=True, text=True, env=env, timeout=3600)
            if r.returncode != 0:
                raise AssertionError("MemorySanitizer build of the emulation, exit code {}:\n{}".format(r.returncode, r.stderr[-6000:]))
            out = np.fromfile(res, dtype=np.float64)
        rc = int(out[:1].view(np.int64)[0])
        if rc != 0:
            return rc
        k = 1
        for p, n in ((z, nscen*nz), (lam, nscen*rpi*N), (st, nscen*16), (hist, cap*8), (self.dual_out if has_out else None, nscen*rec if has_out else 0)):
            v = self._view(p, n) if n else None
            if v is not None:
                v[:] = out[k:k + n]
            k += n
        return 0


def load_emulation():
    "Build (when out of date) and load the host emulation of the kernels."
    src = sorted(EMU.glob('*.cpp')) + sorted(EMU.glob('*.h')) + [EMU / 'build.sh', EMU / 'hip' / 'hip_runtime.h'] \
        + sorted((EMU.parent.parent / 'ms-eetc_amd' / 'csrc').glob('*.hpp')) + [EMU.parent.parent / 'include' / 'mseetc_hip.h']
    import os
    if os.environ.get('MSD_EMU_MSAN'):      # the MemorySanitizer build made by tests/hip_emu/run_msan.sh
        return MsanProxy(os.environ['MSD_EMU_MSAN'])
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


@pytest.mark.parametrize('N,crop,T', [(40, 16000, 804.9041795334854), (60, 30000, 1140.8291957305269), (100, None, 1541.0)])
def test_emulated_second_order_correction_inside_the_fused_iteration(emu, N, crop, T, monkeypatch):
    """
    The first-pass kernels with the second-order correction (W&B section 2.4; IPOPT's default, ocp.py:290) inside the fused iteration (Solver: SOCK,
    msd_kernels_full4.hip -- what the shrinking-horizon loop and a handle that has met corrections launch): the two solves of
    test_emulated_kernel_matches_oracle that take a correction, one node per lane and two, run through the first pass ALONE (no follow-up kernel: EMU_NO_FOLLOW)
    and follow the oracle -- iterations, number of corrections, point; the third case takes none and runs the 64 x 2 kernel with the node constants in LDS.
    """
    from mseetc.ocp import casadiSolver
    from mseetc._device import ST
    from oracle import oracle
    monkeypatch.setenv('EMU_SOCK', '1')
    monkeypatch.setenv('EMU_NO_FOLLOW', '1')
    train, track = cases.train_default(), (cases.track_00(crop) if crop else cases.track_00())
    solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=300, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='profile')
    scen = solver._scenarios(T, 0, 1, 1)
    nz = (4 + int(solver.withPnBrake))*N + 2
    z, lam, st, hist = np.zeros((1, nz)), np.zeros((1, 7*N)), np.full((1, ST['COUNT']), -77.0), np.zeros((8, 8))
    d = lambda a: a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))
    assert emu.emu_solve_batch(ctypes.byref(solver._desc), 1, d(scen), d(z), d(lam), d(st), d(hist), 8) == 0
    prob = cases.oracle_problem(train, track, N)
    ref = oracle.solve(prob, prob.scenario(T), start='profile')
    assert st[0, ST['STATUS']] == 0 and ref['stats']['STATUS'] == 0
    assert int(st[0, ST['ITERS']]) == int(ref['stats']['ITERS'])
    assert int(st[0, ST['N_SOC']]) == int(ref['stats']['N_SOC']) == (0 if N == 100 else 1)
    assert np.max(np.abs(z[0] - ref['z'])/np.maximum(1, np.abs(ref['z']))) < 1e-8
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


@pytest.mark.skipif(not __import__('os').environ.get('RUN_SANITIZERS'), reason="opt-in (an hour of CPU): RUN_SANITIZERS=1")
def test_emulation_under_memory_sanitizer():
    """
    MemorySanitizer (ROCm's clang) over the emulated kernels of every family -- fused first pass, follow-up kernels, restoration phase, watchdog procedure,
    streamed kernel, the other shooting integrators and loss models: no local, LDS word or work-area word is read before it is written where the value
    decides anything (round 5's Solver::evs was of that class and invisible to ASan, UBSan and the poisoned run).  tests/hip_emu/run_msan.sh must pass.
    """
    r = subprocess.run([str(EMU / 'run_msan.sh')], capture_output=True, text=True, timeout=4*3600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])


@pytest.mark.skipif(not __import__('os').environ.get('RUN_SANITIZERS'), reason="opt-in (twenty minutes of CPU): RUN_SANITIZERS=1")
def test_compilers_name_no_uninitialized_read_in_the_device_header():
    "g++ -O2 -Werror=maybe-uninitialized and clang -Werror=sometimes/conditional-uninitialized over every emulation unit: tests/hip_emu/check_uninitialized.sh"
    r = subprocess.run([str(EMU / 'check_uninitialized.sh')], capture_output=True, text=True, timeout=2*3600)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-3000:])
