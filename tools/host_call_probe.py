"""Diagnostic (GPU): where the wall time of one msd_solve_batch call goes at config-1 sizes -- device-resident launch + sync, the C entry point with page-locked
and with pageable result arrays, the Python wrapper.   usage: host_call_probe.py [B ...]"""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('ms-eetc_amd', '', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
from mseetc import workloads as wl, _device
from mseetc.ocp import casadiSolver
from mseetc._device import ST, _d
L = _device.lib()
train, track, N = wl.config('c1')
for B in [int(a) for a in sys.argv[1:]] or [1024, 8192]:
    s = casadiSolver(train, track, wl.options(N))
    pr = s.problem
    scen = s._scenarios(wl.c1_times(B), 0, 1, 1)
    def timeit(f, k=20):
        for _ in range(3): f()
        t0 = time.perf_counter()
        for _ in range(k): f()
        return 1e3*(time.perf_counter() - t0)/k
    # (a) device-resident
    dev = pr.upload(scen) if hasattr(pr, 'upload') else None
    ms = ctypes.c_float(0)
    z_pin = pr._results.empty('zz', (B, pr.nz)); st_pin = pr._results.empty('ss', (B, ST['COUNT']))
    z_pg = np.empty((B, pr.nz)); st_pg = np.empty((B, ST['COUNT']))
    t_pin = timeit(lambda: L.msd_solve_batch(pr._h, B, _d(scen), _d(z_pin), None, _d(st_pin), ctypes.byref(ms)))
    k_ms = ms.value
    t_pg = timeit(lambda: L.msd_solve_batch(pr._h, B, _d(scen), _d(z_pg), None, _d(st_pg), ctypes.byref(ms)))
    t_py = timeit(lambda: pr.solve_batch(scen))
    hold = [None]
    def held():
        hold[0] = pr.solve_batch(scen)
    t_py_held = timeit(held)
    print('B', B, 'kernel_ms %.3f' % k_ms, '| C entry, page-locked results %.3f ms' % t_pin, '| pageable results %.3f ms' % t_pg, '| python wrapper %.3f ms' % t_py, '| wrapper, result held %.3f ms' % t_py_held, flush=True)
    s.close()
