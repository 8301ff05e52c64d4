"""Diagnostic (GPU): what the result download of a config-1 batch costs, three ways -- (A) results stay on the device (msd_solve_batch_device),
(B) the kernels write z* straight into page-locked host memory (the same entry point, the host array's device address as d_z), (C) msd_solve_batch
(device-to-host copies behind the kernels).  Wall time per call in steady state; B and C must return the same bits."""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('ms-eetc_amd', '', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, torch, cases
from mseetc.ocp import casadiSolver
from mseetc._device import ST, lib, _Pinned

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
solver = casadiSolver(cases.train_default(), cases.track_00(), dict(numIntervals=100, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1)))
prob = solver.problem
scen = solver._scenarios(cases.c1_times(B), 0, 1, 1)
nz = prob.nz
hip = ctypes.CDLL('libamdhip64.so')
d_scen = torch.tensor(scen, device='cuda').contiguous()
d_z = torch.empty(B*nz, device='cuda', dtype=torch.float64)
d_st = torch.empty(B*ST['COUNT'], device='cuda', dtype=torch.float64)
torch.cuda.synchronize()
pin = _Pinned(8*B*nz)
dev = ctypes.c_void_p()
rc = hip.hipHostGetDevicePointer(ctypes.byref(dev), pin.ptr, 0)
print('hipHostGetDevicePointer rc', rc, hex(pin.ptr.value), hex(dev.value or 0))
hz = np.ctypeslib.as_array(ctypes.cast(pin.ptr, ctypes.POINTER(ctypes.c_double)), shape=(B, nz))


def timed(fn, n=30):
    for _ in range(5):
        fn()
    t = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    return 1e3*float(np.median(t))


def run_device():
    prob.solve_batch_device(B, d_scen.data_ptr(), d_z.data_ptr(), None, d_st.data_ptr()); prob.synchronize()


def run_zero_copy():
    prob.solve_batch_device(B, d_scen.data_ptr(), dev.value, None, d_st.data_ptr()); prob.synchronize()


a = timed(run_device)
hz[:] = 0
b = timed(run_zero_copy)
zb = hz.copy()
out = None


def run_host():
    global out
    out = prob.solve_batch(scen)


c = timed(run_host)
print('batch %d: device-resident %.4f ms   z* written to page-locked host memory by the kernels %.4f ms (%.3f x)   msd_solve_batch %.4f ms (%.3f x)'
      % (B, a, b, a/b, c, a/c))
print('bit-equal z*:', bool(np.array_equal(zb, out['z'])), ' converged', int((out['stats'][:, 0] == 0).sum()))
