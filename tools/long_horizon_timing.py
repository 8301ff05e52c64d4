"""Kernel time of single long-horizon solves (simulations/table3.py:34 sweeps numIntervals up to 5000): LDS-resident kernels up to 560, streamed above."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, R + '/ms-eetc_amd', R + '/tests']
import cases
from mseetc.ocp import casadiSolver
from mseetc import workloads as wl
for N in (50, 100, 200, 300, 500, 1000, 2000, 5000):
    s = casadiSolver(cases.train_fig10(), wl.track_00(), wl.options(N, maxIterations=1000))
    s.solveBatch([1541.0])
    r = s.solveBatch([1541.0])
    print('N = %4d: geometry %s, %2d iterations, %.2f ms, energy %.4f kWh, status %d' % (N, s.problem.geometry(), r['iterations'][0], r['kernel_ms'], r['cost'][0], r['status'][0]))
    s.close()
