import sys, time
sys.path[:0]=['/root/repo','/root/repo/ms-eetc_amd']
import numpy as np
import bench
for rep in range(2):
  for B in (1024, 8192):
    solver, scen, ov, text = bench.build_workload('c1', B, 0, 0, 'profile', 0, 'rk')
    for _ in range(2): solver.problem.solve_batch(scen)
    t0=time.perf_counter()
    ts=[]
    for _ in range(10):
        t1=time.perf_counter(); r = solver.problem.solve_batch(scen); ts.append(1e3*(time.perf_counter()-t1))
    dt=time.perf_counter()-t0
    print(rep, B, 'ms_per_call %.3f' % (1e3*dt/10), 'kernel_ms %.3f' % r['kernel_ms'], 'calls', ' '.join('%.2f'%t for t in ts), scen.flags['C_CONTIGUOUS'], scen.dtype, flush=True)
    solver.close()
