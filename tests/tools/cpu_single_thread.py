"""Single-thread rate of the CPU oracle on config 1 (SURVEY.md 8d asks for it next to the all-core figure of bench.py)."""
import os, sys, time
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [R + '/tests', R, R + '/ms-eetc_amd']
import cases
from oracle import oracle
prob = cases.oracle_problem(cases.train_default(), cases.track_00(), 100)
T = cases.c1_times(64)
scen = np.array([[0, t, 1, 1] for t in T])
for start in ('profile', 'reference'):
    oracle.solve_batch(prob, scen[:4], nthreads=1, start=start)
    t0 = time.perf_counter()
    z, st, nf = oracle.solve_batch(prob, scen, nthreads=1, start=start)
    dt = time.perf_counter() - t0
    print('%-9s start: %d solves on one thread in %.2f s -> %.1f solves/s (%.1f ms per solve, %.1f iterations)' % (start, len(T), dt, len(T)/dt, 1e3*dt/len(T), st[:, 1].mean()))
