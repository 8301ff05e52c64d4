"""Diagnostic (GPU): kernel time and parity against the oracle at N = 520 / 540 / 556 with whatever geometry the loaded library picks there (MSD_LIB selects
a variant library).  Used for the 256 x 3 experiment of DESIGN.md section 8: in msd_geometry.hpp: pick_geometry_t, in front of the 320 x 2 line,
    if (nodes <= 768 && sizeof(double)*(size_t)lds_doubles(N, 768, DYN != LOSS_STATIC) <= 160*1024) return {256, 3, solve_kernel<256, 3, 1, DYN>};
compile msd_kernels_static.hip with it and link it with the other objects of ms-eetc_amd/lib/obj into a variant library."""
import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/ms-eetc_amd'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, cases
from mseetc.ocp import casadiSolver
from oracle import oracle
train, track = cases.train_default(), cases.track_00()
for N in (520, 540, 556):
    s = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='profile')
    T = 1541*(1 + 0.15*np.random.default_rng(N).random(1024))
    res = s.solveBatch(T); res = s.solveBatch(T)
    prob = cases.oracle_problem(train, track, N)
    scen = np.array([[0.0, t, 1.0, 1.0] for t in T[:32]])
    z, st, nf = oracle.solve_batch(prob, scen, start='profile')
    dobj = np.max(np.abs(res['cost'][:32] - st[:, 2])/np.abs(st[:, 2]))
    print('N', N, 'geometry', s.problem.geometry(), 'kernel ms', res['kernel_ms'], 'converged', int(np.sum(res['status'] == 0)), 'iters', res['iterations'].mean(), 'max dobj vs oracle (32)', dobj, flush=True)
    s.close()
