import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/ms-eetc_amd'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, cases
from mseetc.ocp import casadiSolver
from oracle import oracle
np.set_printoptions(linewidth=220)
train, track, N, B = cases.train_default(), cases.track_00(), 100, 64
solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='profile')
T = 1541*(1 + 0.15*np.random.default_rng(N).random(B))
res = solver.solveBatch(T)
prob = cases.oracle_problem(train, track, N)
scen = np.stack([prob.scenario(float(t))[[oracle.DP['T0'], oracle.DP['TEND'], oracle.DP['V0SQ'], oracle.DP['VNSQ']]] for t in T])
z, st, nfail = oracle.solve_batch(prob, scen, nthreads=0, start='profile')
dit = res['iterations'] - st[:, 1]
k = int(np.argmax(np.abs(dit)))
print('scenario', k, 'T', T[k], 'gpu iters', res['iterations'][k], 'oracle', st[k,1], 'mu', res['stats'][k,4], st[k,4], 'obj', res['cost'][k], st[k,2])
out = solver.problem.solve_batch(solver._scenarios([T[k]], 0, 1, 1), history=64)
ref = oracle.solve(prob, prob.scenario(float(T[k])), start='profile', history=True)
hg, ho = out['hist'], ref['hist']
for i in range(int(max(res['iterations'][k], st[k,1])) + 1):
    g = hg[i] if i < len(hg) else np.zeros(8); o = ho[i] if i < len(ho) else np.zeros(8)
    print('%2d  gpu obj %.10e pr %.3e du %.3e lgmu %6.2f a %.3e %.3e | oracle obj %.10e pr %.3e du %.3e lgmu %6.2f a %.3e %.3e' % (i, g[1], g[2], g[3], g[4], g[6], g[7], o[1], o[2], o[3], o[4], o[6], o[7]))
