import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/tests', R, R + '/ms-eetc_amd']
import numpy as np, cases
from mseetc.ocp import casadiSolver
from oracle import oracle
train, track = cases.train_default(), cases.track_00()
N = 100
solver = casadiSolver(train, track, dict(numIntervals=N, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1)))
prob = cases.oracle_problem(train, track, N)
T = np.array([1500.0, 1560.0, 1620.0, 1700.0])
first = solver.solveBatch(T)
T2 = T*1.005
warm = solver.solveBatch(T2, guess=first['z'], warmMu=1e-2, warmPush=1e-3)
print('warm iters', warm['iterations'], 'kkt', warm['stats'][:, 3])
for k in range(4):
    ref = oracle.solve(prob, prob.scenario(T2[k]), guess=first['z'][k], mu0=1e-2, push=1e-3, history=True)
    print(k, 'oracle iters', ref['stats']['ITERS'], 'kkt', ref['stats']['KKT'])
    print(ref['hist'][-3:, :4])
