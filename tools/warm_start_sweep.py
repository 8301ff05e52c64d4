import sys, time, numpy as np
import os; R=os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0]=[R+'/tests', R, R+'/ms-eetc_amd']
import cases
from mseetc.ocp import casadiSolver
from mseetc.mpc import shrinkingHorizon
train, track = cases.train_default(), cases.track_00()
opts = dict(numIntervals=100, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1))
B=1024
T = 1541*(1 + 0.15*np.random.default_rng(20260612).random(B))
s = casadiSolver(train, track, opts)
first = s.solveBatch(T)
print('cold', first['kernel_ms'], first['iterations'].mean(), (first['status']==0).mean())
T2 = T*(1+0.005*np.random.default_rng(1).standard_normal(B))
cold = s.solveBatch(T2)
for mu in (1e-1,1e-2,1e-3,1e-4):
    for push in (1e-2,1e-3,1e-4):
        w = s.solveBatch(T2, guess=first['z'], warmMu=mu, warmPush=push)
        ok = w['status']==0
        print('mu',mu,'push',push,'ms %.2f'%w['kernel_ms'],'iters %.1f'%w['iterations'].mean(),'max',w['iterations'].max(),'ok',ok.mean(),'dcost', np.max(np.abs(w['cost'][ok]-cold['cost'][ok])/cold['cost'][ok]))
for ws in (False, True):
    t0=time.time()
    log = shrinkingHorizon(train, track, opts, T, numResolves=10, noise=0.01, seed=1, warmStart=ws)
    print('mpc warm',ws,'wall',time.time()-t0,'iters',[round(l['iterations'].mean(),1) for l in log],'fails',[int((l['status']<0).sum()) for l in log])
