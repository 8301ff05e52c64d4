import sys
sys.path[:0]=['/root/repo/ms-eetc_amd','/root/repo','/root/repo/tests']
import numpy as np, cases
from mseetc.ocp import casadiSolver
from mseetc._device import ST
train, track = cases.train_default(), cases.track_00(16000)
s = casadiSolver(train, track, dict(numIntervals=40, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='profile')
T = np.array([804.9041795334854, 762.6780418513658, 700.0])
r = s.solveBatch(T, classifyFailures=False)
print('status', r['status'], 'iters', r['iterations'])
for k in range(3): print({n: float(r['stats'][k, ST[n]]) for n in ('STATUS','ITERS','N_REG','N_SOC','N_BACKTRACK','N_RESTO','N_WATCHDOG','N_FALLBACK','KKT','MU')})
print(s.problem.follow_counts())
