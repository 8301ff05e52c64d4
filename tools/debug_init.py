import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('ms-eetc_amd', '', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, cases
from mseetc.ocp import casadiSolver
from mseetc._device import ST
np.set_printoptions(linewidth=220, precision=5)
N = int(sys.argv[1])
s = casadiSolver(cases.train_fig10(), cases.track_00(), dict(numIntervals=N, maxIterations=1, integrationOptions=dict(numSteps=1, numApproxSteps=1)), startingPoint='reference')

scen = s._scenarios([1541.0], 0, 1, 1)
out = s.problem.solve_batch(scen, history=8)
st = out['stats'][0]; z = out['z'][0]
print('status', st[ST['STATUS']], 'iters', st[ST['ITERS']], 'obj', st[ST['OBJ']])
stp = 5 if s.withPnBrake else 4
zz = z[:stp*N].reshape(N, stp)
print('Fel', zz[:, 0]); print('s', zz[:, stp-3]); print('b', zz[:, stp-1])
print(out['hist'][:2])
os.makedirs(os.path.join(ROOT,'gpurun_out','dbg'),exist_ok=True)
np.save(os.path.join(ROOT,'gpurun_out','dbg','z_init_%d.npy'%N), z)
