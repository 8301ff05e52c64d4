import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/ms-eetc_amd')
import numpy as np, bench
from mseetc._device import ST
np.set_printoptions(linewidth=200)
name = sys.argv[1] if len(sys.argv) > 1 else 'irk_radau2'
solver, scen, ovr, text = bench.build_workload('c1', 1024, 0, 0, 'profile', 0, name)
os.environ['MSD_DEBUG_NO_FOLLOW_UP'] = '0'
ref = solver.problem.solve_batch(scen)
ref = {k: np.array(v) for k, v in ref.items() if hasattr(v, 'shape')}
scen2 = scen.copy(); scen2[:, 1] += 7.0
solver.problem.solve_batch(scen2)      # (what an entry the first pass does not write will still hold)
os.environ['MSD_DEBUG_NO_FOLLOW_UP'] = '1'
out = solver.problem.solve_batch(scen)
diff = np.flatnonzero(out['stats'][:, 2] != ref['stats'][:, 2])
print('scenarios whose statistics differ without the follow-up kernel:', diff.tolist())
for i in diff[:2]:
    o3 = solver.problem.solve_batch(np.ascontiguousarray(scen[i:i + 1]), history=200)
    print('scenario', i, scen[i], 'final (with follow-up) iters', ref['stats'][i, 1], 'status', ref['stats'][i, 0])
    h = o3['hist']
    for r in h[:60]:
        if r[0] == 0 and r[1] == 0 and r[2] == 0: break
        print('   ' + ' '.join('%12.5e' % v for v in r))
