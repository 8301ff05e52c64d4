"""
How many scenarios of the benchmark batches the first-pass kernel hands to the follow-up kernel, and why (msd_problem_follow_counts):
    python tools/follow_probe.py [c1 c1_8192 c2 c3 ...]
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import bench   # noqa: E402

WHY = ['no fused start', 'inertia/scan', 'tiny step', 'soc', 'line search', 'second attempt']
for name in (sys.argv[1:] or ['c1', 'c1_8192', 'c2', 'c3']):
    start = 'reference' if name.endswith('ref') else 'profile'
    wl, B = (name.replace('ref', '').split('_') + [None])[:2]
    B = int(B) if B else bench.PER_GPU_BATCH[wl]
    solver, scen, ovr, text = bench.build_workload(wl, B, None, 0, start, 0)
    before = solver.problem.follow_counts()
    elapsed, ms, st = bench.measure(solver, scen, ovr, 1, 0)
    after = solver.problem.follow_counts()
    print(name, 'B', B, 'kernel ms %.3f' % ms, 'listed', after[0] - before[0], dict(zip(WHY, [a - b for a, b in zip(after[1], before[1])])))
