"""
Diagnostic (GPU): which scenarios of a transcription's config-1 batch the first-pass kernel hands to the follow-up kernel, why, and what the solve
looks like (msd_problem_follow_counts + the per-scenario statistics).   python tools/follow_probe_alt.py [irk_radau2 cvodes_tolerances integrate_losses]
"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import numpy as np
import bench   # noqa: E402
from mseetc._device import ST
WHY = ['no fused start', 'inertia/scan', 'tiny step', 'soc', 'line search', 'second attempt', 'watchdog']
for name in (sys.argv[1:] or ['irk_radau2']):
    solver, scen, ovr, text = bench.build_workload('c1', 1024, 0, 0, 'profile', 0, name)
    before = solver.problem.follow_counts()
    out = solver.problem.solve_batch(scen)
    after = solver.problem.follow_counts()
    st = out['stats']
    print(name, 'kernel ms %.3f' % out['kernel_ms'], 'listed', after[0] - before[0], dict(zip(WHY, [a - b for a, b in zip(after[1], before[1])])))
    odd = np.flatnonzero((st[:, ST['N_RESTO']] > 0) | (st[:, ST['N_WATCHDOG']] > 0) | (st[:, ST['N_SOC']] > 0) | (st[:, ST['N_BACKTRACK']] > 8))
    for i in odd[:10]:
        print('  scenario', i, 'T', scen[i][1] if scen.ndim == 2 else '', {k: st[i, v] for k, v in ST.items() if k in ('STATUS', 'ITERS', 'N_REG', 'N_SOC', 'N_BACKTRACK', 'N_RESTO', 'N_WATCHDOG')})
    # the same batch without the restoration phase: a scenario whose line search breaks down in the first pass ends there (status -2), with its log
    from mseetc import workloads as wl
    from mseetc.ocp import casadiSolver
    extra, io = bench.TRANSCRIPTIONS[name]
    train, track, N = wl.config('c1')
    opts = dict(wl.options(N), **extra)
    if io is not None:
        opts['integrationOptions'] = io
    s2 = casadiSolver(train, track, opts, device=0, startingPoint='profile', restoration=False)
    o2 = s2.problem.solve_batch(scen)
    bad = np.flatnonzero(o2['stats'][:, 0] < 0)
    print('  without restoration: failed', bad.tolist(), o2['stats'][bad, 0].tolist())
    np.set_printoptions(linewidth=200)
    for i in bad[:2]:
        o3 = s2.problem.solve_batch(np.ascontiguousarray(scen[i:i + 1]), history=200)
        st3 = o3['stats'][0]
        print('  scenario', i, scen[i], 'alone: status', st3[0], 'iters', st3[1], 'kkt', st3[3], 'backtracks', st3[ST['N_BACKTRACK']])
        for r in o3['hist'][-14:]:
            print('     ' + ' '.join('%12.5e' % v for v in r))
