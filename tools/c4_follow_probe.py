"""Diagnostic (GPU): config 4's host loop (512 x 50, cold and warm) -- what the first-pass kernels hand to the follow-up kernels, by reason, summed over the loop and for the
first re-solves (msd_problem_follow_counts of every re-solve's handle)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('ms-eetc_amd', '', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np
from mseetc import workloads as wl
from mseetc.mpc import shrinkingHorizon
from mseetc.ocp import casadiSolver
WHY = ['no fused start', 'inertia/scan', 'tiny step', 'soc', 'line search', 'second attempt', 'watchdog']
train, track, N = wl.config('c4')
T = wl.c1_times(512, seed=20260615)
for warm in (False, True):
    seen = []
    def factory(tr, tk, op):
        s = casadiSolver(tr, tk, op, device=0, restoration=False, watchdogTrigger=-1)
        close = s.close
        def closing():
            try:
                seen.append((s.numIntervals, bool(s.opts.energyOptimal) if hasattr(s, 'opts') else None, s.problem.follow_counts()))
            except Exception as e:
                seen.append((s.numIntervals, None, (0, [0]*7)))
            close()
        s.close = closing
        return s
    log = shrinkingHorizon(train, track, wl.options(N), T, numResolves=50, noise=0.01, seed=1, warmStart=warm, solverFactory=factory)
    tot = np.zeros(7, dtype=int); listed = 0
    for k, (Nk, eo, (n, why)) in enumerate(seen):
        tot += np.array(why[:7]); listed += n
        if k < 6 or n > 40:
            print('  ', 'warm' if warm else 'cold', 'handle', k, 'N', Nk, 'listed', n, dict(zip(WHY, why)))
    print('warm' if warm else 'cold', 'handles', len(seen), 'listed', listed, dict(zip(WHY, tot.tolist())), 'failed re-solves', sum(int((l['status'] < 0).sum()) for l in log), 'relaxed', sum(int(l['relaxed'].sum()) for l in log))
