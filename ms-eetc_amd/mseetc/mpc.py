"""
Receding (shrinking) horizon re-solves on top of the batched solver -- BASELINE config 4.

The reference has no MPC loop; its mechanism for a re-solve from the current position is
`Track.updateLimits(positionStart=...)` (track.py:420-450) + a new `casadiSolver` on the cropped track +
`solve(T, initialTime=t_now, initialVelocity=v_now)` (ocp.py:310), always from a cold start (ocp.py:325-339).
This driver does exactly that for a whole batch of scenarios at once: the position sequence only depends on the
grids, so every re-solve is one problem shared by the batch and one kernel launch.
"""

import copy

import numpy as np

from .ocp import casadiSolver


def shrinkingHorizon(train, track, optsDict, terminalTime, numResolves, stride=2, noise=0.0, seed=0,
                     initialTime=0.0, initialVelocity=1.0, terminalVelocity=1.0, device=0, solverFactory=None):
    """
    Re-solve `numResolves` times; after each solve the train advances `stride` intervals of the current grid, the
    measured time and speed at that node are perturbed by `noise` (relative, standard normal) and the remaining
    horizon (stride intervals shorter) is solved again from a cold start.

    terminalTime: array (B,) of arrival times (absolute).
    Returns a list of dicts per re-solve: position [m], numIntervals, t0 (B,), v0 (B,), status, iterations, cost, z.
    `solverFactory(train, track, opts)` lets tests substitute the solver (default: the device solver).
    """

    make = solverFactory or (lambda tr, tk, op: casadiSolver(tr, tk, op, device=device))

    T = np.atleast_1d(np.asarray(terminalTime, dtype=float))
    B = T.shape[0]
    rng = np.random.default_rng(seed)

    N = int(optsDict.get('numIntervals', 100))
    t_now = np.full(B, float(initialTime))
    v_now = np.full(B, float(initialVelocity))
    position = 0.0
    current = copy.deepcopy(track)
    log = []

    for k in range(numResolves):

        Nk = N - stride*k

        if Nk < 1:
            break

        opts = dict(optsDict)
        opts['numIntervals'] = Nk

        solver = make(train, current, opts)
        res = solver.solveBatch(T, initialTime=t_now, terminalVelocity=terminalVelocity, initialVelocity=v_now)

        log.append(dict(position=position, numIntervals=Nk, t0=t_now.copy(), v0=v_now.copy(), status=res['status'].copy(),
                        iterations=res['iterations'].copy(), cost=res['cost'].copy(), z=res['z']))

        if hasattr(solver, 'close'):
            solver.close()

        if Nk - stride < 1:
            break

        # state at node `stride` of this grid (layout ocp.py:376-405): t and b of stage `stride`
        stp = 4 + int(solver.withPnBrake)
        t_meas = res['z'][:, stp*stride + 2 + int(solver.withPnBrake)]
        v_meas = np.sqrt(res['z'][:, stp*stride + 3 + int(solver.withPnBrake)])

        # failed scenarios keep coasting on their last measurement
        ok = res['status'] >= 0
        n1, n2 = rng.standard_normal(B), rng.standard_normal(B)
        t_now = np.where(ok, np.maximum(t_meas*(1 + noise*n1), 0.0), t_now)
        v_now = np.where(ok, v_meas*(1 + noise*n2), v_now)

        advance = float(solver.points.index.values[stride])
        position += advance
        current = copy.deepcopy(current)
        current.updateLimits(positionStart=advance)

    return log
