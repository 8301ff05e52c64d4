"""
Receding (shrinking) horizon re-solves on top of the batched solver -- BASELINE config 4.

The reference has no MPC loop; its mechanism for a re-solve from the current position is
`Track.updateLimits(positionStart=...)` (track.py:420-450) + a new `casadiSolver` on the cropped track +
`solve(T, initialTime=t_now, initialVelocity=v_now)` (ocp.py:310), always from a cold start (ocp.py:325-339).
This driver does exactly that for a whole batch of scenarios at once: the position sequence only depends on the
grids, so every re-solve is one problem shared by the batch and one kernel launch.
"""

import copy
from concurrent.futures import ThreadPoolExecutor

import numpy as np
from ._device import ST as _ST

from .ocp import casadiSolver


def transferSolution(z, positionsOld, positionsNew, withPnBrake):
    """
    Move solutions z (B, nz_old), laid out as ocp.py:166-272 on the grid `positionsOld`, onto the grid `positionsNew`
    (same length unit and origin, covered by the old grid): states t and b = v^2 are interpolated linearly in position,
    the piecewise-constant controls and slacks are taken from the old interval that contains the midpoint of the new
    one.  Used to warm-start a re-solve on a shorter horizon.  Returns (B, nz_new).
    """

    z = np.atleast_2d(np.asarray(z, dtype=float))
    pOld = np.asarray(positionsOld, dtype=float)
    pNew = np.asarray(positionsNew, dtype=float)
    No, Nn = len(pOld) - 1, len(pNew) - 1
    pn = int(bool(withPnBrake))
    stp = 4 + pn

    if z.shape[1] != stp*No + 2:
        raise ValueError("Solution does not match the old grid!")

    # the common case of the shrinking horizon: the new grid is a tail of the old one -> a slice
    off = No - Nn
    if 0 <= off and np.allclose(pNew, pOld[off:], rtol=0, atol=1e-6):
        return np.ascontiguousarray(z[:, stp*off:])

    body = z[:, :stp*No].reshape(-1, No, stp)
    tOld = np.concatenate([body[:, :, 2 + pn], z[:, stp*No:stp*No + 1]], axis=1)
    bOld = np.concatenate([body[:, :, 3 + pn], z[:, stp*No + 1:stp*No + 2]], axis=1)

    k = np.clip(np.searchsorted(pOld, pNew, side='right') - 1, 0, No - 1)
    w = np.clip((pNew - pOld[k])/(pOld[k + 1] - pOld[k]), 0.0, 1.0)
    tNew = tOld[:, k]*(1 - w) + tOld[:, k + 1]*w
    bNew = bOld[:, k]*(1 - w) + bOld[:, k + 1]*w

    mid = 0.5*(pNew[:-1] + pNew[1:])
    km = np.clip(np.searchsorted(pOld, mid, side='right') - 1, 0, No - 1)

    out = np.empty((z.shape[0], stp*Nn + 2))
    nb = out[:, :stp*Nn].reshape(-1, Nn, stp)
    nb[:, :, :2 + pn] = body[:, km, :2 + pn]
    nb[:, :, 2 + pn] = tNew[:, :Nn]
    nb[:, :, 3 + pn] = bNew[:, :Nn]
    out[:, stp*Nn] = tNew[:, Nn]
    out[:, stp*Nn + 1] = bNew[:, Nn]

    return out


class DeviceLoop():
    """
    The shrinking-horizon loop with its bookkeeping on the device (csrc/msd_mpc.hip, include/mseetc_mpc.h): the grid sequence does not depend
    on the solutions, so the problem records of all re-solves are built and uploaded once (here), and `run` executes a whole loop -- measured
    states, scenario records, warm starts, the handling of arrival times that can no longer be met, the log -- as launches on one stream
    without the host in between.  Same loop as shrinkingHorizon(); reusable for any number of batches of arrival times.
    """

    def __init__(self, train, track, optsDict, numResolves, stride=2, noise=0.0, terminalVelocity=1.0, device=0, warmStart=False, warmPush=1e-3,
                 dualMu=1e-4, relaxInfeasible=True, lateMargin=5e-3):

        from . import _device

        N = int(optsDict.get('numIntervals', 100))
        current = copy.deepcopy(track)
        position = 0.0
        self.solvers, self.twins, self.positions = [], [], []
        tail, previous = [], None
        for k in range(numResolves):
            Nk = N - stride*k
            if Nk < 1:
                break
            opts = dict(optsDict); opts['numIntervals'] = Nk
            # (no restoration phase, like shrinkingHorizon: a re-solve that fails is certified and relaxed by the loop itself; no watchdog procedure for the same
            #  reason -- a re-solve that crawls through ten shortened iterations would leave the fused iteration for the follow-up kernel, and a launch lasts as
            #  long as its slowest scenario.  On config 4 it makes no measurable difference either way)
            solver = casadiSolver(train, current, opts, device=device, restoration=False, watchdogTrigger=-1)
            self.solvers.append(solver)
            if relaxInfeasible:
                topts = dict(opts); topts['energyOptimal'] = False; topts.pop('integrateLosses', None)
                self.twins.append(casadiSolver(train, current, topts, device=device, startingPoint='profile', restoration=False, watchdogTrigger=-1))
            pos = position + solver.points.index.values
            tail.append(int(previous is not None and len(previous) - len(pos) == stride and np.allclose(pos, previous[stride:], rtol=0, atol=1e-6)))
            self.positions.append(pos)
            previous = pos
            if Nk - stride < 1:
                break
            nxt = copy.deepcopy(current)
            nxt.updateLimits(positionStart=float(solver.points.index.values[stride]))
            position += float(solver.points.index.values[stride])
            current = nxt
        first = self.solvers[0]
        vlim = first.points['Speed limit [m/s]'].values
        self.stride, self.noise, self.K = stride, noise, len(self.solvers)
        self.energyOptimal = first.energyOptimal
        self.scale = 1.0 if first.energyOptimal else first.scalingFactorObjective
        self.withPnBrake = first.withPnBrake
        self._loop = _device.DeviceLoop(first.problem, self.twins[0].problem if relaxInfeasible else None, [s._desc for s in self.solvers],
                                        [t._desc for t in self.twins] if relaxInfeasible else None, stride,
                                        [s.points['Speed limit [m/s]'].values[0] for s in self.solvers], [s.track.length for s in self.solvers], tail,
                                        first.velocityMin, first._vmaxTrain, min(max(terminalVelocity, first.velocityMin), vlim[-1]), warmStart, dualMu, warmPush,
                                        noise, relaxInfeasible, lateMargin)

    def run(self, terminalTime, seed=0, initialTime=0.0, initialVelocity=1.0, keepZ=True):
        "One loop; returns the log of shrinkingHorizon() (list of dicts per re-solve; 'z' None without keepZ) -- `loop_ms` in every entry is the device time of the whole loop."

        from ._device import MPC

        T = np.array(np.atleast_1d(np.asarray(terminalTime, dtype=float)), copy=True)
        B = T.shape[0]
        rng = np.random.default_rng(seed)
        n1, n2 = np.zeros((max(self.K - 1, 0), B)), np.zeros((max(self.K - 1, 0), B))
        for k in range(self.K - 1):      # (the order of shrinkingHorizon's draws)
            n1[k], n2[k] = rng.standard_normal(B), rng.standard_normal(B)
        log, zs, ms = self._loop.run(T, initialTime, initialVelocity, n1, n2, keepZ=keepZ)
        out = []
        for k in range(self.K):
            r = log[k]
            out.append(dict(position=float(self.positions[k][0]), numIntervals=self.solvers[k].numIntervals, t0=r[:, MPC['T0']].copy(), v0=r[:, MPC['V0']].copy(),
                            T=r[:, MPC['T']].copy(), status=r[:, MPC['STATUS']].astype(int), iterations=r[:, MPC['ITERS']].astype(int),
                            cost=r[:, MPC['OBJ']]*self.scale, z=zs[k] if zs is not None else None, relaxed=r[:, MPC['RELAXED']] > 0,
                            kernel_ms=ms/self.K, loop_ms=ms))
        return out

    def close(self):

        if getattr(self, '_loop', None) is not None:
            self._loop.close()
            self._loop = None
        for s in self.solvers + self.twins:
            s.close()
        self.solvers, self.twins = [], []


def shrinkingHorizon(train, track, optsDict, terminalTime, numResolves, stride=2, noise=0.0, seed=0,
                     initialTime=0.0, initialVelocity=1.0, terminalVelocity=1.0, device=0, solverFactory=None,
                     warmStart=False, warmMu=1e-2, warmPush=1e-3, dualMu=1e-4, relaxInfeasible=True, lateMargin=5e-3, onDevice=False):
    """
    Re-solve `numResolves` times; after each solve the train advances `stride` intervals of the current grid, the
    measured time and speed at that node are perturbed by `noise` (relative, standard normal) and the remaining
    horizon (stride intervals shorter) is solved again -- from a cold start like the reference, or with
    `warmStart=True` from the previous solution moved onto the new grid -- on the device when the new grid is the tail of the old
    one (msd_solve_batch_shifted: primal point and multipliers, barrier parameter `dualMu`), through transferSolution otherwise (primal
    point only, `warmMu`); a warm start that breaks down is repeated cold inside the launch.

    A re-solve that fails because the measured state no longer allows the arrival time (near the end of the horizon a
    late measurement leaves less than the minimum running time) is repeated with the arrival time moved to the scenario's certified
    minimum running time from there (casadiSolver.minimumTime, plus `lateMargin` relative): the train arrives late and keeps
    being controlled (`relaxInfeasible`; off: such a scenario keeps its last measurement).  The moved arrival time stays for the
    rest of the horizon; 'relaxed' in the log marks the scenarios it was applied to.

    terminalTime: array (B,) of arrival times (absolute).
    Returns a list of dicts per re-solve: position [m], numIntervals, t0 (B,), v0 (B,), T (B,), status, iterations, cost, z, relaxed (B,) bool.
    `solverFactory(train, track, opts)` lets tests substitute the solver (default: the device solver).
    """

    if onDevice:
        # the same loop with its bookkeeping on the device (DeviceLoop): one call, no host work between the launches
        if solverFactory is not None:
            raise ValueError("onDevice runs the device solver!")
        loop = DeviceLoop(train, track, optsDict, numResolves, stride=stride, noise=noise, terminalVelocity=terminalVelocity, device=device, warmStart=warmStart,
                          warmPush=warmPush, dualMu=dualMu, relaxInfeasible=relaxInfeasible, lateMargin=lateMargin)
        try:
            return loop.run(terminalTime, seed=seed, initialTime=initialTime, initialVelocity=initialVelocity)
        finally:
            loop.close()

    # (no restoration phase: a re-solve that fails is certified and relaxed below, and a launch lasts as long as its slowest scenario)
    make = solverFactory or (lambda tr, tk, op: casadiSolver(tr, tk, op, device=device, restoration=False, watchdogTrigger=-1))

    T = np.array(np.atleast_1d(np.asarray(terminalTime, dtype=float)), copy=True)
    B = T.shape[0]
    rng = np.random.default_rng(seed)

    N = int(optsDict.get('numIntervals', 100))
    t_now = np.full(B, float(initialTime))
    v_now = np.full(B, float(initialVelocity))
    position = 0.0
    current = copy.deepcopy(track)
    previous = None
    batchOnDevice = False      # the handle's last launch solved the whole batch (its solutions are what a shifted warm start reads)
    last = None
    log = []

    # The sequence of positions, grids and problem descriptions does not depend on the solutions: while the device solves re-solve k a
    # worker thread prepares problem k + 1 (crop of the track, grid, description -- pandas work that releases the interpreter while the
    # main thread waits inside the launch).  Only with the device solver; a substituted factory is called in line.
    pool = ThreadPoolExecutor(max_workers=1) if solverFactory is None else None

    def prepare(trackNow, Nk):
        "(solver for Nk intervals on trackNow, track after the train has advanced `stride` intervals or None)"
        opts = dict(optsDict)
        opts['numIntervals'] = Nk
        solver = make(train, trackNow, opts)
        nxt = None
        if Nk - stride >= 1:
            nxt = copy.deepcopy(trackNow)
            nxt.updateLimits(positionStart=float(solver.points.index.values[stride]))
        return solver, nxt

    pending = None

    try:
        for k in range(numResolves):

            Nk = N - stride*k

            if Nk < 1:
                break

            solver, following = pending.result() if pending is not None else prepare(current, Nk)
            pending = None
            if pool is not None and following is not None and k + 1 < numResolves:
                pending = pool.submit(prepare, following, Nk - stride)

            if hasattr(solver, 'adoptDevice'):
                solver.adoptDevice(last)      # same device handle for every re-solve
                twin = last.__dict__.pop('_twin', None) if last is not None else None
                if twin is not None:
                    if solver.__dict__.get('_twin') is None and hasattr(twin, 'adoptDevice'):
                        # the time-optimal twin of the previous problem hands its device handle to the twin of this one
                        opts = dict(solver._optsDict); opts['energyOptimal'] = False; opts.pop('integrateLosses', None)
                        solver._twin = type(solver)(solver.train, solver.track, opts, device=solver._device, startingPoint='profile', restoration=solver.restoration).adoptDevice(twin)
                    twin.close()
                if warmStart and solverFactory is None:
                    solver.problem.keep_duals(True)      # the multipliers of every solve stay on the device for the next re-solve
                    solver.problem.direct_results(False)      # ... and so do the solutions: the shifted warm start reads them there

            common = dict(initialTime=t_now, terminalVelocity=terminalVelocity, initialVelocity=v_now)

            posNew = position + solver.points.index.values
            tail = (warmStart and previous is not None and batchOnDevice and hasattr(solver, 'adoptDevice') and solverFactory is None and
                    len(previous[1]) - len(posNew) == stride and np.allclose(posNew, previous[1][stride:], rtol=0, atol=1e-6))

            if tail:
                # the new grid is the tail of the old one and the previous solutions are still on the device (same handle): the re-solve
                # warm-starts from them there -- no upload, and scenarios without a usable guess start cold inside the same launch
                res = solver.solveBatch(T, shift=stride, warmMu=dualMu, warmPush=warmPush, classifyFailures=False, **common)
                batchOnDevice = True
            elif warmStart and previous is not None:
                zPrev, posPrev, okPrev = previous
                guess = transferSolution(zPrev, posPrev, position + solver.points.index.values, solver.withPnBrake)
                usable = okPrev & np.isfinite(guess).all(axis=1)
                if usable.any():
                    # scenarios without a usable guess get another scenario's solution as a placeholder: a warm start that breaks down is
                    # repeated from the problem's own starting point inside the launch (solve_kernel), so no second launch is needed
                    if not usable.all():
                        guess[~usable] = guess[np.flatnonzero(usable)[0]]
                    res = solver.solveBatch(T, guess=guess, warmMu=warmMu, warmPush=warmPush, classifyFailures=False, **common)
                else:
                    res = solver.solveBatch(T, classifyFailures=False, **common)
                batchOnDevice = True
            else:
                res = solver.solveBatch(T, classifyFailures=False, **common)
                batchOnDevice = True

            relaxed = np.zeros(B, dtype=bool)
            kernel_extra = 0.0
            bad = np.flatnonzero(res['status'] < 0)
            if relaxInfeasible and bad.size and hasattr(solver, 'minimumTime'):
                # what stops these scenarios is their arrival time: the minimum running time from the measured state (time-optimal twin of
                # this re-solve's problem) tells, and becomes the new arrival time where it is later than the one asked for
                scenBad = solver._scenarios(T[bad], t_now[bad], terminalVelocity, v_now[bad])
                tmin, okMin = solver.minimumTime(scenBad)
                late = okMin & (tmin > (T[bad] - t_now[bad]))
                if late.any():
                    idx, tm = bad[late], tmin[late]
                    relaxed[idx] = True
                    margin = lateMargin
                    for attempt in range(3):
                        # (an interior-point solve needs some room above the minimum running time: a scenario that still breaks down gets four times the margin)
                        T[idx] = t_now[idx] + tm*(1 + margin)
                        again = solver.solveBatch(T[idx], initialTime=t_now[idx], terminalVelocity=terminalVelocity, initialVelocity=v_now[idx], classifyFailures=False)
                        for key in ('z', 'status', 'iterations', 'cost'):
                            res[key][idx] = again[key]
                        kernel_extra += float(again.get('kernel_ms', 0.0))
                        still = again['status'] < 0
                        if not still.any():
                            break
                        idx, tm, margin = idx[still], tm[still], 4*margin
                    batchOnDevice = False      # the handle's last launch held these scenarios only: the next re-solve takes its guess from the host copy

            previous = (res['z'], position + solver.points.index.values, res['status'] >= 0)

            log.append(dict(position=position, numIntervals=Nk, t0=t_now.copy(), v0=v_now.copy(), T=T.copy(), status=res['status'].copy(),
                            iterations=res['iterations'].copy(), cost=res['cost'].copy(), z=res['z'], relaxed=relaxed,
                            kernel_ms=float(res.get('kernel_ms', 0.0)) + kernel_extra,
                            # (inertia corrections of the main launch's solves, where the solver reports its statistics: the device solver)
                            regularisations=(res['stats'][:, _ST['N_REG']].astype(int) if 'stats' in res else None)))

            last = solver

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

            position += float(solver.points.index.values[stride])
            current = following

    finally:
        # also on an exception in a solve or a reconfigure: no worker thread (building the next problem on the shared train) is left behind
        if pool is not None:
            pool.shutdown(wait=True, cancel_futures=True)

    if last is not None and hasattr(last, 'close'):
        last.close()

    return log
