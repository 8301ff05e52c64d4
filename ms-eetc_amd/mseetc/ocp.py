"""
Optimal-control front end of the MI355X solver: same surface as the reference's
`mseetc/ocp.py` (`OptionsCasadiSolver` :12-74, `casadiSolver` :77-409) -- but the NLP is
not built symbolically and handed to IPOPT; `__init__` packs the problem data that the
reference's transcription is made of (ocp.py:96-125, 166-284) into the C-ABI record of
include/mseetc_hip.h and `solve` / `solveBatch` run the HIP interior-point kernels
(csrc/msd_kernel.hpp), one workgroup per scenario.

`solveBatch` is the addition that makes the device worth using: many (T, t0, v0, vN)
scenarios of the same (train, track, options) in one launch.
"""

import numpy as np
import pandas as pd

from .train import OptionsRK, OptionsIRK, OptionsCVODES, collocationTables
from .track import computeDiscretizationPoints
from .utils import Options, classifyLosses, postProcessDataFrame, LOSS_NONE, LOSS_DYNAMIC
from . import _device


class OptionsCasadiSolver(Options):

    def __init__(self, paramsDict):
        # option names, defaults and meanings are the reference's (mseetc/ocp.py:16-28): a drop-in keeps the configuration schema
        self.numIntervals = 100           # N: shooting intervals = pieces of the piecewise-constant controls
        self.maxIterations = 1e3          # interior-point iteration limit (a float in the reference, too)
        self.energyOptimal = True         # objective: traction energy [kWh] (True) or running time [s] (False)
        self.minimumVelocity = 1          # m/s; speeds are bounded below by it (b = v^2 stays away from 0)
        self.integrationMethod = 'RK'     # transcription of the interval dynamics: 'RK' | 'IRK' | 'CVODES'
        self.integrationOptions = {}      # options of that method (OptionsRK / OptionsIRK / OptionsCVODES)
        self.integrateLosses = False      # loss slack from the integrated loss power instead of the mid-point rule

        super().__init__(paramsDict)

    def overwriteDefaults(self, paramsDict):

        super().overwriteDefaults(paramsDict)

        nested = paramsDict['integrationOptions'] if 'integrationOptions' in paramsDict else {}

        if self.integrationMethod == 'RK':
            self.integrationOptions = OptionsRK(nested)

        elif self.integrationMethod == 'IRK':
            self.integrationOptions = OptionsIRK(nested)

        elif self.integrationMethod == 'CVODES':
            self.integrationOptions = OptionsCVODES(nested)

    def checkValues(self):

        self.checkPositiveInteger(self.numIntervals, 'Number of intervals', allowZero=False)

        self.checkPositiveInteger(self.maxIterations, 'Maximum number of iterations', allowZero=False)

        if not isinstance(self.energyOptimal, bool):
            raise ValueError("'energyOptimal' flag must be a boolean!")

        if type(self.minimumVelocity) not in {int, float} or self.minimumVelocity <= 0:
            raise ValueError("Minimum velocity should be a strictly positive number!")

        if self.integrationMethod not in {'RK', 'IRK', 'CVODES'}:
            raise ValueError("Unknown integration method!")

        if not isinstance(self.integrateLosses, bool):
            raise ValueError("'integrateLosses' flag must be a boolean!")


class casadiSolver():
    "Batched multiple-shooting NLP solver on the GPU (name kept from the reference for drop-in use)."

    TOLERANCE = 1e-8   # IPOPT default `tol`; the reference passes only max_iter (ocp.py:290)

    def __init__(self, train, track, optsDict={}, device=0, startingPoint='profile', restoration=True, watchdogTrigger=0):
        """
        Same arguments as the reference (ocp.py:79) plus two that have no counterpart there: `device` (GPU index) and
        `startingPoint`: 'reference' starts every solve from the reference's point (ocp.py:325-339), 'profile' (default)
        from a speed profile built on the device from the limits, the running time and the end speeds -- same optimum,
        about half the interior-point iterations; a scenario that breaks down from it is repeated from the reference's point.
        `restoration` (default True, IPOPT's behaviour): a solve whose filter line search breaks down enters the feasibility restoration
        phase on the device (every loss model, shooting integrator and horizon: in the follow-up kernels, csrc/msd_resto.hpp); False: it ends there
        ('Restoration_Failed') after the restart from the other
        starting point -- what a loop that handles failed scenarios itself wants (mseetc/mpc.py), since a hopeless scenario can spend
        hundreds of iterations in restoration and a launch lasts as long as its slowest scenario.
        """

        if startingPoint not in _device.START:
            raise ValueError("Unknown starting point '{}'!".format(startingPoint))

        self.startingPoint = startingPoint
        self.restoration = bool(restoration)
        self.watchdogTrigger = int(watchdogTrigger)      # IPOPT's watchdog_shortened_iter_trigger; 0 = its default (the reference's setting), < 0 = off
        self._optsDict = dict(optsDict)

        track.checkFields()
        train.checkFields()

        opts = OptionsCasadiSolver(optsDict)


        N = int(opts.numIntervals)

        # train parameters per kg of total mass (ocp.py:96-116)
        rho = train.rho
        totalMass = train.mass*rho

        withRgBrake = train.forceMin != 0
        withPnBrake = train.forceMinPn != 0

        accInf = 10  # acceleration bound when a limit is not defined

        forceMax = train.forceMax/totalMass if train.forceMax is not None else accInf
        forceMin = train.forceMin/totalMass if train.forceMin is not None else -accInf
        forceMinPn = train.forceMinPn/totalMass if train.forceMinPn is not None else -accInf

        powerMax = train.powerMax/totalMass if train.powerMax is not None else None
        powerMin = train.powerMin/totalMass if train.powerMin is not None else None

        accMax = min(accInf, train.accMax if train.accMax is not None else accInf)
        accMin = max(-accInf, -abs(train.accMin) if train.accMin is not None else -accInf)

        hasPower = powerMax is not None or powerMin is not None

        if hasPower:   # ocp.py:186-187
            pwUpper = powerMax if powerMax is not None else forceMax*train.velocityMax
            pwLower = 0 if not withRgBrake else powerMin if powerMin is not None else forceMin*train.velocityMax
        else:
            pwUpper = pwLower = 0.0

        # loss model -> slack rows (ocp.py:99, 225-226)
        if opts.energyOptimal:
            lossKind, ct, cr = classifyLosses(train.lossesCallable())
        else:
            lossKind, ct, cr = LOSS_NONE, 0.0, 0.0

        # shooting grid and profile on it (ocp.py:124-125, 195-196, 266-269)
        self.points = computeDiscretizationPoints(track, N)
        self.steps = np.diff(self.points.index)

        vlim = self.points['Speed limit [m/s]'].values
        bmax = np.zeros(N + 1)
        bmax[1:N] = np.minimum(np.minimum(vlim[1:N], train.velocityMax), vlim[0:N - 1])**2

        scaling = 3.6/(1e-6*totalMass) if opts.energyOptimal else track.length/train.velocityMax   # ocp.py:278,282

        model = train.exportModel()
        io = opts.integrationOptions

        # shooting integrator (ocp.py:92, train.py:294-322)
        if opts.integrationMethod == 'IRK':
            C, D = collocationTables(io.order, io.collMethod)
            integrator = ('IRK', io.order, io.maxIter, C, D)
        elif opts.integrationMethod == 'CVODES':
            integrator = ('CVODES', io.absTol, io.relTol)
        else:
            integrator = None

        # loss slacks from the loss power integrated over the running time of the interval (ocp.py:231-241)
        integrateLosses = bool(opts.integrateLosses) and bool(opts.energyOptimal)
        # (the loss rows of ocp.py:231-241 integrate the loss power along the time-domain model, train.py:367-413: with constant efficiencies that is a
        # multiple of the distance covered -- csrc/msd_lossint.hpp; with a loss table -- the dynamic loss model of efficiency.py, any tabulated loss
        # function -- the loss power L(F, v)/M itself, csrc/msd_lossint_table.hpp (round 6).  Next to any shooting integrator the loss integrals have
        # their own; the table runs with explicit Runge-Kutta shooting: simulations/figure6.py:178, the one place the reference names the combination)
        if integrateLosses and lossKind == LOSS_DYNAMIC and opts.integrationMethod != 'RK':
            raise NotImplementedError("integrateLosses=True with a loss table runs with integrationMethod='RK'.")
        if integrateLosses and lossKind == LOSS_NONE:
            integrateLosses = False      # perfect efficiency: both loss integrals vanish and the rows reduce to s >= 0

        numSteps = io.numSteps if opts.integrationMethod != 'CVODES' else 1
        numApproxSteps = io.numApproxSteps if opts.integrationMethod != 'CVODES' else 0     # train.py:314

        self._desc = _device.make_desc(
            N, withPnBrake, hasPower, opts.energyOptimal, numSteps, numApproxSteps, lossKind, int(opts.maxIterations),
            (model.sr0, model.sr1, model.sr2), train.g, rho, forceMax, forceMin if withRgBrake else 0.0, forceMinPn,
            abs(pwUpper), abs(pwLower), accMin, accMax, ct, cr, float(opts.minimumVelocity)**2, scaling, self.TOLERANCE,
            self.steps, self.points['Gradient [permil]'].values[:N]/1e3, self.points['Curvature [1/m]'].values[:N], bmax,
            lossTable=train.lossesCallable().parameters(totalMass) if lossKind == LOSS_DYNAMIC else None, start=startingPoint, integrator=integrator,
            integrateLosses=integrateLosses, restoration=restoration, watchdogTrigger=watchdogTrigger)

        self._device = device
        self._problem = None   # created on first use: construction stays possible on a machine without GPU

        self.totalMass = totalMass
        self.velocityMin = opts.minimumVelocity
        self.numIntervals = N
        self.withRgBrake = withRgBrake
        self.withPnBrake = withPnBrake
        self.train = train
        self.track = track
        self._vmaxTrain = float(train.velocityMax)
        self.energyOptimal = opts.energyOptimal
        self.scalingFactorObjective = scaling
        self.opts = opts

    # ---- device problem ---------------------------------------------------------------

    @property
    def problem(self):

        if self._problem is None:
            self._problem = _device.DeviceProblem(self._desc, self._device)

        return self._problem

    def _handles(self, devices):
        "Device problems for a multi-device solve: the solver's own handle when its device leads the list, one more per further entry (kept)."

        devices = [int(d) for d in devices]
        if not devices:
            raise ValueError("devices must name at least one device!")
        pool = self.__dict__.setdefault('_pool', [])
        want = []
        used = set()
        for d in devices:
            hit = next((k for k, (dev, _) in enumerate(pool) if dev == d and k not in used), None)
            if hit is None:
                pool.append((d, _device.DeviceProblem(self._desc, d)))
                hit = len(pool) - 1
            used.add(hit)
            want.append(pool[hit][1])
        return want[0], want[1:]

    def adoptDevice(self, other):
        """
        Take over the device handle of another solver (which becomes unusable) instead of creating a new one: the next problem of a
        receding-horizon loop reuses the stream and the device buffers of the previous one (msd_problem_reconfigure).
        """

        if other is not None and other._problem is not None and other._device == self._device and self._problem is None:
            self._problem, other._problem = other._problem.reconfigure(self._desc), None

        return self

    def close(self):

        if self._problem is not None:
            self._problem.close()
            self._problem = None

        for _, prob in self.__dict__.pop('_pool', []):
            prob.close()

        twin = self.__dict__.pop('_twin', None)
        if twin is not None:
            twin.close()

    # ---- scenarios --------------------------------------------------------------------------

    def _scenarios(self, terminalTime, initialTime, terminalVelocity, initialVelocity):
        "(B,4) records (t0, T, v0^2, vN^2) with the reference's checks and clipping (ocp.py:314-320, 343-344)."

        T, t0, vN, v0 = np.broadcast_arrays(*[np.atleast_1d(np.asarray(a, dtype=float)) for a in
                                               (terminalTime, initialTime, terminalVelocity, initialVelocity)])

        if np.any(~np.isfinite(t0)) or np.any(t0 < 0):
            raise ValueError("Initial time must be a positive number, not {}!".format(initialTime))

        if np.any(~np.isfinite(T)) or np.any(T <= 0):
            raise ValueError("Terminal time must be a strictly positive number, not {}!".format(terminalTime))

        vlim = self.points['Speed limit [m/s]'].values

        v0 = np.minimum(np.maximum(v0, self.velocityMin), vlim[0])
        vN = np.minimum(np.maximum(vN, self.velocityMin), vlim[-1])

        return np.stack([t0, T, v0**2, vN**2], axis=1)

    def _overrides(self, B, mass, r0, r1, r2):
        """
        Per-scenario rolling stock: what a new `Train(config={'mass': ..., 'rolling resistance r0': ...})` + `casadiSolver`
        would change (train.py:44-62, ocp.py:96-116, 278).  Limits given in newtons / watts stay, specific ones follow the mass.
        """

        if mass is None and r0 is None and r1 is None and r2 is None:
            return None

        tr = self.train
        full = lambda a, default: np.broadcast_to(np.asarray(default if a is None else a, dtype=float), (B,)).copy()
        mass, r0, r1, r2 = full(mass, tr.mass), full(r0, tr.r0), full(r1, tr.r1), full(r2, tr.r2)

        if np.any(mass <= 0) or np.any(r0 < 0) or np.any(r1 < 0) or np.any(r2 < 0):
            raise ValueError("Train mass must be positive and rolling resistance coefficients non-negative!")

        M = mass*tr.rho
        d = self._desc
        ratio = self.totalMass/M      # specific bounds scale with 1/M; accInf-defaulted bounds (no limit given) do not
        OV = _device.OV
        out = np.zeros((B, OV['COUNT']))
        out[:, OV['SR0']], out[:, OV['SR1']], out[:, OV['SR2']] = r0/M, r1/M, r2/M
        out[:, OV['F_MAX']] = d.f_max*ratio if tr.forceMax is not None else d.f_max
        out[:, OV['F_MIN']] = d.f_min*ratio if tr.forceMin is not None else d.f_min
        out[:, OV['F_MIN_PN']] = d.f_min_pn*ratio if tr.forceMinPn is not None else d.f_min_pn
        # power rows: P/M when a power limit is set, else force*vmax (ocp.py:186-187)
        out[:, OV['PW_UPPER']] = d.pw_upper*ratio if (tr.powerMax is not None or tr.forceMax is not None) else d.pw_upper
        out[:, OV['PW_LOWER']] = d.pw_lower*ratio if (tr.powerMin is not None or tr.forceMin is not None) else d.pw_lower
        out[:, OV['OBJ_DEN']] = 3.6/(1e-6*M) if self.energyOptimal else d.obj_den
        out[:, OV['TOTAL_MASS']] = M      # the dynamic loss model maps specific forces to newtons with it (efficiency.py:108)

        return out

    def solveBatch(self, terminalTime, initialTime=0, terminalVelocity=1, initialVelocity=1, multipliers=False,
                   mass=None, r0=None, r1=None, r2=None, guess=None, warmMu=1e-2, warmPush=1e-3, devices=None, classifyFailures=True, shift=None):
        """
        Solve many scenarios of this problem in one launch.  Arguments broadcast against each other; `mass`, `r0`, `r1`, `r2`
        (SI units, scalars or one value per scenario) perturb the rolling stock per scenario.  `guess` (B, nz) or (nz,), in the
        layout of 'z', warm-starts the solves (barrier parameter `warmMu`, interior push `warmPush`); without it every solve
        cold-starts like the reference (ocp.py:325-339).
        Returns dict: 'z' (B, nz) in the reference's variable layout, 'status' (B,), 'iterations' (B,), 'cost' (B,)
        [kWh or s], 'stats' (raw records), 'kernel_ms', optionally 'lam_g'.
        """

        scen = self._scenarios(terminalTime, initialTime, terminalVelocity, initialVelocity)

        B = max([scen.shape[0]] + [np.size(a) for a in (mass, r0, r1, r2) if a is not None])
        if scen.shape[0] != B:
            scen = np.broadcast_to(scen, (B, scen.shape[1])).copy()

        if guess is not None:
            guess = np.asarray(guess, dtype=float)
            nz = (4 + int(self.withPnBrake))*self.numIntervals + 2
            if guess.shape[-1] != nz or guess.ndim > 2 or (guess.ndim == 2 and guess.shape[0] not in (1, B)):
                raise ValueError("Warm-start guess must have shape ({}, {}) or ({},)!".format(B, nz, nz))
            if not np.all(np.isfinite(guess)):
                raise ValueError("Warm-start guess must be finite!")
            guess = np.broadcast_to(guess.reshape(-1, nz), (B, nz))

        if devices is None:
            out = self.problem.solve_batch(scen, want_multipliers=multipliers, overrides=self._overrides(B, mass, r0, r1, r2),
                                           guess=guess, warmMu=warmMu, warmPush=warmPush, shift=shift)
        else:
            # one handle per entry of `devices` (an index may repeat: two streams on one device), contiguous slices, no collective
            first, others = self._handles(devices)
            out = first.solve_batch_multi(others, scen, want_multipliers=multipliers, overrides=self._overrides(B, mass, r0, r1, r2),
                                          guess=guess, warmMu=warmMu, warmPush=warmPush)

        st = out['stats']
        ST = _device.ST

        if self.energyOptimal and classifyFailures:
            self._classify_failures(scen, st, self._overrides(B, mass, r0, r1, r2))

        cost = st[:, ST['OBJ']]*(1.0 if self.energyOptimal else self.scalingFactorObjective)   # ocp.py:361

        return dict(z=out['z'], status=st[:, ST['STATUS']].astype(int), iterations=st[:, ST['ITERS']].astype(int), cost=cost,
                    stats=st, kernel_ms=out['kernel_ms'], lam_g=out['lam_g'], scenarios=scen)

    def minimumTime(self, scen, overrides=None):
        """
        Minimum running times of the scenarios `scen` (rows t0, T, v0^2, vN^2; T is ignored): the time-optimal twin of the problem
        (same track, train, transcription and integrator, energyOptimal=False, ocp.py:146-150) solved on the device, with the
        scenarios' own rolling stock (`overrides`, rows of _overrides()).  Returns (tmin (B,), ok (B,) bool).
        """

        twin = self.__dict__.get('_twin')

        if twin is None:
            opts = dict(self._optsDict)
            opts['energyOptimal'] = False
            opts.pop('integrateLosses', None)      # (the loss slacks do not exist in the time-optimal problem)
            twin = self._twin = casadiSolver(self.train, self.track, opts, device=self._device, startingPoint='profile', restoration=self.restoration, watchdogTrigger=self.watchdogTrigger)

        sub = np.atleast_2d(np.asarray(scen, dtype=float))
        loose = sub.copy()
        # a running time the twin can certainly meet: three times the run at top speed or three times the one asked for (a much looser
        # bound only costs iterations: the profile start of the twin uses up the time it is given)
        loose[:, 1] = sub[:, 0] + np.maximum(3*self.track.length/self._vmaxTrain, 3*(sub[:, 1] - sub[:, 0]))

        ov = None
        if overrides is not None:
            # the twin's objective is scaled by a constant of the problem (ocp.py:282), not by the mass
            ov = np.array(overrides, dtype=float, copy=True)
            ov[:, _device.OV['OBJ_DEN']] = twin._desc.obj_den

        out = twin.problem.solve_batch(loose, overrides=ov)
        st = out['stats']
        # usable: the twin converged -- or it ended without converging on a feasible point next to its optimum: far down the central path (mu <= 1e-5) or
        # with an optimality error of 1e-4 (its primal point settles long before its multipliers do where both brakes share an active acceleration bound:
        # DESIGN.md section 8).  Feasibility alone is not enough (round 6): the twin is given three times the running time asked for and its profile start
        # uses that time up, so a twin that breaks down early ends on a feasible point whose time says nothing about the minimum -- a scenario was declared
        # late on such a time and its arrival moved by up to a factor of three.  (_classify_failures only calls a running time infeasible when the twin
        # converged; csrc/msd_mpc.hip: mpc_relax applies the same rule on the device)
        viol = st[:, _device.ST['CONSTR_VIOL']]
        near = (st[:, _device.ST['MU']] <= 1e-5) | (st[:, _device.ST['KKT']] <= 1e-4)
        ok = (st[:, _device.ST['STATUS']] >= 0) | (np.isfinite(viol) & (viol <= 1e-6) & near)
        self._twinConverged = st[:, _device.ST['STATUS']] >= 0

        return out['z'][:, -2] - sub[:, 0], ok

    def _classify_failures(self, scen, st, overrides=None):
        """
        IPOPT ends a solve whose constraints cannot be met in its restoration phase with 'Infeasible_Problem_Detected'
        (ocp.py:362-370 prints that status).  The device's restoration phase (csrc/msd_resto.hpp) reports that status itself when it
        converges; on this problem class it usually breaks down first ('Restoration_Failed': its 1-norm objective leaves the distribution
        of the missing time over the intervals open).  The host adds an exact certificate
        for the one infeasibility this problem class knows -- a running time below the minimum: the time-optimal twin of the
        problem is solved for the scenarios that broke down (with their own rolling stock), and those whose minimum running
        time exceeds their T are marked infeasible.  Everything else keeps its status.
        """

        ST = _device.ST
        # (the iteration limit is a verdict of its own -- unless the solve ran into it inside or between restoration phases)
        failed = np.flatnonzero((st[:, ST['STATUS']] < 0) & ((st[:, ST['STATUS']] != _device.STATUS_MAXITER) | (st[:, ST['N_RESTO']] > 0)))

        if failed.size == 0:
            return

        sub = scen[failed]
        tmin, ok = self.minimumTime(sub, None if overrides is None else overrides[failed])
        short = ok & self._twinConverged & (tmin > (sub[:, 1] - sub[:, 0])*(1 + 1e-8))      # (a verdict of infeasibility needs the minimum itself, not an upper bound of it)
        st[failed[short], ST['STATUS']] = _device.STATUS_INFEASIBLE

    def unpack(self, z):
        "z (reference layout, ocp.py:376-405) -> DataFrame indexed by time."

        N = self.numIntervals
        stp = 4 + int(self.withPnBrake)
        body = np.asarray(z[:stp*N]).reshape(N, stp)
        pn = int(self.withPnBrake)

        nan = np.array([np.nan])
        Fel = np.concatenate([body[:, 0], nan])
        Fpb = np.concatenate([body[:, 1], nan]) if pn else np.zeros(N + 1)
        s = np.concatenate([body[:, 1 + pn], nan])
        t = np.concatenate([body[:, 2 + pn], [z[stp*N]]])
        b = np.concatenate([body[:, 3 + pn], [z[stp*N + 1]]])

        # (one construction: every column assigned to an existing frame costs a tenth of a millisecond)
        return pd.DataFrame({'Position [m]': self.points.index.values, 'Velocity [m/s]': np.sqrt(b), 'Force (el) [N]': Fel*self.totalMass,
                             'Force (pnb) [N]': Fpb*self.totalMass, 'Slacks': s*self.totalMass}, index=pd.Index(t, name='Time [s]'))

    def solve(self, terminalTime, initialTime=0, terminalVelocity=1, initialVelocity=1):
        "One scenario; returns (DataFrame or None, stats) like the reference (ocp.py:310-409)."

        if not isinstance(initialTime, (int, float)) or initialTime < 0:
            raise ValueError("Initial time must be a positive number, not {}!".format(initialTime))

        if not isinstance(terminalTime, (int, float)) or terminalTime <= 0:
            raise ValueError("Terminal time must be a strictly positive number, not {}!".format(terminalTime))

        res = self.solveBatch(terminalTime, initialTime, terminalVelocity, initialVelocity)

        status = int(res['status'][0])

        stats = {'Solver status': _device.STATUS_TEXT.get(status, str(status)), 'IP iterations': int(res['iterations'][0]),
                 'CPU time [s]': res['kernel_ms']*1e-3, 'Cost': float(res['cost'][0])}

        if status < 0:

            print("Solver failed with status '{}'".format(stats['Solver status']))

            return None, stats

        print("Solver converged in {:4d} iterations.".format(stats['IP iterations']))

        df = postProcessDataFrame(self.unpack(res['z'][0]), self.points, self.train, device=self._device)   # CVODES=True like ocp.py:407

        return df, stats


OCP = casadiSolver   # BASELINE.json's north_star calls the class OCP; the reference's name is casadiSolver (ocp.py:77)
