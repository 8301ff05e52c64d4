"""
Rolling-stock description for the MI355X train-control solver.

`Train` mirrors the reference loader (`mseetc/train.py:9-219`): a plain, mutable
attribute bag filled from a TTOBench json file, which callers are free to edit
between solver constructions (figure5.py:88, figure10.py:17-22).  The solver
reads the attributes at construction time.  `TrainModel` carries the specific
Davis coefficients (train.py:175-187, 225-277); the ODE itself is integrated on
the device (csrc/), `TrainIntegrator` exposes single-interval integration through
the same C ABI.  Integrator option classes: train.py:457-534.
"""

import json
import math
from pathlib import Path

import numpy as np

from .utils import Options, StaticLosses, checkTTOBenchVersion, closedFormLosses, convertUnit, splitLosses

_DATA_DIR = Path(__file__).resolve().parent.parent / 'data'

# (json key, attribute, sign, mandatory): sign -1 stores -abs(value) (train.py:81-91)
_FIELDS = (
    ('max traction force', 'forceMax', +1),
    ('max reg braking force', 'forceMin', -1),
    ('max pn braking force', 'forceMinPn', -1),
    ('max traction power', 'powerMax', +1),
    ('max reg braking power', 'powerMin', -1),
    ('max acceleration', 'accMax', +1),
    ('max deceleration', 'accMin', -1),
)


def _quantity(entry, sign=+1):
    val = entry['value']
    return convertUnit(-abs(val) if sign < 0 else val, entry['unit'])


class Train():

    def __init__(self, config, pathJSON=_DATA_DIR / 'trains') -> None:

        self.g = 9.81  # acceleration of gravity [m/s^2]

        if not isinstance(config, dict):
            raise ValueError("Train configuration should be provided as a dictionary!")

        if 'id' not in config:
            raise ValueError("Train ID must be specified in configuration!")

        with open(Path(pathJSON) / (config['id'] + '.json')) as file:
            data = json.load(file)

        checkTTOBenchVersion(data, ['1.1', '1.2', '1.3'])

        # config overrides: None drops a limit, {'unit','value'} replaces it (train.py:36-66);
        # the caller's dict is left untouched (the reference pops 'id' from it)
        optional = {"max acceleration", "max deceleration"}
        overrides = {k: v for k, v in config.items() if k != 'id'}
        used = set()

        for key, val in overrides.items():

            if val is None and key in data:
                del data[key]
                used.add(key)
                continue

            if not isinstance(val, dict) or set(val.keys()) != {'unit', 'value'}:
                raise ValueError("Configuration field '{}' should be specified as a dictionary with 'unit' and 'value' keys!".format(key))

            if key in data or key in optional:
                data[key] = val
                used.add(key)

        if set(overrides) != used:
            raise ValueError("Redundant fields in train configuration: {}!".format(', '.join(set(overrides) - used)))

        self.mass = _quantity(data['mass'])  # [kg]

        self.rho = _quantity(data['rho'])  # rotating-mass factor [-]

        if self.rho < 1:
            self.rho += 1  # 6% -> 0.06 -> 1.06

        self.velocityMax = _quantity(data['max speed'])  # [m/s]

        for key, attr, sign in _FIELDS:
            setattr(self, attr, _quantity(data[key], sign) if key in data else None)

        self.r0 = _quantity(data['rolling resistance r0'])  # [N]
        self.r1 = _quantity(data['rolling resistance r1'])  # [N/(m/s)]
        self.r2 = _quantity(data['rolling resistance r2'])  # [N/(m/s)^2]

        hasT, hasR = 'efficiency traction' in data, 'efficiency reg brake' in data

        if hasT or hasR:

            if not (hasT and hasR):
                raise ValueError("Both efficiencies need to be specified in json file!")

            self.etaTraction = _quantity(data['efficiency traction'])
            self.etaRgBrake = _quantity(data['efficiency reg brake'])

        self.checkFields()

    def checkFields(self):
        "Same rules and messages as the reference (train.py:116-172), table driven."

        inf = lambda x: x is not None and np.isinf(x)

        if self.mass is None or self.mass < 0 or inf(self.mass):
            raise ValueError("Train mass must be a positive number, not {}!".format(self.mass))

        if self.g is None or not 9 <= self.g <= 10:
            raise ValueError("Acceleration of gravity must be between 9 and 10 m/s^2, not {}!".format(self.g))

        if self.rho is None or not 1 <= self.rho <= 1.5:
            raise ValueError("Rotation mass factor must be between 1 and 1.5, not {}!".format(self.rho))

        if self.velocityMax is None or self.velocityMax <= 0 or inf(self.velocityMax):
            raise ValueError("Maximum velocity must be a strictly positive number, not {}!".format(self.velocityMax))

        # (attribute, admissible if None, predicate of a bad value, message)
        rules = (
            ('forceMax', lambda x: x <= 0, "Maximum traction force must be strictly positive or free (None), not {}!"),
            ('forceMinPn', lambda x: x > 0, "Maximum pneumatic braking force must be negative, zero or free (None), not {}!"),
            ('forceMin', lambda x: x > 0, "Maximum regenerative braking force must be negative, zero or free (None), not {}!"),
        )

        for name, bad, msg in rules:
            val = getattr(self, name)
            if val is not None and (bad(val) or inf(val)):
                raise ValueError(msg.format(val))

        if self.forceMin == 0 and self.forceMinPn == 0:
            raise ValueError("Both brakes cannot be deactivated simultaneously!")

        rules = (
            ('powerMax', lambda x: x <= 0, "Maximum traction power must be strictly positive or free (None), not {}!"),
            ('powerMin', lambda x: x >= 0, "Maximum regenerative brake power must be strictly negative or free (None), not {}!"),
            ('accMax', lambda x: x <= 0, "Maximum acceleration must be strictly positive or free (None), not {}!"),
            ('accMin', lambda x: x >= 0, "Maximum deceleration must be strictly negative or free (None), not {}!"),
        )

        for name, bad, msg in rules:
            val = getattr(self, name)
            if val is not None and (bad(val) or inf(val)):
                raise ValueError(msg.format(val))

        for name in ('r0', 'r1', 'r2'):
            coef = getattr(self, name)
            if coef is None or coef < 0:
                raise ValueError("Rolling resistance coefficient {} must be positive, not {}!".format(name, coef))

    def exportModel(self):
        "Specific (per kg of mass*rho) Davis coefficients + constants needed by the integrator."

        totalMass = self.mass*self.rho

        return TrainModel(self.r0/totalMass, self.r1/totalMass, self.r2/totalMass, self.rho, self.g, self.forceMinPn != 0)

    def lossesCallable(self):
        "The train's power-loss function L(F [N], v [m/s]) -> [W]: explicit attribute or the two efficiencies."

        if hasattr(self, 'powerLosses'):

            fun = self.powerLosses

            if closedFormLosses(fun) is not None:
                return fun

            # any other function of (F, v): tabulated over the operating range of this train (efficiency.TabulatedLosses), once per
            # function and set of limits; `lossesTableSize = (numForce, numVelocity)` refines the grid
            from .efficiency import TabulatedLosses

            size = tuple(getattr(self, 'lossesTableSize', (24, 32)))
            # a limit the train does not have (None: ocp.py:104-108 bounds the specific force by accInf = 10 then) is that bound in newtons here
            accInf = 10.0
            fmax = self.forceMax if self.forceMax is not None else accInf*self.mass*self.rho
            fmin = self.forceMin if self.forceMin is not None else -accInf*self.mass*self.rho
            if self.velocityMax is None:
                raise ValueError("A tabulated loss function needs the maximum velocity of the train!")
            key = (fmin, fmax, self.velocityMax, size)
            cached = getattr(self, '_lossesTable', None)

            if cached is None or cached[0] is not fun or cached[1] != key:
                cached = (fun, key, TabulatedLosses(fun, fmin, fmax, self.velocityMax, numForce=size[0], numVelocity=size[1]))
                self._lossesTable = cached

            return cached[2]

        if hasattr(self, 'etaTraction') and hasattr(self, 'etaRgBrake'):
            return StaticLosses(self.etaTraction, self.etaRgBrake)

        raise ValueError("Power losses function of train must by either explicitly or implicitly defined!")

    def powerLossesFuns(self, split=True):
        "Specific power losses [W/kg] as function(s) of specific force [N/kg] and speed (train.py:190-219)."

        fun = self.lossesCallable()
        totalMass = self.mass*self.rho

        def specific(f, v):
            return (1/totalMass)*fun(f*totalMass, v)

        return splitLosses(specific) if split else specific


class TrainModel():
    "Data of the space-domain train ODE (train.py:225-277): dt/ds = 1/sqrt(b), db/ds = 2 a(b,u)."

    def __init__(self, sr0, sr1, sr2, rho=1, g=9.81, withPnBrake=True) -> None:

        self.sr0, self.sr1, self.sr2 = sr0, sr1, sr2
        self.rho = rho
        self.g = g
        self.withPnBrake = withPnBrake

    def resistance(self, gradient=0.0, curvature=0.0):
        "Velocity-independent specific resistance g*grad/rho + cr/rho (train.py:252-254)."

        c = abs(curvature)
        cr = self.g*0.5*c/(1 - 30*c) if c <= 1/300 else self.g*0.65*c/(1 - 55*c)

        return self.g*gradient/self.rho + cr/self.rho

    def acceleration(self, velocitySquared, traction=0.0, pnBrake=0.0, gradient=0.0, curvature=0.0):
        "Instantaneous acceleration [m/s^2] (train.py:251-254, accelerationFun :267)."

        rr = self.sr0 + self.sr1*math.sqrt(velocitySquared) + self.sr2*velocitySquared

        return traction + (pnBrake if self.withPnBrake else 0) - rr - self.resistance(gradient, curvature)


def collocationPoints(order, scheme='radau'):
    """
    casadi.collocation_points(order, scheme) restated: the `order` roots, in (0, 1], of the shifted Legendre polynomial
    ('legendre', Gauss points) or of P_{order-1} - P_order ('radau', Radau IIA points, the last one is 1).
    """

    if int(order) != order or not 1 <= order <= 9:
        raise ValueError("Order of implicit Runge-Kutta should be a positive integer between 1 and 9!")

    d = int(order)
    leg = np.polynomial.legendre

    if scheme == 'legendre':
        x = leg.leggauss(d)[0]
    elif scheme == 'radau':
        coef = np.zeros(d + 1)
        coef[d - 1], coef[d] = 1.0, -1.0
        x = np.sort(leg.legroots(coef).real)
        # two Newton steps on P_{d-1} - P_d polish the companion-matrix roots to working precision
        for _ in range(2):
            x = x - leg.legval(x, coef)/leg.legval(x, leg.legder(coef))
        x[-1] = 1.0
    else:
        raise ValueError("Unknown collocation method: {}!".format(scheme))

    return list(0.5*(np.sort(x) + 1.0))


def collocationTables(order, scheme='radau'):
    """
    casadi.collocation_interpolators on the points {0} + collocationPoints: C[r][j] = dL_r/dtau(tau_j), D[r] = L_r(1) for the
    Lagrange polynomials L_r of those order + 1 points.
    """

    tau = np.array([0.0] + collocationPoints(order, scheme))
    n = len(tau)
    C, D = np.zeros((n, n)), np.zeros(n)

    for r in range(n):
        others = np.delete(tau, r)
        basis = np.poly1d(others, r=True)/np.prod(tau[r] - others)
        D[r] = basis(1.0)
        C[r, :] = basis.deriv()(tau)

    return C, D


class TrainIntegrator():
    """
    One shooting interval (reference: train.py:280-364), evaluated on the device.  'RK' is the explicit Runge-Kutta map of the
    OCP transcription (with sensitivities, csrc/msd_kernel.hpp); 'IRK' (collocation, casadi.simpleIRK) and 'CVODES' (adaptive
    integration to tolerances) evaluate single intervals like simulations/figure4.py does (csrc/msd_integrators.hip).
    """

    def __init__(self, model, solver, optsDict={}) -> None:

        if solver not in {'RK', 'IRK', 'CVODES'}:
            raise ValueError("Unknown integration method!")

        self.model = model
        self.solver = solver

        if solver == 'RK':
            self.opts = OptionsRK(optsDict)
        elif solver == 'IRK':
            self.opts = OptionsIRK(optsDict)
            C, D = collocationTables(self.opts.order, self.opts.collMethod)
            self._params = np.concatenate([[self.opts.order, self.opts.numSteps, self.opts.numApproxSteps, self.opts.maxIter], C.ravel(), D])
        else:
            self.opts = OptionsCVODES(optsDict)
            self.opts.numApproxSteps = 0     # train.py:317
            self._params = np.array([self.opts.absTol, self.opts.relTol])

    def solve(self, time, velocitySquared, ds, traction=0, pnBrake=0, gradient=0, curvature=0):

        if not self.model.withPnBrake and pnBrake != 0:
            raise ValueError("Cannot define value for pneumatic braking when this brake is deactivated!")

        from . import _device

        if self.solver == 'RK':
            out = _device.stage_eval(self.model, self.opts, [time], [velocitySquared], [ds], [traction + pnBrake], [gradient], [curvature])
        else:
            out = _device.interval_integrate(self.model, 1 if self.solver == 'CVODES' else 2, self._params, [time], [velocitySquared], [ds],
                                             [traction + pnBrake], [gradient], [curvature])

        return {'time': float(out['time'][0]), 'velSquared': float(out['velSquared'][0])}

    def initRollingResistance(self, solver='CVODES'):
        "Integrator of the energy dissipated by the rolling resistance (train.py:416-442): 'CVODES' (adaptive pair at 1e-8 / 1e-6) or 'RK' (two steps of RK4)."

        if solver not in {'RK', 'CVODES'}:
            raise ValueError("Unknown solver!")

        # train.py:436 (tolerances of the CVODES call) / train.py:431 (casadi.simpleRK(fun, 2, 4): (-numSteps, 0) selects the fixed steps on the device)
        self._rollingParams = np.array([1e-8, 1e-6]) if solver == 'CVODES' else np.array([-2.0, 0.0])

    def calcRollingResistance(self, velocity, ds, traction=0, pnBrake=0, gradient=0, curvature=0):
        "(specific energy [J/kg] lost to the rolling resistance over ds, velocity at the end of the interval) (train.py:445-454)."

        if not hasattr(self, '_rollingParams'):
            raise ValueError("Call initRollingResistance first!")

        from . import _device

        velocity = np.atleast_1d(np.asarray(velocity, dtype=float))
        out = _device.interval_integrate(self.model, 3, self._rollingParams, 0.0, velocity**2, ds,
                                         np.asarray(traction, dtype=float) + (np.asarray(pnBrake, dtype=float) if self.model.withPnBrake else 0.0),
                                         gradient, curvature)
        losses, vEnd = out['time'], np.sqrt(out['velSquared'])

        return (float(losses[0]), float(vEnd[0])) if losses.size == 1 else (losses, vEnd)

    def solveMany(self, time, velocitySquared, ds, force, gradient=0.0, curvature=0.0):
        "Arrays of independent intervals in one launch (no counterpart in the reference); force = traction + pnBrake."

        from . import _device

        if self.solver == 'RK':
            n = len(np.atleast_1d(velocitySquared))
            full = lambda a: np.broadcast_to(np.asarray(a, dtype=float), (n,))
            out = _device.stage_eval(self.model, self.opts, full(time), full(velocitySquared), full(ds), full(force), full(gradient), full(curvature))
        else:
            out = _device.interval_integrate(self.model, 1 if self.solver == 'CVODES' else 2, self._params, time, velocitySquared, ds, force,
                                             gradient, curvature)

        return {'time': out['time'], 'velSquared': out['velSquared']}


class OptionsRK(Options):

    def __init__(self, paramsDict):

        # defaults of the reference's OptionsRK (mseetc/train.py:461-465)
        self.order = 4             # classic RK4 is the only explicit scheme (casadi.simpleRK)
        self.numSteps = 1          # equal sub-steps per shooting interval
        self.numApproxSteps = 0    # 0: (t, b) integrated jointly; k > 0: b only, time by the trapezoidal rule on k pieces
        super().__init__(paramsDict)

    def checkValues(self):

        if self.order != 4:
            raise ValueError("Only explicit Runge-Kutta of order 4 is currently implemented in casadi!")

        self.checkPositiveInteger(self.numSteps, 'Number of integration steps', allowZero=False)

        self.checkPositiveInteger(self.numApproxSteps, 'Number of time approximation steps', allowZero=True)


class OptionsIRK(Options):

    def __init__(self, paramsDict):

        self.order = 2
        self.numSteps = 1
        self.numApproxSteps = 0
        self.collMethod = 'radau'
        self.maxIter = 10
        self.jit = False

        super().__init__(paramsDict)

    def checkValues(self):

        if int(self.order) != self.order or not 1 <= self.order <= 9:
            raise ValueError("Order of implicit Runge-Kutta should be a positive integer between 1 and 9!")

        self.checkPositiveInteger(self.numSteps, 'Number of integration steps', allowZero=False)

        self.checkPositiveInteger(self.numApproxSteps, 'Number of time approximation steps', allowZero=True)

        if self.collMethod not in {'radau', 'legendre'}:
            raise ValueError("Unknown collocation method: {}!".format(self.collMethod))

        self.checkPositiveInteger(self.maxIter, 'Maximum number of iterations', allowZero=False)

        if not isinstance(self.jit, bool):
            raise ValueError("JIT option must be a boolean!")


class OptionsCVODES(Options):

    def __init__(self, paramsDict):

        self.absTol = 1e-8
        self.relTol = 1e-6

        super().__init__(paramsDict)

    def checkValues(self):

        self.checkBounds(self.absTol, 'Absolute tolerance', 1e-20, 1e-1)
        self.checkBounds(self.relTol, 'Relative tolerance', 1e-20, 1e-1)
