"""
Host-side helpers of the MI355X train-control solver: option containers, unit
handling, TTOBench version check, loss-model descriptors and the energy
accounting that is used as the parity metric.

Mirrors the public surface of the reference's `mseetc/utils.py`
(`Options` :45-107, `convertUnit` :367-438, `checkTTOBenchVersion` :339-364,
`splitLosses` :197-220, `postProcessDataFrame` :223-336) without CasADi: loss
models are plain Python callables on the host and small parameter records on
the device.
"""

import math
import re

import numpy as np
import pandas as pd

# --------------------------------------------------------------------------
# units (reference: utils.py:367-438).  NOTE the reference quirk: 'km' DIVIDES
# by 1e3 (utils.py:376-378); shipped data only uses 'm'.  Reproduced as is.
# --------------------------------------------------------------------------

_UNIT_FACTORS = {
    'm': 1.0, 'm/s': 1.0, 'permil': 1.0, 'kg': 1.0, 'W': 1.0, 'N': 1.0, 'm/s^2': 1.0, '-': 1.0,
    'N/(m/s)': 1.0, 'N/(m/s)^2': 1.0, 'kg/m': 1.0,
    'km': 1.0/1e3,
    't': 1e3, 'kW': 1e3, 'MW': 1e6, 'kN': 1e3, 'kN/(m/s)': 1e3, 'kN/(m/s)^2': 1e3, 't/m': 1e3,
}

_IDENTITY_UNITS = {'m', 'm/s', 'permil', 'kg', 'W', 'N', 'm/s^2', '-', 'N/(m/s)', 'N/(m/s)^2', 'kg/m'}


def convertUnit(value, unit):
    "Convert `value` given in `unit` to the internally used SI-style unit."

    if unit in _IDENTITY_UNITS:
        return value                      # untouched (keeps ints ints, like the reference)
    if unit == 'km':
        return value/1e3
    if unit == 'km/h':
        return value/3.6
    if unit == '%':
        return value/100
    if unit == 'kN/(km/h)':
        return value*1e3*3.6
    if unit == 'N/(km/h)':
        return value*3.6
    if unit == 'kN/(km/h)^2':
        return value*1e3*3.6**2
    if unit == 'N/(km/h)^2':
        return value*3.6**2
    if unit in _UNIT_FACTORS:
        return value*_UNIT_FACTORS[unit]

    raise ValueError("Unknown unit: {}!".format(unit))


def checkTTOBenchVersion(jsonDict, supportedVersions):
    "Raise unless the json file advertises one of the supported TTOBench versions."

    if not isinstance(supportedVersions, list) or not all(isinstance(x, str) for x in supportedVersions):
        raise TypeError("'supportedVersions' must be specified a list of strings!")

    meta = jsonDict.get('metadata') if isinstance(jsonDict, dict) else None

    if not isinstance(meta, dict) or 'library version' not in meta:
        raise ValueError("Library version not found in json file!")

    found = re.search(r'v([\d.]+)', meta['library version'])

    if not found:
        raise ValueError("Unexpected format of 'library version' in json file!")

    if found.group(1) not in supportedVersions:
        raise ValueError("Import function works only for library versions {}!".format(','.join(supportedVersions)))


# --------------------------------------------------------------------------
# options (reference: utils.py:45-107)
# --------------------------------------------------------------------------

class Options():
    """
    Attribute-bag options: subclasses set their defaults as attributes and then
    call `super().__init__(paramsDict)`; unknown keys raise ValueError, nested
    Options are overwritten recursively, `checkValues` validates.
    """

    def __init__(self, paramsDict):

        self.overwriteDefaults(paramsDict)
        self.checkValues()

    def checkValues(self):
        pass

    def checkPositiveInteger(self, num, fieldName, allowZero=True):

        ok = (int(num) == num) and (num >= 0 if allowZero else num > 0)

        if not ok:
            raise ValueError("{} must be a {} positive integer!".format(fieldName, 'strictly' if not allowZero else ''))

    def checkBounds(self, num, fieldName, lowerBound, upperBound):

        if not lowerBound <= num <= upperBound:
            raise ValueError("{} must be between {} and {}!".format(fieldName, lowerBound, upperBound))

    def overwriteDefaults(self, paramsDict):

        for key, val in paramsDict.items():

            if not hasattr(self, key):
                raise ValueError("Specified option ({}) does not exist!".format(key))

            cur = getattr(self, key)

            if isinstance(cur, Options):

                if not isinstance(val, dict):
                    raise ValueError("Nested options must be specified as a dictionary!")

                cur.overwriteDefaults(val)

            else:
                setattr(self, key, val)

    def toDict(self):

        out = {}

        for name, val in vars(self).items():

            if name.startswith('__') or name == 'ignoreFields' or callable(val):
                continue

            out[name] = val.toDict() if isinstance(val, Options) else val

        return out


# --------------------------------------------------------------------------
# loss models
# --------------------------------------------------------------------------

LOSS_NONE = 0     # perfect efficiency, L(f,v) = 0
LOSS_STATIC = 1   # constant efficiencies: L = f v (1-eta_t)/eta_t (f>0), -(1-eta_r) f v (f<0)
LOSS_DYNAMIC = 2  # measured motor/converter table + gear + auxiliaries + transformer (efficiency.py)


class StaticLosses():
    """
    Power losses [W] of a drive with constant traction / regenerative-brake
    efficiencies (reference: train.py:199-212).  Callable like the reference's
    lambda `powerLosses(F [N], v [m/s])`.
    """

    def __init__(self, etaTraction, etaRgBrake):
        self.etaTraction = float(etaTraction)
        self.etaRgBrake = float(etaRgBrake)

    def __call__(self, f, v):
        return f*v*(f > 0)*(1 - self.etaTraction)/self.etaTraction - (1 - self.etaRgBrake)*f*v*(f < 0)

    def slopes(self):
        "(ct, cr): slack rows s >= ct*f and s >= -cr*f of the static model (v cancels)."
        return (1 - self.etaTraction)/self.etaTraction, (1 - self.etaRgBrake)


def closedFormLosses(fun):
    """
    (kind, ct, cr) when the callable `fun(F, v)` is one of the loss models the device has in closed form or carries its own parameter
    block (zero, constant efficiencies, efficiency.DynamicLosses / TabulatedLosses), None for any other function.  The reference accepts
    arbitrary CasADi-traceable lambdas (`train.powerLosses = lambda f,v: ...`, e.g. figure5.py:92-93); a plain callable is probed
    on a grid and matched.
    """

    if isinstance(fun, StaticLosses):
        ct, cr = fun.slopes()
        return LOSS_STATIC, ct, cr

    if getattr(fun, 'KIND', None) == LOSS_DYNAMIC:      # efficiency.DynamicLosses, efficiency.TabulatedLosses
        return LOSS_DYNAMIC, 0.0, 0.0

    fs = np.array([-3e5, -1.1e5, -2.5e4, -1.0, 1.0, 3.3e4, 1.2e5, 2.9e5])
    vs = np.array([0.7, 3.0, 11.0, 27.0, 44.0])

    vals = np.array([[float(fun(float(f), float(v))) for v in vs] for f in fs])

    if np.all(vals == 0):
        return LOSS_NONE, 0.0, 0.0

    # static-efficiency shape: L = ct f v for f > 0 and -cr f v for f < 0
    pw = fs[:, None]*vs[None, :]
    ct = vals[-1, -1]/pw[-1, -1]
    cr = -vals[0, -1]/pw[0, -1]
    model = np.where(pw > 0, ct*pw, -cr*pw)

    if ct >= 0 and cr >= 0 and np.allclose(vals, model, rtol=1e-12, atol=1e-9):
        return LOSS_STATIC, float(ct), float(cr)

    return None


def classifyLosses(fun):
    """
    Which device loss model the callable `fun(F, v)` is: (kind, ct, cr).  `Train.lossesCallable()` has already wrapped a function
    without closed form into its table (efficiency.TabulatedLosses); a bare function of that sort cannot be classified without the
    train's operating range.
    """

    found = closedFormLosses(fun)

    if found is None:
        raise NotImplementedError("The power-losses callable is neither zero, nor a constant-efficiency model: pass it through "
                                  "Train.lossesCallable() (train.powerLosses = fun), which tabulates it over the train's operating range.")

    return found


def splitLosses(fun):
    """
    Split a loss function into a traction and a regenerative-brake part, each
    extended linearly through f = 0 (reference: utils.py:197-220).  The slope is
    taken by a central difference of width 1e-10 around +-1e-10 like the
    reference's symbolic derivative evaluated there.
    """

    tol = 1e-10

    def slope(f0, v):
        h = 0.5e-10
        return (fun(f0 + h, v) - fun(f0 - h, v))/(2*h)

    def funTr(f, v):
        return fun(f, v) if f >= 0 else slope(tol, v)*f + fun(0, v)

    def funRgb(f, v):
        return fun(f, v) if f < 0 else slope(-tol, v)*f + fun(0, v)

    return funTr, funRgb


# --------------------------------------------------------------------------
# energy accounting (reference: utils.py:223-259, 291-294, 330)
# --------------------------------------------------------------------------

def curvatureResistance(curv, g, rho):
    "Specific curvature resistance [m/s^2] (reference: train.py:252-253 as written, divided by rho as in :254)."

    c = abs(curv)

    return (g*0.5*c/(1 - 30*c) if c <= 1/300 else g*0.65*c/(1 - 55*c))/rho


def postProcessDataFrame(dfIn, points, train, CVODES=True, integrateLosses=False, integrateRollingResistance=False, device=0):
    """
    Adds the force / power / energy columns the reference computes after a solve (utils.py:223-336).  The two integrations
    run on the GPU (csrc/msd_post.hip): `CVODES=True` re-simulates the trajectory in the time domain with accumulated errors
    (utils.py:164-194), `integrateLosses=True` integrates the losses over every interval instead of the mid-point rule
    (utils.py:261-289), `integrateRollingResistance=True` adds the energy dissipated by the rolling resistance per interval
    (utils.py:296-320; like the reference it integrates with the traction and pneumatic-brake forces only).
    """

    unitScaling = 1e-6/3.6  # Nm -> kWh
    totalMass = train.mass*train.rho

    # the new columns are collected and joined to the frame once at the end: twenty-one insertions into a DataFrame cost more than the solve itself (2.1 of the
    # 2.8 ms of casadiSolver.solve() at N = 100, profiles/r06); same columns, same order as the reference's one-by-one assignments
    df = dfIn
    new = {}

    new['Speed limit [m/s]'] = points['Speed limit [m/s]'].values
    new['Gradient [permil]'] = points['Gradient [permil]'].values
    new['Curvature [1/m]'] = points['Curvature [1/m]'].values

    fel = df['Force (el) [N]'].values.astype(float)
    fpb = df['Force (pnb) [N]'].values.astype(float)
    vel = df['Velocity [m/s]'].values.astype(float)
    pos = df['Position [m]'].values.astype(float)

    with np.errstate(invalid='ignore'):
        facc = fel*(fel >= 0)
        frgb = fel*(fel < 0)

    new['Force (acc) [N]'] = facc
    new['Force (rgb) [N]'] = frgb
    new['Force [N]'] = facc + frgb + fpb

    velNext = np.append(vel[1:], np.nan)
    ds = np.append(np.diff(pos), np.nan)

    new['Max. Power [kW]'] = np.maximum(facc*vel/1e3, facc*velNext/1e3)
    new['Min. Power [kW]'] = np.minimum(frgb*vel/1e3, frgb*velNext/1e3)

    grad = np.asarray(new['Gradient [permil]'], dtype=float)/1000
    curv = np.asarray(new['Curvature [1/m]'])
    times = df.index.values.astype(float)
    model = train.exportModel()

    lossE = np.full(len(df), np.nan)

    if not integrateLosses:

        losses = train.powerLossesFuns(split=False)   # specific: f [N/kg] -> [W/kg]
        vm = 0.5*(vel + velNext)

        for k in range(len(df) - 1):
            lossE[k] = unitScaling*ds[k]*totalMass*losses(fel[k]/totalMass, vm[k])/vm[k]

    else:

        from . import _device

        fun = train.lossesCallable()
        kind, ct, cr = classifyLosses(fun)
        table = fun.parameters(totalMass) if kind == LOSS_DYNAMIC else None
        etr, ebr = _device.integrate_losses(model, kind, ct, cr, table, [fel[:-1]/totalMass], [fpb[:-1]/totalMass], [np.diff(times)],
                                            grad[:-1], curv[:-1], [vel[:-1]], device=device)
        lossE[:-1] = unitScaling*totalMass*np.where(fel[:-1] >= 0, etr[0], ebr[0])     # utils.py:283-287

    new['Losses [kWh]'] = lossE
    new['Energy [kWh]'] = unitScaling*ds*facc + unitScaling*ds*frgb + lossE
    new['Energy (pnb) [kWh]'] = -unitScaling*ds*fpb
    new['Energy (kin) [kWh]'] = unitScaling*0.5*train.mass*vel**2   # train.mass, not mass*rho (utils.py:294)

    rr = (train.r0 + train.r1*vel + train.r2*vel**2)/totalMass
    cr = np.array([curvatureResistance(c, train.g, train.rho) for c in curv])

    new['Acceleration [m/s^2]'] = new['Force [N]']/totalMass - rr - train.g*grad/train.rho - cr

    if integrateRollingResistance:   # utils.py:296-320

        from .train import TrainIntegrator

        integ = TrainIntegrator(model, 'RK')
        integ.initRollingResistance(solver='CVODES')
        loss, _ = integ.calcRollingResistance(vel[:-1], np.diff(pos), facc[:-1]/totalMass, fpb[:-1]/totalMass, grad[:-1], curv[:-1])

        new['Rolling resistance [kWh]'] = np.append(unitScaling*totalMass*np.atleast_1d(loss), np.nan)

    if CVODES:   # simulateCVODES (utils.py:164-194, 332-334)

        from . import _device

        total = (new['Force [N]'][:-1]/totalMass)
        p, v = _device.resimulate(model, [total], [np.diff(times)], grad[:-1], curv[:-1], [pos[0]], [vel[0]], device=device)

        new['Position - cvodes [m]'] = p[0]
        new['Velocity - cvodes [m/s]'] = v[0]
        new['Error position [m]'] = np.abs(p[0] - pos)
        new['Error velocity [m/s]'] = np.abs(v[0] - vel)

    overwritten = [c for c in new if c in dfIn.columns]      # (a frame that already went through here: the reference's assignments overwrite in place)
    if overwritten:
        df = dfIn.copy()
        for c, v in new.items():
            df[c] = v
        return df

    return pd.concat([dfIn, pd.DataFrame(new, index=dfIn.index)], axis=1)

