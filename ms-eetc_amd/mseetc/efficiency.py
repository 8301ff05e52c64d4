"""
Dynamic (speed and load dependent) traction-chain losses -- host side of the model of the reference's
`mseetc/efficiency.py` (7-141): measured motor + converter losses on a (load, speed) grid, interpolated by a tensor
cubic spline, plus gear, auxiliaries and transformer losses.

Differences in mechanics, not in results: CasADi's `interpolant('bspline')` (efficiency.py:30; cubic, not-a-knot) is
replaced by the same interpolating spline built with scipy and converted to piecewise-polynomial form, so that the
device (and the oracle) evaluate plain bicubic patches; `totalLossesFunction` returns a `DynamicLosses` object that is
callable like the reference's closure and also carries the parameter block the solver ships to the GPU.

Like the reference, building the model OVERWRITES limits of the train (efficiency.py:64-71).
"""

import numpy as np
from scipy.interpolate import RectBivariateSpline

from .data import dataLosses


def forceToLoad(force, velocity, forceMax, powerMax):
    "Force [N] (positive) -> load [%]: constant-force region below powerMax/forceMax, constant-power above (efficiency.py:7-12)."

    turningPoint = powerMax/forceMax

    return 100*(force/forceMax) if velocity <= turningPoint else 100*(force*velocity/powerMax)


def loadToForce(load, velocity, forceMax, powerMax):
    "Load [%] -> force [N] (efficiency.py:15-20)."

    turningPoint = powerMax/forceMax

    return (load/100)*(forceMax if velocity <= turningPoint else powerMax/velocity)


class BicubicTable():
    """
    Interpolating tensor cubic spline (not-a-knot in both directions) of values on a (x, y) grid, stored as bicubic
    patches: value = sum_{p,q} c[ix][iy][p][q] (x - xc[ix])^p (y - yc[iy])^q (xc, yc: cell centres).  Zero outside the x range (the reference
    relies on the interpolant returning 0 there, efficiency.py:137); y is clipped by the caller.
    """

    def __init__(self, x, y, values):

        x, y, values = np.asarray(x, float), np.asarray(y, float), np.asarray(values, float)
        spl = RectBivariateSpline(x, y, values, kx=3, ky=3, s=0)
        tx, ty = spl.get_knots()
        c = spl.get_coeffs().reshape(len(tx) - 4, len(ty) - 4)

        self.xb = np.unique(tx)
        self.yb = np.unique(ty)
        nx, ny = len(self.xb) - 1, len(self.yb) - 1

        # bicubic patch of every cell = Taylor expansion of the spline about the cell centre (exact: the spline is a bicubic there)
        self.xc = 0.5*(self.xb[:-1] + self.xb[1:])
        self.yc = 0.5*(self.yb[:-1] + self.yb[1:])
        # the spline is one bicubic per cell: recover it exactly from 4 x 4 samples (normalised local coordinates for conditioning)
        nodes = np.array([-0.9, -0.3, 0.3, 0.9])
        Vinv = np.linalg.inv(np.vander(nodes, 4, increasing=True))
        coef = np.zeros((nx, ny, 4, 4))
        for ix in range(nx):
            hx = 0.5*(self.xb[ix + 1] - self.xb[ix])
            for iy in range(ny):
                hy = 0.5*(self.yb[iy + 1] - self.yb[iy])
                samples = spl(self.xc[ix] + hx*nodes, self.yc[iy] + hy*nodes)          # (4, 4)
                cu = Vinv @ samples @ Vinv.T                                              # powers of u = (x-xc)/hx, v = (y-yc)/hy
                coef[ix, iy] = cu/np.outer(hx**np.arange(4), hy**np.arange(4))
        self.coef = coef                                                        # ascending powers of (x - xc), (y - yc)
        self._check(x, y, values)

    def _check(self, x, y, values):

        for i, xv in enumerate(x):
            for j, yv in enumerate(y):
                assert abs(self.eval(xv, yv)[0] - values[i, j]) <= 1e-8*max(1.0, abs(values[i, j]))

    def cell(self, x, y):
        ix = min(max(int(np.searchsorted(self.xb, np.real(x), side='right')) - 1, 0), len(self.xb) - 2)
        iy = min(max(int(np.searchsorted(self.yb, np.real(y), side='right')) - 1, 0), len(self.yb) - 2)
        return ix, iy

    def eval(self, x, y):
        "(value, d/dx, d/dy, dxx, dxy, dyy); zero outside [xb[0], xb[-1]] in x.  Works for complex x, y (complex-step tests)."

        if np.real(x) < self.xb[0] or np.real(x) > self.xb[-1]:
            return (0.0,)*6

        ix, iy = self.cell(x, y)
        dx, dy = x - self.xc[ix], y - self.yc[iy]
        c = self.coef[ix, iy]
        X = [1, dx, dx*dx, dx*dx*dx]; X1 = [0, 1, 2*dx, 3*dx*dx]; X2 = [0, 0, 2, 6*dx]
        Y = [1, dy, dy*dy, dy*dy*dy]; Y1 = [0, 1, 2*dy, 3*dy*dy]; Y2 = [0, 0, 2, 6*dy]
        acc = lambda A, Bv: sum(c[p, q]*A[p]*Bv[q] for p in range(4) for q in range(4))

        return acc(X, Y), acc(X1, Y), acc(X, Y1), acc(X2, Y), acc(X1, Y1), acc(X, Y2)

    def flat(self):
        "[nx, ny, xb (nx+1), yb (ny+1), coef (nx, ny, 4, 4) row-major about the cell centres]: the block shipped to the device"

        return np.concatenate([[len(self.xb) - 1, len(self.yb) - 1], self.xb, self.yb, self.coef.reshape(-1)])


class DynamicLosses():
    "Callable total losses L(F [N], v [m/s]) -> [W] (efficiency.py:101-141) + its parameters."

    KIND = 2

    def __init__(self, table, forceMax, powerMax, auxiliaries, etaGear, R=10.0, V=15000.0):

        self.table = table
        self.forceMax, self.powerMax = float(forceMax), float(powerMax)
        self.vTurn = self.powerMax/self.forceMax
        self.vMin, self.vMax = float(table.yb[0]), float(table.yb[-1])
        self.auxiliaries, self.etaGear = float(auxiliaries), float(etaGear)
        self.R, self.V = float(R), float(V)

    def motor(self, f, v):
        "Motor + converter losses [W] (efficiency.py:36-49)"

        vr = np.real(v)
        vc = v if self.vMin <= vr <= self.vMax else (self.vMin if vr < self.vMin else self.vMax)
        absf = f if np.real(f) >= 0 else -f
        load = 100*(absf/self.forceMax) if np.real(vc) <= self.vTurn else 100*(absf*vc/self.powerMax)

        return self.table.eval(load, vc)[0]

    def __call__(self, f, v):

        traction = np.real(f) >= 0
        pWheel = f*v if traction else -f*v
        gear = ((1 - self.etaGear)/self.etaGear)*pWheel if traction else (1 - self.etaGear)*pWheel
        motor = self.motor(f, v)

        if not np.real(motor) > 0:      # outside of the table the spline is 0 and so are the total losses (efficiency.py:137)
            return 0.0*f

        R, V = self.R, self.V

        if traction:
            Pm = pWheel + gear + motor + self.auxiliaries
            trafo = (V - np.sqrt(V**2 - 4*R*Pm))**2/(4*R)
        else:
            Pm = pWheel - gear - motor - self.auxiliaries
            trafo = (V - np.sqrt(V**2 + 4*R*Pm))**2/(4*R)

        return gear + motor + self.auxiliaries + trafo

    def parameters(self, totalMass):
        "Flat parameter block for the device: 11 scalars (the last one the total mass the specific losses refer to) + the bicubic table."

        head = [self.forceMax, self.powerMax, self.vTurn, self.vMin, self.vMax, self.auxiliaries,
                (1 - self.etaGear)/self.etaGear, 1 - self.etaGear, self.R, self.V, float(totalMass)]

        return np.concatenate([head, self.table.flat()])


class _JoinedTable():
    "Two BicubicTable objects over the same y grid that meet at x = 0 (a cell edge): one block of bicubic patches for the device."

    def __init__(self, neg, pos):

        assert neg.xb[-1] == 0.0 and pos.xb[0] == 0.0 and np.array_equal(neg.yb, pos.yb)
        self.neg, self.pos = neg, pos
        self.xb = np.concatenate([neg.xb, pos.xb[1:]])
        self.yb = pos.yb
        self.coef = np.concatenate([neg.coef, pos.coef], axis=0)

    def eval(self, x, y):
        return (self.pos if np.real(x) >= 0 else self.neg).eval(x, y)

    def flat(self):
        return np.concatenate([[len(self.xb) - 1, len(self.yb) - 1], self.xb, self.yb, self.coef.reshape(-1)])


class TabulatedLosses():
    """
    Any loss function `L(F [N], v [m/s]) -> [W]` for the device.  The reference hands `train.powerLosses` to CasADi, which traces whatever
    the lambda computes (train.py:190-219, utils.py:197-220); the device has no symbolic layer, so a callable that is neither zero nor a
    constant-efficiency model is sampled on a (force, speed) grid over the train's operating range and shipped as bicubic patches of the
    interpolating (not-a-knot) spline -- the traction side F >= 0 and the braking side F <= 0 as separate splines, because the reference
    splits the function at F = 0 and extends each side linearly (utils.py:197-220), which the device model does on the table.  A function
    that is a polynomial of degree <= 3 in F and in v on each side is represented exactly; otherwise `maxDeviation` holds the largest
    difference found between function and table at the cell centres, relative to the largest loss on the grid (a warning is raised above
    `tolerance`); refine with `numForce` / `numVelocity`.

    Calling the object evaluates the user's function itself (post-processing reports the losses of the function, like the reference);
    `tabulated(F, v)` evaluates what the device sees.  Outside the speed range the table is continued constantly in v; the force range covers
    the train's limits with a margin of 5 %.
    """

    KIND = 2      # utils.LOSS_DYNAMIC: shipped as a parameter block + table like the dynamic model

    def __init__(self, fun, forceMin, forceMax, velocityMax, numForce=24, numVelocity=32, velocityMin=0.25, tolerance=1e-4):

        import warnings

        self.fun = fun
        fpos = np.linspace(0.0, 1.05*float(forceMax), int(numForce) + 1)
        lo = min(float(forceMin), -0.05*float(forceMax))      # a train without regenerative brake still gets a braking side (it stays inactive)
        fneg = np.linspace(1.05*lo, 0.0, int(numForce) + 1)
        vs = np.linspace(float(velocityMin), 1.1*float(velocityMax), int(numVelocity) + 1)

        sample = lambda fs: np.array([[float(fun(float(f), float(v))) for v in vs] for f in fs])
        vpos, vneg = sample(fpos), sample(fneg)

        if not (np.all(np.isfinite(vpos)) and np.all(np.isfinite(vneg))):
            raise ValueError("The power-losses function returns non-finite values on the operating range of the train!")

        self.table = _JoinedTable(BicubicTable(fneg, vs, vneg), BicubicTable(fpos, vs, vpos))
        self.forceMax = float(forceMax)
        self.vMin, self.vMax = float(vs[0]), float(vs[-1])

        # how well does the table follow the function between the samples?
        scale = max(np.max(np.abs(vpos)), np.max(np.abs(vneg)), 1e-300)
        dev = 0.0
        for fs in (fpos, fneg):
            for f in 0.5*(fs[:-1] + fs[1:]):
                for v in 0.5*(vs[:-1] + vs[1:]):
                    dev = max(dev, abs(self.tabulated(f, v) - float(fun(float(f), float(v))))/scale)
        self.maxDeviation = dev

        if dev > tolerance:
            warnings.warn("The tabulated power-losses function deviates from the function by {:.1e} of its largest value: "
                          "increase numForce / numVelocity (train.lossesTableSize).".format(dev))

    def __call__(self, f, v):
        return self.fun(f, v)

    def tabulated(self, f, v):
        "The device's view of the function (before the split at F = 0)."

        vr = np.real(v)
        vc = v if self.vMin <= vr <= self.vMax else (self.vMin if vr < self.vMin else self.vMax)

        return self.table.eval(f, vc)[0]

    def parameters(self, totalMass):
        "Parameter block in the layout of the dynamic model with vTurn = 0: the table is the loss power itself over (signed force, speed)."

        head = [self.forceMax, 0.0, 0.0, self.vMin, self.vMax, 0.0, 0.0, 0.0, 1.0, 1.0, float(totalMass)]

        return np.concatenate([head, self.table.flat()])


def motorLossesFunction(train, detailedOutput=False):
    "Spline of the measured motor + converter losses; updates the train limits to match the data (efficiency.py:54-98)."

    minSpeed, maxSpeed = 20, 160   # [km/h]
    minFreq, maxFreq = 20, 170     # [Hz]
    powFreq = 55                   # frequency where maximum power meets maximum force [Hz]

    HzToKmPerHour = lambda f: ((f - minFreq)/(maxFreq - minFreq))*(maxSpeed - minSpeed) + minSpeed

    forceMax = train.forceMax
    powerMax = forceMax*HzToKmPerHour(powFreq)/3.6

    # update train parameters to match data
    train.powerMax = powerMax
    train.powerMin = -powerMax
    train.forceMin = -forceMax*(train.forceMin != 0)
    train.velocityMax = maxSpeed/3.6

    numMotors = 4

    configA, configB = dataLosses()

    minLosses = np.minimum(np.array(configA['losses']), np.array(configB['losses']))*numMotors   # (loads, frequencies)

    loads = np.array(configB['loads'], dtype=float)
    loads[-1] += 1e-4   # to avoid artifacts when load is 100.000000001 (efficiency.py:27-28)

    velocities = np.array([HzToKmPerHour(f)/3.6 for f in configB['frequencies']])

    table = BicubicTable(loads, velocities, minLosses)

    model = DynamicLosses(table, forceMax, powerMax, auxiliaries=0.0, etaGear=1.0)

    if not detailedOutput:
        return model.motor

    import pandas as pd

    def frame(config):
        df = pd.DataFrame(index=[HzToKmPerHour(f)/3.6 for f in config['frequencies']])
        for i, l in enumerate(config['loads']):
            df[l] = [x*numMotors for x in config['losses'][i]]
        return df

    return {'fun': model.motor, 'dfA': frame(configA), 'dfB': frame(configB), 'table': table, 'forceMax': forceMax, 'powerMax': powerMax}


def totalLossesFunction(train, auxiliaries=27000, etaGear=1):
    "Gear + motor/converter + auxiliaries + transformer losses (efficiency.py:101-141)."

    detail = motorLossesFunction(train, detailedOutput=True)

    return DynamicLosses(detail['table'], detail['forceMax'], detail['powerMax'], auxiliaries, etaGear)
