"""
Independent numpy restatement of the reference NLP (mseetc/ocp.py:166-284 with the
integrator of mseetc/train.py:225-344) used to CERTIFY solutions: it shares no code
with oracle/ms_oracle.c or with the HIP kernels.  First derivatives come from
complex-step differentiation (the expressions are analytic in the variables), so the
stationarity check is accurate to round-off.

kkt_certificate(...) returns the violations of the first-order conditions in the
CasADi/IPOPT sign convention (lam_g > 0: upper bound active).
"""

import numpy as np


class NLP():

    def __init__(self, *, N, withPn, hasPower, energyOptimal, numSteps, numApprox, ds, grad, curv,
                 sr0, sr1, sr2, g, rho, fmax, fmin, fminPn, pwUpper, pwLower, accMin, accMax, ct, cr, vminSq, objDen, bmax, integrateLosses=False):
        self.__dict__.update(locals())
        del self.__dict__['self']
        self.ds = np.asarray(ds, float)
        c = np.abs(np.asarray(curv, float))
        crv = np.where(c <= 1/300, g*0.5*c/(1 - 30*c), g*0.65*c/(1 - 55*c))   # train.py:252-253
        self.G = g*np.asarray(grad, float)/rho + crv/rho                          # train.py:254
        self.stp = 4 + int(withPn)
        self.nz = self.stp*N + 2
        self.rpi = (2 if hasPower else 0) + 3 + (2 if energyOptimal else 0)

    # ---- variable views (ocp.py:166-181, 247-272 ordering) ----------------------

    def split(self, z):
        N, stp, pn = self.N, self.stp, int(self.withPn)
        body = z[:stp*N].reshape(N, stp)
        f = body[:, 0]
        p = body[:, 1] if pn else np.zeros(N, dtype=z.dtype)
        s = body[:, 1 + pn]
        t = np.append(body[:, 2 + pn], z[stp*N])
        b = np.append(body[:, 3 + pn], z[stp*N + 1])
        return f, p, s, t, b

    # ---- integrator (train.py:294-301, 324-344) ---------------------------------------

    def _ode(self, b, w):
        return 2*self.ds*(w - (self.sr0 + self.sr1*np.sqrt(b) + self.sr2*b) - self.G)

    def _rk4(self, b, w, H):
        h = H/self.numSteps
        for _ in range(self.numSteps):
            k1 = self._ode(b, w)
            k2 = self._ode(b + 0.5*h*k1, w)
            k3 = self._ode(b + 0.5*h*k2, w)
            k4 = self._ode(b + h*k3, w)
            b = b + (h/6)*(k1 + 2*k2 + 2*k3 + k4)
        return b

    def interval(self, b, w):
        "(tau, bplus) for all intervals at once."
        if self.numApprox == 0:
            h = 1.0/self.numSteps
            t = np.zeros_like(b)
            for _ in range(self.numSteps):
                k1b, k1t = self._ode(b, w), self.ds/np.sqrt(b)
                b2 = b + 0.5*h*k1b
                k2b, k2t = self._ode(b2, w), self.ds/np.sqrt(b2)
                b3 = b + 0.5*h*k2b
                k3b, k3t = self._ode(b3, w), self.ds/np.sqrt(b3)
                b4 = b + h*k3b
                k4b, k4t = self._ode(b4, w), self.ds/np.sqrt(b4)
                b = b + (h/6)*(k1b + 2*k2b + 2*k3b + k4b)
                t = t + (h/6)*(k1t + 2*k2t + 2*k3t + k4t)
            return t, b
        ns = self.numApprox
        prev, tau = b, 0
        for j in range(1, ns + 1):
            cur = self._rk4(b, w, j/ns)
            tau = tau + 2*self.ds*(1/ns)/(np.sqrt(prev) + np.sqrt(cur))
            prev = cur
        return tau, prev

    # ---- integrateLosses (ocp.py:231-241, train.py:367-413) ------------------------------------------

    def _distance(self, v0, dt, w):
        """
        Distance covered in the time dt by dv/dt = a - sr1 v - sr2 v^2, a = w - sr0 - G, from v0: the closed-form solution of this
        Riccati equation (no integrator involved).  D = sr1^2 + 4 sr2 a > 0: roots v+- of sr2 v^2 + sr1 v - a, k = sr2 (v+ - v-),
        C = (v0 - v+)/(v0 - v-), X = v+ dt + ln((1 - C exp(-k dt))/(1 - C))/sr2.  D < 0 (strong braking): u = v + sr1/(2 sr2) obeys
        du/dt = -sr2 (u^2 + om^2), om^2 = -D/(4 sr2^2): X = ln(cos(th0 - sr2 om dt)/cos(th0))/sr2 - sr1 dt/(2 sr2), th0 = atan(u0/om).
        Every operation is analytic in (v0, dt, w): complex-step differentiation goes through.
        """
        sr1, sr2 = self.sr1, self.sr2
        a = w - self.sr0 - self.G
        D = sr1*sr1 + 4*sr2*a
        pos = D.real > 0
        Dp = np.where(pos, D, 1.0)
        rt = np.sqrt(Dp)
        vp, vm = (-sr1 + rt)/(2*sr2), (-sr1 - rt)/(2*sr2)
        C = (v0 - vp)/(v0 - vm)
        Xp = vp*dt + np.log((1 - C*np.exp(-sr2*(vp - vm)*dt))/(1 - C))/sr2
        Dn = np.where(pos, -1.0, D)
        om = np.sqrt(-Dn)/(2*sr2)
        th0 = np.arctan((v0 + sr1/(2*sr2))/om)
        with np.errstate(invalid='ignore', divide='ignore'):      # (the branch that does not apply may be evaluated outside its domain)
            Xn = np.log(np.cos(th0 - sr2*om*dt)/np.cos(th0))/sr2 - sr1*dt/(2*sr2)
        return np.where(pos, Xp, Xn)

    # ---- NLP functions -----------------------------------------------------------------------

    def rows(self, f, p, s, t0, b0, t1, b1):
        "Constraint rows of every interval, shape (N, rpi), reference order (ocp.py:183-229)."
        tau, bp = self.interval(b0, f + p)
        out = []
        if self.hasPower:
            out += [f*np.sqrt(b0), f*np.sqrt(b1)]
        out += [f + p - (self.sr0 + self.sr1*np.sqrt(b0) + self.sr2*b0) - self.G]
        out += [t1 - (t0 + tau), b1 - bp]
        if self.energyOptimal and self.integrateLosses:
            X = self._distance(np.sqrt(b0), t1 - t0, f + p)       # ocp.py:233 with constant efficiencies: both integrals are multiples of X
            out += [s - self.ct*f*X, s + self.cr*f*X]
        elif self.energyOptimal:
            out += [s - self.ct*f, s + self.cr*f]
        return np.stack(out, axis=1)

    def cons(self, z):
        f, p, s, t, b = self.split(z)
        return self.rows(f, p, s, t[:-1], b[:-1], t[1:], b[1:]).reshape(-1)

    def obj(self, z):
        f, p, s, t, b = self.split(z)
        if self.energyOptimal:
            J = (np.sum(self.ds*f) + np.sum(s) if self.integrateLosses else np.sum(self.ds*(f + s))) + 1e-3*np.sum((f[1:] - f[:-1])**2)      # ocp.py:235 / :223
        else:
            J = t[-1] + 1e-4*(np.sum(f*f) + np.sum(p*p))
        return J/self.objDen

    def bounds(self, t0, T, v0sq, vNsq):
        "lbz, ubz, lbg, ubg (ocp.py:175-272)."
        N, stp, pn = self.N, self.stp, int(self.withPn)
        lbz, ubz = np.zeros(self.nz), np.zeros(self.nz)
        for i in range(N):
            o = stp*i
            lbz[o], ubz[o] = self.fmin, self.fmax
            if pn:
                lbz[o + 1], ubz[o + 1] = self.fminPn, 0.0
            lbz[o + 1 + pn], ubz[o + 1 + pn] = 0.0, np.inf
            if i == 0:
                lbz[o + 2 + pn] = ubz[o + 2 + pn] = t0
                lbz[o + 3 + pn] = ubz[o + 3 + pn] = v0sq
            else:
                lbz[o + 2 + pn], ubz[o + 2 + pn] = t0, T
                lbz[o + 3 + pn], ubz[o + 3 + pn] = self.vminSq, self.bmax[i]
        lbz[stp*N], ubz[stp*N] = t0, T
        lbz[stp*N + 1] = ubz[stp*N + 1] = vNsq
        lo, up = [], []
        if self.hasPower:
            lo += [-abs(self.pwLower)]*2
            up += [abs(self.pwUpper)]*2
        lo += [self.accMin, 0.0, 0.0]
        up += [self.accMax, 0.0, 0.0]
        if self.energyOptimal:
            lo += [0.0, 0.0]
            up += [np.inf, np.inf]
        return lbz, ubz, np.tile(lo, N), np.tile(up, N)

    # ---- derivatives by complex step ------------------------------------------------------------------

    def jac_g(self, z):
        "Dense Jacobian of g (ng x nz)."
        N, stp, pn, rpi = self.N, self.stp, int(self.withPn), self.rpi
        f, p, s, t, b = self.split(z.astype(complex))
        args = [f, p, s, t[:-1], b[:-1], t[1:], b[1:]]
        h = 1e-30
        J = np.zeros((rpi*N, self.nz))
        cols = []
        for i in range(N):
            o = stp*i
            nxt = stp*(i + 1) if i + 1 < N else stp*N - 2 - pn   # so that nxt + 2 + pn -> t_{i+1}
            cols.append([o, (o + 1) if pn else -1, o + 1 + pn, o + 2 + pn, o + 3 + pn, nxt + 2 + pn, nxt + 3 + pn])
        cols = np.array(cols)
        for k in range(7):
            if k == 1 and not pn:
                continue
            pert = [a.copy() for a in args]
            pert[k] = pert[k] + 1j*h
            d = self.rows(*pert).imag/h          # (N, rpi)
            for i in range(N):
                J[rpi*i:rpi*(i + 1), cols[i, k]] += d[i]
        return J

    def grad_obj(self, z):
        h = 1e-30
        gr = np.zeros(self.nz)
        zc = z.astype(complex)
        for k in range(self.nz):
            zc[k] += 1j*h
            gr[k] = self.obj(zc).imag/h
            zc[k] -= 1j*h
        return gr


def kkt_certificate(nlp, z, lam_g, t0, T, v0sq, vNsq, act_tol=1e-6):
    """
    First-order optimality certificate.  Returns a dict of violations:
      stat   : r = grad f + J^T lam_g must be cancelled by bound multipliers of the right sign; reported is the
               largest complementarity product |r_k| * (distance to the bound that sign selects), or |r_k| itself
               where that bound does not exist
      feas_g : max violation of lbg <= g <= ubg, relative to max(1, |bound|)
      feas_z : max violation of lbz <= z <= ubz, relative to max(1, |bound|)
      sign_g : the same complementarity measure for the inequality rows and lam_g
    """

    lbz, ubz, lbg, ubg = nlp.bounds(t0, T, v0sq, vNsq)
    g = nlp.cons(z)
    J = nlp.jac_g(z)
    r = nlp.grad_obj(z) + J.T @ lam_g     # must be cancelled by bound multipliers

    # r_k > 0 needs a lower-bound multiplier zL = r_k, r_k < 0 an upper-bound multiplier zU = -r_k; what is left to check
    # is complementarity of that multiplier with its (relaxed) bound -- an interior-point solution has z*slack ~ mu
    fixed = lbz == ubz
    relax = lambda bnd: 1e-8*np.maximum(1.0, np.abs(np.where(np.isfinite(bnd), bnd, 1.0)))
    with np.errstate(invalid='ignore', over='ignore'):
        compL = np.where(np.isfinite(lbz), np.maximum(r, 0)*np.abs(z - lbz + relax(lbz)), np.where(r > 0, np.inf, 0.0))
        compU = np.where(np.isfinite(ubz), np.maximum(-r, 0)*np.abs(ubz + relax(ubz) - z), np.where(r < 0, np.inf, 0.0))
    stat_v = np.where(fixed, 0.0, np.where(r > 0, compL, compU))
    # variables without the needed bound: the gradient itself must vanish
    stat_v = np.where(np.isinf(stat_v), np.abs(r), stat_v)
    stat = float(np.max(stat_v))

    # relative to max(1, |bound|): IPOPT relaxes every bound by 1e-8*max(1,|bound|) (bound_relax_factor)
    def rel(viol, bound):
        with np.errstate(invalid='ignore'):
            v = viol/np.maximum(1.0, np.abs(np.where(np.isfinite(bound), bound, 1.0)))
        return float(np.max(np.where(np.isfinite(bound), v, -np.inf)))

    feas_g = max(0.0, rel(lbg - g, lbg), rel(g - ubg, ubg))
    feas_z = max(0.0, rel(lbz - z, lbz), rel(z - ubz, ubz))

    eq = lbg == ubg
    with np.errstate(invalid='ignore', over='ignore'):
        cU = np.where(np.isfinite(ubg), np.maximum(lam_g, 0)*np.abs(ubg + relax(ubg) - g), np.where(lam_g > 0, np.inf, 0.0))
        cL = np.where(np.isfinite(lbg), np.maximum(-lam_g, 0)*np.abs(g - lbg + relax(lbg)), np.where(lam_g < 0, np.inf, 0.0))
    sg = np.where(eq, 0.0, np.where(lam_g > 0, cU, cL))
    sg = np.where(np.isinf(sg), np.abs(lam_g), sg)
    sign = float(np.max(sg))

    return dict(stat=stat, feas_g=feas_g, feas_z=feas_z, sign_g=sign)
