"""Diagnostic (CPU oracle): the running times of the figure-5 configuration with the dynamic loss model that do not converge (bench.py: alt.dynamic_losses_N300) --
how far the mid-point speeds of the final iterate are from the kink of the loss model at the turning speed Pmax/Fmax (efficiency.py:7-12).  Restoration phase off:
the plain iteration's stalling point.   usage: dyn_kink_probe.py"""
import os, sys
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in ('ms-eetc_amd','','tests'): sys.path.insert(0, os.path.join(ROOT,p))
import numpy as np
import cases
from oracle import oracle
from mseetc.train import Train
from mseetc.efficiency import totalLossesFunction
from mseetc.track import computeDiscretizationPoints
N=300
train = Train(config={'id': 'NL_Intercity_VIRM6'}); train.forceMinPn = 0
train.powerLosses = totalLossesFunction(train, auxiliaries=27000, etaGear=0.96)
track = cases.track_00(8500)
par = train.powerLosses.parameters(train.mass*train.rho)
oracle.set_loss_table(par)
Fmax, Pmax, vTurn, vMin, vMax = par[0], par[1], par[2], par[3], par[4]
print('Fmax', Fmax, 'Pmax', Pmax, 'vTurn', vTurn, 'vMin', vMin, 'vMax', vMax)
pts = computeDiscretizationPoints(track, N)
prob = oracle.pack_problem(train, pts, dict(numIntervals=N, maxIterations=500, energyOptimal=True, minimumVelocity=1, numSteps=1, numApproxSteps=1), 2, 0.0, 0.0, track.length)
oracle.lib().oracle_set_restoration(0)
for f in (1.2125, 1.2130, 1.2138, 1.2850, 1.2855, 1.10, 1.20, 1.25, 1.30):
    ref = oracle.solve(prob, prob.scenario(272.4726*f, terminalVelocity=100/3.6, initialVelocity=1), start='profile')
    z = ref['z']; body = z[:4*N].reshape(N, 4)      # Fel, s, t, b
    b = np.concatenate([body[:, 3], [z[-1]]]); v = np.sqrt(b); vbar = 0.5*(v[:-1] + v[1:]); fel = body[:, 0]
    d = np.abs(vbar - vTurn); k = int(np.argmin(d))
    dmin = np.abs(vbar - vMin); k2 = int(np.argmin(dmin))
    print('factor', f, 'status', ref['stats']['STATUS'], 'iters', ref['stats']['ITERS'], 'kkt', '%.2e' % ref['stats']['KKT'],
          '| nearest mid-point speed to vTurn: interval', k, 'vbar - vTurn = %.3e' % (vbar[k] - vTurn), 'Fel there %.4f' % fel[k], 'neighbours', ['%.3e' % (vbar[j] - vTurn) for j in (k-1, k+1)],
          '| nearest to vMin: interval', k2, '%.3e' % (vbar[k2] - vMin), 'Fel %.4f' % fel[k2])
