"""
Iterations and optimum of the two starting points (reference cold start vs device-built profile) over the test configurations.
Run on a GPU box: python tools/start_point_survey.py
"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R + '/tests', R, R + '/ms-eetc_amd']
import cases
from mseetc.ocp import casadiSolver
from mseetc.train import Train
from mseetc.efficiency import totalLossesFunction


def survey(label, train, track, N, T, eo=True, **kw):
    opts = dict(numIntervals=N, maxIterations=500, energyOptimal=eo, integrationOptions=dict(numSteps=1, numApproxSteps=1))
    out = {}
    for start in ('reference', 'profile'):
        out[start] = casadiSolver(train, track, opts, startingPoint=start).solveBatch(T, **kw)
    a, b = out['reference'], out['profile']
    ok = (a['status'] >= 0) & (b['status'] >= 0)
    d = np.max(np.abs(a['cost'][ok] - b['cost'][ok])/np.abs(a['cost'][ok])) if ok.any() else float('nan')
    print('%-34s n=%4d  ok ref %4d prof %4d | iters ref %.1f (max %d) prof %.1f (max %d) | kernel ms ref %.2f prof %.2f | max rel dcost %.1e'
          % (label, len(a['status']), (a['status'] >= 0).sum(), (b['status'] >= 0).sum(), a['iterations'].mean(), a['iterations'].max(),
             b['iterations'].mean(), b['iterations'].max(), a['kernel_ms'], b['kernel_ms'], d))


train = cases.train_default()
survey('config1 N100 B1024', train, cases.track_00(), 100, cases.c1_times(1024))
survey('config1 tight/loose T', train, cases.track_00(), 100, np.linspace(1462.0, 2600.0, 512))
survey('config2 CH N200 B1024', train, cases.track_CH(), 200, cases.c2_times(1024))
survey('CH tight/loose T', train, cases.track_CH(), 200, np.linspace(1040.0, 2000.0, 256))
survey('fig10 N100', cases.train_fig10(), cases.track_00(), 100, np.linspace(1480.0, 1800.0, 64))
survey('fig10 N300', cases.train_fig10(), cases.track_00(), 300, np.linspace(1480.0, 1800.0, 64))
t5 = cases.train_fig5(); t5.etaTraction = t5.etaRgBrake = 0.73
survey('fig5 static N300', t5, cases.track_00(8500), 300, 272.4726*np.linspace(1.01, 1.5, 64), terminalVelocity=100/3.6, initialVelocity=1)
survey('fig5 mintime N300', cases.train_fig5(), cases.track_00(8500), 300, np.array([300.0, 400.0]), eo=False, terminalVelocity=100/3.6, initialVelocity=1)
td = Train(config={'id': 'NL_Intercity_VIRM6'}); td.forceMinPn = 0
td.powerLosses = totalLossesFunction(td, auxiliaries=27000, etaGear=0.96)
survey('fig5 dynamic N100', td, cases.track_00(8500), 100, 272.4726*np.linspace(1.02, 1.5, 64), terminalVelocity=100/3.6, initialVelocity=1)
survey('fig5 dynamic N300', td, cases.track_00(8500), 300, 272.4726*np.linspace(1.02, 1.5, 64), terminalVelocity=100/3.6, initialVelocity=1)
rng = np.random.default_rng(3)
survey('mid-journey starts', train, cases.track_00(), 100, 1541 + 200*rng.random(256), initialTime=100*rng.random(256), initialVelocity=1 + 30*rng.random(256),
       terminalVelocity=1 + 20*rng.random(256))
n = np.clip(rng.standard_normal((4, 512)), -2, 2)
survey('config3 perturbed stock', train, cases.track_00(), 100, cases.c1_times(512, seed=20260614), mass=train.mass*(1 + 0.05*n[0]), r0=train.r0*(1 + 0.05*n[1]),
       r1=train.r1*(1 + 0.05*n[2]), r2=train.r2*(1 + 0.05*n[3]))
