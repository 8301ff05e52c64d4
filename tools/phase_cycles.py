"""Diagnostic: per-phase shader-cycle telemetry of the solve kernel for scenario 0 of a config-1 batch (needs a GPU)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('ms-eetc_amd', '', 'tests'):
    sys.path.insert(0, os.path.join(ROOT, p))
import numpy as np, cases
from mseetc.ocp import casadiSolver
from mseetc import _device
N=100
if len(sys.argv) > 1 and sys.argv[1] == 'dyn':      # the figure-5 configuration with the dynamic loss model (bench.py: alt.dynamic_losses_N100)
    from mseetc import workloads as wl
    from mseetc.train import Train
    from mseetc.efficiency import totalLossesFunction
    tr = Train(config={'id': 'NL_Intercity_VIRM6'}); tr.forceMinPn = 0
    tr.powerLosses = totalLossesFunction(tr, auxiliaries=27000, etaGear=0.96)
    solver = casadiSolver(tr, wl.track_00(8500), wl.options(N))
    scen = solver._scenarios(272.4726*(1.05 + 0.25*np.random.default_rng(20260616).random(1024)), 0, 100/3.6, 1)
else:
    solver = casadiSolver(cases.train_default(), cases.track_00(), dict(numIntervals=N, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1)))
    T = cases.c1_times(1024)
    scen = solver._scenarios(T, 0, 1, 1)
out = solver.problem.solve_batch(scen, history=64)
raw = np.zeros((64, 8))
# re-run to fetch the raw buffer including the last rows
import ctypes
L = _device.lib(); h = solver.problem._h
L.msd_set_history(h, raw.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), 64)
out = solver.problem.solve_batch(scen)
names = ['EVAL / R:elements','KKT','ASSEMBLE','RICCATI (roll-out)','READBACK / R:scan P','GPHID','STEPLEN / R:recursion','MERIT','UPDATE','OTHER','R:scan grad','R:feed-forward','R:scan state']
ph = np.concatenate([raw[62], raw[63]])[:13]
st = out['stats'][0]
print("geometry", solver.problem.geometry(), "kernel_ms", out['kernel_ms'], "iters", st[1], "cycles total", st[11])
for nme, v in zip(names, ph):
    print(f"  {nme:22s} {v/1e3:10.0f} kcycles  {100*v/st[11]:5.1f}%   per iter {v/max(st[1],1)/1e3:8.1f} kcyc")
