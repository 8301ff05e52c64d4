"""
The reference's own unit tests (unitTests/curvatureResistance/curvatureResistance.py:94-201), restated:
  1. testMinimumTimeProblem  -- a train on a track with constant curvature 1/300 whose force limits are shifted by the curvature
     resistance must drive the same time-optimal speed profile as the unshifted train on the straight track (1e-3 relative);
  2. testMinimumEnergyProblem -- for no / constant (0.73) / dynamic losses the mechanical energy on the curved track exceeds the one
     on the straight track by F_curv * L within 5 % (energies rounded to 0.1 kWh like the reference does).
Same configuration as the reference: N = 300, RK4 with numSteps = 1, numApproxSteps = 1, first 3475 m of 00_var_speed_limit_100,
v0 = vN = 1 m/s, pneumatic brake off.  Run with the CPU oracle here and with the HIP path under -m gpu.
"""

import copy

import numpy as np
import pytest

import cases
from mseetc.train import Train
from mseetc.track import Track, computeDiscretizationPoints
from mseetc.efficiency import totalLossesFunction
from mseetc.utils import classifyLosses

K = 1/300
L = 3475
OPTS = dict(maxIterations=500, numIntervals=300, integrationMethod='RK', integrationOptions=dict(order=4, numSteps=1, numApproxSteps=1),
            minimumVelocity=1)


def tracks():
    straight = Track(config={'id': '00_var_speed_limit_100'})
    curved = copy.deepcopy(straight)
    curved.importCurvatureTuples(tuples=[[0.0, str(1/K), str(1/K)]])
    return straight, curved


def curvatureForce(g, rho):
    return g*0.5*abs(K)/((1 - 30*abs(K))*rho)*(abs(K) <= 1/300) + g*0.65*abs(K)/((1 - 55*abs(K))*rho)*(abs(K) > 1/300)


# ---- two back ends with the same (df-free) result record ------------------------------------------------------------

def solve_oracle(train, track, energyOptimal, T):
    from oracle import oracle
    track = copy.deepcopy(track)
    track.updateLimits(positionEnd=L)
    pts = computeDiscretizationPoints(track, 300)
    kind, ct, cr = classifyLosses(train.lossesCallable()) if energyOptimal else (0, 0.0, 0.0)
    if kind == 2:
        oracle.set_loss_table(train.lossesCallable().parameters(train.mass*train.rho))
    opts = dict(numIntervals=300, maxIterations=500, energyOptimal=energyOptimal, minimumVelocity=1, numSteps=1, numApproxSteps=1)
    prob = oracle.pack_problem(train, pts, opts, kind, ct, cr, track.length)
    res = oracle.solve(prob, prob.scenario(float(T), terminalVelocity=1, initialVelocity=1))
    assert res['stats']['STATUS'] == 0
    return res['z'], prob.ds


def solve_gpu(train, track, energyOptimal, T):
    from mseetc.ocp import casadiSolver
    track = copy.deepcopy(track)
    track.updateLimits(positionEnd=L)
    solver = casadiSolver(train, track, dict(OPTS, energyOptimal=energyOptimal))
    res = solver.solveBatch(float(T), terminalVelocity=1, initialVelocity=1)
    assert res['status'][0] == 0
    return res['z'][0], solver.steps


def energies(train, z, ds):
    "sum of 'Energy [kWh]' and of 'Losses [kWh]' as postProcessDataFrame computes them (utils.py:243-259, 291)"
    N = 300
    M = train.mass*train.rho
    f, b = z[0:4*N:4], np.append(z[3:4*N:4], z[-1])
    vm = 0.5*(np.sqrt(b[:-1]) + np.sqrt(b[1:]))
    fun = train.lossesCallable()
    losses = (1e-6/3.6)*np.array([dsk*fun(fk*M, vk)/vk for dsk, fk, vk in zip(ds, f, vm)])
    return float(np.sum((1e-6/3.6)*ds*f*M + losses)), float(np.sum(losses))


def run_minimum_time(solve):
    straight, curved = tracks()
    train = Train(config={'id': 'NL_Intercity_VIRM6'})
    train.forceMinPn = 0
    train.powerMax = None
    train.powerMin = None
    train.powerLosses = lambda f, v: 0
    z0, _ = solve(train, straight, False, 180)
    shift = curvatureForce(train.g, train.rho)*train.mass*train.rho
    train.forceMax = train.forceMax + shift
    train.forceMin = train.forceMin + shift
    z1, _ = solve(train, curved, False, 180)
    v0, v1 = np.sqrt(np.append(z0[3:1200:4], z0[-1])), np.sqrt(np.append(z1[3:1200:4], z1[-1]))
    assert np.all(np.abs((v0 - v1)/v0) <= 1e-3)


def run_minimum_energy(solve):
    straight, curved = tracks()
    train = Train(config={'id': 'NL_Intercity_VIRM6'})
    train.forceMinPn = 0
    etaMax = 0.73
    noLosses = lambda f, v: 0
    idealLosses = lambda f, v: f*v*(f > 0)*(1 - etaMax)/etaMax - (1 - etaMax)*f*v*(f < 0)
    realLosses = totalLossesFunction(train, auxiliaries=27000, etaGear=0.96)       # mutates the train for all three models
    expected = curvatureForce(train.g, train.rho)*train.rho*train.mass*L/(3600*1000)
    for lossFunction in (noLosses, idealLosses, realLosses):
        train.powerLosses = lossFunction
        mech = []
        for track in (straight, curved):
            z, ds = solve(train, track, True, 200)
            total, losses = energies(train, z, ds)
            mech.append(round(total, 1) - round(losses, 1))
        assert abs(expected - (mech[1] - mech[0]))/expected <= 5e-2


def test_minimum_time_problem_oracle():
    run_minimum_time(solve_oracle)


def test_minimum_energy_problem_oracle():
    run_minimum_energy(solve_oracle)


@pytest.mark.gpu
def test_minimum_time_problem_gpu():
    run_minimum_time(solve_gpu)


@pytest.mark.gpu
def test_minimum_energy_problem_gpu():
    run_minimum_energy(solve_gpu)
