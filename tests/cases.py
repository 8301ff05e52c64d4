"""
Problem configurations used across the tests and adapters that
turn them into (a) the oracle's flat problem and (b) the independent numpy NLP.
"""

import numpy as np

from mseetc.track import Track, computeDiscretizationPoints
from mseetc.train import Train

from oracle import oracle
from oracle.oracle import DP, IP

import nlp_numpy

CONFIG_JSON = dict(maxIterations=500, numIntervals=300, integrationMethod='RK',
                   integrationOptions=dict(order=4, numSteps=1, numApproxSteps=1))   # simulations/config.json


from mseetc.workloads import train_default, track_00, track_CH, c1_times, c2_times      # the benchmark's definitions


def train_fig10():
    "figure10.py:16-22 / gpops/trainMain.m configuration"
    train = train_default()
    train.forceMinPn = 0
    train.forceMin = -train.forceMax
    train.powerMax = 3129277
    train.powerMin = -train.powerMax
    train.etaTraction = 0.73
    train.etaRgBrake = 0.73
    return train


def train_fig5():
    "figure5.py:87-94 after the side effects of totalLossesFunction (efficiency.py:56-71)"
    train = train_default()
    train.forceMinPn = 0
    hz = lambda f: ((f - 20)/(170 - 20))*(160 - 20) + 20
    pmax = train.forceMax*hz(55)/3.6
    train.powerMax = pmax
    train.powerMin = -pmax
    train.forceMin = -train.forceMax*(train.forceMin != 0)
    train.velocityMax = 160/3.6
    return train


def oracle_problem(train, track, N, energyOptimal=True, losses='static', numSteps=1, numApproxSteps=1, maxIterations=500, vmin=1, integration=None, watchdogTrigger=0):
    """
    integration: None ('RK') or the reference's options of the other shooting integrators, e.g. dict(integrationMethod='IRK', order=2,
    collMethod='radau', maxIter=10) / dict(integrationMethod='CVODES', absTol=1e-8, relTol=1e-6).  The oracle keeps ONE set of
    collocation tables (module state, like its loss table): the last problem packed with 'IRK' owns them.
    """
    pts = computeDiscretizationPoints(track, N)
    opts = dict(numIntervals=N, maxIterations=maxIterations, energyOptimal=energyOptimal, minimumVelocity=vmin,
                numSteps=numSteps, numApproxSteps=numApproxSteps, watchdogTrigger=watchdogTrigger)
    if integration:
        opts.update(integration)
        if integration.get('integrationMethod') == 'IRK':
            from mseetc.train import collocationTables
            oracle.set_collocation(*collocationTables(integration['order'], integration.get('collMethod', 'radau')))
    if losses == 'static':
        kind, ct, cr = 1, (1 - train.etaTraction)/train.etaTraction, 1 - train.etaRgBrake
    else:
        kind, ct, cr = 0, 0.0, 0.0
    return oracle.pack_problem(train, pts, opts, kind, ct, cr, track.length)


def numpy_nlp(prob):
    ip, dp = prob.ip, prob.dp
    return nlp_numpy.NLP(N=int(ip[IP['N']]), withPn=bool(ip[IP['WITH_PN']]), hasPower=bool(ip[IP['HAS_POWER']]),
                         energyOptimal=bool(ip[IP['ENERGY_OPT']]), numSteps=int(ip[IP['NUM_STEPS']]), numApprox=int(ip[IP['NUM_APPROX']]),
                         ds=prob.ds, grad=prob.grad, curv=prob.curv, sr0=dp[DP['SR0']], sr1=dp[DP['SR1']], sr2=dp[DP['SR2']],
                         g=dp[DP['G']], rho=dp[DP['RHO']], fmax=dp[DP['FMAX']], fmin=dp[DP['FMIN']], fminPn=dp[DP['FMIN_PN']],
                         pwUpper=dp[DP['PW_UPPER']], pwLower=dp[DP['PW_LOWER']], accMin=dp[DP['ACC_MIN']], accMax=dp[DP['ACC_MAX']],
                         ct=dp[DP['LOSS_CT']], cr=dp[DP['LOSS_CR']], vminSq=dp[DP['VMIN_SQ']], objDen=dp[DP['OBJ_DEN']], bmax=prob.bmax,
                         integrateLosses=bool(ip[IP['INTEGRATE_LOSSES']]))




def gpops_profile_deviation(z, ds, stp=4):
    """
    A multiple-shooting solution of the figure-10 configuration against the GPOPS-II trajectory the reference holds (gpops/00_var_speed_limit_100_GPOPSII.csv,
    278 rows of t, s, v; figure10.py:50-55,81-85 overlays it on the DMS solution): v(s) and t(s) of the solution interpolated at the GPOPS nodes
    (b = v^2 linear in s between shooting nodes).  Returns max |dv|, rms dv, max |dt|.
    """
    import pandas as pd
    from pathlib import Path
    g = pd.read_csv(Path(__file__).resolve().parent / 'golden' / '00_var_speed_limit_100_GPOPSII.csv').drop_duplicates(subset='Position [m]')      # (figure10.py:52)
    gs, gv, gt = g['Position [m]'].values, g['Velocity [m/s]'].values, g['Time [s]'].values
    N = len(ds)
    t, b = np.r_[z[stp - 2:stp*N:stp], z[-2]], np.r_[z[stp - 1:stp*N:stp], z[-1]]
    pos = np.r_[0.0, np.cumsum(ds)]
    dv, dt = np.abs(np.sqrt(np.interp(gs, pos, b)) - gv), np.abs(np.interp(gs, pos, t) - gt)
    return float(dv.max()), float(np.sqrt((dv**2).mean())), float(dt.max())
