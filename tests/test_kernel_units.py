"""
Pieces of the device header checked in isolation on the CPU (host build of ms-eetc_amd/csrc/msd_kernel.hpp through tests/hip_emu, no GPU, no oracle):
the last interval's elimination against a dense KKT solve in 50-digit arithmetic (mpmath) -- also where the two forces' curvatures differ by
twenty orders of magnitude or share a stiff common part, the cases round 5's rewrite is for.
"""

import subprocess
from pathlib import Path

import numpy as np
import pytest

EMU = Path(__file__).resolve().parent / 'hip_emu'
S = dict(TB=0, TW=1, BB=2, BW=3, RT=4, RB=5, HTT=6, HBB=7, HBQ=8, HBF=9, HBP=10, HQQ=11, HQF=12, HFF=13, HFP=14, HPP=15, OA=16, OB=17, HT=18, HB=19, HQ=20, HF=21, HP=22,
         GFS=23, IS=24, GS=25)


@pytest.fixture(scope='module')
def harness(tmp_path_factory):
    exe = tmp_path_factory.mktemp('unit') / 'last_interval'
    subprocess.run(['g++', '-std=c++17', '-O1', '-pthread', '-ffp-contract=off', '-I', str(EMU), '-o', str(exe), str(EMU / 'unit' / 'last_interval.cpp')], check=True)
    return exe


def _reference(H, h, dyn, Ptt, pt, x):
    "(u, lam, V(x) - V(0)) of the last interval's equality-constrained QP in 50 digits"
    import mpmath as mp
    mp.mp.dps = 50
    Tb, Tw, Bb, Bw, rt, rb = [mp.mpf(float(v)) for v in dyn]
    Hm = mp.matrix(6, 6); hv = mp.matrix(6, 1)
    for a in range(6):
        hv[a] = mp.mpf(float(h[a]))
        for b in range(6):
            Hm[a, b] = mp.mpf(H[a][b]) if isinstance(H[a][b], str) else mp.mpf(float(H[a][b]))
    a = mp.matrix([1, Tb, 0, Tw, Tw, 0]); c = mp.matrix([0, Bb, 0, Bw, Bw, 0])
    Hf = Hm + mp.mpf(Ptt)*(a*a.T); hf = hv + (mp.mpf(Ptt)*rt + mp.mpf(pt))*a

    def solve(xx):
        xx = mp.matrix([mp.mpf(float(v)) for v in xx])
        M = mp.matrix(4, 4); r = mp.matrix(4, 1)
        for i in range(3):
            for j in range(3):
                M[i, j] = Hf[3 + i, 3 + j]
            M[i, 3] = -c[3 + i]; M[3, i] = c[3 + i]
            r[i] = -(hf[3 + i] + sum(Hf[3 + i, j]*xx[j] for j in range(3)))
        r[3] = -(rb + sum(c[j]*xx[j] for j in range(3)))
        sol = mp.lu_solve(M, r)
        y = mp.matrix(list(xx) + [sol[0], sol[1], sol[2]])
        val = (y.T*Hf*y)[0]/2 + (hf.T*y)[0]
        return [sol[0], sol[1], sol[2]], sol[3], val
    u, lam, val = solve(x)
    _, _, v0 = solve([0, 0, 0])
    return [float(v) for v in u], float(lam), float(val - v0)


@pytest.mark.parametrize('case', ['f stiff', 'p stiff', 'both free', 'stiff common part', 'both on bounds'])
def test_last_interval_against_a_dense_solve_in_50_digits(harness, case):
    """
    msd::last_interval: value function of stage N-1, feedback of (Fel, Fpb, s) and the multiplier of the b row.  `f stiff` / `p stiff`: one force on a
    bound (barrier curvature 1e11) -- rounds 1-4 always eliminated Fel and lost every digit of P_bb in the first case (random sweep seed 176).
    `stiff common part`: the acceleration row active with both forces free -- 1e12 in Hff, Hfp and Hpp alike, the forces' own curvatures 1e-4:
    the pivot of the kept force comes from the own curvatures (S_OA, S_OB), not from Hff - 2 Hfp + Hpp.
    """
    rng = np.random.default_rng(abs(hash(case)) % 1000)
    A = rng.standard_normal((6, 6)); H = A@A.T + np.eye(6)
    H[0, 1:] = 0; H[1:, 0] = 0; H[2, 4] = H[4, 2] = 0; H[2, 5] = H[5, 2] = 0; H[1, 5] = H[5, 1] = 0; H[4, 5] = H[5, 4] = 0      # the block's sparsity (static loss rows)
    h = rng.standard_normal(6)
    oa, ob, common = {'f stiff': (1e11, 2e-3, 0.3), 'p stiff': (3e-3, 1e11, 0.3), 'both free': (0.7, 0.4, 0.2), 'stiff common part': (3e-4, 1e-4, 1e12),
                      'both on bounds': (2e10, 5e10, 7.0)}[case]
    H[3, 3], H[3, 4], H[4, 3], H[4, 4] = common + oa, common, common, common + ob
    H[3, 5] = H[5, 3] = 0.3*np.sqrt(min(oa, 1.0)*H[5, 5])      # (the slack's coupling with Fel within what keeps the problem convex on the b row's null space)
    dyn = (-0.4, -150.0, 0.98, 430.0, 1e-3, -2e-3)
    s = np.zeros(31)
    s[0:6] = dyn
    s[S['HTT']], s[S['HBB']], s[S['HBQ']], s[S['HBF']], s[S['HBP']], s[S['HQQ']], s[S['HQF']] = H[0, 0], H[1, 1], H[1, 2], H[1, 3], H[1, 4], H[2, 2], H[2, 3]
    s[S['HFF']], s[S['HFP']], s[S['HPP']], s[S['OA']], s[S['OB']] = H[3, 3], H[3, 4], H[4, 4], oa, ob
    s[S['HT']], s[S['HB']], s[S['HQ']], s[S['HF']], s[S['HP']] = h[:5]
    s[S['GFS']], s[S['IS']], s[S['GS']] = H[3, 5], 1/H[5, 5], h[5]
    Ptt, pt = 0.7, 0.3
    out = subprocess.run([str(harness)], input=' '.join('%.17g' % v for v in list(s) + [Ptt, pt]) + ' 1', capture_output=True, text=True, check=True).stdout.split('\n')
    assert int(out[0]) == 1
    Pn, pvn, K, KS, LG = [np.array(out[k].split(), float) for k in range(1, 6)]
    x = rng.standard_normal(3)
    # exact entries of the control block for the reference: common + own as strings of full precision
    import mpmath as mp
    mp.mp.dps = 50
    Hs = [[H[a][b] for b in range(6)] for a in range(6)]
    Hs[3][3] = mp.nstr(mp.mpf(common) + mp.mpf(oa), 40); Hs[4][4] = mp.nstr(mp.mpf(common) + mp.mpf(ob), 40)
    u, lam, dV = _reference(Hs, h, dyn, Ptt, pt, x)
    df, dp, ds = K[0:3]@x + K[6], K[3:6]@x + K[7], KS[:3]@x + KS[3]
    lb = (LG[0]*x[0] + LG[1]*x[1] + LG[2]*x[2] + LG[3]*df + LG[4]*dp + LG[5]*ds + LG[6])/dyn[3]
    Pm = np.array([[Pn[0], Pn[1], Pn[2]], [Pn[1], Pn[3], Pn[4]], [Pn[2], Pn[4], Pn[5]]])
    scale = max(1.0, np.max(np.abs(u)))
    assert np.max(np.abs(np.array([df, dp, ds]) - u)) <= 1e-9*scale, (case, df, dp, ds, u)
    assert abs((0.5*x@Pm@x + pvn@x) - dV) <= 1e-9*max(1.0, abs(dV)), (case, 0.5*x@Pm@x + pvn@x, dV)
    # the multiplier comes out as a sum of terms of the size of the stiff curvature times the step: absolute accuracy eps * that size
    assert abs(lb - lam) <= 1e-9*max(1.0, abs(lam)) + 1e-15*max(abs(LG[3]*df), abs(LG[4]*dp)), (case, lb, lam)
