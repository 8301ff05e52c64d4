"""
CPU-only checks of the drop-in boundary: the HIP library builds for gfx950, loads, and exports every
symbol include/mseetc_hip.h declares; the host-side front end validates options like the reference and
refuses -- loudly -- to solve without a device.  No compute call is made here.
"""

import ctypes
import subprocess
import re
from pathlib import Path

import numpy as np
import pytest

import cases

ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope='module')
def built():
    import __graft_entry__ as g
    return g.build()


def test_library_exports_every_declared_symbol(built):
    header = ''.join((ROOT / 'include' / h).read_text() for h in ('mseetc_hip.h', 'mseetc_mpc.h', 'mseetc_aux.h'))
    names = set(re.findall(r'\b(msd_[a-z_]+)\s*\(', header))
    assert len(names) >= 21 and {'msd_problem_first_pass_ms', 'msd_mpc_create', 'msd_mpc_run', 'msd_mpc_destroy', 'msd_mpc_nz', 'msd_problem_follow_counts'} <= names
    lib = ctypes.CDLL(str(built))
    for n in sorted(names):
        assert hasattr(lib, n), n


def test_desc_struct_matches_header(built, tmp_path):
    # the ctypes mirror against the C compiler's layout of the header: size and the offset of every field
    from mseetc._device import ProblemDesc, ABI_VERSION
    header = (ROOT / 'include' / 'mseetc_hip.h').read_text()
    assert '#define MSD_ABI_VERSION {}'.format(ABI_VERSION) in header
    fields = [f[0] for f in ProblemDesc._fields_]
    src = tmp_path / 'layout.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mseetc_hip.h"\nint main(void) {\n'
                   + '  printf("%zu\\n", sizeof(msd_problem_desc));\n'
                   + ''.join('  printf("%zu\\n", offsetof(msd_problem_desc, {}));\n'.format(f) for f in fields) + '  return 0;\n}\n')
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', '-I', str(ROOT / 'include'), '-o', str(exe), str(src)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(ProblemDesc)
    assert out[1:] == [getattr(ProblemDesc, f).offset for f in fields]


def test_mpc_plan_struct_matches_header(built, tmp_path):
    "the ctypes mirror of msd_mpc_plan (include/mseetc_mpc.h) against the C compiler's layout, and the log's column indices"
    from mseetc._device import MpcPlan, MPC
    fields = [f[0] for f in MpcPlan._fields_]
    src = tmp_path / 'layout_mpc.c'
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "mseetc_mpc.h"\nint main(void) {\n'
                   + '  printf("%zu\\n", sizeof(msd_mpc_plan));\n'
                   + ''.join('  printf("%zu\\n", offsetof(msd_mpc_plan, {}));\n'.format(f) for f in fields)
                   + '  printf("%d %d %d %d %d %d %d %d\\n", MSD_MPC_T0, MSD_MPC_V0, MSD_MPC_T, MSD_MPC_STATUS, MSD_MPC_ITERS, MSD_MPC_OBJ, MSD_MPC_RELAXED, MSD_MPC_COUNT);\n  return 0;\n}\n')
    exe = tmp_path / 'layout_mpc'
    subprocess.run(['gcc', '-I', str(ROOT / 'include'), '-o', str(exe), str(src)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    n = len(fields)
    assert out[0] == ctypes.sizeof(MpcPlan)
    assert out[1:1 + n] == [getattr(MpcPlan, f).offset for f in fields]
    assert out[1 + n:] == [MPC[k] for k in ('T0', 'V0', 'T', 'STATUS', 'ITERS', 'OBJ', 'RELAXED', 'COUNT')]


def test_no_device_fails_loudly(built):
    from mseetc import _device
    if _device.lib().msd_device_count() > 0:
        pytest.skip("a GPU is present")
    from mseetc.ocp import casadiSolver
    from mseetc._device import DeviceError
    solver = casadiSolver(cases.train_default(), cases.track_00(), {'numIntervals': 100, 'integrationOptions': {'numApproxSteps': 1}})
    with pytest.raises(DeviceError):
        solver.solveBatch([1541.0])


def test_front_end_option_validation():
    from mseetc.ocp import casadiSolver, OptionsCasadiSolver, OCP
    assert OCP is casadiSolver
    train, track = cases.train_default(), cases.track_00()
    for bad in ({'numIntervals': 0}, {'numIntervals': 10.5}, {'maxIterations': 0}, {'energyOptimal': 1}, {'minimumVelocity': -1},
                {'integrationMethod': 'LINEAR'}, {'integrateLosses': 'yes'}, {'noSuchOption': 1},
                {'integrationOptions': {'order': 3}}, {'integrationOptions': {'numSteps': 0}}, {'integrationOptions': {'bogus': 1}}):
        with pytest.raises(ValueError):
            casadiSolver(train, track, bad)
    o = OptionsCasadiSolver({'maxIterations': 500, 'numIntervals': 300, 'integrationMethod': 'RK',
                             'integrationOptions': {'order': 4, 'numSteps': 1, 'numApproxSteps': 1}})   # simulations/config.json
    assert o.maxIterations == 500 and o.integrationOptions.numApproxSteps == 1
    assert OptionsCasadiSolver({}).maxIterations == 1e3    # float default accepted (ocp.py:18)
    s = casadiSolver(train, track, {'numIntervals': 100})
    assert s.numIntervals == 100 and len(s.points) == 101 and len(s.steps) == 100
    with pytest.raises(ValueError):
        s.solve(-5)
    with pytest.raises(ValueError):
        s.solve(100, initialTime=-1)
    with pytest.raises(ValueError):
        s.solve('100')


def test_front_end_packs_the_same_problem_as_the_oracle_packer():
    # two independent restatements of ocp.py:96-125,266-269: the product's front end and oracle.pack_problem
    from mseetc.ocp import casadiSolver
    from oracle.oracle import DP, IP
    for train, track, N, eo in [(cases.train_default(), cases.track_00(), 100, True), (cases.train_fig10(), cases.track_CH(), 200, True),
                                (cases.train_fig5(), cases.track_00(8500), 300, False)]:
        opts = dict(numIntervals=N, energyOptimal=eo, maxIterations=500, integrationOptions=dict(numSteps=1, numApproxSteps=1))
        if not eo:
            train.powerLosses = lambda f, v: 0
        d = casadiSolver(train, track, opts)._desc
        p = cases.oracle_problem(train, track, N, energyOptimal=eo, losses='static' if eo else 'none')
        assert (d.num_intervals, d.with_pn_brake, d.has_power_rows, d.energy_optimal) == tuple(int(p.ip[IP[k]]) for k in ('N', 'WITH_PN', 'HAS_POWER', 'ENERGY_OPT'))
        for a, k in [('sr0', 'SR0'), ('sr1', 'SR1'), ('sr2', 'SR2'), ('f_max', 'FMAX'), ('f_min', 'FMIN'), ('f_min_pn', 'FMIN_PN'), ('pw_upper', 'PW_UPPER'),
                     ('pw_lower', 'PW_LOWER'), ('acc_min', 'ACC_MIN'), ('acc_max', 'ACC_MAX'), ('loss_ct', 'LOSS_CT'), ('loss_cr', 'LOSS_CR'),
                     ('vmin_sq', 'VMIN_SQ'), ('obj_den', 'OBJ_DEN'), ('tol', 'TOL')]:
            assert getattr(d, a) == p.dp[DP[k]], a
        for arr, ref in zip(d._keep, (p.ds, p.grad, p.curv, p.bmax)):
            assert np.array_equal(arr, ref)
