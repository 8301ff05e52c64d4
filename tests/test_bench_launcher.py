"""
bench.py's launch logic on a machine without a GPU: `--gpus N` never fails on WORLD_SIZE, it reports that there is no device;
the command it would start the ranks with is the torch.distributed.run line the driver uses; and that launcher really runs
the given script as N ranks over 127.0.0.1 (checked with a gloo group on CPU).
"""

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import bench      # noqa: E402


def _clean_env():
    env = dict(os.environ)
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    return env


@pytest.mark.parametrize('gpus', [1, 2, 8])
def test_no_device_is_reported_with_its_own_exit_code(gpus):
    import torch
    if torch.cuda.device_count() >= gpus:
        pytest.skip("this machine has the devices")
    r = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', str(gpus), '--no-build'], capture_output=True, text=True, env=_clean_env(), timeout=600)
    # no JSON line can be produced here: a non-zero exit code with an explicit message, never a stack trace and never a silent success
    # (--no-build exits with a message of its own when the library is stale: also a clean, explicit exit)
    assert r.returncode != 0, "nothing was measured: the exit code must say so"
    assert r.returncode == 2 or 'stale' in r.stderr, r.stderr
    assert 'WORLD_SIZE' not in r.stderr and 'Traceback' not in r.stderr
    if r.returncode == 2:
        assert 'HIP device' in r.stderr and r.stdout.strip() == ''


def test_rank_command_is_the_drivers_launch_line():
    cmd = bench.rank_command(['--gpus', '4', '--steps', '5'], 4, 29511)
    assert cmd[1:3] == ['-m', 'torch.distributed.run']
    assert '--nnodes=1' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[cmd.index('--master-port') + 1] == '29511'
    assert cmd[-5].endswith('bench.py') and cmd[-4:] == ['--gpus', '4', '--steps', '5']


def test_launcher_runs_the_ranks(tmp_path):
    "The same launcher line with a stand-in script: two ranks form a gloo group, reduce MAX like bench.py's timing, rank 0 prints one JSON line."
    script = tmp_path / 'ranks.py'
    script.write_text(
        "import json, os, torch, torch.distributed as dist\n"
        "dist.init_process_group('gloo')\n"
        "r, w = dist.get_rank(), dist.get_world_size()\n"
        "t = torch.tensor([1.0 + r], dtype=torch.float64)\n"
        "dist.all_reduce(t, op=dist.ReduceOp.MAX)\n"
        "n = torch.tensor([10 + r]); dist.all_reduce(n, op=dist.ReduceOp.SUM)\n"
        "dist.barrier()\n"
        "if r == 0: print(json.dumps({'world': w, 'max': float(t.item()), 'sum': int(n.item())}), flush=True)\n"
        "dist.destroy_process_group()\n")
    cmd = bench.rank_command([], 2, bench.free_port())
    cmd[cmd.index(str(ROOT / 'bench.py'))] = str(script)
    r = subprocess.run(cmd, capture_output=True, text=True, env=_clean_env(), timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    assert json.loads(line) == {'world': 2, 'max': 2.0, 'sum': 21}


@pytest.mark.gpu
def test_two_ranks_on_the_gpu_box():
    """
    The whole multi-rank path of bench.py on a box with one GPU: the parent starts torch.distributed.run, two ranks share device 0
    (MSD_BENCH_SHARE_DEVICES=1: gloo instead of RCCL for the barrier and the reductions, everything else as in an 8-GPU run) and rank 0
    prints the aggregate line.
    """
    import json, os, subprocess, sys
    env = dict(os.environ, MSD_BENCH_SHARE_DEVICES='1')
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '1', '--no-build'], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['scaling'] == 'weak' and line['config']['scenarios'] == 2048 and line['config']['converged'] == 2048
    assert line['value'] > 1e5 and 'alt' not in line and 'cpu_baseline' not in line
    # the line says what carried the barrier and the reductions, how many ranks a reduction saw and which device every rank drove
    cfg = line['config']
    assert cfg['process_group_backend'].startswith('gloo (MSD_BENCH_SHARE_DEVICES=1') and cfg['world_size_seen'] == 2 and cfg['device_of_rank'] == [0, 0]
    assert line['roofline']['bound'] == 'hbm' and 0 < line['roofline']['frac_useful'] < 1
    # without the sharing switch two ranks on one device cannot form an RCCL group: the run fails loudly instead of falling back to gloo on its own
    env2 = {k: v for k, v in os.environ.items() if k not in ('MSD_BENCH_SHARE_DEVICES', 'MSD_BENCH_GLOO_FALLBACK')}
    import torch
    if torch.cuda.device_count() == 1:
        out2 = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--no-build'], env=env2, capture_output=True, text=True, timeout=600)
        assert out2.returncode != 0 and not [l for l in out2.stdout.splitlines() if l.startswith('{')]


@pytest.mark.gpu
@pytest.mark.parametrize('how', ['in_process', 'under_the_launcher'])
def test_rccl_process_group_at_world_size_one(how):
    """
    The RCCL branch of bench.py -- `nccl` process group, the probing all-reduce, the barrier in front of and behind the timed region, the MAX and SUM
    reductions -- on the one GPU of the box: `--process-group` forms the group at world size 1, directly and under the driver's launcher line
    (torch.distributed.run --nproc-per-node 1), so that an 8-GPU run does not meet that code for the first time.  No scaling claim comes out of it.
    """
    env = _clean_env()
    env['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'
    args = ['--gpus', '1', '--steps', '3', '--warmup', '1', '--no-build', '--no-alt', '--no-cpu-baseline', '--process-group']
    cmd = [sys.executable, str(ROOT / 'bench.py')] + args if how == 'in_process' else bench.rank_command(args, 1, bench.free_port())
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['config']['converged'] == 1024 and line['value'] > 1e5
    assert line['config']['process_group_backend'] == 'nccl (RCCL)' and line['config']['world_size_seen'] == 1 and line['config']['device_of_rank'] == [0]
    assert line['config']['by_rank']['ip_iterations_mean'][0] == pytest.approx(line['config']['ip_iterations_mean'])


@pytest.mark.gpu
def test_eight_ranks_config3_and_config4_on_the_gpu_box():
    """
    The launcher path of an 8-GPU run for BASELINE configs 3 and 4 at their full sizes, on the one device of the box (MSD_BENCH_SHARE_DEVICES=1:
    rank r -> device r % 1, gloo for the barrier and the reductions): config 3 = 8 x 8192 scenarios with per-scenario rolling stock -- the
    aggregate line parses, every one of the 65 536 scenarios converges, the ranks solved different batches (seed + rank) --, config 4 =
    8 x 512 scenarios x 50 re-solves through the device-resident loop.  No scaling number comes out of this: eight ranks share one GPU.
    """
    import json, os, subprocess, sys
    env = dict(os.environ, MSD_BENCH_SHARE_DEVICES='1')
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '8', '--workload', 'c3', '--steps', '2', '--warmup', '1', '--no-build'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 8 and line['scaling'] == 'weak' and line['config']['scenarios'] == 65536 and line['config']['converged'] == 65536
    br = line['config']['by_rank']
    assert len(br['ip_iterations_mean']) == 8 and len(set(br['first_running_time'])) == 8 and len(set(br['ip_iterations_mean'])) > 1
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '8', '--workload', 'c4', '--steps', '1', '--warmup', '1', '--no-build'], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 8 and line['config']['batch_per_gpu'] == 512
    assert line['config']['resolves_successful'] + line['config']['resolves_failed'] == 8*512*50 and line['config']['resolves_failed'] <= 8*4


def test_default_step_counts_and_options():
    "about a second of launches per workload without flags; explicit flags win; the transcriptions map onto the reference's options"
    a = bench.parse_args([])
    assert (a.workload, a.steps, a.warmup, a.gpus, a.transcription, a.single_process) == ('c1', 500, 10, 1, 'rk', False)
    assert bench.parse_args(['--workload', 'c4']).steps == 5 and bench.parse_args(['--workload', 'c4']).warmup == 1
    assert bench.parse_args(['--workload', 'c2']).steps == 50
    b = bench.parse_args(['--gpus', '8', '--steps', '20', '--warmup', '3', '--single-process', '--transcription', 'irk_radau2'])
    assert (b.gpus, b.steps, b.warmup, b.single_process, b.transcription) == (8, 20, 3, True, 'irk_radau2')
    extra, io = bench.TRANSCRIPTIONS['cvodes_tolerances']
    assert extra == dict(integrationMethod='CVODES') and io == {}
    assert bench.TRANSCRIPTIONS['integrate_losses'][0] == dict(integrateLosses=True)
    assert set(bench.PER_GPU_BATCH) == {'c1', 'c2', 'c3', 'c4'} and bench.PER_GPU_BATCH['c4']*8 == 4096 and bench.PER_GPU_BATCH['c3']*8 == 65536


def test_roofline_block_fields():
    "the roofline object carries the contract's fields and the three fractions; without a matching traffic record the measured one is null"
    class E:
        @staticmethod
        def hip_digest():
            return 'no such digest'
    r = bench.roofline_block(E, 'c1', 1024, 100, 502, 100*1024*20.6, 1.15, (64, 2))
    for key in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'frac_model_S', 'frac_compulsory', 'frac_measured', 'frac_useful', 'valu_issue', 'limiter', 'launch_ms', 'kernel_ms'):
        assert key in r
    # the top-level fields are always SURVEY 8(d)'s pricing against the HBM roof (the issue roof sits in `valu_issue` when counters are on file)
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and r['valu_issue'] is None
    assert abs(r['frac_useful'] - 400*100*1024*20.6/1.15e-3/1e12/78.6) < 1e-12
    assert abs(r['achieved'] - 904*100*1024*20.6/1.15e-3/1e9) < 1e-6*r['achieved'] and abs(r['frac'] - r['achieved']/8000.0) < 1e-12
    assert r['traffic'] is None and r['frac_measured'] is None
    assert abs(r['compulsory_bytes_per_launch'] - 1024*(8*502 + 168)) < 1e-9
