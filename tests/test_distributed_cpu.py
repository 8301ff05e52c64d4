"""
The N > 1 path on CPU: two processes, gloo backend.  The HIP library cannot run here, so the per-rank compute is done by
the CPU oracle standing in for the device (the oracle is the checker of the sharding logic, not a product fallback):
sharded + gathered result == unsharded result, bit for bit, for even and ragged splits.
"""

import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import cases
from mseetc.sharding import shard_bounds, solve_sharded


def test_shard_bounds_cover_the_batch_exactly():
    for B in (0, 1, 7, 8, 1024, 65536, 65537):
        for W in (1, 2, 3, 4, 8):
            cuts = [shard_bounds(B, W, r) for r in range(W)]
            assert cuts[0][0] == 0 and cuts[-1][1] == B
            assert all(cuts[r][1] == cuts[r + 1][0] for r in range(W - 1))
            sizes = [hi - lo for lo, hi in cuts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(8, 2, 2)


def _oracle_slice(prob):
    from oracle import oracle

    def run(scen):
        z, st, nfail = oracle.solve_batch(prob, scen, nthreads=1)
        return dict(z=z, stats=st)
    return run


def _worker(rank, world, port, B, out_path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    here = os.path.dirname(os.path.abspath(__file__))
    for p in (os.path.join(here, '..', 'ms-eetc_amd'), os.path.join(here, '..'), here):
        if p not in sys.path:
            sys.path.insert(0, p)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import cases as cs
    prob = cs.oracle_problem(cs.train_default(), cs.track_00(crop=20000), 40)
    T = 700 + 300*np.random.default_rng(5).random(B)
    scen = np.stack([np.zeros(B), T, np.ones(B), np.ones(B)], axis=1)
    res = solve_sharded(_oracle_slice(prob), scen, rank=rank, world_size=world)
    if rank == 0:
        np.savez(out_path, z=res['z'], stats=res['stats'])
    else:
        assert res is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize('B,world', [(6, 2), (5, 2), (1, 2), (11, 4), (3, 4), (17, 8)])
def test_gloo_ranks_equal_single_process(tmp_path, B, world):
    "2, 4 and 8 ranks (the node sizes of BASELINE's metric), even, ragged and emptier-than-ranks splits: sharded + gathered == unsharded, bit for bit"
    out = str(tmp_path / 'res.npz')
    port = 29500 + (os.getpid() + 7*B + world) % 2000
    mp.spawn(_worker, args=(world, port, B, out), nprocs=world, join=True)
    got = np.load(out)
    prob = cases.oracle_problem(cases.train_default(), cases.track_00(crop=20000), 40)
    T = 700 + 300*np.random.default_rng(5).random(B)
    scen = np.stack([np.zeros(B), T, np.ones(B), np.ones(B)], axis=1)
    ref = _oracle_slice(prob)(scen)
    assert got['z'].shape == ref['z'].shape
    assert np.array_equal(got['z'], ref['z'])
    assert np.array_equal(got['stats'][:, :3], ref['stats'][:, :3])
