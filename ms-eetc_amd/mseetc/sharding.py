"""
Multi-GPU use of the batched solver: scenarios are independent, so a batch is cut into contiguous slices, one per
rank (= one process per GPU), and every rank solves its slice with its own device.  There is no collective on the
data path; `torch.distributed` is only used to collect the results on rank 0 (NCCL/RCCL on GPUs, gloo in the CPU tests).
"""

import numpy as np


def shard_bounds(num_scenarios, world_size, rank):
    "Contiguous slice [lo, hi) of rank `rank`; sizes differ by at most one."

    if not 0 <= rank < world_size:
        raise ValueError("rank out of range")

    base, extra = divmod(int(num_scenarios), int(world_size))
    lo = rank*base + min(rank, extra)

    return lo, lo + base + (1 if rank < extra else 0)


def solve_sharded(solve_slice, scenarios, rank=0, world_size=1, group=None):
    """
    `solve_slice(scen) -> dict(z=(b, nz), stats=(b, ns))` is called with this rank's slice of `scenarios` (B, 4).
    Returns the assembled dict(z, stats) on rank 0 and None elsewhere.
    """

    scenarios = np.ascontiguousarray(scenarios, dtype=np.float64)
    B = scenarios.shape[0]
    lo, hi = shard_bounds(B, world_size, rank)
    mine = solve_slice(scenarios[lo:hi]) if hi > lo else None

    if world_size == 1:
        return dict(z=mine['z'], stats=mine['stats'])

    import torch
    import torch.distributed as dist

    # shapes are known on every rank: (hi-lo, nz) and (hi-lo, ns); empty slices send zero rows
    meta = [None]*world_size
    dist.all_gather_object(meta, None if mine is None else (mine['z'].shape[1], mine['stats'].shape[1]), group=group)
    nz, ns = next(m for m in meta if m is not None)

    # gather needs equally sized tensors: every slice is padded to the largest one and trimmed on rank 0
    sizes = [shard_bounds(B, world_size, r)[1] - shard_bounds(B, world_size, r)[0] for r in range(world_size)]
    rows = max(sizes)

    def pack(a, cols):
        out = np.zeros((rows, cols))
        if a is not None:
            out[:a.shape[0]] = a
        return torch.from_numpy(out)

    device = torch.device('cuda', torch.cuda.current_device()) if dist.get_backend(group) == 'nccl' else torch.device('cpu')

    zt = pack(None if mine is None else mine['z'], nz).to(device)
    st = pack(None if mine is None else mine['stats'], ns).to(device)
    zs = [torch.zeros((rows, nz), dtype=torch.float64, device=device) for _ in range(world_size)] if rank == 0 else None
    ss = [torch.zeros((rows, ns), dtype=torch.float64, device=device) for _ in range(world_size)] if rank == 0 else None

    dist.gather(zt, zs, dst=0, group=group)
    dist.gather(st, ss, dst=0, group=group)

    if rank != 0:
        return None

    return dict(z=torch.cat([t[:m] for t, m in zip(zs, sizes)]).cpu().numpy(), stats=torch.cat([t[:m] for t, m in zip(ss, sizes)]).cpu().numpy())
