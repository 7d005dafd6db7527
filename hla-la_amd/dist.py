"""Multi-GPU plumbing: read pairs shard embarrassingly (one process per GPU); the only exchange is one gather of
fixed-size per-pair records to rank 0 (RCCL over xGMI on GPUs, gloo in the CPU tests).  No data-path collective."""
from __future__ import annotations

import numpy as np

_PER_CHAIN = ("chain_contig", "chain_pos", "chain_offset", "chain_as", "chain_reverse")


def shard_bounds(n_units: int, rank: int, world: int):
    """Contiguous block partition: the first n_units % world ranks get one extra unit."""
    q, r = divmod(n_units, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_pairs(batch: dict, rank: int, world: int):
    """Slice a hlala_batch_in dict to this rank's block of pairs.  Returns (sub_batch, first_pair, first_chain);
    pass rng_seed + 2 * first_chain to the context so every DP draws the seed it would draw in the unsharded run."""
    p0, p1 = shard_bounds(batch["n_pairs"], rank, world)
    r0, r1 = 2 * p0, 2 * p1
    c0, c1 = int(batch["chain_off"][r0]), int(batch["chain_off"][r1])
    b0, b1 = int(batch["read_off"][r0]), int(batch["read_off"][r1])
    g0, g1 = int(batch["cigar_off"][c0]), int(batch["cigar_off"][c1])
    sub = dict(n_pairs=p1 - p0,
               read_off=(np.asarray(batch["read_off"][r0:r1 + 1]) - b0).astype(np.int32),
               read_bases=np.asarray(batch["read_bases"][b0:b1]), read_quals=np.asarray(batch["read_quals"][b0:b1]),
               chain_off=(np.asarray(batch["chain_off"][r0:r1 + 1]) - c0).astype(np.int32),
               read_primary=(np.asarray(batch["read_primary"][r0:r1]) - c0).astype(np.int32),
               n_chains=c1 - c0,
               cigar_off=(np.asarray(batch["cigar_off"][c0:c1 + 1]) - g0).astype(np.int32),
               cigar=np.asarray(batch["cigar"][g0:g1]))
    for k in _PER_CHAIN:
        sub[k] = np.asarray(batch[k][c0:c1])
    for k in ("insert_mean", "insert_sd"):
        if k in batch:
            sub[k] = batch[k]
    return sub, p0, c0


def gather_records(local, dst: int = 0):
    """Gather equally-shaped per-pair record tensors to `dst` (torch.distributed; backend nccl == RCCL on ROCm).
    Ragged shards are padded to the largest block; returns the list of per-rank tensors on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    mx = int(max(int(s.item()) for s in sizes))
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    out = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, out, dst=dst)
    if rank != dst:
        return None
    return [o[:int(s.item())] for o, s in zip(out, sizes)]


def reduce_coverage(local, dst: int = 0):
    """Sum the per-rank coverage counters (bases_per_level, int32[L-1]) on `dst`: the one reduction of the path
    (the reference sums its per-thread counters the same way, processBAM.cpp:1866-1887).  `local` is a torch tensor
    on the device the process group works on (cuda for nccl == RCCL, cpu for gloo); returns the total on dst, None elsewhere."""
    import torch.distributed as dist
    t = local.clone()
    dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)
    return t if dist.get_rank() == dst else None
