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
    pass rng_seed + 2 * first_chain to the context (or first_chain to hlala_batch_set_first_chain) so every DP draws the seed it would
    draw in the unsharded run."""
    p0, p1 = shard_bounds(batch["n_pairs"], rank, world)
    return shard_pairs_range(batch, p0, p1)


def shard_pairs_range(batch: dict, p0: int, p1: int):
    """Pairs [p0, p1) of a hlala_batch_in dict as a batch of their own: (sub_batch, first_pair, first_chain)."""
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


# --------------------------------------------------------------------------------------------------------------------------------
# From the shards to the call.  What HLATypeInference consumes per locus is not the per-pair scalars but the EXON POSITIONS of the
# pairs that overlap the locus (hla/HLATyper.cpp:1386-1428) -- a few hundred to a few thousand reads per locus, ragged.  Every rank
# extracts them from its own resident batch; ONE ragged gather (sizes first, then one padded byte payload: RCCL gather over xGMI on
# GPUs, gloo in the CPU tests) brings them to rank 0, which runs filters -> likelihoods -> all pairs -> call exactly as the unsharded
# run does.  Pairs are sharded in contiguous blocks, so concatenating the ranks' lists in rank order reproduces the unsharded order.

_EXON_ARRAYS = ("read_pair", "read_weighted_ok", "read_fraction_ok", "read_distance", "read_cols_nongap", "pos_off", "pos_exon", "pos_level", "pos_mate",
                "pos_mapq", "pos_novel_gap", "geno_off", "geno_chars", "qual_chars", "read_reverse", "read_mapq")


def pack_arrays(arrays: dict) -> np.ndarray:
    """dict of 1-D numpy arrays -> one uint8 buffer: int64 header (count, then per array: name length, dtype code length, elements), names /
    dtype strings, then the raw bytes of every array padded to 8."""
    names = sorted(arrays)
    head = [len(names)]; blobs = []; meta = b""
    for k in names:
        a = np.ascontiguousarray(arrays[k]); assert a.ndim == 1, k
        nb = k.encode(); db = a.dtype.str.encode()
        head += [len(nb), len(db), a.shape[0]]; meta += nb + db
        raw = a.tobytes(); blobs.append(raw + b"\0" * (-len(raw) % 8))
    meta += b"\0" * (-len(meta) % 8)
    return np.frombuffer(np.asarray(head, np.int64).tobytes() + meta + b"".join(blobs), np.uint8).copy()


def unpack_arrays(buf: np.ndarray) -> dict:
    raw = np.ascontiguousarray(buf, np.uint8).tobytes()
    n = int(np.frombuffer(raw[:8], np.int64)[0]); head = np.frombuffer(raw[8:8 + 24 * n], np.int64).reshape(n, 3)
    p = 8 + 24 * n; metas = []
    for ln, ld, cnt in head:
        metas.append((raw[p:p + ln].decode(), np.dtype(raw[p + ln:p + ln + ld].decode()), int(cnt))); p += int(ln + ld)
    p += -(p - (8 + 24 * n)) % 8
    out = {}
    for name, dt, cnt in metas:
        nbytes = cnt * dt.itemsize
        out[name] = np.frombuffer(raw[p:p + nbytes], dt).copy(); p += nbytes + (-nbytes % 8)
    return out


def gather_ragged(arrays: dict, dst: int = 0, device=None):
    """Gather a dict of 1-D arrays of rank-dependent lengths to `dst`: an all_gather of the payload sizes, then ONE gather of the packed
    bytes padded to the largest payload.  Returns the list of per-rank dicts on dst, None elsewhere.  `device`: where the process group
    works (None: cpu / gloo; a cuda device for nccl == RCCL)."""
    import torch
    import torch.distributed as dist
    world, rank = dist.get_world_size(), dist.get_rank()
    payload = torch.from_numpy(pack_arrays(arrays))
    if device is not None:
        payload = payload.to(device)
    n = torch.tensor([payload.shape[0]], dtype=torch.int64, device=payload.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    mx = max(int(s.item()) for s in sizes)
    pad = torch.zeros(mx, dtype=torch.uint8, device=payload.device); pad[:payload.shape[0]] = payload
    out = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
    dist.gather(pad, out, dst=dst)
    if rank != dst:
        return None
    return [unpack_arrays(o[:int(s.item())].cpu().numpy()) for o, s in zip(out, sizes)]


def merge_exon_positions(parts, first_pairs):
    """Concatenate the exon-position lists of the ranks (dicts as Batch.exon_positions returns them) into the list of the unsharded run:
    pair indices back to the global numbering, offset arrays chained."""
    out = {k: [] for k in _EXON_ARRAYS}
    pbase = gbase = 0; n_ok = n_broken = 0
    pos_off = [np.zeros(1, np.int32)]; geno_off = [np.zeros(1, np.int32)]
    for e, p0 in zip(parts, first_pairs):
        for k in _EXON_ARRAYS:
            if k == "read_pair":
                out[k].append(np.asarray(e[k], np.int32) + np.int32(p0))
            elif k == "pos_off":
                pos_off.append(np.asarray(e[k][1:], np.int32) + np.int32(pbase))
            elif k == "geno_off":
                geno_off.append(np.asarray(e[k][1:], np.int32) + np.int32(gbase))
            else:
                out[k].append(np.asarray(e[k]))
        pbase += int(e["pos_off"][-1]) if len(e["pos_off"]) else 0
        gbase += int(e["geno_off"][-1]) if len(e["geno_off"]) else 0
        n_ok += int(np.asarray(e["counts"])[0]); n_broken += int(np.asarray(e["counts"])[1])
    m = {k: np.concatenate(v) for k, v in out.items() if k not in ("pos_off", "geno_off")}
    m["pos_off"] = np.concatenate(pos_off); m["geno_off"] = np.concatenate(geno_off)
    m.update(n_reads=len(m["read_pair"]), n_pos=len(m["pos_exon"]), n_chars=len(m["geno_chars"]), n_pairs_ok=n_ok, n_pairs_broken=n_broken)
    return m


def gather_exon_positions(local: dict, first_pair: int, dst: int = 0, device=None):
    """The one exchange of the typing path: this rank's exon positions of one locus -> the merged list on `dst` (None elsewhere)."""
    arrays = {k: np.asarray(local[k]) for k in _EXON_ARRAYS}
    arrays["counts"] = np.asarray([local["n_pairs_ok"], local["n_pairs_broken"], first_pair], np.int64)
    parts = gather_ragged(arrays, dst=dst, device=device)
    if parts is None:
        return None
    return merge_exon_positions(parts, [int(p["counts"][2]) for p in parts])


def call_locus_sharded(engine, local_positions: dict, first_pair: int, cluster_seq, n_clusters: int, n_columns: int, filter_params, dst: int = 0, device=None):
    """One locus across ranks: gather the exon positions, then on `dst` the unsharded chain -- read / allele filters (host), per-cluster x
    per-read likelihoods, all cluster pairs, the call.  `engine` supplies the four steps (the product: libhlala_gpu.so through the ctypes
    binding; the CPU test: the oracle): filter_positions(e, params) -> (use, ignored, stats); exon_loglik(exon_in) -> (LL, mism);
    pair_loglik(LL, mism) -> (pairLL, misAvg, misMin); call_locus(pairLL, misAvg, misMin) -> dict.  Returns the call dict on dst."""
    e = gather_exon_positions(local_positions, first_pair, dst=dst, device=device)
    if e is None:
        return None
    use, ignored, stats = engine.filter_positions(e, filter_params)
    xin = engine.exon_in(e, use, cluster_seq, n_clusters, n_columns)
    LL, M = engine.exon_loglik(xin)
    pl = engine.pair_loglik(LL, M)
    call = dict(engine.call_locus(*pl))
    call.update(positions=e, pos_use=use, pair_ll=pl[0], mis_avg=pl[1], mis_min=pl[2])
    return call
