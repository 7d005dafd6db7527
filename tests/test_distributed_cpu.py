"""CPU, world_size 2 over gloo: the sharding + gather plumbing of the N > 1 path reproduces the unsharded result.
The per-rank engine here is the oracle (test infrastructure) -- on GPUs it is libhlala_gpu.so; the plumbing is identical."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _records(pairs):
    n = len(pairs["pair_ll"])
    rec = np.zeros((n, 8))
    rec[:, 0] = pairs["pair_status"]; rec[:, 1] = pairs["best_chain"][0::2]; rec[:, 2] = pairs["best_chain"][1::2]
    rec[:, 3] = pairs["n_combinations"]; rec[:, 4] = pairs["pair_ll"]; rec[:, 5] = pairs["pair_mapq"]
    rec[:, 6] = pairs["mate_mapq"][0::2]; rec[:, 7] = pairs["mate_mapq"][1::2]
    return rec


def _worker(rank, world, port, n_pairs, tmp):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import load_package
    from tools import synth
    from oracle_binding import Oracle
    P = load_package()
    import importlib.util
    spec = importlib.util.spec_from_file_location("hla_la_amd.dist", os.path.join(ROOT, "hla-la_amd", "dist.py"))
    D = importlib.util.module_from_spec(spec); spec.loader.exec_module(D)
    w = synth.make_world(seed=7, G=6000, k=1)
    b = synth.make_batch(w, n_pairs, seed=8)
    sub, p0, c0 = D.shard_pairs(b, rank, world)
    o = Oracle(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=(99 + 2 * c0) & 0xFFFFFFFF)
    pr = o.align_batch(sub)["pairs"]
    rec = _records(pr)
    rec[:, 1:3] += c0                     # chain indices back to the global numbering
    got = D.gather_records(torch.from_numpy(rec), dst=0)
    # per-pair post-processing: the coverage counters of the shards sum to the counters of the unsharded run
    import oracle_binding as ob
    nlev = int(w["graph"]["n_levels"]); genes = (np.array([100, 3000], np.int32), np.array([400, 3300], np.int32))
    cov, inc = ob.postprocess_pairs(pr, sub["n_pairs"], o.max_columns, genes[0], genes[1], nlev - 1)
    tot = D.reduce_coverage(torch.from_numpy(cov), dst=0)
    incs = D.gather_records(torch.from_numpy(inc.astype(np.int64)), dst=0)
    if rank == 0:
        fullp = Oracle(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=99).align_batch(b)["pairs"]
        full = _records(fullp)
        cat = torch.cat(got).numpy()
        cov_full, inc_full = ob.postprocess_pairs(fullp, b["n_pairs"], o.max_columns, genes[0], genes[1], nlev - 1)
        ok = np.array_equal(cat, full) and np.array_equal(tot.numpy(), cov_full) and np.array_equal(torch.cat(incs).numpy(), inc_full) and cov_full.sum() > 0
        np.save(os.path.join(tmp, "ok.npy"), np.array([int(ok), cat.shape[0]]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [61])      # odd: ragged shards (31 + 30)
def test_sharded_gather_equals_unsharded(tmp_path, oracle, n_pairs):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, n_pairs, str(tmp_path)), nprocs=2, join=True)
    ok = np.load(tmp_path / "ok.npy")
    assert ok[0] == 1 and ok[1] == n_pairs


def test_shard_bounds_cover_everything():
    import importlib.util
    spec = importlib.util.spec_from_file_location("d", os.path.join(ROOT, "hla-la_amd", "dist.py"))
    D = importlib.util.module_from_spec(spec); spec.loader.exec_module(D)
    for n in (0, 1, 7, 8, 1000003):
        for world in (1, 2, 4, 8):
            edges = [D.shard_bounds(n, r, world) for r in range(world)]
            assert edges[0][0] == 0 and edges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(edges, edges[1:]))
            assert max(e[1] - e[0] for e in edges) - min(e[1] - e[0] for e in edges) <= 1


# ------------------------------------------------------------------------------------------------ shards -> gather -> call on rank 0

LOCUS = dict(lmin=2000, exons=((100, 400, 0), (800, 1100, 300)), width=1300, P_ex=600, Cn=40)


def _locus_arrays():
    l2e = np.full(LOCUS["width"], -1, np.int32)
    for a, b, o in LOCUS["exons"]:
        l2e[a:b] = np.arange(o, o + (b - a))
    rng = np.random.default_rng(3); nuc = np.frombuffer(b"ACGT", np.uint8)
    cons = nuc[rng.integers(0, 4, LOCUS["P_ex"])]
    seqs = np.tile(cons, (LOCUS["Cn"], 1)); snp = rng.random(seqs.shape) < 0.03; seqs[snp] = nuc[rng.integers(0, 4, int(snp.sum()))]
    return l2e, seqs


class OracleEngine:
    """The four rank-0 steps of hla-la_amd/dist.py:call_locus_sharded on the CPU checker (the product engine is tests/test_distributed_gpu.py)."""
    def __init__(self, P, ob):
        self.P, self.ob = P, ob

    def filter_positions(self, e, prm):
        return self.ob.filter_positions(e, prm)

    def exon_in(self, e, use, seqs, Cn, Pex):
        return self.P.exon_in_from_positions(e, use, seqs, Cn, Pex)

    def exon_loglik(self, xin):
        return self.ob.exon_loglik(xin)

    def pair_loglik(self, LL, M):
        return self.ob.pair_loglik(LL, M)

    def call_locus(self, a, b, c):
        return self.ob.call_locus(a, b, c)


def _typing_worker(rank, world, port, n_pairs, tmp):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import load_package
    from tools import synth
    import oracle_binding as ob
    P = load_package()
    import importlib.util
    spec = importlib.util.spec_from_file_location("hla_la_amd.dist", os.path.join(ROOT, "hla-la_amd", "dist.py"))
    D = importlib.util.module_from_spec(spec); spec.loader.exec_module(D)
    w = synth.make_world(seed=11, G=6000, k=1)
    b = synth.make_batch(w, n_pairs, seed=12)
    l2e, seqs = _locus_arrays(); lmin = LOCUS["lmin"]
    gene = (np.array([lmin], np.int32), np.array([lmin + len(l2e) - 1], np.int32)); n_cov = int(w["graph"]["n_levels"]) - 1
    prm = P.default_filter_params(first20_n=6, first20_limit_per_read=0)
    eng = OracleEngine(P, ob)

    def positions(batch, rng_seed):
        o = ob.Oracle(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=rng_seed & 0xFFFFFFFF)
        pr = o.align_batch(batch)["pairs"]
        _, inc = ob.postprocess_pairs(pr, batch["n_pairs"], o.max_columns, gene[0], gene[1], n_cov)
        return ob.exon_positions(pr, batch, o.max_columns, lmin, l2e, b["insert_mean"], b["insert_sd"], pair_mask=inc)

    sub, p0, c0 = D.shard_pairs(b, rank, world)
    local = positions(sub, 99 + 2 * c0)
    local.setdefault("read_reverse", np.zeros(2 * local["n_reads"], np.uint8))
    call = D.call_locus_sharded(eng, local, p0, seqs, LOCUS["Cn"], LOCUS["P_ex"], prm, dst=0)
    if rank == 0:
        full = positions(b, 99)
        full.setdefault("read_reverse", np.zeros(2 * full["n_reads"], np.uint8))
        ok = True
        for k in D._EXON_ARRAYS:
            ok = ok and np.array_equal(np.asarray(call["positions"][k]), np.asarray(full[k]), equal_nan=True)
        ok = ok and call["positions"]["n_pairs_ok"] == full["n_pairs_ok"] and call["positions"]["n_pairs_broken"] == full["n_pairs_broken"]
        use, ign, st = eng.filter_positions(full, prm)
        LL, M = eng.exon_loglik(eng.exon_in(full, use, seqs, LOCUS["Cn"], LOCUS["P_ex"]))
        pl = eng.pair_loglik(LL, M); ref = eng.call_locus(*pl)
        ok = ok and np.array_equal(call["pos_use"], use) and np.array_equal(call["pair_ll"], pl[0]) and np.array_equal(call["order"], ref["order"])
        ok = ok and (call["first_cluster"], call["second_cluster"]) == (ref["first_cluster"], ref["second_cluster"]) and call["first_marginal"] == ref["first_marginal"]
        np.save(os.path.join(tmp, "typing_ok.npy"), np.array([int(ok), full["n_reads"], call["positions"]["n_reads"]]))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_typing_equals_unsharded_call(tmp_path, oracle):
    """north_star: "a single gather of per-read best-path records back to rank 0 for the final allele-pair call".  What the call consumes are
    the exon positions per locus (hla/HLATyper.cpp:1386-1428): ragged gather, then filters -> likelihoods -> pairs -> call on rank 0."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_typing_worker, args=(2, port, 301, str(tmp_path)), nprocs=2, join=True)
    ok = np.load(tmp_path / "typing_ok.npy")
    assert ok[0] == 1 and ok[1] == ok[2] and ok[1] > 20
