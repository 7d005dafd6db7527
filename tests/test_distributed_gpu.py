"""Two ranks, PRODUCT engine on every rank -- over RCCL (backend nccl, one device per rank) when the node has two or more GPUs, over gloo with both
ranks sharing the one GPU of the box otherwise: pairs sharded in blocks, per-rank alignment with the
random seeds of the unsharded run, one gather of the per-pair records, one ragged gather of the exon positions of the locus, the call on
rank 0 -- equal to the call of the unsharded product run.  (On an 8-GPU node the same code runs over RCCL; tests/test_distributed_cpu.py
covers the plumbing with the oracle engine where there is no GPU.)"""
import ctypes as C
import os
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from test_distributed_cpu import LOCUS, _locus_arrays

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


class ProductEngine:
    def __init__(self, P, ctx):
        self.P, self.ctx, self.lib = P, ctx, C.CDLL(P.LIB_PATH)

    def filter_positions(self, e, prm):
        return self.P.filter_positions(self.lib, e, prm)

    def exon_in(self, e, use, seqs, Cn, Pex):
        return self.P.exon_in_from_positions(e, use, seqs, Cn, Pex)

    def exon_loglik(self, xin):
        return self.ctx.exon_loglik(xin)

    def pair_loglik(self, LL, M):
        return self.ctx.pair_loglik(LL, M)

    def call_locus(self, a, b, c):
        return self.ctx.call_locus(a, b, c)


def _worker(rank, world, port, n_pairs, tmp, backend):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    my_dev = rank if backend == "nccl" else 0
    dev = torch.device("cuda", my_dev) if backend == "nccl" else None          # where the process group works
    if backend == "nccl":
        torch.cuda.set_device(my_dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from conftest import load_package
    from tools import synth
    P = load_package()
    import importlib.util
    spec = importlib.util.spec_from_file_location("hla_la_amd.dist", os.path.join(ROOT, "hla-la_amd", "dist.py"))
    D = importlib.util.module_from_spec(spec); spec.loader.exec_module(D)
    w = synth.make_world_m(seed=7, n_levels=60_000, n_windows=3, alleles=(400, 3000))
    b = synth.make_batch_m(w, n_pairs, seed=12, frac_gene=0.6)
    # the locus: the first gene window, its hypervariable exon and the next one as "exon 2 / exon 3"
    M, ex = synth.window_matrix(w, 0)
    first = int(w["windows"]["first_level"][0]); cols = np.nonzero(ex > 0)[0]
    l2e = np.full(len(ex), -1, np.int32); l2e[cols] = np.arange(len(cols), dtype=np.int32)
    seqs = np.ascontiguousarray(M[:, cols][: LOCUS["Cn"] * 4: 4])                     # 40 allele rows as clusters
    Cn, Pex = seqs.shape
    gene = ([first], [first + len(ex) - 1])
    prm = P.default_filter_params(first20_n=6, first20_limit_per_read=0)

    def run(batch, rng_seed):
        ctx = P.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=rng_seed & 0xFFFFFFFF, device=my_dev)
        gb = ctx.batch(batch); gb.align()
        assert gb.stats().n_errors == 0
        ctx.set_gene_intervals(*gene)
        inc = gb.postprocess()
        e = gb.exon_positions(first, l2e, b["insert_mean"], b["insert_sd"], pair_mask=inc)
        pr = gb.pairs()
        rec = np.stack([pr["pair_status"], pr["best_chain"][0::2], pr["best_chain"][1::2], pr["n_combinations"], pr["pair_ll"], pr["pair_mapq"],
                        pr["mate_mapq"][0::2], pr["mate_mapq"][1::2]], axis=1).astype(np.float64)
        return ctx, e, rec, ctx.coverage()

    sub, p0, c0 = D.shard_pairs(b, rank, world)
    ctx, local, rec, cov = run(sub, 99 + 2 * c0)
    rec[:, 1:3] += c0
    to_pg = (lambda a: torch.from_numpy(a).to(dev)) if dev is not None else torch.from_numpy
    got = D.gather_records(to_pg(rec), dst=0)
    tot = D.reduce_coverage(to_pg(cov), dst=0)
    call = D.call_locus_sharded(ProductEngine(P, ctx), local, p0, seqs, Cn, Pex, prm, dst=0, device=dev)
    if rank == 0:
        ctxF, full, recF, covF = run(b, 99)
        eng = ProductEngine(P, ctxF)
        ok = np.array_equal(torch.cat(got).cpu().numpy(), recF) and np.array_equal(tot.cpu().numpy(), covF)
        for k in D._EXON_ARRAYS:
            ok = ok and np.array_equal(np.asarray(call["positions"][k]), np.asarray(full[k]), equal_nan=True)
        use, ign, st = eng.filter_positions(full, prm)
        LL, Mm = eng.exon_loglik(eng.exon_in(full, use, seqs, Cn, Pex))
        pl = eng.pair_loglik(LL, Mm); ref = eng.call_locus(*pl)
        ok = ok and np.array_equal(call["pos_use"], use) and np.array_equal(call["pair_ll"], pl[0]) and np.array_equal(call["order"], ref["order"])
        ok = ok and (call["first_cluster"], call["second_cluster"]) == (ref["first_cluster"], ref["second_cluster"])
        np.save(os.path.join(tmp, "ok.npy"), np.array([int(ok), full["n_reads"], call["positions"]["n_reads"], int(backend == "nccl")]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_reach_the_unsharded_call(tmp_path):
    import torch
    # (counting devices does not initialise the GPU in this process: the ranks are spawned first)
    backend = "nccl" if torch.cuda.device_count() >= 2 else "gloo"
    port = 32500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, 1201, str(tmp_path), backend), nprocs=2, join=True)
    ok = np.load(tmp_path / "ok.npy")
    assert ok[0] == 1 and ok[1] == ok[2] and ok[1] > 50 and ok[3] == int(backend == "nccl")
