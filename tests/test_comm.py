"""GPU: the exchange steps of the path for several contexts in ONE process (include/hlala_gpu.h: hlala_comm_*): the gather of the per-pair records and the sum of the
coverage counters -- what the reference does on the host when it merges its threads' results (mapper/processBAM.cpp:1866-1887, :1902-1913).  On a one-GPU box: a
communicator of one context (device copies), of one context with RCCL forced (ncclCommInitAll on one device, a grouped send / receive to itself, ncclReduce), and of
two contexts that share the device (copies); with two or more GPUs the last case runs over RCCL.  Every result equals what the single-context calls give."""
import numpy as np
import pytest

from tools import synth

pytestmark = pytest.mark.gpu


def _world_and_batches():
    w = synth.make_world(seed=21, G=6000, k=1)
    return w, [synth.make_batch(w, 300, seed=31), synth.make_batch(w, 200, seed=32)]


def _records_via_pairs(gb):
    p = gb.pairs_scalars()
    n = gb.n_pairs
    r = np.zeros((n, 8))
    r[:, 0] = p["pair_status"]; r[:, 1] = p["best_chain"][0::2]; r[:, 2] = p["best_chain"][1::2]; r[:, 3] = p["n_combinations"]
    r[:, 4] = p["pair_ll"]; r[:, 5] = p["pair_mapq"]; r[:, 6] = p["mate_mapq"][0::2]; r[:, 7] = p["mate_mapq"][1::2]
    return r


@pytest.mark.timeout(600)
@pytest.mark.parametrize("force_rccl", [None, "1"])
def test_communicator_of_one_context(pkg, monkeypatch, force_rccl):
    if force_rccl is not None:
        monkeypatch.setenv("HLALA_COMM_RCCL", force_rccl)
    w, bs = _world_and_batches()
    b = bs[0]
    ctx = pkg.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=5)
    gb = ctx.batch(b); gb.align(); gb.postprocess()
    comm = pkg.Comm([ctx])
    assert comm.uses_rccl == (force_rccl == "1")
    rec, counts = comm.gather_pair_records([gb])
    assert counts.tolist() == [b["n_pairs"]] and np.array_equal(rec, _records_via_pairs(gb))
    rec2, counts2 = comm.gather_pair_records([None])
    assert counts2.tolist() == [0] and rec2.shape == (0, 8)
    cov = ctx.coverage()
    assert cov.sum() > 0 and np.array_equal(comm.reduce_coverage(), cov)
    assert np.array_equal(comm.reduce_coverage(reset=True), cov) and comm.reduce_coverage().sum() == 0
    comm.close()


@pytest.mark.timeout(600)
def test_two_contexts_gather_in_context_order_and_sum_their_coverage(pkg):
    import torch
    w, bs = _world_and_batches()
    two = torch.cuda.device_count() >= 2          # two GPUs: RCCL between them; one: the contexts share it and the communicator copies
    kw = dict(insert_mean=bs[0]["insert_mean"], insert_sd=bs[0]["insert_sd"], rng_seed=5)
    ctxs = [pkg.Context(w["graph"], w["contigs"], device=(i if two else 0), **kw) for i in range(2)]
    gbs = [ctxs[i].batch(bs[i]) for i in range(2)]
    for g in gbs:
        g.align(); g.postprocess()
    comm = pkg.Comm(ctxs)
    assert comm.uses_rccl == two
    rec, counts = comm.gather_pair_records(gbs)
    assert counts.tolist() == [bs[0]["n_pairs"], bs[1]["n_pairs"]]
    assert np.array_equal(rec, np.concatenate([_records_via_pairs(g) for g in gbs]))
    rec1, counts1 = comm.gather_pair_records([None, gbs[1]])
    assert counts1.tolist() == [0, bs[1]["n_pairs"]] and np.array_equal(rec1, _records_via_pairs(gbs[1]))
    cov = ctxs[0].coverage() + ctxs[1].coverage()
    assert np.array_equal(comm.reduce_coverage(), cov)
    with pytest.raises(RuntimeError):
        comm.gather_pair_records([gbs[1], gbs[0]])          # a batch handed in for the wrong context
    comm.close()
