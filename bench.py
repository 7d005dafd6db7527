#!/usr/bin/env python3
"""bench.py -- paired reads/s aligned to a PRG on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (stages A+B+C = processBAM::alignOneReadPair) over one batch of
synthetic 2x150 bp pairs that is already resident in HBM.  N > 1: one process per GPU (torchrun), the
pairs shard embarrassingly (weak scaling: every rank aligns its own batch), and the only exchange is one
RCCL gather of the fixed-size per-pair records to rank 0, inside the timed region.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
from __future__ import annotations

import argparse
import importlib.util
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def load_package():
    if "hla_la_amd" in sys.modules:
        return sys.modules["hla_la_amd"]
    spec = importlib.util.spec_from_file_location("hla_la_amd", os.path.join(ROOT, "hla-la_amd", "__init__.py"),
                                                  submodule_search_locations=[os.path.join(ROOT, "hla-la_amd")])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["hla_la_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


def algorithmic_bytes_per_pair(read_len, chains_ext_per_pair, edges_per_chain, out_cols_per_mate):
    """B_pair = 2(Lr/2 + Lr) + sum_chains[32 + 5 Lr + 5 * edges touched] + 2*7*(Lr+G) + 64,
    with the measured number of extended chains per pair, CSR edge records touched per chain (= e(Lr+E)) and
    output columns per mate (= Lr + G).  Every byte counted once."""
    return (2 * (read_len / 2 + read_len)
            + chains_ext_per_pair * (32 + 5 * read_len + 5 * edges_per_chain)
            + 2 * 7 * out_cols_per_mate + 64)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--pairs", type=int, default=1_048_576, help="read pairs per GPU per step (BASELINE config 2: 1M)")
    ap.add_argument("--levels", type=int, default=5_000_000, help="levels of the synthetic MHC-scale stand-in graph")
    ap.add_argument("--cpu-pairs", type=int, default=4096, help="pairs of the same workload timed on the CPU oracle")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the measured path")
    # one process per GPU; HLALA_BENCH_BACKEND=gloo (+ ranks sharing a device) exists only to dry-run the N > 1 plumbing on a 1-GPU box
    backend = os.environ.get("HLALA_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)
    assert args.gpus == world, f"--gpus {args.gpus} but WORLD_SIZE={world}"

    P = load_package()
    from tools import synth

    t0 = time.time()
    # the real PRG_MHC_GRCh38_withIMGT is not available offline: synthetic stand-in, same on every rank
    w = synth.make_world(seed=2, G=args.levels, k=1, n_mut=3, n_largegap=1)
    b = synth.make_batch_fast(w, args.pairs, seed=1000 + rank)
    t_gen = time.time() - t0
    stream = torch.cuda.current_stream().cuda_stream
    ctx = P.Context(w["graph"], w["contigs"], insert_mean=b["insert_mean"], insert_sd=b["insert_sd"],
                    rng_seed=12345, max_columns=384, device=local_rank, stream=stream)
    gb = ctx.batch(b)            # inputs resident in HBM from here on
    rec = torch.empty((args.pairs, 8), dtype=torch.float64, device="cuda")
    gathered = [torch.empty_like(rec) for _ in range(world)] if (world > 1 and rank == 0) else None

    def step():
        gb.align()
        gb.export_pair_records(rec.data_ptr())
        if world > 1:
            if backend == "nccl":
                dist.gather(rec, gathered, dst=0)          # RCCL over xGMI: the one exchange of the path
            else:
                dist.gather(rec.cpu(), [g.cpu() for g in gathered] if gathered is not None else None, dst=0)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t1 = time.perf_counter()
    ev0.record()
    ext_ms = []
    for _ in range(args.steps):
        step()
        ext_ms.append(None)
    ev1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t1
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    st = gb.stats()          # HIP events of the LAST step on the ctx stream + device work counters
    if rank == 0:
        n_ok = int((rec[:, 0] == 0).sum().item())
        ms_per_step = elapsed / args.steps * 1e3
        value = args.pairs * world * args.steps / elapsed
        chains_pp = st.n_chains_extended / args.pairs
        edges_pc = st.n_edges_touched / max(1, st.n_chains_extended)
        cols_pm = st.n_out_columns / max(1, st.n_chains_extended)
        bpp = algorithmic_bytes_per_pair(150, chains_pp, edges_pc, cols_pm)
        ext_s = st.ms_dp_main * 1e-3      # the dominant kernel alone: k_dp<DpTiny, 0> (HIP events around it on the ctx stream)
        achieved = bpp * args.pairs / ext_s / 1e9
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "r01_traffic.json")
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("pairs") == args.pairs and tj.get("levels") == args.levels:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "paired reads/sec aligned to PRG graph", "value": value, "unit": "read pairs/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": f"{args.pairs} synthetic 2x150bp pairs per GPU (BASELINE config 2) vs a synthetic "
                                   f"{args.levels}-level PRG stand-in for PRG_MHC_GRCh38_withIMGT",
                       "pairs_per_gpu": args.pairs, "graph_levels": args.levels, "parallelism": f"shard{world}",
                       "chains_per_pair": b["n_chains"] / args.pairs, "extended_chains_per_pair": chains_pp,
                       "dp_calls_per_pair": st.n_dp_calls / args.pairs, "dp_iterations_per_call": st.n_dp_iterations / max(1, st.n_dp_calls),
                       "dp_cells_per_s": st.n_dp_cells / ((st.ms_dp_main + st.ms_extend_retry) * 1e-3), "pairs_ok": n_ok, "chain_errors": int(st.n_errors),
                       "stage_ms": {"project": st.ms_project, "extend": st.ms_extend, "extend_dp_16lane_kernel": st.ms_dp_main,
                                    "extend_dp_retry_classes": st.ms_extend_retry, "pair": st.ms_pair},
                       "dp_calls_sharing_a_dp": int(st.n_dp_shared), "dp_calls_retried_wider_class": int(st.n_chains_retried), "dp_calls_retried_large_class": int(st.n_dp_retried_large),
                       "generation_s": t_gen},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "kernel": "k_dp<DpTiny, 0>", "kernel_ms": st.ms_dp_main,
                         "algorithmic_bytes_per_pair": bpp},
        }
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import subprocess
            subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
            from oracle_binding import Oracle
            nb = min(args.cpu_pairs, args.pairs)
            sb = synth.make_batch_fast(w, nb, seed=1000)       # same generator, same seed: a prefix-like sample of rank 0's workload
            o = Oracle(w["graph"], w["contigs"], insert_mean=sb["insert_mean"], insert_sd=sb["insert_sd"], rng_seed=12345, max_columns=384)
            tc = time.perf_counter()
            o.align_batch(sb)
            dtc = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": nb / dtc, "unit": "read pairs/s", "cores": 1, "kind": "port",
                                   "sample": f"{nb} pairs of the same synthetic workload, oracle/hlala_oracle.cpp (C++ restatement, "
                                             f"-O2, single thread as in HLA-LA.cpp:799), {dtc:.1f} s"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
