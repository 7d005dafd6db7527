#!/usr/bin/env python3
"""bench.py -- paired reads/s aligned to a PRG on N MI355X (BASELINE.json metric).

A "step" is one pass of the hot path (stages A+B+C = processBAM::alignOneReadPair) over one batch of synthetic 2x150 bp pairs that
is already resident in HBM.  N > 1: one process per GPU, the pairs shard embarrassingly (weak scaling: every rank aligns its own
batch), and the only exchange is one RCCL gather of the fixed-size per-pair records to rank 0, inside the timed region.
`python bench.py --gpus N` without a launcher starts its N ranks itself (torch.distributed.run as a child process, before anything
touches the GPU); under torchrun (WORLD_SIZE set) it is one of the ranks.

Workload: "Graph M" -- 5 M levels, 8 backbone haplotypes at 0.3 % divergence with 2 % gap stretches, 40 gene windows
of 3-6 kb carrying 500-5000 allele paths merged by the suffix-10 rule -- and 2x150 bp pairs with qualities / errors from the
reference's empirical matrix, Poisson indels, start-to-start jump N(350, 35), >= 30 % of the pairs from allele rows of the gene
windows, bwa-like seeds (soft clips, secondary alignments on other contigs).  `--graph simple` selects the round-1 stand-in.

Prints ONE JSON line on rank 0 with `roofline` and `cpu_baseline`.  `value` is the resident rate (inputs in HBM when the timed region starts: the
contract of this bench).  Rank 0 at N = 1 additionally measures, each with its own timed loop:
  host_inclusive  the boundary of the C ABI -- hlala_batch_create (H2D) -> hlala_align_batch -> hlala_batch_get_pairs_packed (D2H of every column of
                  the selected alignments) per step, page-locked caller buffers (hlala_pinned_alloc), two batches in flight on ONE context: uploads run
                  on the context's upload stream, downloads on its reader stream, both beside the alignment of the other batch;
  end_to_end      BAM bytes -> hla/*: the `HLA-LA --action HLA` host program on a Graph M graph directory and a multi-million-pair BAM (decode on all
                  host threads, batches through the GPU two in flight, typing of six loci, result files), its own End-to-end line;
and, as reports: gene-window and backbone pairs separately, the typer kernels at C = 3000, the CPU oracle on a bounded sample.
"""
from __future__ import annotations

import argparse
import hashlib
import importlib.util
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
TAIL_POOL_DEFAULT = 1          # hlala_set_tail_pool of the bench's loops (--tail-pool)
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


def kernel_source_hash():
    """sha1 over the kernel / API sources (line comments and blank lines left out: a reworded comment is not a new kernel): measurements kept
    under profiles/ are tagged with it and dropped when it differs."""
    import re
    h = hashlib.sha1()
    d = os.path.join(ROOT, "hla-la_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h", ".hpp")):
            for line in open(os.path.join(d, f), "r", errors="replace"):
                code = re.sub(r"//.*$", "", line).strip()
                if code:
                    h.update(code.encode() + b"\n")
    return h.hexdigest()[:16]


def algorithmic_bytes_per_pair(read_len, chains_ext_per_pair, mean_out_degree, cols_per_chain, out_cols_per_mate):
    """B_pair = 2(Lr/2 + Lr) + sum_chains[32 + 5 Lr + 5 e (Lr + E)] + 2*7*(Lr + G) + 64, every byte counted ONCE:
    a chain touches the CSR edge records (label 1 B + target 4 B) of the levels it spans -- (Lr + E) = its columns -- times the
    graph's mean out-degree e, however often the DP re-reads them."""
    return (2 * (read_len / 2 + read_len)
            + chains_ext_per_pair * (32 + 5 * read_len + 5 * mean_out_degree * cols_per_chain)
            + 2 * 7 * out_cols_per_mate + 64)


def spawn_ranks(n):
    """`python bench.py --gpus N` outside a launcher: start N ranks as a child process tree (nothing has touched the GPU yet)."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def make_workload(args, synth, rank):
    if args.graph == "m":
        w = synth.make_world_m(seed=2, n_levels=args.levels)
        mk = lambda n, seed, **kw: synth.make_batch_m(w, n, seed=seed, **kw)       # noqa: E731
    else:
        w = synth.make_world(seed=2, G=args.levels, k=1, n_mut=3, n_largegap=1)
        mk = lambda n, seed, **kw: synth.make_batch_fast(w, n, seed=seed)          # noqa: E731
    return w, mk


def write_batch(d, r, j, bb):
    np.save(os.path.join(d, "b%d_%d_scalars.npy" % (r, j)), np.asarray([bb["n_pairs"], bb["n_chains"]], np.int64))
    np.save(os.path.join(d, "b%d_%d_insert.npy" % (r, j)), np.asarray([bb["insert_mean"], bb["insert_sd"]], np.float64))
    for k in BATCH_KEYS:
        np.save(os.path.join(d, "b%d_%d_%s.npy" % (r, j, k)), bb[k])


def gen_worker(argv):
    """`bench.py --gen-worker <dir> <ranks> --pairs P --levels L --graph g`: the two batches of each listed rank, written where shared_workload's readers expect them
    (no torch, no GPU: a child of rank 0)."""
    d = argv[0]; ranks = [int(x) for x in argv[1].split(",") if x]
    ap = argparse.ArgumentParser(); ap.add_argument("--pairs", type=int); ap.add_argument("--levels", type=int); ap.add_argument("--graph")
    a = ap.parse_args(argv[2:])
    sys.path.insert(0, ROOT)
    from tools import synth
    w, mk = make_workload(a, synth, 0)
    for r in ranks:
        for j, seed in enumerate((1000 + r, 5000 + r)):
            write_batch(d, r, j, mk(a.pairs, seed))


def workload_desc(args, w):
    if args.graph == "m":
        return (f"{args.pairs} synthetic 2x150bp pairs per GPU (BASELINE config 2) vs Graph M, the SURVEY 8(d) stand-in for PRG_MHC_GRCh38_withIMGT: "
                f"{args.levels} levels, 8 backbone haplotypes (0.3 % divergence, 2 % gap stretches), 40 gene windows of 3-6 kb with 500-5000 allele paths "
                f"(suffix-10 node merging, up to {int(w['max_nodes_per_level'])} nodes per level); reads: I101_NA12878 quality matrix stretched to 150, Poisson indels, "
                f"start-to-start jump N(350,35) = inner distance N(200,35), 30 % of the pairs drawn from allele rows of the gene windows")
    return (f"{args.pairs} synthetic 2x150bp pairs per GPU vs the round-1 stand-in: {args.levels} levels, 5 haplotypes "
            f"(simpleGraphSimulator recipe), geometric qualities, no read indels, inner distance N(200,35)")


BATCH_KEYS = ("read_off", "read_bases", "read_quals", "chain_off", "read_primary", "chain_contig", "chain_pos", "chain_offset", "chain_as", "chain_reverse", "cigar_off", "cigar")


def shared_workload(args, synth, rank, world, dist):
    """N > 1: rank 0 generates the world ONCE, and every rank's two batches with it (the generator keeps the allele matrices of the gene windows: eight
    copies of it side by side are eight 10 s generations and eight working sets); the other ranks read the arrays from a directory in shared memory."""
    import shutil
    import tempfile
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    d = os.path.join(base, "hlala_bench_%s" % os.environ.get("MASTER_PORT", "0"))
    w = mk = None
    if rank == 0:
        shutil.rmtree(d, ignore_errors=True); os.makedirs(d)
        w, mk = make_workload(args, synth, rank)
        for k, v in w["graph"].items():
            np.save(os.path.join(d, "graph_%s.npy" % k), np.asarray(v))
        for k, v in w["contigs"].items():
            np.save(os.path.join(d, "contigs_%s.npy" % k), np.asarray(v))
        np.save(os.path.join(d, "max_nodes_per_level.npy"), np.asarray(int(w.get("max_nodes_per_level", 0))))
        # the other ranks' batches: generated side by side by child processes that never touch the GPU (round 6: sixteen generations one after the other were
        # 3.5 minutes at N = 8 before the first timed step); every worker builds the same world again (deterministic, ~10 s) and writes its ranks' batches
        nw = min(world - 1, max(1, (os.cpu_count() or 8) // 2), 7)
        shares = [[r for r in range(1, world) if (r - 1) % nw == k] for k in range(nw)]
        workers = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--gen-worker", d, ",".join(str(r) for r in sh), "--pairs", str(args.pairs), "--levels", str(args.levels), "--graph", args.graph])
                   for sh in shares if sh]
        bs = [mk(args.pairs, 1000), mk(args.pairs, 5000)]
        for wk in workers:
            if wk.wait() != 0:
                raise SystemExit("bench.py: a workload generator process failed")
    dist.barrier()
    if rank != 0:
        ld = lambda n: np.load(os.path.join(d, n + ".npy"))          # noqa: E731
        def unbox(a): return a.item() if a.ndim == 0 else a          # noqa: E704
        g = {k[6:-4]: unbox(ld(k[:-4])) for k in sorted(os.listdir(d)) if k.startswith("graph_")}
        c = {k[8:-4]: unbox(ld(k[:-4])) for k in sorted(os.listdir(d)) if k.startswith("contigs_")}
        w = {"graph": g, "contigs": c, "max_nodes_per_level": int(ld("max_nodes_per_level"))}
        bs = []
        for j in range(2):
            sc = ld("b%d_%d_scalars" % (rank, j)); ins = ld("b%d_%d_insert" % (rank, j))
            bb = {"n_pairs": int(sc[0]), "n_chains": int(sc[1]), "insert_mean": float(ins[0]), "insert_sd": float(ins[1])}
            for k in BATCH_KEYS:
                bb[k] = ld("b%d_%d_%s" % (rank, j, k))
            bs.append(bb)
    dist.barrier()
    if rank == 0:
        shutil.rmtree(d, ignore_errors=True)
    return w, mk, bs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=6)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--pairs", type=int, default=1_048_576, help="read pairs per GPU per step (BASELINE config 2: 1M)")
    ap.add_argument("--levels", type=int, default=5_000_000, help="levels of the synthetic MHC-scale stand-in graph")
    ap.add_argument("--graph", choices=["m", "simple"], default="m")
    ap.add_argument("--cpu-pairs", type=int, default=16384, help="pairs of the same workload timed on the CPU oracle (1 thread; 4x as many on all cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-overlap", action="store_true", help="resident loop: two batches alternate, but each step's export follows its alignment at once (a batch's tail does not run beside the next batch)")
    ap.add_argument("--single-batch", action="store_true", help="one batch at a time in both loops: no overlap of a batch's tail / transfers with the next batch")
    ap.add_argument("--no-extras", action="store_true", help="skip the measurements outside the timed region (gene / backbone split, long reads, typer, end to end)")
    ap.add_argument("--ascii-bases", action="store_true", help="boundary loop: hand the read bases over as ASCII (1 B per base) instead of 4-bit packed as the BAM decoder has them")
    ap.add_argument("--resident-only", action="store_true", help="kernel A/B mode: only the resident loop (inputs already in HBM) is run and timed; `value` is then the RESIDENT rate and the line says so")
    ap.add_argument("--resident-steps", type=int, default=8, help="steps of the resident loop that follows the timed boundary loop (config.resident; 0 = skip)")
    ap.add_argument("--e2e-pairs", type=int, default=8_388_608, help="pairs of the sample pushed through `HLA-LA --action HLA` for the end-to-end rate (0 = skip)")
    ap.add_argument("--e2e-frac-gene", type=float, default=0.04, help="share of the end-to-end sample drawn from the gene windows: 0.04 = the windows' share of the graph, i.e. the uniform "
                    "coverage of a whole-genome sample (the resident workload keeps 0.3: its gene-window pairs are the expensive ones; at 0.3 every typed locus would see 2000x coverage)")
    ap.add_argument("--e2e-threads", default="0,128", help="--decodeThreads values of the end-to-end runs (0 = the decoder's default: twice the CPUs the process may use, at most 32 threads)")
    ap.add_argument("--e2e-samples", default="2,4", help="end to end: further runs with this many samples (the same BAM; 'n:threads' = with --decodeThreads threads) in ONE call on one device, the decode of sample k + 1 beside the alignment of sample k ('' = none)")
    ap.add_argument("--no-extras-but-e2e", action="store_true", help="of the measurements outside the timed region only the end-to-end run")
    ap.add_argument("--e2e-variants", default="", help="experiments: further end-to-end runs of the same sample under other environments, 'label:ENV=1 ENV2=x;label2:...'")
    ap.add_argument("--long-reads", type=int, default=50_000, help="reads of the long-read record (BASELINE config 5: 50 000 reads of ~10 kb; 0 = skip)")
    ap.add_argument("--in-flight", type=int, default=3, choices=(2, 3), help="batches whose outputs are live at one time in the boundary loop: 2 = the alignment of batch i+2 is queued after batch i has been read back and destroyed; 3 (default) = as soon as batch i is COMPLETE, before its read-back (two alignments queued on the GPU either way, plus one upload ahead)")
    ap.add_argument("--tail-pool", type=int, default=TAIL_POOL_DEFAULT, help="hlala_set_tail_pool(k): the broad / large / in-memory DP classes of k consecutive alignments run in one launch per class "
                                                                             "(1 = every alignment runs its own tail, as in rounds 2-5); the loops keep k alignments queued ahead and k + 1 batches' outputs live")
    ap.add_argument("--resident-lag", type=int, default=0, help="resident loop with a tail pool: the export of step i follows the alignment of step i + lag (default: tail pool + 1); lag + 1 resident batches")
    ap.add_argument("--long-reads-batch", type=int, default=50000, help="reads per batch of the long-read record (16 384-column rows: 17 GB of column arrays for 50 000 reads; five batches of 10 000 take 2.5 times as long -- every batch ends on its slowest wavefronts)")
    ap.add_argument("--long-reads-check", type=int, default=2048, help="reads of the long-read record compared with the CPU oracle after the clock has stopped (0 = none)")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the measured path")
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # one process per GPU; HLALA_BENCH_BACKEND=gloo (+ ranks sharing a device) exists only to dry-run the N > 1 plumbing on a 1-GPU box
    backend = os.environ.get("HLALA_BENCH_BACKEND", "nccl")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=backend)

    P = load_package()
    from tools import synth

    t0 = time.time()
    if world > 1:
        w, mk, bsrc = shared_workload(args, synth, rank, world, dist)
    else:
        w, mk = make_workload(args, synth, rank)
        bsrc = [mk(args.pairs, 1000), mk(args.pairs, 5000)]
    if args.single_batch:
        bsrc = bsrc[:1]
    b = bsrc[0]
    desc = workload_desc(args, w)
    t_gen = time.time() - t0
    stream = torch.cuda.current_stream().cuda_stream
    ckw = dict(insert_mean=b["insert_mean"], insert_sd=b["insert_sd"], rng_seed=12345, max_columns=384, device=local_rank)
    ctx = P.Context(w["graph"], w["contigs"], stream=stream, **ckw)
    if args.tail_pool > 1:
        ctx.set_tail_pool(args.tail_pool)
    recs = [torch.empty((args.pairs, 8), dtype=torch.float64, device="cuda") for _ in range(2)]
    gathered = [torch.empty_like(recs[0]) for _ in range(world)] if (world > 1 and rank == 0) else None
    n_gathers = [0]

    # the gather runs on a stream of its own: the library's export returns with the records in place, and the collective must not queue behind the
    # alignment of the NEXT batch, which the main stream already holds
    gstream = torch.cuda.Stream() if world > 1 and backend == "nccl" else None

    def before_export():
        if gstream is not None:
            gstream.synchronize()          # the previous gathers have read their record buffers

    def gather(k):
        # the ONE exchange of the path: the fixed-size per-pair records to rank 0 (RCCL over xGMI)
        if world > 1:
            if backend == "nccl":
                with torch.cuda.stream(gstream):
                    dist.gather(recs[k], gathered, dst=0)
            else:
                dist.gather(recs[k].cpu(), [g.cpu() for g in gathered] if gathered is not None else None, dst=0)
            n_gathers[0] += 1

    def timed(fn, n):
        """barrier + synchronize on both sides of n steps, MAX over ranks"""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fn(n)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        el = mine = time.perf_counter() - t1
        per_rank = [mine]
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            allt = [torch.zeros_like(tt) for _ in range(world)]
            dist.all_gather(allt, tt)
            per_rank = [float(x.item()) for x in allt]
            el = max(per_rank)
        return el, per_rank

    # ---- the timed region: the boundary of the C ABI.  Per step: hlala_batch_create (H2D from page-locked caller buffers) -> hlala_align_batch ->
    # export of the per-pair records (+ the gather to rank 0 for N > 1) -> hlala_batch_get_pairs + hlala_batch_get_pairs_packed (D2H of every column
    # of the selected alignments) -> hlala_batch_destroy; two batches in flight on ONE context and one host thread: the next batch is uploaded and
    # queued before the current one is read back (uploads on the context's upload stream, downloads on its reader stream).
    boundary = None
    if not args.resident_only:
        bnd = Boundary(args, P, ctx, bsrc)
        try:
            def run_boundary(n):
                if n <= 0:
                    return
                if args.single_batch:
                    for i in range(n):
                        h = bnd.start(i); before_export()
                        bnd.finish(h, recs[0], lambda: gather(0))
                    return
                # Two alignments in flight and one upload ahead: while the GPU aligns batches i and i+1 the host thread uploads the INPUTS of batch i+2 (0.9 GB;
                # hlala_batch_create allocates no outputs), then reads batch i back and destroys it, then queues the alignment of batch i+2, whose output arrays
                # are the pool blocks batch i just gave back.  (With the upload AFTER the read-back -- round 3's order -- the one host thread was the critical path:
                # create 63 ms + wait for the batch 127 + read-back 67 = the 258 ms of a step, 30 ms of them with the GPU's main stream idle.)
                # --in-flight 3 (round 5): a batch is complete when its tail classes have run on the side stream, beside the main-stream kernels of the NEXT batch;
                # with main-stream steps of ~205 ms the chain [tail of batch i: ~135 ms] -> [read-back of batch i: 62 ms] -> [launch of batch i+2] ends where the main
                # stream runs dry.  With 3, the alignment of batch i+2 is queued as soon as batch i is COMPLETE, before it is read back: three batches' outputs are
                # live at that moment (a third set of pool blocks, 50 GB), two alignments are queued on the GPU as before.  (Queueing a third alignment AHEAD instead
                # made hlala_align_batch block for a whole step: profiles/r05_experiments.txt.)
                # --tail-pool k (round 6): k alignments are queued ahead instead of two -- batch i is complete once the pooled tail of ITS group of k has run, and the
                # library launches that tail with the k-th alignment of the group; k + 1 batches' outputs are live.
                na = max(2, args.tail_pool)
                ahead = [bnd.start(k) for k in range(min(na, n))]
                for i in range(n):
                    up = bnd.upload(i + na) if i + na < n else None
                    before_export()
                    h = ahead.pop(0)
                    bnd.wait_export(h, recs[i % 2], lambda k=i % 2: gather(k))
                    if args.in_flight == 3 and up is not None:
                        ahead.append(bnd.launch(up)); up = None
                    bnd.readback(h)
                    if up is not None:
                        ahead.append(bnd.launch(up))
            run_boundary(max(args.warmup, (max(2, args.tail_pool) + 1) if args.in_flight == 3 else 1))          # (at least one: pool blocks, first touch of the page-locked buffers; three with three sets of outputs live, so that the third set exists before the clock starts)
            g0 = n_gathers[0]
            bnd.host_s = {}; bnd.host_n = {}
            elapsed, per_rank = timed(run_boundary, args.steps)
            boundary = {"host_thread_ms_per_call": bnd.host_ms(), "elapsed": elapsed, "per_rank_s": per_rank, "gathers": n_gathers[0] - g0, "pairs_ok": bnd.pairs_ok(), "columns": bnd.last_cols,
                        "bytes_up": bnd.bytes_up, "bytes_down": bnd.bytes_down()}
            # the same with pageable caller buffers (what a caller that does not use hlala_pinned_alloc gets): a report
            if world == 1 and not args.no_extras:
                bnd.pageable()
                run_boundary(1)
                el_p, _ = timed(run_boundary, 3)
                boundary["pageable"] = {"value": args.pairs * 3 / el_p, "ms_per_step": el_p / 3 * 1e3, "steps": 3}
        finally:
            bnd.free()

    # ---- the resident loop: inputs in HBM, two batch objects of the workload alternate, a step = hlala_align_batch + the export of the per-pair records
    # (+ the gather); the export of step i is issued after the alignment of step i+1, so the side-stream tail of one batch runs beside the bulk of the next.
    # --tail-pool k > 1: k + 2 resident batch objects of the two source batches take turns, the export of step i follows the alignment of step i + k (the tail of a
    # group of k runs with the group's last alignment; one more alignment is queued before the host waits for it)
    lag = 1 if args.tail_pool <= 1 else (args.resident_lag if args.resident_lag > 0 else args.tail_pool + 1)
    gbs = [ctx.batch(x) for x in bsrc] if lag == 1 else [ctx.batch(bsrc[i % len(bsrc)]) for i in range(lag + 1)]
    gb = gbs[0]

    def finish(k):
        before_export()
        gbs[k].export_pair_records(recs[k % len(recs)].data_ptr())
        gather(k % len(recs))

    def run(n):
        for i in range(n):
            k = i % len(gbs)
            gbs[k].align()
            if len(gbs) == 1 or args.no_overlap:
                finish(k)
            elif i >= lag:
                finish((i - lag) % len(gbs))
        if len(gbs) > 1 and n > 0 and not args.no_overlap:
            for j in range(max(0, n - lag), n):
                finish(j % len(gbs))

    def device_memory(batch):
        """bytes of one batch's device blocks and of the whole context (hlala_debug_memory), chains with column rows"""
        import ctypes as C_
        out = (C_.c_ulonglong * 4)()
        try:
            ctx.lib.hlala_debug_memory.argtypes = [C_.c_void_p, C_.c_void_p, C_.POINTER(C_.c_ulonglong)]
            ctx.lib.hlala_debug_memory(ctx.h, batch.b, out)
        except AttributeError:
            return None
        return {"batch_bytes": int(out[0]), "chains": int(bsrc[0]["n_chains"]), "chains_with_column_rows": int(out[1]), "context_bytes_with_two_resident_batches": int(out[3]),
                "what": "device blocks of one aligned 1 M-pair batch (inputs + outputs); column rows exist for the chains that passed the filters (batch.h: chain_row)"}

    rsteps = args.steps if args.resident_only else args.resident_steps
    resident = None
    if rsteps > 0:
        run(max(args.warmup if args.resident_only else 2, len(gbs) if lag > 1 else 0))
        g0 = n_gathers[0]
        el_r, per_rank_r = timed(run, rsteps)
        resident = {"value": args.pairs * world * rsteps / el_r, "unit": "read pairs/s", "steps": rsteps, "ms_per_step": el_r / rsteps * 1e3, "per_rank_s": per_rank_r,
                    "what": "inputs resident in HBM: hlala_align_batch + device-to-device export of the per-pair records (+ the gather for N > 1) per step, "
                            + ("one batch at a time" if len(gbs) == 1 else ("two resident batches alternate" if lag == 1 else f"tail pool of {lag}: {len(gbs)} resident batches take turns, the export of a step follows the alignment {lag} steps later"))}
        if world > 1:
            assert n_gathers[0] - g0 == rsteps, "every step of the resident loop gathers once"
    else:
        gb.align()
    st = gb.stats()          # HIP events of THIS batch's last alignment (the events belong to the batch) + device work counters
    torch.cuda.synchronize()
    rec = recs[((rsteps - 1) % len(gbs)) % len(recs)] if rsteps > 0 else recs[0]          # the records of the last step (the ones the last gather carried)
    n_ok_local = int((rec[:, 0] == 0).sum().item())
    if world > 1:
        # every rank's share arrived: the gathered records of the last step hold `pairs` valid rows per rank
        oks = [None] * world
        dist.all_gather_object(oks, n_ok_local)
        if rank == 0:
            assert dist.get_world_size() == world and len(gathered) == world
            if backend == "nccl":
                got = [int((g[:, 0] == 0).sum().item()) for g in gathered]
                assert got == [int(x) for x in oks], f"gathered records differ from the ranks' own counts: {got} vs {oks}"
    else:
        oks = [n_ok_local]

    if rank == 0:
        if boundary is not None:
            elapsed = boundary["elapsed"]; steps = args.steps
        else:
            elapsed = resident["ms_per_step"] * 1e-3 * resident["steps"]; steps = resident["steps"]
        ms_per_step = elapsed / steps * 1e3
        value = args.pairs * world * steps / elapsed
        g = w["graph"]
        chains_pp = st.n_chains_extended / args.pairs
        cols_pc = st.n_out_columns / max(1, st.n_chains_extended)
        e_mean = g["n_edges"] / max(1, g["n_nodes"] - 1)
        bpp = algorithmic_bytes_per_pair(150, chains_pp, e_mean, cols_pc, cols_pc)
        cls_ms = [float(x) for x in st.ms_dp_class]; cls_n = [int(x) for x in st.n_dp_class]
        # The first DP class is five kernels launched back to back (round 5): the band kernels (k_dp_band<16 / 32 / 64>: calls on linear stretches of the graph), then the
        # jump-free and the general instantiation of the 16-lane template, whose events span both.  Every kernel is named on its own; the single dominant kernel among
        # the classes of the main stream is what `roofline` prices (the times of the side-stream classes -- a few hundred long DP calls at low priority beside the next
        # batch, hlala_align_batch -- are waiting times, not work), and `roofline.first_class` holds the sum of the five.
        ms_band = float(st.ms_dp_band); ms_jf = float(st.ms_dp_jump_free); ms_gen = max(0.0, cls_ms[0] - ms_jf)
        kern = [("k_dp_band<16> + <32> + <64>", ms_band), ("k_dp<DpTinyJF, 0>", ms_jf), ("k_dp<DpTiny, 0>", ms_gen), ("k_dp<DpMid, 1>", cls_ms[1]), ("k_dp<DpSmall, 2>", cls_ms[2]), ("k_dp<DpWide, 3>", cls_ms[3])]
        names = [k for k, _ in kern]
        dom = int(np.argmax([m for _, m in kern]))
        dom_ms = kern[dom][1]
        achieved = bpp * args.pairs / (dom_ms * 1e-3) / 1e9
        first_class_ms = ms_band + cls_ms[0]
        khash = kernel_source_hash()
        traffic, traffic_note, secondary = None, "no PMC pass on file for this build", {}
        tfile = next((f for f in (os.path.join(ROOT, "profiles", t + "_traffic.json") for t in ("r06", "r05", "r04", "r03", "r02")) if os.path.exists(f)), "")
        tname = os.path.relpath(tfile, ROOT) if tfile else ""
        if tfile:
            try:
                tj = json.load(open(tfile))
                same = (tj.get("pairs") == args.pairs and tj.get("levels") == args.levels and tj.get("graph") == args.graph and tj.get("kernel") == names[dom])
                if same and tj.get("kernel_source_hash") == khash:
                    traffic = tj.get("hbm_bytes_per_launch"); traffic_note = f"{tname} (kernel sources {khash})"
                elif same:
                    traffic_note = f"{tname} was measured on kernel sources {tj.get('kernel_source_hash')}, this build is {khash}: dropped"
                if same:
                    secondary = dict(tj.get("secondary", {})); secondary["measured_on_kernel_source_hash"] = tj.get("kernel_source_hash")
            except Exception:
                pass
        secondary["dp_cells_per_s"] = st.n_dp_cells / ((sum(cls_ms) + ms_band) * 1e-3)
        if secondary.get("valu_insts_all_dp_classes_per_launch"):
            secondary["valu_wave_insts_per_dp_cell"] = secondary["valu_insts_all_dp_classes_per_launch"] / max(1, st.n_dp_cells)
        headline = ("host-inclusive: hlala_batch_create (H2D, page-locked caller buffers) + hlala_align_batch + export of the per-pair records (+ gather) + hlala_batch_get_pairs + "
                    "hlala_batch_get_pairs_packed (D2H) + hlala_batch_destroy per step, " + ("one batch at a time" if args.single_batch else f"{args.in_flight} alignments in flight and one upload ahead on one context, one host thread")) \
            if boundary is not None else "RESIDENT rate (--resident-only: kernel A/B mode, the boundary loop was not run)"
        out = {
            "metric": "paired reads/sec aligned to PRG graph", "value": value, "unit": "read pairs/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": desc, "timed_region": headline, "graph": args.graph, "pairs_per_gpu": args.pairs, "graph_levels": args.levels, "graph_nodes": int(g["n_nodes"]), "graph_edges": int(g["n_edges"]),
                       "parallelism": f"shard{world}", "tail_pool": args.tail_pool, "batches_in_flight": 1 if args.single_batch else (args.in_flight if boundary is not None else 2), "chains_per_pair": b["n_chains"] / args.pairs, "extended_chains_per_pair": chains_pp,
                       "columns_per_chain": cols_pc, "mean_out_degree": e_mean,
                       "dp_calls_per_pair": st.n_dp_calls / args.pairs, "dp_iterations_per_call": st.n_dp_iterations / max(1, st.n_dp_calls),
                       "pairs_ok": int(oks[0]), "pairs_ok_per_rank": [int(x) for x in oks], "chain_errors": int(st.n_errors),
                       "resident": resident, "device_memory": device_memory(gbs[0]),
                       "stage_ms": {"project": st.ms_project, "extend": st.ms_extend, "pair": st.ms_pair, "side_stream": st.ms_side,
                                    "dp_16lane": cls_ms[0], "dp_32lane": cls_ms[1], "dp_64lane": cls_ms[2], "dp_wide": cls_ms[3], "dp_broad": cls_ms[4], "dp_large": cls_ms[5], "dp_in_memory": cls_ms[6],
                                    "dp_16lane_jump_free_part": ms_jf, "dp_16lane_general_part": ms_gen, "dp_band": ms_band},
                       "stage_ms_source": "HIP events of one batch of the resident loop, on the streams its kernels ran on (the kernels of the boundary loop are the same)",
                       "dp_calls_entering_class": {"16lane": cls_n[0], "32lane": cls_n[1], "64lane": cls_n[2], "wide": cls_n[3], "broad": cls_n[4], "large": cls_n[5], "in_memory": cls_n[6], "16lane_jump_free": int(st.n_dp_jump_free),
                                                   "band": int(st.n_dp_band), "band_failed_over_to_16lane": int(st.n_dp_band_failed), "jump_free_met_a_jump_and_went_to_the_general_list": int(st.n_dp_jump_free_failed)},
                       "dp_calls_sharing_a_dp": int(st.n_dp_shared), "generation_s": t_gen, "kernel_source_hash": khash},
            "roofline": {"bound": "latency", "bound_note": "dependent LDS / L2 round trips inside one iteration of an integer frontier DP, which four wavefronts per SIMD do not cover: the kernel's SIMDs issue vector instructions 28 % of the time (profiles/r06_experiments.txt 11), HBM is 1-2 % busy -- achieved / peak / frac are the HBM figures the metric asks for, not the binding resource", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0,
                         "traffic": traffic, "traffic_source": traffic_note, "kernel": names[dom], "kernel_ms": dom_ms,
                         "first_class": {"kernels": "k_dp_band<16> + <32> + <64> + k_dp<DpTinyJF, 0> + k_dp<DpTiny, 0>", "ms": first_class_ms, "achieved": bpp * args.pairs / (first_class_ms * 1e-3) / 1e9,
                                         "by_kernel_ms": {k: m for k, m in kern[:3]}},
                         "algorithmic_bytes_per_pair": bpp, "whole_step_achieved": bpp * args.pairs / (ms_per_step * 1e-3) / 1e9,
                         "note": "dominant kernel only (HIP events on the ctx stream); the path is bound by instruction issue and dependent LDS / global "
                                 "round trips, not by HBM (SURVEY 8(d)): see `secondary`",
                         "secondary": secondary},
        }
        if boundary is not None:
            out["host_inclusive"] = {"value": value, "unit": "read pairs/s", "steps": steps, "ms_per_step": ms_per_step, "per_rank_s": boundary["per_rank_s"], "gathers_in_timed_region": boundary["gathers"],
                                     "columns_returned_per_step": boundary["columns"], "bytes_down_per_step": boundary["bytes_down"], "bytes_up_per_step": boundary["bytes_up"],
                                     "pairs_ok_last_step": boundary["pairs_ok"], "host_thread_ms_per_call": boundary["host_thread_ms_per_call"], "pageable": boundary.get("pageable"), "what": "this IS the headline: `value` / `ms_per_step` of this line"}
            if world > 1:
                assert boundary["gathers"] == args.steps, "every step of the timed region gathers once"
        if world == 1 and not args.no_extras and not args.resident_only and not args.no_extras_but_e2e:
            try:
                out["config"].update(extras(args, P, synth, w, mk, b, ctx, gb, ckw))
            except Exception as e:          # the extras are reports, never a reason to lose the bench line
                out["config"]["extras_error"] = repr(e)
        for x in gbs:
            x.close()
        ctx.close()
        if world == 1 and not args.no_extras and not args.resident_only:
            if args.long_reads > 0:
                try:
                    out["long_reads"] = long_reads(args, P, synth, w)
                except AssertionError:          # the timed reads differ from the oracle: a throughput of wrong results is not reported
                    raise
                except Exception as e:
                    out["long_reads"] = {"error": repr(e)}
            if args.e2e_pairs > 0 and args.graph == "m":
                try:
                    out["end_to_end"] = end_to_end(args, P, synth, w, mk)
                except Exception as e:
                    out["end_to_end"] = {"error": repr(e)}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args, synth, w, mk)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


class Pinned:
    """numpy views of page-locked host memory from the library (hlala_pinned_alloc)"""

    def __init__(self, lib):
        import ctypes as C
        self.lib = lib; self.ptrs = []
        lib.hlala_pinned_alloc.restype = C.c_void_p; lib.hlala_pinned_alloc.argtypes = [C.c_size_t]
        lib.hlala_pinned_free.argtypes = [C.c_void_p]; lib.hlala_pinned_free.restype = None

    def empty(self, n, dtype):
        import ctypes as C
        dt = np.dtype(dtype); nb = max(1, int(n)) * dt.itemsize
        p = self.lib.hlala_pinned_alloc(nb)
        if not p:
            raise RuntimeError("hlala_pinned_alloc failed")
        self.ptrs.append(p)
        return np.frombuffer((C.c_char * nb).from_address(p), dtype=dt, count=int(n))

    def copy(self, a):
        o = self.empty(a.size, a.dtype); o[:] = a.reshape(-1)
        return o

    def free(self):
        for p in self.ptrs:
            self.lib.hlala_pinned_free(p)
        self.ptrs = []


class Boundary:
    """The C-ABI boundary of one step: hlala_batch_create (H2D) -> hlala_align_batch | export of the per-pair records (+ gather) -> per-pair scalars +
    hlala_batch_get_pairs_packed (D2H: level 4 B + graph char + read char + mapQ char of every column of the selected alignments, the 7 B per column
    SURVEY 8(d) counts as output) -> hlala_batch_destroy.  Caller buffers are page-locked (hlala_pinned_alloc) until pageable() swaps them."""

    DTS = dict(read_bases=np.uint8, read_quals=np.uint8, chain_contig=np.int32, chain_pos=np.int32, chain_offset=np.int32, chain_as=np.int32, chain_reverse=np.uint8, cigar=np.uint32,
               read_off=np.int64, chain_off=np.int64, cigar_off=np.int64, read_primary=np.int32)

    def __init__(self, args, P, ctx, batches):
        import ctypes as C
        self.C = C; self.P = P; self.ctx = ctx; self.lib = lib = ctx.lib; self.batches = batches
        self.pin = pin = Pinned(lib)
        self.ins = []
        lib.hlala_pack_bases.argtypes = [P.c_u8p, P.c_i64p, C.c_int64, P.c_u8p]
        for b in batches:
            d = {k: b[k] for k in ("n_pairs", "n_chains")}
            for k, dt in self.DTS.items():
                d[k] = pin.copy(np.ascontiguousarray(b[k], dt))
            if not args.ascii_bases:
                # the bases as a BAM decoder has them (hlala_bam_extract_seeds_opt with HLALA_SEEDS_PACKED: what the host program hands over): 4-bit packed, unpacked on the device
                ro = d["read_off"]; nr = len(ro) - 1
                pk = pin.empty((int(ro[-1]) + nr + 1) // 2 + 2, np.uint8)
                if lib.hlala_pack_bases(d["read_bases"].ctypes.data_as(P.c_u8p), ro.ctypes.data_as(P.c_i64p), nr, pk.ctypes.data_as(P.c_u8p)) != 0:
                    raise RuntimeError("hlala_pack_bases failed")
                d["read_bases_packed"] = pk; d["read_bases"] = None; d["first_read"] = 0
            st, keep = P.fill_struct(P.BatchIn, d)
            self.ins.append((st, keep, d))
        self.bytes_up = int(sum(v.nbytes for v in self.ins[0][2].values() if isinstance(v, np.ndarray)))
        self.n = n = args.pairs; self.nr = nr = 2 * n; self.cap = cap = nr * 184
        self.off = pin.empty(nr + 1, np.int64)
        self.cols = dict(col_level=pin.empty(cap, np.int32), col_gchar=pin.empty(cap, np.uint8), col_schar=pin.empty(cap, np.uint8), col_mapq=pin.empty(cap, np.uint8))
        self.scal = dict(pair_status=pin.empty(n, np.int32), best_chain=pin.empty(nr, np.int32), n_combinations=pin.empty(n, np.int32), pair_ll=pin.empty(n, np.float64),
                         pair_mapq=pin.empty(n, np.float64), mate_mapq=pin.empty(nr, np.float64), strands_valid=pin.empty(n, np.uint8))
        self.po, self._keep_po = P.fill_struct(P.PairsOut, self.scal)
        self.pk = P.PairsPackedOut(); self.pk.cap_cols = cap
        self._bind(self.off, self.cols)
        lib.hlala_batch_get_pairs_packed.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(P.PairsPackedOut)]
        lib.hlala_batch_get_pairs.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(P.PairsOut)]
        lib.hlala_batch_export_pair_records.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self.last_cols = 0
        self.host_s = {}; self.host_n = {}

    def _bind(self, off, cols):
        types = dict(self.P.PairsPackedOut._fields_)
        self.pk.col_off = off.ctypes.data_as(self.P.c_i64p)
        for k, v in cols.items():
            setattr(self.pk, k, v.ctypes.data_as(types[k]))
        self._bound = (off, cols)

    def pageable(self):
        P = self.P
        self._bind(np.zeros(self.nr + 1, np.int64), {k: np.zeros_like(v) for k, v in self.cols.items()})
        ins = []
        for b in self.batches:
            st, keep = P.fill_struct(P.BatchIn, {k: np.ascontiguousarray(b[k], dt) for k, dt in self.DTS.items()} | {k: b[k] for k in ("n_pairs", "n_chains")})
            ins.append((st, keep, None))
        self.ins = ins

    def _t(self, name, t0):
        t1 = time.perf_counter(); self.host_s[name] = self.host_s.get(name, 0.0) + (t1 - t0); self.host_n[name] = self.host_n.get(name, 0) + 1
        return t1

    def upload(self, i):
        C = self.C; h = C.c_void_p()
        t = time.perf_counter()
        self.ctx._check(self.lib.hlala_batch_create(self.ctx.h, C.byref(self.ins[i % len(self.ins)][0]), C.byref(h)), "hlala_batch_create")
        self._t("create", t)
        return h

    def launch(self, h):
        t = time.perf_counter()
        self.ctx._check(self.lib.hlala_align_batch(self.ctx.h, h), "hlala_align_batch")
        self._t("align_launch", t)
        return h

    def start(self, i):
        return self.launch(self.upload(i))

    def wait_export(self, h, rec, gather):
        C = self.C
        t = time.perf_counter()
        self.ctx._check(self.lib.hlala_batch_export_pair_records(self.ctx.h, h, C.c_void_p(rec.data_ptr())), "hlala_batch_export_pair_records")
        gather()
        self._t("wait_and_export", t)

    def readback(self, h):
        C = self.C
        t = time.perf_counter()
        self.ctx._check(self.lib.hlala_batch_get_pairs(self.ctx.h, h, C.byref(self.po)), "hlala_batch_get_pairs")
        t = self._t("get_pairs", t)
        self.ctx._check(self.lib.hlala_batch_get_pairs_packed(self.ctx.h, h, C.byref(self.pk)), "hlala_batch_get_pairs_packed")
        t = self._t("get_pairs_packed", t)
        self.lib.hlala_batch_destroy(h)
        self._t("destroy", t)
        self.last_cols = int(self.pk.n_cols_total)

    def finish(self, h, rec, gather):
        self.wait_export(h, rec, gather)
        self.readback(h)

    def host_ms(self):
        """mean wall clock of the host thread per call (ms): where the one host thread of the boundary loop spends a step"""
        return {k: 1e3 * self.host_s[k] / max(1, self.host_n[k]) for k in self.host_s}

    def pairs_ok(self):
        return int((self.scal["pair_status"] == 0).sum())

    def bytes_down(self):
        return 7 * self.last_cols + 8 * (self.nr + 1) + 37 * self.n

    def free(self):
        self.pin.free()


def end_to_end(args, P, synth, w, mk):
    """BAM bytes -> hla/*: `HLA-LA --action HLA` (hla-la_amd/host/HLA-LA.cpp) on a Graph M graph directory; bwa / samtools stand-ins hand over the BAM of a
    synthetic sample (their command lines run unchanged).  The figure is the program's own End-to-end line: units / (BAM decode + page-locking and insert size +
    alignment and typing).  One sample, one run per --decodeThreads value of --e2e-threads (0 = the decoder's default, at most 32 threads: the share of a
    256-thread host one of eight samples gets in BASELINE config 4 and, measured, its best count; 128 beside it)."""
    import re
    import shutil
    import stat
    import tempfile
    from tools import graphm_dir
    exe = os.path.join(ROOT, "hla-la_amd", "bin", "HLA-LA")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "hla-la_amd", "csrc"), "../bin/HLA-LA"])
    tmp = tempfile.mkdtemp(prefix="hlala_e2e_")
    try:
        t0 = time.time()
        gdir = os.path.join(tmp, "graph"); os.makedirs(gdir)
        loci = ["A", "B", "C", "DQA1", "DQB1", "DRB1"]
        graphm_dir.write_graph_dir_m(gdir, w, P, loci=loci)
        t_dir = time.time() - t0; t0 = time.time()
        ch = min(args.pairs, 1 << 20); nch = max(1, args.e2e_pairs // ch)
        clen = np.diff(w["contigs"]["contig_off"])
        bam = os.path.join(tmp, "sample.bam")
        bw = synth.BamWriter(bam, [(nm, int(clen[i])) for i, nm in enumerate(graphm_dir.ref_names(w))], threads=0, level=1)
        for k in range(nch):
            bk = mk(ch, 3000 + k, frac_gene=args.e2e_frac_gene); names, _ = synth.scrambled_names(k, ch)
            bw.append_batch(bk, names, order="coordinate"); del bk
        size = bw.close()
        t_bam = time.time() - t0

        def stub(path, text):
            with open(path, "w") as f:
                f.write(text)
            os.chmod(path, os.stat(path).st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
        stub(os.path.join(tmp, "bwa"), "#!/bin/bash\nif [ \"$1\" = index ]; then touch $2.sa $2.ann $2.bwt; fi\nexit 0\n")
        stub(os.path.join(tmp, "samtools"), f"#!/bin/bash\ncase \"$1\" in\n view) cat > /dev/null ;;\n sort) while [ $# -gt 0 ]; do if [ \"$1\" = -o ]; then ln -f {bam} \"$2\" || cp {bam} \"$2\"; fi; shift; done ;;\n"
                                             " index) touch \"$2.bai\" ;;\nesac\nexit 0\n")
        for fq in ("r1.fq", "r2.fq"):
            with open(os.path.join(tmp, fq), "w") as f:
                f.write("@r\nA\n+\nI\n")

        def one(threads, extra_env=None):
            outd = os.path.join(tmp, "out%d" % threads)
            cmd = [exe, "--action", "HLA", "--maxThreads", "2", "--sampleID", "S", "--outputDirectory", outd, "--PRG_graph_dir", gdir, "--FASTQU", os.path.join(tmp, "r1.fq"),
                   "--FASTQ1", os.path.join(tmp, "r1.fq"), "--FASTQ2", os.path.join(tmp, "r2.fq"), "--bwa_bin", os.path.join(tmp, "bwa"), "--samtools_bin", os.path.join(tmp, "samtools"),
                   "--mapAgainstCompleteGenome", "0", "--longReads", "0", "--loci", ",".join(loci), "--rngSeed", "12345", "--batchPairs", str(ch)]
            if threads > 0:
                cmd += ["--decodeThreads", str(threads)]
            t0 = time.time()
            r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp, timeout=1500, env=dict(os.environ, **(extra_env or {})))
            t_run = time.time() - t0
            if r.returncode != 0:
                return {"error": (r.stdout + r.stderr)[-1500:]}
            m = re.search(r"End-to-end: ([0-9.e+]+) units per s \(BAM decode ([0-9.e+-]+) s on (\d+) threads \+ page-locking and insert size ([0-9.e+-]+) s \+ alignment and typing ([0-9.e+-]+) s, of which the host spent ([0-9.e+-]+) s filling", r.stdout)
            sp = re.search(r"Speed: ([0-9.e+]+) protoSeeds", r.stdout)
            ph = re.search(r"Typing phases: (.*)", r.stdout)
            files = sorted(os.listdir(os.path.join(outd, "hla")))
            calls = [ln for ln in r.stdout.splitlines() if ln.startswith("Locus ")]
            shutil.rmtree(outd, ignore_errors=True)
            return {"value": float(m.group(1)), "unit": "read pairs/s", "pairs": nch * ch, "bam_bytes": int(size), "decode_s": float(m.group(2)), "decode_threads": int(m.group(3)),
                    "page_locking_and_insert_size_s": float(m.group(4)), "alignment_and_typing_s": float(m.group(5)), "window_fill_beside_the_gpu_s": float(m.group(6)), "speed_line_pairs_per_s": float(sp.group(1)) if sp else None,
                    "process_wall_s": t_run, "whole_process_pairs_per_s": nch * ch / t_run, "typing_phases": ph.group(1) if ph else None,
                    "log": [ln[:700] for ln in r.stdout.splitlines() if ("Seed extraction:" in ln or ln.startswith("End-to-end:"))] + [ln[:300] for ln in r.stderr.splitlines() if ln.startswith(("bam-debug:", "host-debug:"))][:60],
                    "loci": loci, "result_files": len(files), "calls": calls[:6]}

        def several(ns, threads=0):
            """ns samples (the same BAM) in ONE call on one device: the decode of sample k + 1 runs on the host threads while the GPU aligns sample k (round 6: HLA-LA.cpp
            SampleSchedule).  pairs / wall of the program's `Samples:` line (from the end of the graph directory's load to the last sample's result files), every sample's
            hla/* compared byte for byte with the first one's."""
            import hashlib
            outs = [os.path.join(tmp, "outs%d_%d" % (ns, i)) for i in range(ns)]
            rep_ = lambda x: ",".join([x] * ns)
            cmd = [exe, "--action", "HLA", "--maxThreads", "2", "--sampleID", ",".join("S%d" % i for i in range(ns)), "--outputDirectory", ",".join(outs), "--PRG_graph_dir", gdir,
                   "--FASTQU", rep_(os.path.join(tmp, "r1.fq")), "--FASTQ1", rep_(os.path.join(tmp, "r1.fq")), "--FASTQ2", rep_(os.path.join(tmp, "r2.fq")), "--bwa_bin", os.path.join(tmp, "bwa"),
                   "--samtools_bin", os.path.join(tmp, "samtools"), "--mapAgainstCompleteGenome", "0", "--longReads", "0", "--loci", ",".join(loci), "--rngSeed", "12345", "--batchPairs", str(ch)]
            if threads > 0:
                cmd += ["--decodeThreads", str(threads)]
            t0 = time.time()
            r = subprocess.run(cmd, capture_output=True, text=True, cwd=tmp, timeout=3000)
            t_run = time.time() - t0
            if r.returncode != 0:
                return {"error": (r.stdout + r.stderr)[-1500:]}
            m = re.search(r"Samples: (\d+) on (\d+) device\(s\), (\d+) decoding at a time, in ([0-9.e+-]+) s", r.stdout)
            def digest(d):
                h = hashlib.sha256()
                for fn in sorted(os.listdir(os.path.join(d, "hla"))):
                    h.update(fn.encode()); h.update(open(os.path.join(d, "hla", fn), "rb").read())
                return h.hexdigest()
            dg = [digest(o) for o in outs]
            for o in outs:
                shutil.rmtree(o, ignore_errors=True)
            wall = float(m.group(4))
            return {"samples": ns, "decode_threads_asked": threads, "pairs": ns * nch * ch, "wall_s": wall, "value": ns * nch * ch / wall, "unit": "read pairs/s", "decoding_at_a_time": int(m.group(3)), "process_wall_s": t_run,
                    "whole_process_pairs_per_s": ns * nch * ch / t_run, "hla_files_identical_across_samples": len(set(dg)) == 1,
                    "per_sample_lines": [ln[:260] for ln in r.stdout.splitlines() if ln.startswith("End-to-end:")],
                    "what": "one HLA-LA call, %d samples on one device: sample k + 1 is decoded while sample k is aligned and typed; pairs of all samples / seconds from the loaded graph directory to the last result file" % ns}

        runs = [one(int(t)) for t in str(args.e2e_threads).split(",") if t.strip() != ""]
        res = runs[0]
        if args.e2e_samples:
            res["several_samples"] = [several(int(x.split(":")[0]), int(x.split(":")[1]) if ":" in x else 0) for x in str(args.e2e_samples).split(",") if x.strip() != ""]
        if args.e2e_variants:          # experiments: the same sample again under other environments ("label:ENV=1 ENV2=x;label2:...")
            res["variants"] = {}
            for v in args.e2e_variants.split(";"):
                label, _, envs = v.partition(":")
                rv = one(0, dict(kv.split("=", 1) for kv in envs.split()))
                res["variants"][label] = {k: rv.get(k) for k in ("value", "decode_s", "page_locking_and_insert_size_s", "alignment_and_typing_s", "window_fill_beside_the_gpu_s", "typing_phases", "process_wall_s", "error") if k in rv}
        res.update({"host_cpus": {"hardware_threads": os.cpu_count(), "cgroup_cpu_quota": host_cpu_quota()},
                    "gene_window_share_of_the_sample": args.e2e_frac_gene, "setup_s": {"graph_directory": t_dir, "sample_generation_and_bam": t_bam},
                    "what": "HLA-LA --action HLA: BAM bytes -> hla/* (decode on the stated host threads, batches of %d pairs two in flight on one GPU, typing of %d loci, result files); "
                            "value = pairs / (decode + page-locking and insert size + alignment and typing), the program's End-to-end line; graph loading and context creation are per process" % (ch, len(loci))})
        if len(runs) > 1:
            res["other_thread_counts"] = [{k: r.get(k) for k in ("value", "decode_s", "decode_threads", "page_locking_and_insert_size_s", "alignment_and_typing_s", "process_wall_s", "whole_process_pairs_per_s", "error") if k in r} for r in runs[1:]]
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def long_reads(args, P, synth, w):
    """BASELINE config 5: long-read mode (processBAM::alignOneLongRead, mapper/processBAM.cpp:3618-3838: projection of the one primary alignment +
    extendToFullSequenceLength + scoreOneAlignment; the reference runs no extension DP in this mode, :3732-3734).  --long-reads distinct reads over 6-14 kb of
    reference (mean read length ~10.2 kb) drawn from the backbone haplotypes of the bench's graph -- crossing the gene windows where they lie --, substitutions 5 %, insertions 4 %,
    deletions 4 %, in batches of 10 000 reads through a context with 16 384-column rows.  Inputs resident; reads/s and bases/s, stage times."""
    n = args.long_reads; per = getattr(args, "long_reads_batch", 50000)
    ctx = P.Context(w["graph"], w["contigs"], insert_mean=200.0, insert_sd=35.0, rng_seed=12345, long_read_mode=1, max_columns=16384, device=0)
    try:
        t0 = time.time()
        bs = synth.make_long_batches_parallel(w, n, per_batch=per, seed=700, len_lo=6000, len_hi=14000, procs=min(16, max(1, (os.cpu_count() or 8) // 2)))
        t_gen = time.time() - t0
        gbs = [ctx.batch_unpaired(x) for x in bs]
        bases = int(sum(int(x["read_off"][-1]) for x in bs))
        gbs[0].align(); gbs[0].stats()                       # warm-up (stats() waits for the batch)
        t = time.perf_counter()
        for g in gbs:
            g.align()
        sts = [g.stats() for g in gbs]                     # (the batches run in order on the context's main stream)
        dt = time.perf_counter() - t
        ok = int(sum(int((g.pairs_scalars()["pair_status"] == 0).sum()) for g in gbs))
        # ---- after the clock has stopped: some of the reads that were just timed against the CPU oracle (checker only; VERDICT r04: the timed reads were never
        # compared with anything) -- chosen chain, every column of the selected alignment, the per-position qualities, the log likelihood
        checked = 0
        try:
            checked = long_reads_parity(args, gbs[0], bs[0], w, min(args.long_reads_check, bs[0]["n_pairs"]))
        except (OSError, ImportError, subprocess.CalledProcessError) as e:      # no oracle on this machine (not built / no compiler): the check is skipped and says so;
            checked = {"skipped": repr(e)}                                      # a DIFFERENCE from the oracle (AssertionError) is not caught: the bench fails, no line is printed
        for g in gbs:
            g.close()
        # roofline of the leg: SURVEY 8(d)'s per-unit bytes for ONE read and ONE chain (packed bases + qualities, the record, translation + reference base per column, the
        # CSR edges of the levels the chain spans counted once, the output columns) over the time of the dominant kernel, k_project_chains<ProjLdsLong>
        cols = float(sum(int(s_.n_out_columns) for s_ in sts)) / max(1, n); rl = bases / max(1, n)
        e_mean = w["graph"]["n_edges"] / max(1, w["graph"]["n_nodes"] - 1)
        b_read = (rl / 2 + rl) + (32 + 5 * rl + 5 * e_mean * cols) + 7 * cols + 64
        proj_s = float(sum(s_.ms_project for s_ in sts)) * 1e-3
        ach = b_read * n / max(proj_s, 1e-9) / 1e9
        roof = {"bound": "latency", "bound_note": "one wavefront per read: column passes against its HBM slab and the level loops of the re-threading DP (round 6: long segments level by level out of LDS-staged chunks); achieved / peak / frac are the HBM figures the metric asks for", "kernel": "k_project_chains<ProjLdsLong>", "kernel_ms_sum": proj_s * 1e3, "algorithmic_bytes_per_read": b_read, "columns_per_read": cols, "achieved": ach, "peak": 8000.0, "unit": "GB/s",
                "frac": ach / 8000.0, "traffic": None, "note": "HIP events of the batches' projection stage; the measured HBM traffic and the SQ counters of the kernel: profiles/r0N_long_*"}
        tl = next((f for f in (os.path.join(ROOT, "profiles", t + "_long_traffic.json") for t in ("r06", "r05")) if os.path.exists(f)), "")
        if tl:
            try:
                tj = json.load(open(tl)); roof["traffic_per_read"] = tj.get("hbm_bytes_per_read"); roof["traffic"] = tj.get("hbm_bytes_per_read", 0) * n / max(1, len(gbs)); roof["secondary"] = tj.get("secondary")
                roof["traffic_source"] = "%s (kernel sources %s)" % (os.path.relpath(tl, ROOT), tj.get("kernel_source_hash"))
            except Exception:
                pass
        return {"reads": n, "bases": bases, "parity_checked": checked, "reads_per_batch": per, "roofline": roof, "mean_read_length": bases / max(1, n), "reads_per_s": n / dt, "bases_per_s": bases / dt, "seconds": dt, "batches": len(gbs), "reads_ok": ok,
                "stage_ms_sum": {"project": float(sum(s.ms_project for s in sts)), "pad_and_score": float(sum(s.ms_extend for s in sts)), "select": float(sum(s.ms_pair for s in sts))},
                "chain_errors": int(sum(int(s.n_errors) for s in sts)), "generation_s": t_gen,
                "what": "hlala_batch_create_unpaired batches resident in HBM, hlala_align_batch each; distinct reads, one primary alignment each (the reference takes primaries only, processBAM.cpp:732-738)"}
    finally:
        ctx.close()


def long_reads_parity(args, gb, u, w, m):
    """The first m reads of a timed long-read batch, product (already aligned: gb) against oracle/hlala_oracle.cpp (alignOneLongRead, mapper/processBAM.cpp:3618-3838).
    Returns m; raises on the first difference."""
    if m <= 0:
        return 0
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    from oracle_binding import Oracle
    c1 = int(u["chain_off"][m]); b1 = int(u["read_off"][m]); g1 = int(u["cigar_off"][c1])
    sub = dict(n_pairs=m, read_off=u["read_off"][:m + 1], read_bases=u["read_bases"][:b1], read_quals=u["read_quals"][:b1], chain_off=u["chain_off"][:m + 1],
               read_primary=u["read_primary"][:m], n_chains=c1, chain_contig=u["chain_contig"][:c1], chain_pos=u["chain_pos"][:c1], chain_offset=u["chain_offset"][:c1],
               chain_as=u["chain_as"][:c1], chain_reverse=u["chain_reverse"][:c1], cigar_off=u["cigar_off"][:c1 + 1], cigar=u["cigar"][:g1])
    e = Oracle(w["graph"], w["contigs"], insert_mean=200.0, insert_sd=35.0, rng_seed=12345, long_read_mode=1, max_columns=16384).align_long_reads(sub)["pairs"]
    pk = gb.pairs_packed(); sc = gb.pairs_scalars(); off = pk["col_off"]; ncols = np.diff(off); stride = 16384
    for r in range(m):
        k0 = int(e["n_cols"][r])
        if int(e["pair_status"][r]) != int(sc["pair_status"][r]) or k0 != int(ncols[r]):
            raise AssertionError(f"long read {r}: status / columns differ from the oracle")
        for key in ("col_level", "col_edge", "col_gchar", "col_schar", "col_mapq"):
            if not np.array_equal(np.asarray(e[key])[r * stride:r * stride + k0], pk[key][off[r]:off[r] + k0]):
                raise AssertionError(f"long read {r}: {key} differs from the oracle")
    if not (np.allclose(sc["pair_ll"][:m], e["pair_ll"][:m], rtol=1e-12, atol=0) and np.array_equal(sc["best_chain"][:m], e["best_chain"][:m])):
        raise AssertionError("long reads: likelihood / selection differs from the oracle")
    return int(m)


def host_cpu_quota():
    """CPUs the process may keep busy under its control group's CFS quota (cgroup v2 cpu.max, v1 cpu.cfs_quota_us); None: no quota.  The GPU boxes of this
    pool show 256 hardware threads and a quota of 16 CPUs: every host-side figure of this file (decoder, end to end, cpu_baseline) is a 16-CPU figure."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if q == "max" else float(q) / float(per)
    except Exception:
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return q / per if q > 0 and per > 0 else None
    except Exception:
        return None


def cpu_baseline(args, synth, w, mk):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    from oracle_binding import Oracle
    nb = min(args.cpu_pairs, args.pairs)
    sb = mk(nb, 1000)              # same generator, same seed: the first pairs of rank 0's workload
    o = Oracle(w["graph"], w["contigs"], insert_mean=sb["insert_mean"], insert_sd=sb["insert_sd"], rng_seed=12345, max_columns=384)
    tc = time.perf_counter(); o.align_batch(sb); dtc = time.perf_counter() - tc
    nm = min(4 * nb, args.pairs)
    quota = host_cpu_quota()
    mb = mk(nm, 1000)
    tc = time.perf_counter(); r = o.align_batch_mt(mb, 0, pairs_only=True); dtm = time.perf_counter() - tc
    return {"value": nb / dtc, "unit": "read pairs/s", "cores": 1, "kind": "port",
            "sample": f"first {nb} pairs of the same synthetic workload, oracle/hlala_oracle.cpp (C++ restatement, -O2, single thread as in HLA-LA.cpp:799), {dtc:.1f} s",
            "all_cores": {"value": nm / dtm, "unit": "read pairs/s", "cores": int(min(r["threads"], quota)) if quota else int(r["threads"]), "threads": int(r["threads"]), "host_cpus": os.cpu_count(),
                          "cgroup_cpu_quota": quota,
                          "sample": f"first {nm} pairs, OpenMP parallel for schedule(dynamic,64) over pairs, {dtm:.1f} s"}}


def extras(args, P, synth, w, mk, b, ctx, gb, ckw):
    """Measurements outside the timed region (rank 0, one GPU)."""
    ex = {}
    # ---- gene-window and backbone pairs separately (resident, one step each after a warm-up)
    if args.graph == "m":
        ns = min(args.pairs, 262144)
        for name, fg in (("gene_window_pairs", 1.0), ("backbone_pairs", 0.0)):
            sb = mk(ns, 77, frac_gene=fg)
            g2 = ctx.batch(sb); g2.align(); g2.stats()
            t = time.perf_counter(); g2.align(); s2 = g2.stats(); dt = time.perf_counter() - t
            ex[name] = {"pairs": ns, "pairs_per_s": ns / dt, "chains_per_pair": sb["n_chains"] / ns, "extended_chains_per_pair": s2.n_chains_extended / ns,
                        "dp_ms_by_class": [float(x) for x in s2.ms_dp_class], "dp_calls_entering_class": [int(x) for x in s2.n_dp_class], "chain_errors": int(s2.n_errors)}
            # the same batch as the headline's loops run theirs: two resident batch objects alternate, a batch's side-stream tail beside the next one's main stream
            # (`pairs_per_s` above is ONE batch alone: main stream, then its tail classes one after the other with nothing beside them)
            try:
                g3 = ctx.batch(sb); g3.align(); g3.stats()
                pr = [g2, g3]; nst = 8
                t = time.perf_counter()
                for i in range(nst):
                    pr[i % 2].align()
                    if i >= 1: pr[(i - 1) % 2].stats()
                pr[(nst - 1) % 2].stats()
                dt2 = time.perf_counter() - t
                ex[name]["pairs_per_s_two_in_flight"] = ns * nst / dt2; ex[name]["steps_two_in_flight"] = nst
                g3.close()
            except Exception as e:
                ex[name]["pairs_per_s_two_in_flight"] = None; ex[name]["two_in_flight_error"] = repr(e)
            g2.close()
    # ---- typer kernels at the size of a real class-I locus (C = 3000 clusters, R = 400 reads, 546 exon columns)
    try:
        loc = synth.make_locus(seed=5, n_clusters=3000, exon_length=546, n_reads=400)
        Cn, R = 3000, 400
        ctx.exon_loglik(loc)
        t = time.perf_counter(); LL, mism = ctx.exon_loglik(loc); t_exon = time.perf_counter() - t
        ctx.pair_loglik(LL, mism)
        t = time.perf_counter(); pl = ctx.pair_loglik(LL, mism); t_pair = time.perf_counter() - t
        ctx.call_locus(*pl)
        t = time.perf_counter(); ctx.call_locus(*pl); t_call = time.perf_counter() - t
        ex["typer"] = {"clusters": Cn, "reads": R, "exon_loglik_ms": t_exon * 1e3, "pair_loglik_ms": t_pair * 1e3, "call_locus_ms": t_call * 1e3,
                       "g_logavg_per_s": Cn * (Cn + 1) / 2 * R / t_pair / 1e9, "what": "host wall clock of the C-ABI calls incl. their transfers"}
    except Exception as e:      # the typer record is a report, never a reason to lose the bench line
        ex["typer"] = {"error": str(e)}
    return ex


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--gen-worker":
        gen_worker(sys.argv[2:]); sys.exit(0)
    main()
