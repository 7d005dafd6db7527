#!/bin/bash
# round 4, session 14: the host side of a sample under the box's 16-CPU quota -- hop beside the inflate, kept reads for the k-mer questions, page-locking
# window by window: the tests that touch them, then the end-to-end run with its variants (page-locking of the whole sample up front as before)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
make -s -C oracle 2>&1 | tail -1; make -s -C tools/graphm 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_typer.py tests/test_typer_files.py tests/test_end_to_end.py tests/test_hla_la_binary.py tests/test_bam.py tests/test_bam_scale.py tests/test_insert_size.py -m gpu -q -x > gpurun_out/r4_host_tests.log 2>&1
echo "host tests rc=$?"; tail -4 gpurun_out/r4_host_tests.log
HLALA_HOST_DEBUG=1 HLALA_BAM_DEBUG=1 timeout 1500 python bench.py --steps 6 --warmup 2 --resident-steps 0 --long-reads 0 --no-cpu-baseline --no-extras-but-e2e --e2e-threads 0,16 --e2e-variants "zlib:HLALA_BAM_ZLIB=1;again:HLALA_X=1" > gpurun_out/r4_e2e_host.log 2> gpurun_out/r4_e2e_host.err
echo "bench rc=$?"
python3 - <<'PY'
import json
l = [x for x in open('gpurun_out/r4_e2e_host.log') if x.startswith('{')]
if l:
    d = json.loads(l[-1]); e = d.get("end_to_end", {})
    keys = ("value", "decode_s", "decode_threads", "page_locking_and_insert_size_s", "alignment_and_typing_s", "window_fill_beside_the_gpu_s", "typing_phases", "process_wall_s", "error")
    print("e2e", {k: e.get(k) for k in keys}); print(e.get("host_cpus"))
    for ln in e.get("log", []): print("   ", ln[:400])
    for r in e.get("other_thread_counts", []): print("other", r)
    for k, v in e.get("variants", {}).items(): print("variant", k, v)
else:
    print(open('gpurun_out/r4_e2e_host.err').read()[-2000:])
PY
# the same program on a sample with 30 % of its pairs in the gene windows
HLALA_HOST_DEBUG=1 timeout 1500 python bench.py --steps 2 --warmup 1 --resident-steps 0 --long-reads 0 --no-cpu-baseline --no-extras-but-e2e --e2e-threads 0 --e2e-frac-gene 0.3 > gpurun_out/r4_e2e_gene30.log 2> gpurun_out/r4_e2e_gene30.err
python3 - <<'PY'
import json
l = [x for x in open('gpurun_out/r4_e2e_gene30.log') if x.startswith('{')]
if l:
    e = json.loads(l[-1]).get("end_to_end", {})
    print("e2e 30% gene share", {k: e.get(k) for k in ("value", "decode_s", "alignment_and_typing_s", "typing_phases", "process_wall_s", "error")})
    for ln in e.get("log", []):
        if "host-debug: locus" in ln or "whole action" in ln: print("   ", ln[-300:])
else:
    print(open('gpurun_out/r4_e2e_gene30.err').read()[-1500:])
PY
