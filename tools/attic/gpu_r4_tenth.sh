#!/bin/bash
# round 4, tenth session: packed bases + lazy window fill + lazy outputs (tests that walk samples window by window), fewer tail blocks per CU (A/B), end to end under four environments
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
make -s -C oracle 2>&1 | tail -1; make -s -C tools/graphm 2>&1 | tail -1
timeout 1500 python -m pytest tests/test_bam.py tests/test_bam_scale.py tests/test_hla_la_binary.py tests/test_end_to_end.py tests/test_config3_stream.py tests/test_gpu_align.py tests/test_insert_size.py -m gpu -q -x > gpurun_out/r4_parity10.log 2>&1
echo "parity rc=$?"; tail -4 gpurun_out/r4_parity10.log
bash tools/gpu_r4_ab.sh --modes two "default:HLALA_X=1" "broad1:HLALA_BROAD_PER_CU=1" "broad1-large2:HLALA_BROAD_PER_CU=1 HLALA_LARGE_GRID_DIV=2" "broad2-large2:HLALA_BROAD_PER_CU=2 HLALA_LARGE_GRID_DIV=2"
timeout 1500 python bench.py --steps 6 --warmup 2 --resident-steps 0 --long-reads 0 --no-cpu-baseline --e2e-threads 0 --e2e-variants "eager-ascii:HLALA_BAM_EAGER=1 HLALA_SEEDS_ASCII=1;eager-packed:HLALA_BAM_EAGER=1;lazy-ascii:HLALA_SEEDS_ASCII=1;lazy-packed-again:HLALA_X=1" > gpurun_out/r4_e2e_variants.log 2> gpurun_out/r4_e2e_variants.err
python - <<'PY'
import json
l = [x for x in open('gpurun_out/r4_e2e_variants.log') if x.startswith('{')]
if l:
    d = json.loads(l[-1]); e = d.get("end_to_end") or {}
    print("boundary value %d ms %.1f bytes up %d | host ms %s" % (d["value"], d["ms_per_step"], d["host_inclusive"]["bytes_up_per_step"], {k: round(v, 1) for k, v in d["host_inclusive"]["host_thread_ms_per_call"].items()}))
    print("lazy-packed", {k: e.get(k) for k in ("value", "decode_s", "page_locking_and_insert_size_s", "alignment_and_typing_s", "window_fill_beside_the_gpu_s", "typing_phases", "error")})
    for k, v in (e.get("variants") or {}).items():
        print(k, v)
else:
    print(open('gpurun_out/r4_e2e_variants.err').read()[-2000:])
PY
