#!/bin/bash
# the N > 1 path of bench.py on a 1-GPU box: two ranks share the device, gloo carries the gather (on an 8-GPU node the same code runs over RCCL);
# also `python bench.py --gpus 2` outside a launcher (it starts its own ranks), and HLA-LA's end-to-end line after the decoder changes
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HLALA_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --pairs 65536 --levels 500000 > gpurun_out/r3_bench_2ranks_dryrun.log 2>&1
tail -1 gpurun_out/r3_bench_2ranks_dryrun.log | cut -c1-700
timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --pairs 65536 --levels 500000 > gpurun_out/r3_bench_2ranks_selfspawn.log 2>&1
tail -1 gpurun_out/r3_bench_2ranks_selfspawn.log | cut -c1-400
unset HLALA_BENCH_BACKEND
timeout 900 python bench.py --steps 2 --warmup 1 --host-steps 0 --no-cpu-baseline > gpurun_out/r3_bench_e2e_b.log 2>&1
python - <<'PY'
import json
l=[x for x in open('gpurun_out/r3_bench_e2e_b.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); e=d.get("end_to_end") or {}
    print("end_to_end", {k: e.get(k) for k in ("value","pairs","decode_s","decode_threads","alignment_and_typing_s","typing_phases","process_wall_s","error")})
PY
