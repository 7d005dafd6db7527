#!/bin/bash
# The N > 1 paths on a 1-GPU box (an 8-GPU node is the driver's to launch):
#  1. bench.py under torch.distributed.run, 2 and 4 ranks sharing the device, gloo carrying the gather (on a multi-GPU node the same code runs over RCCL):
#     rank 0 generates the workload once, the others read it from shared memory; the line carries per-rank seconds and every rank's pairs_ok
#  2. `python bench.py --gpus 2` outside a launcher (it starts its own ranks before anything touches the GPU)
#  3. HLA-LA --devices 0,0,0,0: four contexts in the one host process, windows dealt round-robin (tests/test_hla_la_binary.py drives the program the same way)
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HLALA_BENCH_BACKEND=gloo
for n in 2 4; do
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2961$n bench.py --gpus $n --steps 2 --warmup 1 --pairs 131072 --levels 1000000 > gpurun_out/r5_bench_${n}ranks_dryrun.log 2>&1
  echo "ranks=$n rc=$?"; grep '^{' gpurun_out/r5_bench_${n}ranks_dryrun.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print({k: d[k] for k in ('value', 'n_gpus', 'steps', 'ms_per_step')}, 'pairs_ok_per_rank', c['pairs_ok_per_rank'], 'gathers', d['host_inclusive']['gathers_in_timed_region'], 'per_rank_s', d['host_inclusive']['per_rank_s'], 'resident', {k: c['resident'][k] for k in ('value', 'ms_per_step')})
" || tail -5 gpurun_out/r5_bench_${n}ranks_dryrun.log
done
timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --pairs 131072 --levels 1000000 > gpurun_out/r5_bench_2ranks_selfspawn.log 2>&1
echo "selfspawn rc=$?"; grep '^{' gpurun_out/r5_bench_2ranks_selfspawn.log | tail -1 | cut -c1-300
unset HLALA_BENCH_BACKEND
make -s -C oracle 2>&1 | tail -1
timeout 900 python -m pytest tests/test_hla_la_binary.py -m gpu -q -x 2>&1 | tail -3
