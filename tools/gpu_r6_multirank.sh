#!/bin/bash
# round 6: bench.py with EIGHT ranks sharing the one device of this box (gloo carries the gather; on an 8-GPU node the same code runs over RCCL): the ranks' batches are
# generated side by side by child processes of rank 0 -- generation_s in the line is the start-up the driver would see
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export HLALA_BENCH_BACKEND=gloo
export HLALA_POOL_CAP_GB=2          # eight processes share ONE device here: their contexts (13 GB of graph and slabs each) and three batches each must fit side by side
for n in ${RANKS:-8}; do
  t0=$(date +%s)
  timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2971$n bench.py --gpus $n --steps 2 --warmup 1 --pairs ${PAIRS:-65536} --levels ${LEVELS:-5000000} > gpurun_out/r6_bench_${n}ranks_dryrun.log 2>&1
  echo "ranks=$n rc=$? wall $(( $(date +%s) - t0 )) s"; grep '^{' gpurun_out/r6_bench_${n}ranks_dryrun.log | tail -1 | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); c = d['config']
    print({k: d[k] for k in ('value', 'n_gpus', 'steps', 'ms_per_step')}, 'generation_s', c['generation_s'], 'pairs_ok_per_rank', c['pairs_ok_per_rank'], 'gathers', d['host_inclusive']['gathers_in_timed_region'], 'per_rank_s', d['host_inclusive']['per_rank_s'])
" || tail -5 gpurun_out/r6_bench_${n}ranks_dryrun.log

done
