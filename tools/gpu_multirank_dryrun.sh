#!/bin/bash
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
export HLALA_BENCH_BACKEND=gloo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 2 --warmup 1 --pairs 65536 --levels 500000 2>&1 | tail -3 | cut -c1-900
unset HLALA_BENCH_BACKEND
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --steps 2 --warmup 1 --pairs 65536 --levels 500000 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
timeout 300 python -m pytest tests/test_host_cpp.py -q -m gpu 2>&1 | tail -2
