#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite (without the 8 M-pair streaming test), then that test with its report, then a short bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
nproc > gpurun_out/r3_host.txt; free -g >> gpurun_out/r3_host.txt; df -h /tmp . >> gpurun_out/r3_host.txt 2>&1
timeout 1200 python -m pytest tests -m gpu -q -x --durations=25 --deselect tests/test_config3_stream.py > gpurun_out/r3_pytest_a.log 2>&1
echo "pytest A rc=$?" >> gpurun_out/r3_pytest_a.log
tail -45 gpurun_out/r3_pytest_a.log
timeout 1200 python -m pytest tests/test_config3_stream.py -m gpu -q -x -s > gpurun_out/r3_pytest_b.log 2>&1
echo "pytest B rc=$?" >> gpurun_out/r3_pytest_b.log
tail -25 gpurun_out/r3_pytest_b.log
timeout 900 python bench.py --steps 10 --warmup 3 > gpurun_out/r3_bench_a.log 2>&1
echo "bench rc=$?" >> gpurun_out/r3_bench_a.log
tail -c 3000 gpurun_out/r3_bench_a.log
