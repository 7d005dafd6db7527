cd $GRAFT_REPO_ROOT
make -C oracle 2>&1 | tail -1
mkdir -p gpurun_out
# full default bench (1M pairs, 5M-level graph)
( time timeout 1500 python bench.py --steps 2 --warmup 1 ) > gpurun_out/bench_r01_full.log 2>&1
tail -2 gpurun_out/bench_r01_full.log | cut -c1-2500
# kernel trace of a smaller run (same command shape) for profiles/
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -- python3 $GRAFT_REPO_ROOT/bench.py --pairs 262144 --levels 5000000 --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_r01.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_r01.log | cut -c1-1500
find $GRAFT_REPO_ROOT/gpurun_out/prof_r01 -name "*stats*" | head
