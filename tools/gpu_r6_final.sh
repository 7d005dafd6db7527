#!/bin/bash
# round 6, closing session: the whole GPU suite, the profile artefacts of the final kernels (short reads: tools/gpu_profile.sh r06; long reads: tools/gpu_r5_long.sh r06_long), the
# bench line as the driver runs it, the eight-rank dry run.  The derived summaries are made HERE first, so that the bench line that follows finds profiles/r06_traffic.json tagged
# with this build's sources; only gpurun_out/ travels back: `python tools/derive_profiles.py r06; python tools/derive_long_profile.py r06_long 20000` make them again afterwards.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
make -s -C oracle 2>&1 | tail -1; make -s -C tools/graphm 2>&1 | tail -1
( time timeout 2700 python -m pytest tests -m gpu -x -q ) > gpurun_out/r6_pytest_full.log 2>&1
tail -6 gpurun_out/r6_pytest_full.log
bash tools/gpu_profile.sh r06 2>&1 | tail -3 | cut -c1-300
python tools/derive_profiles.py r06 2>&1 | tail -12
bash tools/gpu_r5_long.sh r06_long 50000 2>&1 | tail -8 | cut -c1-400
python tools/derive_long_profile.py r06_long 20000 2>&1 | tail -2
( time timeout 1800 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r6_bench_full.log 2> gpurun_out/r6_bench_full.err
tail -c 1500 gpurun_out/r6_bench_full.log | cut -c1-600; tail -4 gpurun_out/r6_bench_full.err
bash tools/gpu_r6_multirank.sh 2>&1 | tail -4 | cut -c1-700
