#!/bin/bash
# round 5, closing session: the whole GPU suite, the profile artefacts of the final kernels (short reads: tools/gpu_profile.sh r05; long reads: tools/gpu_r5_long.sh), the bench
# line as the driver runs it, the N > 1 dry runs.  The derived summaries are made HERE first, so that the bench line that follows finds profiles/r05_traffic.json tagged with
# this build's sources; only gpurun_out/ travels back: `python tools/derive_profiles.py r05; python tools/derive_long_profile.py` make them again afterwards.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
make -s -C oracle 2>&1 | tail -1; make -s -C tools/graphm 2>&1 | tail -1
( time timeout 2700 python -m pytest tests -m gpu -x -q ) > gpurun_out/r5_pytest_full.log 2>&1
tail -6 gpurun_out/r5_pytest_full.log
bash tools/gpu_profile.sh r05 2>&1 | tail -3 | cut -c1-300
python tools/derive_profiles.py r05 2>&1 | tail -12
bash tools/gpu_r5_long.sh r05_long 50000 2>&1 | tail -8 | cut -c1-400
python tools/derive_long_profile.py r05_long 20000 2>&1 | tail -2
( time timeout 1800 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r5_bench_full.log 2> gpurun_out/r5_bench_full.err
tail -c 1500 gpurun_out/r5_bench_full.log | cut -c1-600; tail -4 gpurun_out/r5_bench_full.err
bash tools/gpu_multirank_dryrun.sh 2>&1 | tail -8 | cut -c1-700
