#!/bin/bash
# round 4, closing session: the whole GPU suite, the bench line as the driver runs it, the profile artefacts of the final kernels, the N > 1 dry runs
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
make -s -C oracle 2>&1 | tail -1; make -s -C tools/graphm 2>&1 | tail -1
( time timeout 2400 python -m pytest tests -m gpu -x -q ) > gpurun_out/r4_pytest_full.log 2>&1
tail -6 gpurun_out/r4_pytest_full.log
# (the profile passes first and their summaries derived HERE, so that the bench line that follows finds profiles/r04_traffic.json tagged with this build's sources;
#  the derived files are made again from gpurun_out/ after the session -- only gpurun_out/ travels back)
bash tools/gpu_profile.sh r04 2>&1 | tail -3 | cut -c1-300
python tools/derive_profiles.py r04 2>&1 | tail -2
( time timeout 1500 python bench.py --steps 20 --warmup 5 ) > gpurun_out/r4_bench_full.log 2> gpurun_out/r4_bench_full.err
tail -c 1500 gpurun_out/r4_bench_full.log | cut -c1-600; tail -4 gpurun_out/r4_bench_full.err
bash tools/gpu_multirank_dryrun.sh 2>&1 | tail -8 | cut -c1-700
