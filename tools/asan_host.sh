#!/bin/bash
# AddressSanitizer / UBSan over the host-side code of the library (parsers, filters, typer files): CPU build only.
# Builds hla-la_amd/csrc/host_*.cpp + flat_graph.cpp into /tmp/libhlala_host_asan.so and runs the parser robustness tests against it.
set -e
cd "$(dirname "$0")/.."
S=hla-la_amd/csrc
g++ -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined -fno-omit-frame-pointer -o /tmp/libhlala_host_asan.so \
    $S/host_check.cpp $S/flat_graph.cpp $S/host_filters.cpp $S/host_loaders.cpp $S/host_bam.cpp $S/host_typer.cpp -lz -pthread
ASAN_LIB=$(g++ -print-file-name=libasan.so)
# (libstdc++ is preloaded too: the sanitizer resolves __cxa_throw at start-up, before Python loads the library that throws)
STDCXX=$(g++ -print-file-name=libstdc++.so)
LD_PRELOAD="$ASAN_LIB $STDCXX" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 HLALA_LIB_PATH=/tmp/libhlala_host_asan.so HLALA_HOST_LIB=/tmp/libhlala_host_asan.so \
    python -m pytest tests/test_host_flatten.py -k "matches_oracle or linear_steps or rejects_bad" -x -q -m "not gpu"       # (the graph flatten incl. the linear steps and the device's jump tables; the ABI tests need the GPU library)
LD_PRELOAD="$ASAN_LIB $STDCXX" ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 HLALA_LIB_PATH=/tmp/libhlala_host_asan.so \
    python -m pytest tests/test_parsers_robust.py tests/test_typer_files.py tests/test_filters.py tests/test_bam.py tests/test_bam_scale.py tests/test_graph_files.py -x -q -m "not gpu" "$@"
