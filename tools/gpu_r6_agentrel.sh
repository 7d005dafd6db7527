#!/bin/bash
# round 6: the library built with -DHLALA_DP_AGENT_RELEASE (the in-memory DP class releases at agent scope, as a device in threadgroup-split mode needs) -- run from a
# tree whose libhlala_gpu.so was built that way (make -C hla-la_amd/csrc EXTRA=-DHLALA_DP_AGENT_RELEASE): the build flag is reported, the tests that reach that class are green
set -u
: "${GRAFT_REPO_ROOT:?run through gpurun}"
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sys; sys.path.insert(0, "tests")
from conftest import load_package
P = load_package(); f = P.load_library().hlala_build_flags()
print("hlala_build_flags() =", f, "(agent-scope release: %s)" % bool(f & P.BUILD_AGENT_RELEASE))
assert f & P.BUILD_AGENT_RELEASE
PY
timeout 1800 python -m pytest tests/test_graph_m.py tests/test_gpu_align.py tests/test_full_scale.py -x -q -m gpu 2>&1 | tail -4
