cd $GRAFT_REPO_ROOT
timeout 300 python tools/pcie_timing.py 1048576 5000000 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_align.py -x -q -m gpu 2>&1 | tail -3
