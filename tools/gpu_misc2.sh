cd $GRAFT_REPO_ROOT
timeout 900 python tools/long_timing.py 5000 2000000 2>&1 | tail -4
