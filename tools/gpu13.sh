cd $GRAFT_REPO_ROOT
timeout 300 python tools/dbg_timing.py 262144 5000000 2>&1 | tail -5
