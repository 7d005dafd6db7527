cd $GRAFT_REPO_ROOT
HLALA_DEBUG=1 timeout 300 python tools/dbg_timing.py 262144 5000000 2>&1 | tail -3
