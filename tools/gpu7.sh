cd $GRAFT_REPO_ROOT
export HLALA_DEBUG=1
timeout 300 python tools/dbg_timing.py 65536 1000000 2>&1 | tail -4
