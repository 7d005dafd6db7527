cd $GRAFT_REPO_ROOT
make -C oracle 2>&1 | tail -1
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
timeout 60 python tools/dbg_align.py 2 8000 0 100 2>&1 | grep -E "PARITY|MISMATCH|HUNG|ms_project" | cut -c1-400
timeout 600 python bench.py --pairs 131072 --levels 1000000 --steps 2 --warmup 1 --cpu-pairs 1024 2>&1 | tail -3
