cd $GRAFT_REPO_ROOT
make -C oracle 2>&1 | tail -1
timeout 900 python -m pytest tests/test_unpaired.py -x -q -m gpu 2>&1 | grep -v "^  File" | tail -25
