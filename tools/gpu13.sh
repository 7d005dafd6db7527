cd $GRAFT_REPO_ROOT
make -C oracle 2>&1 | tail -1
timeout 600 python -m pytest tests/test_gpu_extend.py -x -q -m gpu -s 2>&1 | grep -E "k=|passed|failed|Error" | head
