cd $GRAFT_REPO_ROOT
make -C oracle 2>&1 | tail -1
export PYTHONFAULTHANDLER=1
for t in "seed1" "seed2" "seed3" "seed4"; do
  echo "=== $t"
  timeout 150 python -m pytest "tests/test_gpu_extend.py" -x -q -m gpu -k "$t" 2>&1 | tail -15
  echo "rc=$?"
done
