cd $GRAFT_REPO_ROOT
make -C oracle 2>&1 | tail -1
for dbg in 1 0; do
for cfg in "1 5000 1 200" "3 8000 3 200" "2 8000 0 100" "4 3000 10 150"; do
echo "=== dbg=$dbg $cfg"
if [ $dbg = 1 ]; then export HLALA_DEBUG=1; else unset HLALA_DEBUG; fi
timeout 60 python tools/dbg_align.py $cfg 2>&1 | grep -E "PARITY|MISMATCH|HUNG|ms_project|Error|error" | cut -c1-500; echo "rc=$?"
done; done
timeout 300 python -m pytest tests/test_gpu_extend.py -x -q -m gpu 2>&1 | tail -5
