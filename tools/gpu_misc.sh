cd $GRAFT_REPO_ROOT
make -C oracle 2>&1 | tail -1
timeout 900 python -m pytest tests/test_call.py -x -q -m gpu 2>&1 | tail -15
python - <<'PY'
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from tools import synth
from conftest import load_package
import oracle_binding as ob
P = load_package()
w = synth.make_world(seed=1, G=300, k=1); ctx = P.Context(w["graph"], w["contigs"])
C = 3000; nP = C*(C+1)//2
rng = np.random.default_rng(1)
LL = -rng.random(nP) * 500; MA = rng.integers(0, 200, nP) / 2.0; MM = rng.integers(0, 100, nP).astype(float)
ctx.call_locus(LL, MA, MM)
t = time.time(); g = ctx.call_locus(LL, MA, MM); tg = time.time() - t
t = time.time(); e = ob.call_locus(LL, MA, MM); te = time.time() - t
print('C=3000 (4.5M pairs): hlala_call_locus %.1f ms incl. transfers, oracle (std::sort, 1 core) %.1f ms; same call: %s' % (tg*1e3, te*1e3, (g['first_cluster'], g['second_cluster']) == (e['first_cluster'], e['second_cluster'])))
PY
