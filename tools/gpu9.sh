cd $GRAFT_REPO_ROOT
make -C oracle 2>&1 | tail -1
timeout 600 python -m pytest tests/test_typer.py tests/test_host_flatten.py -x -q -m gpu 2>&1 | tail -12
python - <<'PY'
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from tools import synth
from conftest import load_package
import oracle_binding as ob
P = load_package()
w = synth.make_world(seed=1, G=300, k=1); ctx = P.Context(w["graph"], w["contigs"])
loc = synth.make_locus(seed=9, n_clusters=3000, n_reads=400)
t=time.time(); LL, m = ctx.exon_loglik(loc); t1=time.time()-t
t=time.time(); out = ctx.pair_loglik(LL, m); t2=time.time()-t
t=time.time(); LL2, m2 = ctx.exon_loglik(loc); t1b=time.time()-t
t=time.time(); out = ctx.pair_loglik(LL, m); t2b=time.time()-t
print('C=3000 R=400: exon_loglik %.1f ms (2nd %.1f), pair_loglik %.1f ms (2nd %.1f) incl. transfers; %.2f G logAvg/s' % (t1*1e3, t1b*1e3, t2*1e3, t2b*1e3, 3000*3001/2*400/t2b/1e9))
t=time.time(); o = ob.pair_loglik(LL[:300], m[:300]); t3=time.time()-t
print('oracle pair_loglik C=300: %.1f ms -> %.3f G logAvg/s (1 core)' % (t3*1e3, 300*301/2*400/t3/1e9))
PY
