import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import make_problem, make_engine, oracle_net
from oracle import net_np as N
from dca_amd.ops import HipOps
from dca_amd.train import fit_engine
ops = HipOps()
ae = sys.argv[1] if len(sys.argv) > 1 else 'zinb'
n, G, hs = 2000, 1000, (64, 32, 64)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 21
X, Y, sf, p = make_problem(n, G, hs, ae, True, seed=seed)
ref = oracle_net(ae, p, hs, True)
rh = N.fit(ref, X.astype(np.float64), Y.astype(np.float64), sf.astype(np.float64), epochs=2, batch_size=32, shuffle_rng=np.random.RandomState(5))
print('oracle', rh['loss'], rh['val_loss'])
for fused in (True, False):
    for graph in (True,):
        eng = make_engine(ops, ae, G, hs, True, 0.0, p, X, Y, sf)
        eng.use_fused = fused
        h = fit_engine(eng, 1800, 200, 1800, 200, 0, epochs=2, batch_size=32, shuffle_rng=np.random.RandomState(5), use_graph=graph)
        print('fused', fused, 'graph', graph, h.history['loss'], h.history['val_loss'])
        pp = eng.get_params()
        print('    rel', np.abs(np.array(h.history['loss'])/np.array(rh['loss'])-1).max(), np.abs(np.array(h.history['val_loss'])/np.array(rh['val_loss'])-1).max())
        print('    theta_w', pp['theta_w'][:4], 'mm0', pp['mm0'][:3], 'mv0', pp['mv0'][:3])
