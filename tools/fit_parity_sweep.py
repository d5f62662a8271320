"""Per-epoch loss of the GPU fit loop against the fp64 oracle over several problem seeds (BASELINE config 2 shape).
  python tools/fit_parity_sweep.py nb,zinb-conddisp 23,24,25,26,27
"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import make_problem, oracle_net, make_engine
from oracle import net_np as N
from dca_amd.ops import HipOps
from dca_amd.train import fit_engine
ops = HipOps()
n, G, hs = 2000, 1000, (64, 32, 64)
for ae in sys.argv[1].split(','):
    for seed in [int(x) for x in sys.argv[2].split(',')]:
        X, Y, sf, p = make_problem(n, G, hs, ae, True, seed=seed)
        ref = oracle_net(ae, p, hs, True)
        rh = N.fit(ref, X.astype(np.float64), Y.astype(np.float64), sf.astype(np.float64), epochs=2, batch_size=32,
                   shuffle_rng=np.random.RandomState(5))
        eng = make_engine(ops, ae, G, hs, True, 0.0, p, X, Y, sf)
        n_train = int(n * 0.9)
        h = fit_engine(eng, n_train, n - n_train, n_train, n - n_train, 0, epochs=2, batch_size=32,
                       shuffle_rng=np.random.RandomState(5), use_graph=True)
        rl = np.abs(np.array(h.history['loss']) / np.array(rh['loss']) - 1)
        rv = np.abs(np.array(h.history['val_loss']) / np.array(rh['val_loss']) - 1)
        print(ae, 'seed', seed, 'loss rel', rl, 'val rel', rv, flush=True)
