import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo')); sys.path.insert(0, os.path.join(sys.path[0], 'tests'))
from oracle.torch_ref import TorchAE
from oracle import net_np as N
G, hs = 20000, (64, 32, 64)
p = N.init_params('zinb-conddisp', G, hs, seed=0, dtype=np.float32)
rng = np.random.RandomState(0)
for B in (32, 1024):
    X = torch.tensor(rng.normal(size=(B, G)).astype(np.float32)); Y = torch.tensor(rng.poisson(0.1, (B, G)).astype(np.float32)); S = torch.ones(B)
    for th in (8, 16, 32, 64, 128, 256):
        torch.set_num_threads(th)
        net = TorchAE('zinb-conddisp', p, hs, True)
        net.train_step(X, Y, S)
        t0 = time.perf_counter(); k = 0
        while time.perf_counter() - t0 < 3.0 and k < 50:
            net.train_step(X, Y, S); k += 1
        el = time.perf_counter() - t0
        print('B', B, 'threads', th, 'ms/step', round(1e3 * el / k, 1), 'cells/s', round(B * k / el), flush=True)
