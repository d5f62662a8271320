"""Debug: same steps through the fused and the separate heads kernels; report the first divergence."""
import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import make_problem, make_engine
from dca_amd.ops import HipOps

ops = HipOps()
ae = sys.argv[1] if len(sys.argv) > 1 else 'zinb'
n, G, hs = 2000, 1000, (64, 32, 64)
X, Y, sf, p = make_problem(n, G, hs, ae, True, seed=21)
engs = []
for fused in (True, False):
    e = make_engine(ops, ae, G, hs, True, 0.0, p, X, Y, sf)
    e.use_fused = fused
    e.reserve(200)
    e.set_lr(1e-3)
    e.hist = torch.zeros(64, dtype=torch.float32, device=e.dev)
    engs.append(e)
rng = np.random.RandomState(0)
for step, B in enumerate([32, 32, 8, 32, 200, 32]):
    rows = rng.permutation(1800)[:B].astype(np.int32)
    res = []
    for e in engs:
        e.perm = torch.as_tensor(rows).to(e.dev)
        e.cursor.zero_(); e.acc.zero_()
        e.train_step(B, rows_per_slot=B)
        torch.cuda.synchronize()
        res.append((float(e.hist[0].item()), e.get_grads(), e.get_params()))
    print('step', step, 'B', B, 'loss', res[0][0], res[1][0])
    for name in res[0][1]:
        a, b = res[0][1][name], res[1][1][name]
        d = np.abs(a - b).max(); s = np.abs(b).max()
        flag = ' <<<<' if d > 1e-4 * max(s, 1e-30) + 1e-12 else ''
        print('   grad %-10s maxdiff %.3e scale %.3e%s' % (name, d, s, flag))
    for name in res[0][2]:
        a, b = res[0][2][name], res[1][2][name]
        d = np.abs(a - b).max(); s = np.abs(b).max()
        if d > 1e-5 * max(s, 1e-30):
            print('   param %-10s maxdiff %.3e scale %.3e' % (name, d, s))
for e in engs:
    e.acc.zero_()
    e.eval_loss_sum(1800, 2000, 1.0 / (200 * G))
    torch.cuda.synchronize()
    print('val', e.use_fused, e.acc.cpu().numpy())
    e.acc.zero_()
    e.eval_loss_sum(1800, 2000, 1.0 / (200 * G), chunk=32)
    torch.cuda.synchronize()
    print('val chunk32', e.use_fused, e.acc.cpu().numpy())
