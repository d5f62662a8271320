import sys, os
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import make_problem, make_engine
from dca_amd.ops import HipOps
ops = HipOps()
ae = sys.argv[1] if len(sys.argv) > 1 else 'zinb'
SYNC = len(sys.argv) > 2 and sys.argv[2] == 'sync'
REV = len(sys.argv) > 2 and sys.argv[2] == 'rev'
n, G, hs = 2000, 1000, (64, 32, 64)
X, Y, sf, p = make_problem(n, G, hs, ae, True, seed=21)
engs = []
for fused in (True, False):
    e = make_engine(ops, ae, G, hs, True, 0.0, p, X, Y, sf)
    e.use_fused = fused
    e.reserve(200); e.set_lr(1e-3)
    e.hist = torch.zeros(64, dtype=torch.float32, device=e.dev)
    engs.append(e)
idx = np.arange(1800); np.random.RandomState(5).shuffle(idx)
for e in engs:
    e.perm = torch.as_tensor(idx.astype(np.int32)).to(e.dev)
    e.cursor.zero_()
ef, eu = engs
lay = ef.lay
for t in range(57):
    B = min(32, 1800 - 32 * t)
    # identical state going into the step
    if SYNC:
        ef.w.copy_(eu.w); ef.ms.copy_(eu.ms)
        for i in range(3):
            ef.mm[i].copy_(eu.mm[i]); ef.mv[i].copy_(eu.mv[i])
    if REV:
        eu.w.copy_(ef.w); eu.ms.copy_(ef.ms)
        for i in range(3):
            eu.mm[i].copy_(ef.mm[i]); eu.mv[i].copy_(ef.mv[i])
    wd = (ef.w - eu.w).abs().max().item(); msd = (ef.ms - eu.ms).abs().max().item()
    for e in engs:
        e.train_step(B, rows_per_slot=32)
    torch.cuda.synchronize()
    gf, gu = ef.g.cpu().numpy(), eu.g.cpu().numpy()
    d = np.abs(gf - gu)
    rel = d.max() / np.abs(gu).max()
    print('t', t, 'wdiff_in %.3e msdiff_in %.3e gdiff %.3e loss %.8f %.8f' % (wd, msd, d.max(), gf[lay.P], gu[lay.P]))
    if rel > 1e-5 or t < 2:
        print('step', t, 'B', B, 'max abs diff', d.max(), 'gmax', np.abs(gu).max(), 'at', int(d.argmax()))
        for name, (off, shape) in lay.seg.items():
            sz = int(np.prod(shape))
            dd = d[off:off + sz]
            if dd.max() > 1e-6 * np.abs(gu).max():
                ii = np.unravel_index(int(dd.argmax()), shape)
                print('    %-8s maxdiff %.3e at %s  fused %.6e unfused %.6e  nbad %d' % (
                    name, dd.max(), ii, gf[off:off+sz].reshape(shape)[ii], gu[off:off+sz].reshape(shape)[ii],
                    int((dd > 1e-6 * np.abs(gu).max()).sum())))
        rows = idx[32 * t:32 * t + B]
        if rel > 1e-5:
            yb = Y[rows]
            print('    y max', yb.max(), 'n>16', int((yb > 16).sum()), 'nonint', int((yb != np.floor(yb)).sum()))
