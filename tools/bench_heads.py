"""Times dcahip_heads_fused alone (C3 shape by default) next to the separate kernels it replaces.
  python tools/bench_heads.py [B] [G] [hL] [flags] [iters]          (COMPACT=1: counts read from the byte store)
"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get('DCA_AMD_LIB'):
    from dca_amd import build as _b
    _b.LIB = os.environ['DCA_AMD_LIB']
    _b.needs_build = lambda: False
from dca_amd.ops import HipOps
from dca_amd import synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
hL = int(sys.argv[3]) if len(sys.argv) > 3 else 64
flags = int(sys.argv[4]) if len(sys.argv) > 4 else 1
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 20
only_fused = os.environ.get('ONLY_FUSED', '0') == '1'
ops = HipOps()
dev = torch.device('cuda')
nh = 1 + (0 if flags & 2 else 1) + (1 if flags & 1 else 0)
Gp = (G + 3) // 4 * 4
NH = nh * Gp
n = B + 64
Y = synth.generate_counts(n, G, device=dev)
X, sf = synth.normalize_on_device(Y, G, None)
del X
g = torch.Generator(device='cpu'); g.manual_seed(0)
H = torch.relu(torch.randn(B, hL, generator=g)).to(dev)
lim = (6.0 / (hL + G)) ** 0.5
Wh = ((torch.rand(hL + 1, NH, generator=g) * 2 - 1) * lim).to(dev)
Wh[hL].zero_()
tw = torch.zeros(Gp, device=dev)
perm = torch.randperm(n, generator=g, dtype=torch.int32)[:B].to(dev)
cur = torch.zeros(1, dtype=torch.int64, device=dev)
gW = torch.zeros(hL + 1, NH, device=dev); gth = torch.zeros(Gp, device=dev)
dH = torch.zeros(B, hL, device=dev)
part = torch.zeros(ops.max_partials, dtype=torch.float64, device=dev)
ws = torch.zeros(ops.heads_fused_workspace_bytes(B, hL, G, Gp, flags) // 4, device=dev)
inv_n = 1.0 / (B * G)

order = None
if os.environ.get('TILE_ORDER', '1') != '0':          # pair gene tiles of similar non-zero load (engine._set_tile_order)
    ntg = (G + 31) // 32
    nz = torch.zeros(ntg * 32, device=dev)
    nz[:G] = (Y[:, :G] != 0).sum(dim=0)
    o = torch.argsort(nz.view(ntg, 32).sum(dim=1), descending=True).to(torch.int32)
    order = torch.cat([o, torch.arange(ntg, ops.heads_tile_order_len(G), dtype=torch.int32, device=dev)]).contiguous()


cc = None
if os.environ.get('COMPACT', '0') == '1':             # the counts as bytes (what the engine hands K-HEADS on count data)
    from dca_amd import compact
    cc = compact.build(ops, Y, n, G)
    assert cc is not None


def fused():
    return ops.heads_fused(H, hL, Wh, NH, Wh[hL], Gp, tw if flags & 2 else None, Y, Gp, sf, perm, cur, B, hL, G,
                           0.0, inv_n, flags, gW, NH, gth if flags & 2 else None, dH, hL, part, ws, tile_order=order,
                           **({'compact': cc} if cc is not None else {}))

def timeit(fn, n):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n

t = timeit(fused, iters)
fl = 6.0 * B * hL * NH
print('heads_fused B=%d G=%d hL=%d flags=%d: %.3f ms  (%.1f TFLOP/s of MFMA work, %.2f M cells/s)' % (B, G, hL, flags, t, fl / t / 1e9, B / t / 1e3))
loss = torch.zeros(1, device=dev); ops.loss_finalize(part, fused(), inv_n, loss); torch.cuda.synchronize()
print('loss', loss.item())
