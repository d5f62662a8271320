"""Per-phase cycle counters (s_memtime) of the pipelined K-HEADS kernel: -DDCA_HEADS_TIMING build of the library beside the product's.
    python tools/timing_heads_p4.py [B=4096] [G=20000] [flags=1]"""
import os, sys, ctypes, subprocess
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
so = os.path.join(ROOT, 'tools', '_dbg', 'libdcahip_timing_p4.so')     # the pipelined kernel lives in an experiment build only
from dca_amd import build as b
if not os.path.exists(so):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    b.build_hip(force=False, verbose=False, defines=('DCA_HEADS_TIMING', 'DCA_EXP_HEADS_P4'), out=so)
b.LIB = so
b.needs_build = lambda: False
from dca_amd import hip, synth, compact
from dca_amd.ops import HipOps
ops = HipOps()
L = hip.lib()
L.dcahip_heads_set_timing.argtypes = [ctypes.c_void_p]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
flags = int(sys.argv[3]) if len(sys.argv) > 3 else 1
hL = 64
dev = torch.device('cuda')
nh = 1 + (0 if flags & 2 else 1) + 1
Gp = (G + 3) // 4 * 4
NH = nh * Gp
n = 68579
Y = synth.generate_counts(n, G, device=dev)
X, sf = synth.normalize_on_device(Y, G, None)
del X
cc = compact.build(ops, Y, n, G)
g = torch.Generator(device='cpu'); g.manual_seed(0)
lim = (6.0 / (hL + G)) ** 0.5
Wh = ((torch.rand(hL + 1, NH, generator=g) * 2 - 1) * lim).to(dev)
tw = torch.zeros(Gp, device=dev)
ntg = (G + 31) // 32
nz = torch.zeros(ntg * 32, device=dev)
nz[:G] = (Y[:8192, :G] != 0).sum(dim=0)
o = torch.argsort(nz.view(ntg, 32).sum(dim=1), descending=True).to(torch.int32)
order = torch.cat([o, torch.arange(ntg, ops.heads_tile_order_len(G), dtype=torch.int32, device=dev)]).contiguous()
part = torch.zeros(ops.max_partials, dtype=torch.float64, device=dev)
H = torch.relu(torch.randn(B, hL, generator=g)).to(dev)
perm = torch.randperm(n, generator=g, dtype=torch.int32)[:B].to(dev)
cur = torch.zeros(1, dtype=torch.int64, device=dev)
ws = torch.zeros(ops.heads_fused_workspace_bytes(B, hL, G, Gp, flags) // 4, device=dev)
gW = torch.zeros(hL + 1, NH, device=dev); gth = torch.zeros(Gp, device=dev); dH = torch.zeros(B, hL, device=dev)
loss = torch.zeros(1, device=dev)
ops.heads_set_p4_min_tiles(5)


def fused():
    return ops.heads_fused(H, hL, Wh, NH, Wh[hL], Gp, tw if flags & 2 else None, Y, Gp, sf, perm, cur, B, hL, G,
                           0.0, 1.0 / (B * G), flags, gW, NH, gth if flags & 2 else None, dH, hL, part, ws, tile_order=order,
                           loss_out=loss, compact=cc)


fused(); torch.cuda.synchronize()
tim = torch.zeros(2048 * 4 * 10, dtype=torch.int64, device='cuda')
L.dcahip_heads_set_timing(tim.data_ptr())
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record(); fused(); e.record(); torch.cuda.synchronize()
ms = s.elapsed_time(e)
L.dcahip_heads_set_timing(None)
t = tim.cpu().numpy().reshape(-1, 10).astype(np.float64)
t = t[t.sum(1) > 0]
names = ['steady: iteration prologue (requests, accumulator init, first operands)', 'steady: region 0 (9 blocks)', 'steady: region 1',
         'steady: region 2', 'steady: region 3', 'steady: non-zero flushes + next group loads', 'steady: staging of F, bookkeeping',
         'fill / drain iterations (sequential stages)', 'item prologue (weights -> LDS)', 'item epilogue (dW tree, stores)']
tot = t.sum(1).mean()
nt = (B + 31) // 32
tiles_per_wave = nt * ntg / (len(t))
print('launch %.4f ms; %d waves; mean counted cycles per wave %.0f (%.0f per row tile x gene tile, %.1f such tiles per wave)'
      % (ms, len(t), tot, tot / tiles_per_wave, tiles_per_wave))
for i, nme in enumerate(names):
    print('  %-75s %12.0f  %5.1f%%   %8.0f per tile' % (nme, t[:, i].mean(), 100 * t[:, i].mean() / tot, t[:, i].mean() / tiles_per_wave))
