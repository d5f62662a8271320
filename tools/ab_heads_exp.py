"""K-HEADS (8-wave kernel) timed with the product library and with experiment builds of dcahip_heads.hip (wrong results on
purpose: each removes one cost to price it), one subprocess per library, alternating.
    python tools/ab_heads_exp.py tools/_dbg/libdcahip_X.so ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, os
sys.path.insert(0, %r)
lib = %r
if lib:
    from dca_amd import build as b; b.LIB = lib; b.needs_build = lambda: False
import torch
from dca_amd.ops import HipOps
from dca_amd import synth, compact
ops = HipOps(); dev = torch.device('cuda')
B, G, hL, flags = 4096, 20000, 64, 1
Gp = G; NH = 3 * Gp; n = 68579
Y = synth.generate_counts(n, G, device=dev); X, sf = synth.normalize_on_device(Y, G, None); del X
cc = compact.build(ops, Y, n, G)
g = torch.Generator(device='cpu'); g.manual_seed(0)
lim = (6.0 / (hL + G)) ** 0.5
Wh = ((torch.rand(hL + 1, NH, generator=g) * 2 - 1) * lim).to(dev)
ntg = (G + 31) // 32
nz = torch.zeros(ntg * 32, device=dev); nz[:G] = (Y[:8192, :G] != 0).sum(dim=0)
o = torch.argsort(nz.view(ntg, 32).sum(dim=1), descending=True).to(torch.int32)
order = torch.cat([o, torch.arange(ntg, ops.heads_tile_order_len(G), dtype=torch.int32, device=dev)]).contiguous()
part = torch.zeros(ops.max_partials, dtype=torch.float64, device=dev)
H = torch.relu(torch.randn(B, hL, generator=g)).to(dev)
perm = torch.randperm(n, generator=g, dtype=torch.int32)[:B].to(dev)
cur = torch.zeros(1, dtype=torch.int64, device=dev)
ws = torch.zeros(ops.heads_fused_workspace_bytes(B, hL, G, Gp, flags) // 4, device=dev)
gW = torch.zeros(hL + 1, NH, device=dev); dH = torch.zeros(B, hL, device=dev); loss = torch.zeros(1, device=dev)
def fused():
    return ops.heads_fused(H, hL, Wh, NH, Wh[hL], Gp, None, Y, Gp, sf, perm, cur, B, hL, G, 0.0, 1.0 / (B * G), flags, gW, NH, None,
                           dH, hL, part, ws, tile_order=order, loss_out=loss, compact=cc)
for _ in range(3): fused()
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(30): fused()
e.record(); torch.cuda.synchronize()
print('RESULT %%.4f' %% (s.elapsed_time(e) / 30))
'''
libs = [''] + [os.path.abspath(a) for a in sys.argv[1:]]
for r in range(2):
    for lib in libs:
        out = subprocess.run([sys.executable, '-c', CODE % (ROOT, lib)], capture_output=True, text=True, cwd=ROOT, stdin=subprocess.DEVNULL)
        line = [l for l in out.stdout.splitlines() if l.startswith('RESULT')]
        print('%-40s %s' % (os.path.basename(lib) or 'product (8-wave kernel)', line[0] if line else 'FAILED ' + out.stderr[-300:]), flush=True)
