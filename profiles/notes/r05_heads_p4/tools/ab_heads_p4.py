"""K-HEADS alone: the 8-wave kernel against the pipelined 4-wave kernel (dca_amd/csrc/heads_p4.inc) on one box, alternating,
over batch sizes; same inputs, counts from the byte store (what the engine hands the kernel).
    python tools/ab_heads_p4.py [flags=1] [G=20000] [B ...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dca_amd import build as _b
_P4 = os.path.join(ROOT, 'tools', '_dbg', 'libdcahip_DCA_EXP_HEADS_P4.so')      # the pipelined kernel lives in an experiment build only
if not os.path.exists(_P4):
    os.makedirs(os.path.dirname(_P4), exist_ok=True)
    _b.build_hip(defines=('DCA_EXP_HEADS_P4',), out=_P4)
_b.LIB = _P4
_b.needs_build = lambda: False
from dca_amd.ops import HipOps
from dca_amd import synth, compact

flags = int(sys.argv[1]) if len(sys.argv) > 1 else 1
G = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
Bs = [int(a) for a in sys.argv[3:]] or [4096, 2048, 1024, 512, 256, 281]
hL = 64
ops = HipOps()
dev = torch.device('cuda')
nh = 1 + (0 if flags & 2 else 1) + (1 if flags & 1 else 0)
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
for B in Bs:
    H = torch.relu(torch.randn(B, hL, generator=g)).to(dev)
    perm = torch.randperm(n, generator=g, dtype=torch.int32)[:B].to(dev)
    cur = torch.zeros(1, dtype=torch.int64, device=dev)
    ws = torch.zeros(ops.heads_fused_workspace_bytes(B, hL, G, Gp, flags) // 4, device=dev)
    inv_n = 1.0 / (B * G)
    res = {}
    for name, thr in (('8-wave', 1 << 30), ('p4', 5), ('8-wave', 1 << 30), ('p4', 5)):
        ops.heads_set_p4_min_tiles(thr)
        gW = torch.zeros(hL + 1, NH, device=dev); gth = torch.zeros(Gp, device=dev); dH = torch.zeros(B, hL, device=dev)
        loss = torch.zeros(1, device=dev)

        def fused():
            return ops.heads_fused(H, hL, Wh, NH, Wh[hL], Gp, tw if flags & 2 else None, Y, Gp, sf, perm, cur, B, hL, G,
                                   0.0, inv_n, flags, gW, NH, gth if flags & 2 else None, dH, hL, part, ws, tile_order=order,
                                   loss_out=loss, compact=cc)
        fused(); torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20):
            fused()
        e.record(); torch.cuda.synchronize()
        t = s.elapsed_time(e) / 20
        res.setdefault(name, []).append((t, loss.item(), gW.double().abs().sum().item(), dH.double().abs().sum().item()))
    a, b = res['8-wave'], res['p4']
    print('B=%5d flags=%d: 8-wave %.4f / %.4f ms   p4 %.4f / %.4f ms   ratio %.3f | loss %.8f vs %.8f  sum|gW| %.6e vs %.6e  sum|dH| %.6e vs %.6e'
          % (B, flags, a[0][0], a[1][0], b[0][0], b[1][0], min(x[0] for x in b) / min(x[0] for x in a),
             a[0][1], b[0][1], a[0][2], b[0][2], a[0][3], b[0][3]), flush=True)
