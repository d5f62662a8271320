"""The 64-unit first-layer weight gradient from the byte store: first kernel (form 0) against the ring kernel (form 2)
on one box, alternating, over batch sizes; plus their agreement.     python tools/ab_enc0_dw.py [B ...]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get('DCA_DW_LIB'):                    # an experiment build (tools/_dbg/libdcahip_<variant>.so) in place of the product
    from dca_amd import build as _b
    _b.LIB = os.path.join(ROOT, 'tools', '_dbg', os.environ['DCA_DW_LIB'])
    _b.needs_build = lambda: False
from dca_amd import synth, prep, compact
from dca_amd.ops import HipOps
ops = HipOps()
dev = torch.device('cuda')
Bs = [int(a) for a in sys.argv[1:]] or [4096, 2048, 1024, 512, 281]
G, h, n = 20000, 64, 68579
Y = synth.generate_counts(n, G, device=dev)
counts = prep.cell_counts(ops, Y, n, G)
sf = counts / counts.median()
X, norm = prep.transform(ops, Y, n, G, sf, True, True, return_norm=True)
del X
cc = compact.build(ops, Y, n, G).with_input(norm['fac'], norm['do_log'], norm['mean'], norm['std'], ops=ops)
cc.ensure_lut(ops)
cur = torch.zeros(1, dtype=torch.int64, device=dev)
for B in Bs:
    perm = torch.randperm(n, device=dev, dtype=torch.int32)[:B].contiguous()
    dZ = torch.randn(B, h, device=dev) * 1e-3
    wsd = torch.zeros(ops.enc0_dw_sparse_workspace_bytes(B, G, h) // 4 + 4, device=dev)
    res, out = {1: [], 2: []}, {}
    for form in (1, 2, 1, 2):             # the call's `form` argument: 1 = first kernel, 2 = ring kernel
        gW = torch.zeros(G + 1, h, device=dev)
        for _ in range(3):
            ops.enc0_dw_sparse(cc, perm, cur, 0, B, G, h, dZ, h, gW, h, wsd, form=form)
        torch.cuda.synchronize()
        s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(30):
            ops.enc0_dw_sparse(cc, perm, cur, 0, B, G, h, dZ, h, gW, h, wsd, form=form)
        e.record(); torch.cuda.synchronize()
        res[form].append(s.elapsed_time(e) / 30)
        out[form] = gW.clone()
    d = (out[1] - out[2]).abs().max().item() / out[1].abs().max().item()
    print('B=%5d: first %.4f / %.4f ms   ring %.4f / %.4f ms   ratio %.3f   max |diff| / max |gW| %.1e'
          % (B, res[1][0], res[1][1], res[2][0], res[2][1], min(res[2]) / min(res[1]), d), flush=True)
