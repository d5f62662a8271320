"""Sweeps split_k of dcahip_sgemm on the step's skinny GEMM shapes (C3, B=4096)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get('DCA_AMD_LIB'):
    from dca_amd import build as _b
    _b.LIB = os.environ['DCA_AMD_LIB']
    _b.needs_build = lambda: False
from dca_amd.ops import HipOps
ops = HipOps()
dev = torch.device('cuda')
B, G, h = 4096, 20000, 64
n = 61721
X = torch.randn(n, G, device=dev)
W0 = torch.randn(G + 1, h, device=dev) * 0.01
Z = torch.zeros(B, h, device=dev)
dZ = torch.randn(B, h, device=dev)
gW = torch.zeros(G + 1, h, device=dev)
perm = torch.randperm(n, device=dev, dtype=torch.int32)[:B].contiguous()
cur = torch.zeros(1, dtype=torch.int64, device=dev)
ws = torch.zeros(256 * 1024 * 1024 // 4, device=dev)

def timeit(fn, it=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it

fl = 2.0 * B * G * h
for sk in (0, 16, 32, 48):
    if ops.sgemm_workspace_bytes(0, 0, B, h, G, False, sk) > ws.numel() * 4: continue
    t = timeit(lambda: ops.sgemm(0, 0, B, h, G, X, G, W0, h, Z, h, bias=W0[G], perm=perm, cursor=cur, split_k=sk, ws=ws))
    print('enc0 fwd  NN M=%d N=%d K=%d split_k=%3d: %.3f ms %.1f TF/s' % (B, h, G, sk, t, fl / t / 1e9))
W0T = W0[:G].t().contiguous()
for sk in (0, 16, 32, 48):
    if ops.sgemm_workspace_bytes(0, 1, B, h, G, False, sk) > ws.numel() * 4: continue
    t = timeit(lambda: ops.sgemm(0, 1, B, h, G, X, G, W0T, G, Z, h, bias=W0[G], perm=perm, cursor=cur, split_k=sk, ws=ws))
    print('enc0 fwd  NT M=%d N=%d K=%d split_k=%3d: %.3f ms %.1f TF/s' % (B, h, G, sk, t, fl / t / 1e9))
for sk in (0, 4, 6, 8):
    if ops.sgemm_workspace_bytes(1, 0, G, h, B, True, sk) > ws.numel() * 4: continue
    t = timeit(lambda: ops.sgemm(1, 0, G, h, B, X, G, dZ, h, gW, h, perm=perm, cursor=cur, colsum_row=True, split_k=sk, ws=ws))
    print('enc0 dW   TN M=%d N=%d K=%d split_k=%3d: %.3f ms %.1f TF/s' % (G, h, B, sk, t, fl / t / 1e9))

# heads-shaped GEMMs of the wide network (C5: hL = 512, G = 25000, B = 2048) and of C3 (separate-kernel path)
for (Bh, hL, NHh) in ((2048, 512, 75000), (4096, 64, 60000)):
    Hh = torch.randn(Bh, hL, device=dev); Wh = torch.randn(hL + 1, NHh, device=dev) * 0.01
    A = torch.zeros(Bh, NHh, device=dev); D = torch.randn(Bh, NHh, device=dev) * 1e-3
    gWh = torch.zeros(hL + 1, NHh, device=dev); dHh = torch.zeros(Bh, hL, device=dev)
    flh = 2.0 * Bh * hL * NHh
    t = timeit(lambda: ops.sgemm(0, 0, Bh, NHh, hL, Hh, hL, Wh, NHh, A, NHh, bias=Wh[hL], ws=ws), 5)
    print('heads fwd NN M=%d N=%d K=%d: %.3f ms %.1f TF/s' % (Bh, NHh, hL, t, flh / t / 1e9))
    t = timeit(lambda: ops.sgemm(1, 0, hL, NHh, Bh, Hh, hL, D, NHh, gWh, NHh, colsum_row=True, ws=ws), 5)
    print('heads dW  TN M=%d N=%d K=%d: %.3f ms %.1f TF/s' % (hL, NHh, Bh, t, flh / t / 1e9))
    t = timeit(lambda: ops.sgemm(0, 1, Bh, hL, NHh, D, NHh, Wh, NHh, dHh, hL, ws=ws), 5)
    print('heads dH  NT M=%d N=%d K=%d: %.3f ms %.1f TF/s' % (Bh, hL, NHh, t, flh / t / 1e9))
    del Hh, Wh, A, D, gWh, dHh
