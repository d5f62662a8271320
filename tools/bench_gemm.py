"""Sweeps split_k of dcahip_sgemm on the step's skinny GEMM shapes (C3, B=4096)."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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
for sk in (0, 6, 12, 16, 24, 32, 48, 64, 96):
    if ops.sgemm_workspace_bytes(0, 0, B, h, G, False, sk) > ws.numel() * 4: continue
    t = timeit(lambda: ops.sgemm(0, 0, B, h, G, X, G, W0, h, Z, h, bias=W0[G], perm=perm, cursor=cur, split_k=sk, ws=ws))
    print('enc0 fwd  NN M=%d N=%d K=%d split_k=%3d: %.3f ms %.1f TF/s' % (B, h, G, sk, t, fl / t / 1e9))
for sk in (0, 1, 2, 3, 4, 6, 8, 12, 16):
    if ops.sgemm_workspace_bytes(1, 0, G, h, B, True, sk) > ws.numel() * 4: continue
    t = timeit(lambda: ops.sgemm(1, 0, G, h, B, X, G, dZ, h, gW, h, perm=perm, cursor=cur, colsum_row=True, split_k=sk, ws=ws))
    print('enc0 dW   TN M=%d N=%d K=%d split_k=%3d: %.3f ms %.1f TF/s' % (G, h, B, sk, t, fl / t / 1e9))
