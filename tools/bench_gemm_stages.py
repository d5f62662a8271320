"""enc0 GEMMs (C3, B = 4096) over split_k; run with DCA_GEMM_STAGES=1 / 2."""
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
Z = torch.zeros(B, h, device=dev); dZ = torch.randn(B, h, device=dev); gW = torch.zeros(G + 1, h, device=dev)
cur = torch.zeros(1, dtype=torch.int64, device=dev)
ws = torch.zeros(256 * 1024 * 1024 // 4, device=dev)
perm = torch.randperm(n, device=dev, dtype=torch.int32)[:B].contiguous()


def timeit(fn, it=30):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it


print('big', os.environ.get('DCA_GEMM_BIG', '0'))
print('fwd ', ' '.join('S=%d: %.3f' % (sk, timeit(lambda: ops.sgemm(0, 0, B, h, G, X, G, W0, h, Z, h, bias=W0[G], perm=perm, cursor=cur, split_k=sk, ws=ws)))
                       for sk in (0, 0, 16, 24, 32, 48, 64)))
print('dW  ', ' '.join('S=%d: %.3f' % (sk, timeit(lambda: ops.sgemm(1, 0, G, h, B, X, G, dZ, h, gW, h, perm=perm, cursor=cur, colsum_row=True, split_k=sk, ws=ws)))
                       for sk in (0, 0, 3, 4, 6, 8, 12)))
