"""enc0 GEMMs (C3, B = 4096): how much of their time is the row gather?  Random perm vs sorted perm vs contiguous
rows vs a small matrix that stays in the Infinity Cache (no HBM at all)."""
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
cur = torch.zeros(1, dtype=torch.int64, device=dev)
ws = torch.zeros(256 * 1024 * 1024 // 4, device=dev)
rp = torch.randperm(n, device=dev, dtype=torch.int32)[:B].contiguous()
perms = {'random': rp, 'sorted': torch.sort(rp)[0].contiguous(), 'contiguous': torch.arange(B, device=dev, dtype=torch.int32),
         'cache-resident (512 rows repeated)': (torch.arange(B, device=dev, dtype=torch.int32) % 512).contiguous()}


def timeit(fn, it=20):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it


for name, perm in perms.items():
    t1 = timeit(lambda: ops.sgemm(0, 0, B, h, G, X, G, W0, h, Z, h, bias=W0[G], perm=perm, cursor=cur, ws=ws))
    t2 = timeit(lambda: ops.sgemm(1, 0, G, h, B, X, G, dZ, h, gW, h, perm=perm, cursor=cur, colsum_row=True, ws=ws))
    print('%-36s fwd %.3f ms (%.2f TB/s of X)   dW %.3f ms (%.2f TB/s)' % (name, t1, B * G * 4 / t1 / 1e9, t2, B * G * 4 / t2 / 1e9))
