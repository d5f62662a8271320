"""Only the first-layer weight gradient from the byte store (for rocprofv3 runs): B x 20 000 x 64, ITERS launches."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get('DCA_AMD_LIB'):
    from dca_amd import build as _b
    _b.LIB = os.environ['DCA_AMD_LIB']
    _b.needs_build = lambda: False
from dca_amd import synth, prep, compact
from dca_amd.ops import HipOps
ops = HipOps()
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
G, h, n = 20000, 64, 20000
Y = synth.generate_counts(n, G, device=dev)
counts = prep.cell_counts(ops, Y, n, G)
sf = counts / counts.median()
X, norm = prep.transform(ops, Y, n, G, sf, True, True, return_norm=True)
cc = compact.build(ops, Y, n, G).with_input(norm['fac'], norm['do_log'], norm['mean'], norm['std'], ops=ops)
cur = torch.zeros(1, dtype=torch.int64, device=dev)
perm = torch.randperm(n, device=dev, dtype=torch.int32)[:B].contiguous()
dZ = torch.randn(B, h, device=dev) * 1e-3
gW = torch.zeros(G + 1, h, device=dev)
wsd = torch.zeros(ops.enc0_dw_sparse_workspace_bytes(B, G, h) // 4 + 4, device=dev)
for _ in range(iters):
    ops.enc0_dw_sparse(cc, perm, cur, 0, B, G, h, dZ, h, gW, h, wsd)
torch.cuda.synchronize()
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    ops.enc0_dw_sparse(cc, perm, cur, 0, B, G, h, dZ, h, gW, h, wsd)
e.record(); torch.cuda.synchronize()
print('B=%d dW from the byte store: %.4f ms' % (B, s.elapsed_time(e) / iters))
