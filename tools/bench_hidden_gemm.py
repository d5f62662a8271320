"""The twelve products of the hidden stack of BASELINE configs[4]'s network (512-256-128-256-512, batch 2048) through
dcahip_sgemm, automatic plan and explicit split-K factors (the split's reduce launch is part of the call).
  python tools/bench_hidden_gemm.py [iters]          (DCA_AMD_LIB=<path> times an ablation build)
"""
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
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
B = 2048
hs = [512, 256, 128, 256, 512]
shapes = []
for i in range(1, len(hs)):
    shapes.append(('fwd%d' % i, 0, 0, B, hs[i], hs[i - 1]))
for i in range(1, len(hs)):
    shapes.append(('dW%d ' % i, 1, 0, hs[i - 1], hs[i], B))
for i in range(1, len(hs)):
    shapes.append(('dH%d ' % i, 0, 1, B, hs[i - 1], hs[i]))


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


tot = {}
for name, ta, tb, M, N, K in shapes:
    ra, ca = (K, M) if ta else (M, K)
    rb, cb = (N, K) if tb else (K, N)
    A = torch.randn(ra, ca, device=dev); Bm = torch.randn(rb, cb, device=dev)
    C = torch.zeros(M + 1, N, device=dev)
    ws = torch.empty(64 * (M + 1) * N, device=dev)
    ref = (A.t() if ta else A).double() @ (Bm.t() if tb else Bm).double()
    out = []
    for sk in (0, 1, 2, 4, 8):
        t = timeit(lambda: ops.sgemm(ta, tb, M, N, K, A, ca, Bm, cb, C, N, split_k=sk, ws=ws))
        err = float((C[:M].double() - ref).abs().max() / ref.abs().max())
        assert err < 1e-5, (name, sk, err)
        out.append('%s %5.1f' % ('auto' if sk == 0 else 'S=%d' % sk, t))
        tot[sk] = tot.get(sk, 0.0) + t
    print('%s %4d x %4d x %4d   us: %s' % (name, M, N, K, '   '.join(out)), flush=True)
print('sum of the twelve: ' + '   '.join('%s %.0f us' % ('auto' if k == 0 else 'S=%d' % k, v) for k, v in tot.items()))
