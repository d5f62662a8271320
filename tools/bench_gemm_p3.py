"""dcahip_gemm_p3 (products from pre-split bf16 planes) next to dcahip_sgemm on the shapes of BASELINE configs[4]'s
network (512-256-128-256-512 on 25 000 genes, batch 2048) and of configs[2]'s first layer.
  python tools/bench_gemm_p3.py [iters]
"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get('DCA_AMD_LIB'):                      # an ablation build of the library
    from dca_amd import build as _b
    _b.LIB = os.environ['DCA_AMD_LIB']
    _b.needs_build = lambda: False
from dca_amd.ops import HipOps

ops = HipOps()
dev = torch.device('cuda')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10


def timeit(fn):
    fn(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def r8(x):
    return (x + 7) // 8 * 8


SHAPES = [  # name, ta, tb, M, N, K
    ('C5 heads fwd   H Wh          ', 0, 0, 2048, 75000, 512),
    ('C5 heads dH    D Wh^T        ', 0, 1, 2048, 512, 75000),
    ('C5 heads dW    H^T D         ', 1, 0, 512, 75000, 2048),
    ('C5 enc0 fwd    X W0          ', 0, 0, 2048, 512, 25000),
    ('C5 enc0 dW     X^T dZ        ', 1, 0, 25000, 512, 2048),
    ('C3 enc0 fwd    X W0          ', 0, 0, 4096, 64, 20000),
    ('C3 enc0 dW     X^T dZ        ', 1, 0, 20000, 64, 4096),
]
g = torch.Generator(device='cpu'); g.manual_seed(0)
for name, ta, tb, M, N, K in SHAPES:
    K = (K + 15) // 16 * 16                              # (zero padding: what the engine allocates)
    ra, ca = (K, M) if ta else (M, K)
    rb, cb = (N, K) if tb else (K, N)
    A = torch.randn(ra, r8(ca), device=dev); B = torch.randn(rb, r8(cb), device=dev)
    C = torch.zeros(M + 1, r8(N), device=dev)
    pa = ops.planes_alloc(ra, ca, dev); pb = ops.planes_alloc(rb, cb, dev)
    ops.split_planes(A, A.shape[1], ra, ca, pa); ops.split_planes(B, B.shape[1], rb, cb, pb)
    ws = torch.empty(max(ops.gemm_p3_workspace_bytes(M, N, K, False, 0), ops.sgemm_workspace_bytes(ta, tb, M, N, K, False, 0), 4) // 4, device=dev)
    t3 = timeit(lambda: ops.gemm_p3(ta, tb, M, N, K, pa, pb, C, C.shape[1], ws=ws))
    C3 = C[:M, :N].clone()
    t1 = timeit(lambda: ops.sgemm(ta, tb, M, N, K, A, A.shape[1], B, B.shape[1], C, C.shape[1], ws=ws))
    err = float((C3 - C[:M, :N]).abs().max() / C[:M, :N].abs().max())
    tsa = timeit(lambda: ops.split_planes(A, A.shape[1], ra, ca, pa))
    tsb = timeit(lambda: ops.split_planes(B, B.shape[1], rb, cb, pb))
    fl = 2.0 * M * N * K
    print('%s planes %.3f ms (%.0f TF/s)   sgemm %.3f ms (%.0f TF/s)   split A %.3f B %.3f ms   max diff %.1e'
          % (name, t3, fl / t3 / 1e9, t1, fl / t1 / 1e9, tsa, tsb, err), flush=True)
    del A, B, C, pa, pb, ws
