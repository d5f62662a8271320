"""Latency of the small-batch kernels of the hidden stack, one by one: each is captured 50x into a hipGraph (dependent
launches, as in the training step) and replayed; time per launch = replay time / 50.
  python tools/bench_small_layers.py [B]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dca_amd.ops import HipOps

ops = HipOps()
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
f32 = dict(dtype=torch.float32, device=dev)
hs = (64, 32, 64)
Z = [torch.randn(B, h, **f32) for h in hs]
XH = [torch.zeros(B, h, **f32) for h in hs]
H = [torch.zeros(B, h, **f32) for h in hs]
dH = [torch.randn(B, h, **f32) for h in hs]
dZ = [torch.zeros(B, h, **f32) for h in hs]
W = [None] + [torch.randn(hs[i - 1], hs[i], **f32) * 0.1 for i in (1, 2)]
gW = [None] + [torch.zeros(hs[i - 1] + 1, hs[i], **f32) for i in (1, 2)]
bias = [torch.zeros(h, **f32) for h in hs]
beta = [torch.zeros(h, **f32) for h in hs]
mm = [torch.zeros(h, **f32) for h in hs]
mv = [torch.ones(h, **f32) for h in hs]
inv = [torch.ones(h, **f32) for h in hs]
cnt = torch.zeros(1, dtype=torch.int64, device=dev)


def entry(j):
    d = dict(H=hs[j], beta=beta[j], moving_mean=mm[j], moving_var=mv[j], Z=Z[j], ldz=hs[j], xhat=XH[j], ldx=hs[j],
             Hout=H[j], ldh=hs[j], inv_std=inv[j])
    if j > 0:
        d.update(W=W[j], ldw=hs[j], bias=bias[j], K=hs[j - 1])
    return d


def timed(name, fn, reps=50, replays=20):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(reps):
                fn()
    torch.cuda.current_stream().wait_stream(st)
    g.replay(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(replays):
        g.replay()
    e.record(); torch.cuda.synchronize()
    print('%-58s %6.2f us per launch' % (name, s.elapsed_time(e) * 1e3 / (reps * replays)))


timed('counter_add (launch floor)', lambda: ops.counter_add(cnt, 1))
timed('bn_relu_train_small, layer 0', lambda: ops.bn_relu_train_small(Z[0], 64, B, 64, beta[0], mm[0], mv[0], 0.99, 1e-3, 1,
                                                                      H[0], 64, XH[0], 64, inv[0]))
timed('dense_bn_small, layer 1 (64 -> 32)', lambda: ops.dense_bn_small(H[0], 64, W[1], 32, bias[1], B, 64, 32, True, beta[1], mm[1],
                                                                        mv[1], 0.99, 1e-3, 1, Z[1], 32, XH[1], 32, H[1], 32, inv[1]))
for n in (1, 2, 3):
    timed('hidden_small_chain, %d entries' % n, lambda n=n: ops.hidden_small_chain([entry(j) for j in range(n)], None, 0, B, True,
                                                                                 0.99, 1e-3, 1))
timed('dense_bn_bwd_small, layer 2 (32 -> 64)', lambda: ops.dense_bn_bwd_small(dH[2], 64, H[2], 64, XH[2], 64, inv[2], H[1], 32, W[2], 64,
                                                                            B, 32, 64, True, float(B), 1, gW[2], 64, beta[2], dH[1], 32))
timed('bn_bwd_small, layer 0', lambda: ops.bn_bwd_small(dH[0], 64, H[0], 64, XH[0], 64, inv[0], float(B), B, 64, dZ[0], 64, beta[0]))
