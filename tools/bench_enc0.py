"""First layer at the benchmark shape (G = 20 000, 64 units): dense K-GEMM on the fp32 input against K-SPARSE on the
compact counts, forward and weight gradient, at several batch sizes.  Prints one line per (batch, kernel)."""
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
# (the four-wave shape of the matrix-pipe forward and the non-zero-only forward are experiment builds: DCA_DW_LIB = a library built
# with -DDCA_EXP_FWD_FORM2 / -DDCA_EXP_ENC0_SPARSE_FWD, dca_amd.build.build_hip(defines=..., out=...))
dev = torch.device('cuda')
G = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
h = int(sys.argv[2]) if len(sys.argv) > 2 else 64
n = 20000
Y = synth.generate_counts(n, G, device=dev)
counts = prep.cell_counts(ops, Y, n, G)
sf = counts / counts.median()
X, norm = prep.transform(ops, Y, n, G, sf, True, True, return_norm=True)
cc = compact.build(ops, Y, n, G).with_input(norm['fac'], norm['do_log'], norm['mean'], norm['std'], ops=ops)
print('nonzero fraction %.4f  max count %d  escapes %s' % ((Y[:, :G] != 0).float().mean().item(), int(Y.max().item()),
                                                            0 if cc.ovf_col is None else cc.ovf_col.numel()))
W0 = torch.randn(G + 1, h, device=dev) * 0.01
W0T = torch.zeros(h, Y.shape[1], device=dev)
cur = torch.zeros(1, dtype=torch.int64, device=dev)
ws = torch.zeros(256 * 1024 * 1024 // 4, device=dev)
HAS_SP = ops.has('dcahip_enc0_fwd_sparse')          # the non-zero-only forward: experiment builds only
wsf = torch.zeros(ops.enc0_fwd_sparse_workspace_bytes(h) // 4 + 4, device=dev) if HAS_SP else None
ldx = X.shape[1]


def timeit(fn, it=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it


for B in ([int(b) for b in os.environ['BENCH_B'].split(',')] if os.environ.get('BENCH_B') else (32, 128, 512, 1024, 2048, 4096, 8192)):
    perm = torch.randperm(n, device=dev, dtype=torch.int32)[:B].contiguous()
    Z = torch.zeros(B, h, device=dev); Z2 = torch.zeros(B, h, device=dev)
    dZ = torch.randn(B, h, device=dev) * 1e-3
    gW = torch.zeros(G + 1, h, device=dev); gW2 = torch.zeros(G + 1, h, device=dev)
    wsd = torch.zeros(ops.enc0_dw_sparse_workspace_bytes(B, G, h) // 4 + 4, device=dev)

    def fwd_nn():
        ops.sgemm(0, 0, B, h, G, X, ldx, W0, h, Z, h, bias=W0[G], perm=perm, cursor=cur, ws=ws)

    def fwd_nt():
        ops.transpose(W0, h, G, h, W0T, ldx)
        ops.sgemm(0, 1, B, h, G, X, ldx, W0T, ldx, Z, h, bias=W0[G], perm=perm, cursor=cur, ws=ws)

    def fwd_sp():
        ops.enc0_fwd_sparse(cc, perm, cur, 0, B, G, h, W0, h, W0[G], Z2, h, wsf)

    nbl = ops.enc0_fwd_lut_workspace_bytes(B, G, h)
    wsl = torch.zeros(max(nbl, 16) // 4 + 4, device=dev)
    Z3 = torch.zeros(B, h, device=dev)

    def fwd_lut():
        ops.enc0_fwd_lut(cc, perm, cur, 0, B, G, h, W0, h, W0[G], Z3, h, wsl)

    def dw_tn():
        ops.sgemm(1, 0, G, h, B, X, ldx, dZ, h, gW, h, perm=perm, cursor=cur, colsum_row=True, ws=ws)

    def dw_sp():
        ops.enc0_dw_sparse(cc, perm, cur, 0, B, G, h, dZ, h, gW2, h, wsd)

    t = {'fwd dense NN': timeit(fwd_nn), 'fwd dense NT+transpose': timeit(fwd_nt) if B >= 256 else float('nan'),
         'fwd sparse': timeit(fwd_sp) if HAS_SP else float('nan'), 'fwd lut': timeit(fwd_lut) if nbl else float('nan'), 'dW dense TN': timeit(dw_tn), 'dW sparse': timeit(dw_sp)}
    fwd_nn(); (fwd_sp() if HAS_SP else fwd_lut()); dw_tn(); dw_sp(); torch.cuda.synchronize()
    ez = (Z - Z2).abs().max().item() / Z.abs().max().item() if HAS_SP else 0.0
    if nbl:
        fwd_lut(); torch.cuda.synchronize()
        ez = max(ez, (Z - Z3).abs().max().item() / Z.abs().max().item())
    eg = (gW - gW2).abs().max().item() / gW.abs().max().item()
    print('B=%5d  ' % B + '  '.join('%s %.4f ms' % kv for kv in t.items()) + '   max diff fwd %.1e dW %.1e' % (ez, eg), flush=True)
