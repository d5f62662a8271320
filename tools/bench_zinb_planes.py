"""K-ZINB with plane output (dcahip_zinb_nll_planes) alone at BASELINE configs[4]'s decoder shape: batch 2048 x 25 000 genes,
the three likelihood layouts, ~7 % non-zero counts.  Prints ms per launch and the algorithmic rate (zinb-conddisp: 34 B per element, 3 pre-activations +
count in, 3 x 3 bf16 pieces out; constant dispersion: 2 + count in, 2 x 3 pieces + the fp32 dispersion gradient out).
  python tools/bench_zinb_planes.py [iters] [nonzero fraction]        (DCA_AMD_LIB=<path> times an ablation build)
"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if os.environ.get('DCA_AMD_LIB'):
    from dca_amd import build as _b
    _b.LIB = os.environ['DCA_AMD_LIB']
    _b.needs_build = lambda: False
from dca_amd import hip
from dca_amd.ops import HipOps

ops = HipOps()
dev = torch.device('cuda')
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
nzf = float(sys.argv[2]) if len(sys.argv) > 2 else 0.07
B, G = 2048, 25000
ld = (G + 7) // 8 * 8
g = torch.Generator(device=dev); g.manual_seed(0)
for name, flags in (('zinb-conddisp', hip.NLL_HAS_PI), ('zinb', hip.NLL_HAS_PI | hip.NLL_CONST_DISP), ('nb', hip.NLL_CONST_DISP)):
    has_pi, cdisp = bool(flags & hip.NLL_HAS_PI), bool(flags & hip.NLL_CONST_DISP)
    A = torch.randn(B, 3 * ld, device=dev, generator=g) * 0.5
    Y = torch.where(torch.rand(B, ld, device=dev, generator=g) < nzf, torch.randint(1, 9, (B, ld), device=dev, generator=g).float(),
                    torch.zeros((), device=dev))
    sf = torch.rand(B, device=dev, generator=g) + 0.5
    tw = torch.randn(ld, device=dev, generator=g) * 0.1
    P = ops.planes_alloc(B, 3 * ld, dev)
    Dth = torch.zeros(B, ld, device=dev)
    partials = torch.zeros(8192, dtype=torch.float64, device=dev)
    def run():
        return ops.zinb_nll_planes(A, None if cdisp else A[:, ld:], A[:, 2 * ld:] if has_pi else None, 3 * ld, tw if cdisp else None,
                                   Y, ld, sf, None, None, B, G, 0.0, 1.0 / B, flags, P, 0, 0 if cdisp else ld, 2 * ld if has_pi else 0,
                                   Dth if cdisp else None, ld, partials)
    run(); torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        run()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    heads = 1 + (0 if cdisp else 1) + (1 if has_pi else 0)
    byts = 4 * heads + 4 + 6 * heads + (4 if cdisp else 0)
    print('%-14s %.3f ms   %d B/element -> %.2f TB/s' % (name, ms, byts, B * G * byts / ms / 1e9), flush=True)
