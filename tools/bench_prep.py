"""K-PREP at BASELINE configs[2] size: counts resident, time the four streaming passes."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dca_amd.ops import HipOps
from dca_amd import synth, prep
ops = HipOps()
n, G = 68579, 20000
dev = torch.device('cuda')
Y = synth.generate_counts(n, G, device=dev)
torch.cuda.synchronize()
def run():
    cc = prep.cell_counts(ops, Y, n, G)
    gc = prep.gene_counts(ops, Y, n, G)
    fac = cc / cc.median()
    X = prep.transform(ops, Y, n, G, fac, True, True)
    return X, cc, gc
run(); torch.cuda.synchronize()
t0 = time.perf_counter(); X, cc, gc = run(); torch.cuda.synchronize(); t = time.perf_counter() - t0
el = n * G
print('K-PREP %d x %d: %.2f ms (row sums + gene counts + log-normalise + scale); 24 B/element algorithmic -> %.0f GB/s' % (n, G, t * 1e3, 24.0 * el / t / 1e9))
# cross-check with the torch formulation used by the bench's synthetic generator
X2, sf2 = synth.normalize_on_device(Y, G, None)
print('max |X - X_torch| = %.3e' % (X - X2).abs().max().item())
