"""End-to-end dca() on a host AnnData of BASELINE configs[2] size: host buffers in, host buffers out
(PCIe and host work included), phases timed.  python tools/bench_e2e.py [cells] [genes] [epochs] [batch]"""
import os, sys, time
import numpy as np, pandas as pd, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 68579
G = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 32
from dca_amd import synth, io, prep, train as T
from dca_amd._anndata import AnnData
from dca_amd.api import dca
t0 = time.perf_counter()
Y = synth.generate_counts(n, G, device=torch.device('cuda'))[:, :G].cpu().numpy()
torch.cuda.empty_cache()
print('synthetic counts on host: %.1f GB (%.1f s to generate + download)' % (Y.nbytes / 1e9, time.perf_counter() - t0))
ad = AnnData(Y, obs=pd.DataFrame(index=np.arange(n).astype(str)), var=pd.DataFrame(index=np.arange(G).astype(str)))

marks = {}
orig_norm, orig_train = io.normalize, T.train
import dca_amd.api as A
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); s = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); marks[name] = marks.get(name, 0.0) + time.perf_counter() - s
        return r
    return w
A.resident_counts = timed('upload of the counts + per-gene totals', A.resident_counts)
A.normalize = timed('normalize (K-PREP + X download + .raw copy)', orig_norm)
A.train = timed('train', orig_train)
s = time.perf_counter()
net = dca(ad, ae_type='zinb-conddisp', epochs=epochs, batch_size=batch, early_stop=0, reduce_lr=0,
          return_info=True, return_model=True, verbose=False)
torch.cuda.synchronize()
total = time.perf_counter() - s
for k, v in marks.items():
    print('  %-48s %7.2f s' % (k, v))
print('  %-48s %7.2f s' % ('predict (denoise + dispersion + dropout, download)', total - sum(marks.values())))
hist = ad.uns['dca_loss_history']
print('dca() total %.2f s for %d x %d, %d epochs at batch %d: loss %s' % (total, n, G, epochs, batch, ['%.4f' % x for x in hist['loss']]))
n_train = int(n * 0.9)
print('training alone: %.0f cells/s (train rows x epochs / train time, validation pass included)' % (n_train * epochs / marks['train']))
