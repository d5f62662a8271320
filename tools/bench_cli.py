"""The command line end to end on one MI355X: a gene x cell count TSV in, the reference's result files out
(dca/__main__.py:23-24, dca/train.py:105-176).  python tools/bench_cli.py [cells] [genes] [epochs]
Phases are timed by wrapping the functions the CLI calls (no change to them)."""
import os, sys, time, shutil, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
G = int(sys.argv[2]) if len(sys.argv) > 2 else 5000
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
import torch
from dca_amd import synth, hostlib, io as dio, train as T, network as NW
tmp = tempfile.mkdtemp(prefix='dca_cli_')
inp, out = os.path.join(tmp, 'counts.tsv'), os.path.join(tmp, 'out')
Y = synth.generate_counts(n, G, device=torch.device('cuda'))[:, :G].cpu().numpy()
t0 = time.perf_counter()
hostlib.write_tsv(inp, Y.T, rownames=['g%d' % j for j in range(G)], colnames=['c%d' % i for i in range(n)])   # gene x cell
print('input: %d genes x %d cells, %.0f MB of text (written in %.1f s)' % (G, n, os.path.getsize(inp) / 1e6, time.perf_counter() - t0))
marks = {}
def timed(name, fn):
    def w(*a, **k):
        torch.cuda.synchronize(); s = time.perf_counter()
        r = fn(*a, **k)
        torch.cuda.synchronize(); marks[name] = marks.get(name, 0.0) + time.perf_counter() - s
        return r
    return w
dio.read_text = timed('read the text matrix', dio.read_text)
dio.normalize = timed('normalize (upload, K-PREP, download)', dio.normalize)
T.train = timed('train', T.train)
NW.Autoencoder.predict = timed('predict', NW.Autoencoder.predict)
NW.write_text_matrix = timed('write the result files', NW.write_text_matrix)
NW.Autoencoder.predict_write = timed('predict + write, one call (contains the two above when not fused)', NW.Autoencoder.predict_write)
from dca_amd.__main__ import main
sys.argv = ['dca', inp, out, '-e', str(epochs), '--earlystop', '0', '--reducelr', '0']
s = time.perf_counter()
main()
torch.cuda.synchronize()
total = time.perf_counter() - s
for k, v in marks.items():
    print('  %-40s %7.2f s' % (k, v))
print('  %-40s %7.2f s' % ('everything else', total - sum(v for k, v in marks.items() if not k.startswith('predict + write'))
                            - (marks.get([k for k in marks if k.startswith('predict + write')][0], 0.0) if os.environ.get('DCA_AMD_FUSED_WRITE', '1') != '0' else 0.0)
                            + (sum(v for k, v in marks.items() if k in ('predict', 'write the result files')) if os.environ.get('DCA_AMD_FUSED_WRITE', '1') != '0' else 0.0)))
files = sorted(os.listdir(out))
print('dca CLI total %.2f s; wrote %s (%.0f MB)' % (total, files, sum(os.path.getsize(os.path.join(out, f)) for f in files) / 1e6))
shutil.rmtree(tmp)
