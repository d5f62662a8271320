"""One epoch of BASELINE configs[2] at the reference-default batch (68 579 x 20 000, zinb-conddisp 64-32-64,
batch 32 = 1 929 steps + the validation pass): the MI355X engine against the oracle's torch-CPU twin of the
reference step (oracle/torch_ref.py: loss graph as dca/loss.py:122-156 composes it, fit semantics of
dca/train.py:91-98) in fp64 (truth) and in fp32 (what an fp32 evaluation of the reference graph gives), from
identical weights and the identical shuffled order.

    python tools/c3_epoch_parity.py [--out profiles/r02_c3_epoch_parity.json] [--cells 68579] [--genes 20000]

Writes: relative error of the epoch's loss / val_loss, of every step's batch loss (max and median), and the
element-wise relative error of mean / dispersion / dropout (and absolute error of the latent) on 1 024 cells after
the epoch -- each for  engine vs oracle-fp64  and, as the yardstick,  oracle-fp32 vs oracle-fp64.

The oracle workers are separate CPU processes reading the host copy of the device-generated matrices from
/dev/shm (the oracle never sees the GPU; the engine never sees the oracle).
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
N_OUT = 1024
BATCH, SHUFFLE_SEED, HIDDEN = 32, 5, (64, 32, 64)


def oracle_worker(shm, dtype_name, threads):
    import torch
    from oracle.torch_ref import TorchAE
    from oracle import net_np as N
    torch.set_num_threads(threads)
    tdt = torch.float64 if dtype_name == 'f64' else torch.float32
    ndt = np.float64 if dtype_name == 'f64' else np.float32
    X = np.load(os.path.join(shm, 'X.npy'), mmap_mode='r')
    Y = np.load(os.path.join(shm, 'Y.npy'), mmap_mode='r')
    sf = np.load(os.path.join(shm, 'sf.npy'))
    with np.load(os.path.join(shm, 'params.npz')) as z:
        p = {k: z[k] for k in z.files}
    n, G = X.shape
    n_train = int(n * 0.9)
    net = TorchAE('zinb-conddisp', p, HIDDEN, True, dtype=tdt)
    idx = np.arange(n_train)
    np.random.RandomState(SHUFFLE_SEED).shuffle(idx)
    t0 = time.time()
    step_loss = []
    tot = 0.0
    for s in range(0, n_train, BATCH):
        b = idx[s:s + BATCH]
        xb = torch.as_tensor(np.asarray(X[b], dtype=ndt))
        yb = torch.as_tensor(np.asarray(Y[b], dtype=ndt))
        sb = torch.as_tensor(np.asarray(sf[b], dtype=ndt))
        loss = float(net.train_step(xb, yb, sb))
        step_loss.append(loss)
        tot += loss * len(b)
    t_train = time.time() - t0
    vt = 0.0
    with torch.no_grad():
        for s in range(n_train, n, 1024):
            e = min(n, s + 1024)
            xb = torch.as_tensor(np.asarray(X[s:e], dtype=ndt)); yb = torch.as_tensor(np.asarray(Y[s:e], dtype=ndt))
            sb = torch.as_tensor(np.asarray(sf[s:e], dtype=ndt))
            vt += float(net.loss(xb, yb, sb, training=False, n_total=float((n - n_train) * G)))
    pn = {k: v.detach().numpy().astype(ndt) for k, v in net.p.items()}
    out = N.OracleAE('zinb-conddisp', pn, HIDDEN, True).predict(np.asarray(X[:N_OUT], dtype=ndt), np.asarray(sf[:N_OUT], dtype=ndt))
    np.savez(os.path.join(shm, 'oracle_%s.npz' % dtype_name), loss=tot / n_train, val_loss=vt, step_loss=np.asarray(step_loss),
             seconds=t_train, **{'out_' + k: np.asarray(v, np.float64 if dtype_name == 'f64' else np.float32) for k, v in out.items()})


def rel(a, b):
    return float(abs(a / b - 1))


def out_stats(got, ref):
    st = {}
    for k in ('mean', 'dispersion', 'dropout'):
        r = np.abs(np.asarray(got[k], np.float64) / np.asarray(ref[k], np.float64) - 1)
        st[k] = {'max_rel': float(r.max()), 'p999_rel': float(np.quantile(r, 0.999)), 'median_rel': float(np.median(r))}
    d = np.abs(np.asarray(got['latent'], np.float64) - np.asarray(ref['latent'], np.float64))
    st['latent'] = {'max_abs': float(d.max()), 'scale': float(np.abs(ref['latent']).max())}
    return st


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--out', default=os.path.join(ROOT, 'gpurun_out', 'c3_epoch_parity.json'))
    ap.add_argument('--cells', type=int, default=68579)
    ap.add_argument('--genes', type=int, default=20000)
    ap.add_argument('--worker', nargs=3, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.worker:
        return oracle_worker(args.worker[0], args.worker[1], int(args.worker[2]))
    import torch
    from dca_amd import synth, prep
    from dca_amd.engine import Engine
    from dca_amd.ops import HipOps
    from dca_amd.train import fit_engine
    from oracle import net_np as N
    n, G = args.cells, args.genes
    ops = HipOps()
    dev = torch.device('cuda')
    Y = synth.generate_counts(n, G, device=dev)
    counts = prep.cell_counts(ops, Y, n, G)
    sf = counts / counts.median()
    X = prep.transform(ops, Y, n, G, sf, True, True)
    p = {k: np.asarray(v, np.float32) for k, v in N.init_params('zinb-conddisp', G, HIDDEN, batchnorm=True, seed=0).items()}
    shm = tempfile.mkdtemp(prefix='dca_c3_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    t0 = time.time()
    np.save(os.path.join(shm, 'X.npy'), X[:, :G].cpu().numpy())
    np.save(os.path.join(shm, 'Y.npy'), Y[:, :G].cpu().numpy())
    np.save(os.path.join(shm, 'sf.npy'), sf.cpu().numpy())
    np.savez(os.path.join(shm, 'params.npz'), **p)
    t_copy = time.time() - t0
    cores = os.cpu_count() or 1
    th = max(1, min(32, cores // 2))
    workers = [subprocess.Popen([sys.executable, os.path.abspath(__file__), '--worker', shm, tag, str(th)]) for tag in ('f64', 'f32')]
    # ---- the engine (HIP kernels through the C ABI), same weights, same order
    eng = Engine('zinb-conddisp', G, G, HIDDEN, True, 0.0, ops=ops)
    eng.set_params(p)
    eng.attach_device_data(X, Y, sf)
    n_train = int(n * 0.9)
    torch.cuda.synchronize(); t0 = time.time()
    h = fit_engine(eng, n_train, n - n_train, n_train, n - n_train, 0, epochs=1, batch_size=BATCH,
                   shuffle_rng=np.random.RandomState(SHUFFLE_SEED), use_graph=True)
    torch.cuda.synchronize(); t_gpu = time.time() - t0
    steps = (n_train + BATCH - 1) // BATCH
    gpu_steps = eng.hist[:steps].cpu().numpy().astype(np.float64)
    eng.reserve(max(N_OUT, 1024))
    out = eng.predict_chunk(0, N_OUT, {'mean', 'dispersion', 'dropout', 'latent'})
    gpu_out = {k: v.cpu().numpy() for k, v in out.items()}
    for w in workers:
        assert w.wait() == 0
    o64 = dict(np.load(os.path.join(shm, 'oracle_f64.npz')))
    o32 = dict(np.load(os.path.join(shm, 'oracle_f32.npz')))

    def block(loss, val, steps_, outs):
        sr = np.abs(np.asarray(steps_) / o64['step_loss'] - 1)
        return {'loss_rel': rel(loss, float(o64['loss'])), 'val_loss_rel': rel(val, float(o64['val_loss'])),
                'step_loss_max_rel': float(sr.max()), 'step_loss_median_rel': float(np.median(sr)),
                'step_loss_first10_max_rel': float(sr[:10].max()),
                'outputs_%d_cells' % N_OUT: out_stats(outs, {k[4:]: v for k, v in o64.items() if k.startswith('out_')})}

    res = {
        'config': 'BASELINE configs[2]: zinb-conddisp 64-32-64 on synthetic %d x %d (dca_amd/synth.py, K-PREP), batch %d, '
                  '1 epoch = %d steps + validation on the last %d cells; weights glorot seed 0; order RandomState(%d)'
                  % (n, G, BATCH, steps, n - n_train, SHUFFLE_SEED),
        'oracle': 'oracle/torch_ref.py (torch-CPU autograd of the loss graph as written in dca/loss.py), %d threads per worker, '
                  'host has %d cores' % (th, cores),
        'loss': {'engine': h.history['loss'][0], 'oracle_f64': float(o64['loss']), 'oracle_f32': float(o32['loss'])},
        'val_loss': {'engine': h.history['val_loss'][0], 'oracle_f64': float(o64['val_loss']), 'oracle_f32': float(o32['val_loss'])},
        'engine_vs_oracle_f64': block(h.history['loss'][0], h.history['val_loss'][0], gpu_steps, gpu_out),
        'oracle_f32_vs_oracle_f64': block(float(o32['loss']), float(o32['val_loss']), o32['step_loss'],
                                          {k[4:]: v for k, v in o32.items() if k.startswith('out_')}),
        'seconds': {'engine_epoch_incl_validation': t_gpu, 'oracle_f64_train': float(o64['seconds']),
                    'oracle_f32_train': float(o32['seconds']), 'device_to_shm_copy': t_copy},
    }
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, 'w') as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res, indent=1))
    for fn in os.listdir(shm):
        os.remove(os.path.join(shm, fn))
    os.rmdir(shm)


if __name__ == '__main__':
    main()
