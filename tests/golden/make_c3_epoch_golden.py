"""Generates tests/golden/c3_epoch_oracle.npz: ONE WHOLE EPOCH of BASELINE configs[2] at the reference's default batch
(zinb-conddisp 64-32-64 on 68 579 x 20 000, batch 32 = 1 929 steps, then the validation loss on the last 6 858 cells),
by the oracle's torch-CPU twin of the reference step (oracle/torch_ref.py: the loss graph as dca/loss.py:122-156 composes
it, Dense -> BatchNorm -> ReLU as dca/network.py:92-141, Keras RMSprop + clipvalue as dca/train.py:54-59, fit semantics
of dca/train.py:91-98) in fp64 (truth) and in fp32 (what an fp32 evaluation of the reference graph gives: the yardstick
for any fp32 implementation), for several (shuffle seed, initialisation seed) pairs.

    python tests/golden/make_c3_epoch_golden.py [--seeds 3] [--threads 4] [--jobs 2]      (CPU only; ~1 h on 8 cores)

The count matrix is dca_amd.synth.generate_counts_portable: a pure function of (n, G, seed) on any device (integer
hashing + thresholds from IEEE basic arithmetic), so the GPU test regenerates the identical matrix in HBM -- the fixture
carries its checksum.  The oracle's inputs are computed here on the host: size factors = library size / median
(dca/io.py:99-101), X = z-scored log1p(y / sf) with fp32 element operations and fp64 accumulation, as K-PREP computes it
(dca/io.py:103-109 semantics, ddof = 1).  The engine test uses K-PREP's own output: the two X differ by fp32 rounding.
The fixture stores, per seed pair: the 1 929 batch losses, the epoch loss (sample-weighted mean) and val_loss, each for
fp64, fp32 and a second fp32 realisation (inputs perturbed by an ulp) -- about 150 KB.
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

N_CELLS, N_GENES, HIDDEN, BATCH = 68579, 20000, (64, 32, 64), 32
DATA_SEED, LR = 20260925, 1e-3
TAGS = ('f64', 'f32', 'f32b')
SEED_PAIRS = [(5, 0), (6, 1), (7, 2), (8, 3)]          # (numpy RandomState seed of the shuffle, glorot initialisation seed)


def host_inputs(Y):
    """(X fp32 [n, G], sf fp32 [n]) from uint8 counts, K-PREP's arithmetic on the host."""
    n, G = Y.shape
    lib = Y.sum(1, dtype=np.int64).astype(np.float64)
    sf = (lib / np.median(lib)).astype(np.float32)
    X = np.empty((n, G), np.float32)
    s1 = np.zeros(G); s2 = np.zeros(G)
    for s in range(0, n, 4096):
        x = np.log1p((Y[s:s + 4096].astype(np.float32) / sf[s:s + 4096, None]).astype(np.float32)).astype(np.float32)
        X[s:s + 4096] = x
        s1 += x.sum(0, dtype=np.float64)
        s2 += (x * x).astype(np.float32).sum(0, dtype=np.float64)
    mean = s1 / n
    var = (s2 / n - mean * mean) * (n / (n - 1.0))
    std = np.sqrt(np.maximum(var, 0)); std[std == 0] = 1.0
    mean32, std32 = mean.astype(np.float32), std.astype(np.float32)
    for s in range(0, n, 4096):
        X[s:s + 4096] = (X[s:s + 4096] - mean32[None, :]) / std32[None, :]
    return X, sf


def worker(shm, shuffle_seed, init_seed, dtype_name, threads, max_steps):
    import torch
    from oracle.torch_ref import TorchAE
    from oracle import net_np as N
    torch.set_num_threads(threads)
    tdt = torch.float64 if dtype_name == 'f64' else torch.float32
    ndt = np.float64 if dtype_name == 'f64' else np.float32
    # 'f32b': a second, equally valid fp32 evaluation -- every input perturbed by about an ulp (x * (1 + 2^-23) moves about
    # half of the values to their neighbour), the size of the difference between two correct fp32 z-score routines.  The
    # pool of fp32 realisations is the yardstick the engine is held to (a single one is one draw of a chaotic quantity).
    bump = np.float32(1.0 + 2.0 ** -23) if dtype_name == 'f32b' else None
    X = np.load(os.path.join(shm, 'X.npy'), mmap_mode='r')
    Y = np.load(os.path.join(shm, 'Y.npy'), mmap_mode='r')
    sf = np.load(os.path.join(shm, 'sf.npy'))
    n, G = X.shape
    n_train = int(n * 0.9)
    p = {k: np.asarray(v, np.float32) for k, v in N.init_params('zinb-conddisp', G, HIDDEN, batchnorm=True, seed=init_seed).items()}
    net = TorchAE('zinb-conddisp', p, HIDDEN, True, dtype=tdt)
    idx = np.arange(n_train)
    np.random.RandomState(shuffle_seed).shuffle(idx)
    t0 = time.time()
    step_loss, tot = [], 0.0
    for s in range(0, n_train, BATCH):
        if max_steps and len(step_loss) >= max_steps:
            break
        b = np.sort(idx[s:s + BATCH])                       # (sorted for the memory map; the batch loss does not depend on row order)
        order = np.argsort(np.argsort(idx[s:s + BATCH]))
        xb = np.asarray(X[b], dtype=ndt)[order]
        if bump is not None:
            xb = (xb * bump).astype(np.float32)
        xb = torch.as_tensor(xb)
        yb = torch.as_tensor(np.asarray(Y[b], dtype=ndt)[order])
        sb = torch.as_tensor(np.asarray(sf[b], dtype=ndt)[order])
        loss = float(net.train_step(xb, yb, sb, lr=float(np.float32(LR))))
        step_loss.append(loss)
        tot += loss * len(b)
        if len(step_loss) % 200 == 0:
            print('%s seeds (%d, %d): step %d loss %.6f (%.0f s)' % (dtype_name, shuffle_seed, init_seed, len(step_loss), loss,
                                                                    time.time() - t0), flush=True)
    vt = 0.0
    with torch.no_grad():
        for s in range(n_train, n, 1024):
            e = min(n, s + 1024)
            xb = np.array(X[s:e], dtype=ndt)
            if bump is not None:
                xb = (xb * bump).astype(np.float32)
            xb = torch.as_tensor(xb); yb = torch.as_tensor(np.array(Y[s:e], dtype=ndt))
            sb = torch.as_tensor(np.array(sf[s:e], dtype=ndt))
            vt += float(net.loss(xb, yb, sb, training=False, n_total=float((n - n_train) * G)))
    np.savez(os.path.join(shm, 'oracle_%d_%d_%s.npz' % (shuffle_seed, init_seed, dtype_name)),
             loss=tot / n_train, val_loss=vt, step_loss=np.asarray(step_loss), seconds=time.time() - t0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--seeds', type=int, default=3)
    ap.add_argument('--threads', type=int, default=4)
    ap.add_argument('--jobs', type=int, default=2)
    ap.add_argument('--max-steps', type=int, default=0, help='(timing trials) stop the training loop early')
    ap.add_argument('--cells', type=int, default=N_CELLS)
    ap.add_argument('--shm', default=None, help='reuse the inputs of an earlier run in this directory')
    ap.add_argument('--worker', nargs=5, default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.worker:
        w = args.worker
        return worker(w[0], int(w[1]), int(w[2]), w[3], int(w[4]), args.max_steps)
    import torch
    from dca_amd import synth
    n, G = args.cells, N_GENES
    shm = args.shm or tempfile.mkdtemp(prefix='dca_c3_epoch_', dir='/dev/shm' if os.path.isdir('/dev/shm') else None)
    t0 = time.time()
    if not os.path.exists(os.path.join(shm, 'meta.npz')):
        Yt = synth.generate_counts_portable(n, G, seed=DATA_SEED, device='cpu', dtype=torch.uint8)
        checksum = synth.counts_checksum(Yt, G)
        Y = Yt[:, :G].numpy()
        print('counts: %.0f s, zeros %.4f, max %d, checksum %d' % (time.time() - t0, (Y == 0).mean(), Y.max(), checksum), flush=True)
        X, sf = host_inputs(Y)
        np.save(os.path.join(shm, 'X.npy'), X); np.save(os.path.join(shm, 'Y.npy'), Y); np.save(os.path.join(shm, 'sf.npy'), sf)
        np.savez(os.path.join(shm, 'meta.npz'), checksum=np.int64(checksum), sf_sum=np.float64(sf.astype(np.float64).sum()))
        del X, Y, Yt
        print('inputs written to %s: %.0f s' % (shm, time.time() - t0), flush=True)
    meta = np.load(os.path.join(shm, 'meta.npz'))
    jobs = [(ss, si, tag) for ss, si in SEED_PAIRS[:args.seeds] for tag in TAGS]
    jobs = [j for j in jobs if not os.path.exists(os.path.join(shm, 'oracle_%d_%d_%s.npz' % j))]
    running = []
    while jobs or running:
        while jobs and len(running) < args.jobs:
            ss, si, tag = jobs.pop(0)
            cmd = [sys.executable, os.path.abspath(__file__), '--worker', shm, str(ss), str(si), tag, str(args.threads)]
            if args.max_steps:
                cmd += ['--max-steps', str(args.max_steps)]
            running.append(subprocess.Popen(cmd))
        time.sleep(2)
        for pr in list(running):
            if pr.poll() is not None:
                assert pr.returncode == 0
                running.remove(pr)
    out = {'checksum': meta['checksum'], 'shape': np.asarray([n, G, BATCH]), 'data_seed': np.int64(DATA_SEED),
           'seed_pairs': np.asarray(SEED_PAIRS[:args.seeds])}
    for ss, si in SEED_PAIRS[:args.seeds]:
        for tag in TAGS:
            z = np.load(os.path.join(shm, 'oracle_%d_%d_%s.npz' % (ss, si, tag)))
            for k in ('loss', 'val_loss', 'step_loss'):
                out['%s_%d_%s' % (k, ss, tag)] = z[k]
            print('seeds (%d, %d) %s: loss %.8f val_loss %.8f (%.0f s)' % (ss, si, tag, float(z['loss']), float(z['val_loss']),
                                                                          float(z['seconds'])))
    if not args.max_steps and n == N_CELLS:
        np.savez_compressed(os.path.join(HERE, 'c3_epoch_oracle.npz'), **out)
        print('wrote c3_epoch_oracle.npz')


if __name__ == '__main__':
    main()
