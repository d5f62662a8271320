"""Generates tests/golden/c3_steps_oracle.npz: the first 64 training steps (batch 32, the reference default) and the
validation loss on 512 held-out cells of BASELINE configs[2] (zinb-conddisp 64-32-64 on a 68 579 x 20 000 count matrix),
computed by the fp64 oracle (oracle/net_np.py) -- the driver-visible form of the "per-epoch loss to 1e-4 on the
68k x 20k config" clause (tests/test_engine_gpu.py::test_c3_first_steps_match_oracle holds the MI355X engine to 1e-5 per
step and 1e-4 on the means).

    python tests/golden/make_c3_steps_golden.py          (about 10 minutes and 6 GB on 8 cores)

The whole 68 579 x 20 000 matrix is generated here (numpy PCG64, the Gamma-Poisson + dropout model of SURVEY 8d) because
the inputs of every cell depend on it: the size factors need the median library size, the z-score the mean and standard
deviation of every gene over ALL cells (dca/io.py:99-109).  The fixture then stores only what the 64 steps touch: the
counts of the 2 048 + 512 cells involved (as sparse triplets), their size factors, and the per-gene statistics; the
engine test rebuilds X for those cells with the same fp32 operations K-PREP uses.  Initial weights: glorot-uniform from
oracle.net_np.init_params(seed 0), rebuilt by the test.  Shuffle: numpy RandomState(5) over the 61 721 training cells.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

N_CELLS, N_GENES, HIDDEN, BATCH, STEPS, N_VAL = 68579, 20000, (64, 32, 64), 32, 64, 512
DATA_SEED, SHUFFLE_SEED, INIT_SEED, LR, CLIP = 20260925, 5, 0, 1e-3, 5.0


def generate(n, G, seed, chunk=2048):
    """uint8 counts [n, G] (the model keeps them far below 255 at these gene means), every gene and cell >= 1 count."""
    rng = np.random.Generator(np.random.PCG64(seed))
    m = rng.normal(-3.2, 1.6, size=G)
    em = np.exp(m)
    Y = np.zeros((n, G), dtype=np.uint8)
    for s in range(0, n, chunk):
        b = min(chunk, n - s)
        lib = rng.lognormal(0.0, 0.4, size=b)
        lam = lib[:, None] * em[None, :] * rng.gamma(2.0, 0.5, size=(b, G))
        y = rng.poisson(lam)
        y *= rng.random((b, G)) >= 0.3
        Y[s:s + b] = np.minimum(y, 250)
    rows = np.nonzero(Y.sum(1) == 0)[0]
    Y[rows, rng.integers(0, G, len(rows))] = 1
    cols = np.nonzero(Y.sum(0) == 0)[0]
    Y[rng.integers(0, n, len(cols)), cols] += 1
    return Y


def kprep_input(Yr, fac, mean32, std32):
    """X of the given cells with the fp32 operations of K-PREP (dcahip_prep_*): division, log1p, (x - mean) / std."""
    x = np.log1p((Yr.astype(np.float32) / fac.astype(np.float32)[:, None]).astype(np.float32)).astype(np.float32)
    return ((x - mean32[None, :]) / std32[None, :]).astype(np.float32)


def main():
    from oracle import net_np as N
    t0 = time.time()
    Y = generate(N_CELLS, N_GENES, DATA_SEED)
    print('counts generated: %.0f s, non-zero %.4f, max %d' % (time.time() - t0, (Y != 0).mean(), Y.max()), flush=True)
    lib = Y.sum(1, dtype=np.int64).astype(np.float64)
    sf = (lib / np.median(lib)).astype(np.float32)                       # size factors = normalisation divisor (io.py:99-101)
    # per-gene mean / std (ddof = 1) of log1p(y / sf) over all cells, fp32 values accumulated in fp64 as K-PREP does
    s1 = np.zeros(N_GENES); s2 = np.zeros(N_GENES)
    for s in range(0, N_CELLS, 4096):
        x = np.log1p((Y[s:s + 4096].astype(np.float32) / sf[s:s + 4096, None]).astype(np.float32)).astype(np.float32)
        s1 += x.sum(0, dtype=np.float64)
        s2 += (x * x).astype(np.float32).sum(0, dtype=np.float64)
    mean = s1 / N_CELLS
    var = (s2 / N_CELLS - mean * mean) * (N_CELLS / (N_CELLS - 1.0))
    std = np.sqrt(np.maximum(var, 0)); std[std == 0] = 1.0
    mean32, std32 = mean.astype(np.float32), std.astype(np.float32)
    n_train = int(N_CELLS * 0.9)
    idx = np.arange(n_train)
    np.random.RandomState(SHUFFLE_SEED).shuffle(idx)
    train_rows = idx[:STEPS * BATCH]
    val_rows = np.arange(n_train, n_train + N_VAL)
    rows = np.concatenate([train_rows, val_rows])
    Yr = Y[rows]
    X = kprep_input(Yr, sf[rows], mean32, std32).astype(np.float64)
    Yd = Yr.astype(np.float64); sfd = sf[rows].astype(np.float64)
    p = N.init_params('zinb-conddisp', N_GENES, HIDDEN, batchnorm=True, seed=INIT_SEED, dtype=np.float64)
    p = {k: np.asarray(v, np.float32).astype(np.float64) for k, v in p.items()}      # the engine holds fp32 weights
    v = slice(STEPS * BATCH, STEPS * BATCH + N_VAL)

    def run(dtype):
        net = N.OracleAE('zinb-conddisp', {k: a.astype(dtype) for k, a in p.items()}, HIDDEN, True, 0.0)
        Xd, Yq, sq = X.astype(dtype), Yd.astype(dtype), sfd.astype(dtype)
        ms, out = {}, []
        for st in range(STEPS):
            b = slice(st * BATCH, (st + 1) * BATCH)
            loss, g = net.loss_and_grads(Xd[b], Yq[b], sq[b])
            N.rmsprop_step(net.p, g, ms, float(np.float32(LR)), clip=CLIP)
            out.append(float(loss))
            if st % 8 == 0:
                print('%s step %d loss %.8f (%.0f s)' % (np.dtype(dtype).name, st, loss, time.time() - t0), flush=True)
        return out, float(net.eval_loss_sum(Xd[v], Yq[v], sq[v])) / (N_VAL * N_GENES)

    losses, val = run(np.float64)
    # the same 64 steps by an fp32 evaluation of the same restatement: the yardstick for any fp32 implementation -- Keras'
    # RMSprop (epsilon outside the root) takes sign-like steps while an accumulator is small, so fp32 noise in small
    # gradients grows into O(lr) differences of those parameters within a few steps
    losses32, val32 = run(np.float32)
    print('val_loss on %d held-out cells: %.8f (fp32 oracle %.8f)' % (N_VAL, val, val32))
    print('fp32 oracle vs fp64 oracle, per-step loss: max rel %.2e' % np.abs(np.asarray(losses32) / np.asarray(losses) - 1).max())
    r, c = np.nonzero(Yr)
    np.savez_compressed(os.path.join(HERE, 'c3_steps_oracle.npz'),
                        nz_row=r.astype(np.uint16), nz_col=c.astype(np.uint16), nz_val=Yr[r, c].astype(np.uint8),
                        sf=sf[rows], gene_mean=mean32, gene_std=std32, step_loss=np.asarray(losses, np.float64),
                        val_loss=np.float64(val), step_loss_f32=np.asarray(losses32, np.float64),
                        val_loss_f32=np.float64(val32), shape=np.asarray([len(rows), N_GENES, STEPS, BATCH, N_VAL]),
                        rows=rows.astype(np.int32))
    print('wrote c3_steps_oracle.npz: %.1f MB' % (os.path.getsize(os.path.join(HERE, 'c3_steps_oracle.npz')) / 1e6))


if __name__ == '__main__':
    main()
