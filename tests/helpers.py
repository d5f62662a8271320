"""Shared pieces of the engine parity tests (CPU with the oracle-backed ops, GPU with HipOps)."""
import numpy as np
import torch

from conftest import synth_counts
from oracle import net_np as N


def make_problem(n, G, hs, ae_type, batchnorm=True, seed=0, dtype=np.float64):
    """Synthetic counts -> (X z-scored log-normalised, Y raw counts, sf) + glorot params with the
    zero-initialised vectors perturbed (so their gradients are exercised)."""
    rng = np.random.RandomState(seed)
    Y = synth_counts(n, G, seed + 1)
    lib = Y.sum(1)
    sf = lib / np.median(lib)
    Xn = np.log1p(Y / sf[:, None])
    X = (Xn - Xn.mean(0)) / np.maximum(Xn.std(0, ddof=1), 1e-12)
    p = N.init_params(ae_type, G, hs, batchnorm=batchnorm, seed=seed + 2, dtype=np.float64)
    for k in p:
        if k[0] in 'bt':
            p[k] = rng.normal(0, .1, p[k].shape)
    f32 = lambda a: np.asarray(a, np.float32)
    X, Y, sf = f32(X), f32(Y), f32(sf)
    p = {k: f32(v) for k, v in p.items()}
    return X, Y, sf, p


def oracle_net(ae_type, p, hs, batchnorm, ridge=0.0, dtype=np.float64, activation='relu', **dropout):
    return N.OracleAE(ae_type, {k: np.asarray(v, dtype).copy() for k, v in p.items()}, hs, batchnorm, ridge,
                      activation=activation, **dropout)


def make_engine(ops, ae_type, G, hs, batchnorm, ridge, p, X, Y, sf, comm=None, activation='relu', **dropout):
    from dca_amd.engine import Engine
    eng = Engine(ae_type, G, G, hs, batchnorm, ridge, ops=ops, comm=comm, activation=activation, **dropout)
    eng.set_params(p)
    eng.load_data(X, Y, sf)
    return eng


def assert_grads_close(got, ref, rtol=2e-3, atol_scale=2e-5, skip=()):
    gscale = max(float(np.abs(np.asarray(v)).max()) for v in ref.values())
    for k, r in ref.items():
        if k in skip:
            continue
        g = got[k]
        r = np.asarray(r, np.float64)
        if np.abs(r).max() < 1e-9 * gscale:
            # bias of a Dense feeding BatchNormalization: its gradient is identically zero
            # (round-off in both implementations)
            assert np.abs(g).max() < 1e-5 * gscale, (k, float(np.abs(g).max()))
            continue
        tol = rtol * np.abs(r) + atol_scale * max(np.abs(r).max(), 1e-12)
        err = np.abs(g.astype(np.float64) - r)
        assert (err <= tol).all(), (k, float(err.max()), float(np.abs(r).max()),
                                    np.argwhere(err > tol)[:4].tolist())


def run_single_step(eng, rows, lr=1e-3):
    """Runs one training step on the given storage rows; returns (loss, grads, params)."""
    B = len(rows)
    eng.reserve(B)
    eng.perm = torch.as_tensor(np.asarray(rows, np.int32)).to(eng.dev)
    eng.hist = torch.zeros(4, dtype=torch.float32, device=eng.dev)
    eng.cursor.zero_(); eng.acc.zero_()
    eng.set_lr(lr)
    if eng.comm.dp:                       # data parallel with one rank: this rank holds the whole (global) batch
        eng.train_step(B, B, [B], B)
    else:
        eng.train_step(B, rows_per_slot=B)
    if eng.dev.type == 'cuda':
        torch.cuda.synchronize()
    return float(eng.hist[0].item()), eng.get_grads(), eng.get_params()
