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


class _FlatState:
    """A named parameter set (oracle names) <-> the engine's flat fp32 layout, assembled on the host in one buffer so that
    a whole state crosses to the device in ONE copy (Engine.set_params round-trips the flat buffer per call)."""

    def __init__(self, lay):
        self.lay = lay
        self.flat = torch.zeros(lay.total, dtype=torch.float32)
        self.views = {}
        for name in lay.seg:
            if name in ('Wh', 'bh'):
                continue
            v = lay.view(self.flat, name)
            self.views[name] = v[:lay.G_out] if name in ('theta_w', 'pi_k', 'pi_c') else v
        Wh, bh = lay.view(self.flat, 'Wh'), lay.view(self.flat, 'bh')
        for hd in lay.dense_heads:
            c0, c1 = lay.head_cols(hd)
            self.views['W_' + hd] = Wh[:, c0:c1]
            self.views['b_' + hd] = bh[c0:c1]

    def fill(self, named):
        for k, v in self.views.items():
            v.copy_(named[k].detach() if isinstance(named[k], torch.Tensor) else torch.as_tensor(np.asarray(named[k])))
        return self.flat


def run_reseeded_epoch(eng, tnet, order, B, lr=1e-3, clip=5.0, report_every=0):
    """ONE EPOCH in which EVERY step of the engine starts from the fp64 oracle's state (parameters, RMSprop accumulators,
    batch-norm moving statistics) -- the statement about the whole epoch that the chaotic growth of fp32 round-off under
    Keras' RMSprop cannot blur (a free-running fp32 trajectory leaves the fp64 one within ten steps).

    eng: an Engine with its data attached; tnet: oracle.torch_ref.TorchAE (fp64) holding the same initial parameters;
    order: the epoch's shuffled storage rows.  Per step (train.py:91-98: batch = order[s : s + B], the last one partial):
      * the engine's flat buffers are overwritten with the oracle's state, both take the step on the same rows (the oracle
        reads the engine's own device-resident fp32 inputs);
      * batch loss: |engine / oracle - 1|;
      * every gradient element against the oracle's: |g - r| <= 2e-3 |r| + 2e-5 max|r| over its tensor (the single-step
        tolerance of assert_grads_close; biases in front of a batch norm, whose gradient is round-off, are held to 1e-5 of
        the largest gradient) -> number of violations;
      * the optimizer: the engine's new parameters / accumulators against clipvalue + RMSprop (train.py:54-57) evaluated
        in fp64 from the ENGINE's gradient and the state it started from, as error / tolerance with tolerance = 2e-6 of
        the update + one fp32 rounding of the stored value;
      * batch-norm moving statistics against the oracle's.
    Returns a dict of per-step arrays; nothing is asserted here."""
    from dca_amd.engine import RMS_RHO, RMS_EPS
    lay, dev = eng.lay, eng.dev
    P = lay.P
    n_train = len(order)
    steps = (n_train + B - 1) // B
    eng.reserve(B)
    eng.perm = torch.as_tensor(np.asarray(order, np.int32)).to(dev)
    eng.hist = torch.zeros(steps + 4, dtype=torch.float32, device=dev)
    eng.cursor.zero_(); eng.acc.zero_()
    eng.set_lr(lr)
    eng.clip = clip
    st_w, st_ms, st_g = _FlatState(lay), _FlatState(lay), _FlatState(lay)
    nbn = len(eng.mm)
    zero_bias = [lay.seg['b%d' % i] for i in range(len(lay.hidden))] if lay.batchnorm else []
    segs = [(name,) + lay.seg[name] for name in lay.seg]
    out = {k: np.zeros(steps) for k in ('loss_eng', 'loss_or', 'grad_viol', 'grad_err', 'upd_err', 'ms_err', 'bn_err')}
    lr32 = float(np.float32(lr))
    order_t = torch.as_tensor(np.asarray(order, np.int64)).to(dev)
    for k in range(steps):
        rows = order_t[k * B:(k + 1) * B]
        b = int(rows.numel())
        # ---- the engine starts from the oracle's state
        w0 = st_w.fill(tnet.p).to(dev)
        ms0 = st_ms.fill(tnet.ms).to(dev)
        eng.w[:lay.total].copy_(w0); eng.ms[:lay.total].copy_(ms0)
        for i in range(nbn):
            eng.mm[i].copy_(tnet.p['mm%d' % i].to(torch.float32)); eng.mv[i].copy_(tnet.p['mv%d' % i].to(torch.float32))
        eng.train_step(b, rows_per_slot=B)
        # ---- the oracle takes the same step on the engine's own inputs
        xb = eng.X[rows][:, :lay.G_in].to('cpu', torch.float64)
        yb = eng.Y[rows][:, :lay.G_out].to('cpu', torch.float64)
        sb = eng.sf[rows].to('cpu', torch.float64)
        loss_or, g_or = tnet.grads(xb, yb, sb)
        g_ref = st_g.fill(g_or).to(dev)
        with torch.no_grad():
            for name, gk in g_or.items():
                gk = gk.clamp(-clip, clip)
                tnet.ms[name].mul_(RMS_RHO).addcmul_(gk, gk, value=1 - RMS_RHO)
                tnet.p[name] -= lr32 * gk / (torch.sqrt(tnet.ms[name]) + RMS_EPS)
        # ---- comparisons, on the device
        g_eng = eng.g[:P]
        gscale = g_ref[:P].abs().max()
        viol = torch.zeros((), dtype=torch.float64, device=dev)
        worst = torch.zeros((), dtype=torch.float64, device=dev)
        for name, off, shape in segs:
            n_el = int(np.prod(shape))
            ge, gr = g_eng[off:off + n_el].double(), g_ref[off:off + n_el].double()
            if (off, shape) in zero_bias:
                tol = torch.full_like(gr, 1.0) * 1e-5 * gscale
            else:
                tol = 2e-3 * gr.abs() + 2e-5 * torch.clamp(gr.abs().max(), min=1e-12)
            e = (ge - gr).abs()
            viol += (e > tol).sum()
            worst = torch.maximum(worst, (e / tol).max())
        gc = g_eng.clamp(-clip, clip).double() if clip else g_eng.double()
        ms1 = RMS_RHO * ms0[:P].double() + (1 - RMS_RHO) * gc * gc
        w1 = w0[:P].double() - lr32 * gc / (torch.sqrt(ms1) + RMS_EPS)
        dw = (w1 - w0[:P].double()).abs()
        # error / tolerance: 2e-6 of the update itself + one fp32 rounding of the stored value
        upd = ((eng.w[:P].double() - w1).abs() / (2e-6 * dw + 1.2e-7 * w1.abs() + 1e-12)).max()
        mse = ((eng.ms[:P].double() - ms1).abs() / (5e-7 * ms1 + 1e-36)).max()
        bn = torch.zeros((), dtype=torch.float64, device=dev)
        for i in range(nbn):
            for a, r in ((eng.mm[i], tnet.p['mm%d' % i]), (eng.mv[i], tnet.p['mv%d' % i])):
                r = r.to(dev)
                bn = torch.maximum(bn, ((a.double() - r).abs() / (1e-6 + 1e-5 * r.abs())).max())
        res = torch.stack([eng.g[P].double(), viol, worst, upd, mse, bn]).cpu().numpy()
        out['loss_eng'][k], out['loss_or'][k] = res[0], float(loss_or)
        out['grad_viol'][k], out['grad_err'][k], out['upd_err'][k], out['ms_err'][k], out['bn_err'][k] = res[1:]
        if report_every and (k + 1) % report_every == 0:
            rel = np.abs(out['loss_eng'][:k + 1] / out['loss_or'][:k + 1] - 1)
            print('step %d / %d: batch loss %.6f, |rel| max so far %.1e, gradient violations %d, worst gradient error / '
                  'tolerance %.2f' % (k + 1, steps, res[0], rel.max(), int(out['grad_viol'][:k + 1].sum()),
                                      out['grad_err'][:k + 1].max()), flush=True)
    return out
