"""Dropout parity cases shared by the CPU (oracle ops) and GPU (HIP) suites."""
import numpy as np

from helpers import make_problem, oracle_net, make_engine, assert_grads_close, run_single_step
from oracle import net_np as N
from test_dp_gloo import FixedOrders

DROP = dict(hidden_dropout=[0.2, 0.0, 0.35], input_dropout=0.15, dropout_seed=0x1234567890abcdef)


def step_parity(ops, ae='zinb-conddisp', bn=True, n=96, G=150, hs=(64, 32, 64), B=48, steps=3):
    """Training steps with dropout on every kind of site: loss and every gradient against the fp64
    oracle drawing the same masks; the step counter moves the masks."""
    X, Y, sf, p = make_problem(n, G, hs, ae, bn, seed=5)
    ref = oracle_net(ae, p, hs, bn, **DROP)
    eng = make_engine(ops, ae, G, hs, bn, 0.0, p, X, Y, sf, **DROP)
    rng = np.random.RandomState(1)
    losses = []
    for s in range(steps):
        rows = rng.permutation(n)[:B]
        assert ref.step == s
        rl, rg = ref.loss_and_grads(X[rows], Y[rows], sf[rows])
        N.rmsprop_step(ref.p, rg, ref.__dict__.setdefault('_ms', {}), 1e-3)
        loss, g, _ = run_single_step(eng, rows)
        assert abs(loss - rl) < 2e-5 * abs(rl), (s, loss, rl)
        assert_grads_close(g, rg)
        assert int(eng.drop_iter.item()) == s + 1
        losses.append(rl)
    return losses


def fit_parity(ops, ae='zinb', bn=True, n=120, G=60, hs=(16, 8, 16), B=32, epochs=2, rtol=2e-4):
    from dca_amd.engine import Engine
    from dca_amd.train import fit_engine
    X, Y, sf, p = make_problem(n, G, hs, ae, bn, seed=6)
    drop = dict(hidden_dropout=0.25, input_dropout=0.0, dropout_seed=99)
    n_train = int(n * 0.9)
    rs = np.random.RandomState(4)
    orders = []
    for _ in range(epochs):
        idx = np.arange(n_train)
        rs.shuffle(idx)
        orders.append(idx)
    ref = oracle_net(ae, p, hs, bn, **drop)
    rh = N.fit(ref, X, Y, sf, epochs=epochs, batch_size=B, shuffle_rng=FixedOrders(orders), reduce_lr=1, early_stop=0)
    eng = Engine(ae, G, G, hs, bn, 0.0, ops=ops, **drop)
    eng.set_params(p)
    eng.load_data(X, Y, sf)
    h = fit_engine(eng, n_train, n - n_train, n_train, n - n_train, 0, epochs=epochs, batch_size=B,
                   shuffle_rng=FixedOrders(orders), reduce_lr=1, early_stop=0)
    np.testing.assert_allclose(h.history['loss'], rh['loss'], rtol=rtol)
    np.testing.assert_allclose(h.history['val_loss'], rh['val_loss'], rtol=rtol)
    return h.history


def inference_ignores_dropout(ops, ae='zinb-conddisp', n=40, G=50, hs=(16, 8, 16)):
    X, Y, sf, p = make_problem(n, G, hs, ae, True, seed=7)
    a = make_engine(ops, ae, G, hs, True, 0.0, p, X, Y, sf, hidden_dropout=0.5, input_dropout=0.5)
    b = make_engine(ops, ae, G, hs, True, 0.0, p, X, Y, sf)
    for e in (a, b):
        e.reserve(n)
    oa = {k: v.cpu().numpy().copy() for k, v in a.predict_chunk(0, n, {'mean', 'dispersion', 'dropout', 'latent'}).items()}
    ob = {k: v.cpu().numpy().copy() for k, v in b.predict_chunk(0, n, {'mean', 'dispersion', 'dropout', 'latent'}).items()}
    for k in oa:
        np.testing.assert_array_equal(oa[k], ob[k])
