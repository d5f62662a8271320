"""K-DROP on the MI355X: masks bit-identical to the oracle's generator, training steps / fit with
dropout against the fp64 oracle drawing the same masks."""
import numpy as np
import pytest
import torch

from oracle import net_np as N
import _dropout_cases as C

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from dca_amd.ops import HipOps
    return HipOps()


@pytest.mark.parametrize('B,h,ld,rate,gather', [(37, 64, 64, 0.3, False), (5, 6, 8, 0.5, False), (300, 1001, 1004, 0.15, True),
                                                 (1, 1, 4, 0.9, False), (4096, 20000, 20000, 0.1, True)])
def test_kernel_equals_oracle_generator(ops, B, h, ld, rate, gather):
    dev = torch.device('cuda')
    rng = np.random.RandomState(B + h)
    nsrc = B + 13 if gather else B
    x = rng.normal(size=(nsrc, ld)).astype(np.float32)
    xd = torch.as_tensor(x).to(dev)
    out = torch.full((B, ld), 7.0, dtype=torch.float32, device=dev)
    step = torch.tensor([41], dtype=torch.int64, device=dev)
    seed, layer, row0 = 0xfeedfacecafebeef, 3, 1000003
    perm = cursor = None
    rows = np.arange(B)
    if gather:
        order = rng.permutation(nsrc).astype(np.int32)
        perm = torch.as_tensor(order).to(dev)
        cursor = torch.tensor([7], dtype=torch.int64, device=dev)
        rows = order[7:7 + B]
    ops.dropout_apply(xd, ld, perm, cursor, B, h, rate, seed, step, layer, row0, out, ld)
    torch.cuda.synchronize()
    keep = N.dropout_keep(seed, 41, layer, row0, B, h, rate)
    want = np.where(keep, x[rows, :h] * N.dropout_scale(rate), np.float32(0))
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got[:, :h], want)
    assert (got[:, h:] == 7.0).all()                       # padding columns untouched
    if not gather:                                         # in place == the backward call
        ops.dropout_apply(xd, ld, None, None, B, h, rate, seed, step, layer, row0, xd, ld)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(xd.cpu().numpy()[:, :h], want)


def test_bad_arguments_are_refused(ops):
    dev = torch.device('cuda')
    x = torch.zeros(4, 8, device=dev)
    for kw in (dict(rate=1.0), dict(rate=-0.1), dict(h=0), dict(ld=4)):
        a = dict(rate=0.5, h=8, ld=8)
        a.update(kw)
        with pytest.raises(RuntimeError):
            ops.dropout_apply(x, a['ld'], None, None, 4, a['h'], a['rate'], 1, None, 0, 0, x, a['ld'])
    perm = torch.zeros(4, dtype=torch.int32, device=dev)
    cur = torch.zeros(1, dtype=torch.int64, device=dev)
    with pytest.raises(RuntimeError):                      # gather cannot run in place
        ops.dropout_apply(x, 8, perm, cur, 4, 8, 0.5, 1, None, 0, 0, x, 8)


def test_engine_steps_match_oracle(ops):
    C.step_parity(ops)


def test_engine_steps_match_oracle_no_batchnorm_const_disp(ops):
    C.step_parity(ops, ae='zinb', bn=False)


def test_fit_matches_oracle(ops):
    C.fit_parity(ops)


def test_fit_with_step_graphs_matches_eager(ops):
    """The step counter lives in device memory: replayed step graphs draw fresh masks every step."""
    from dca_amd.engine import Engine
    from dca_amd.train import fit_engine
    from helpers import make_problem
    from test_dp_gloo import FixedOrders
    n, G, hs, B, epochs = 200, 80, (16, 8, 16), 32, 2
    X, Y, sf, p = make_problem(n, G, hs, 'zinb-conddisp', True, seed=8)
    n_train = int(n * 0.9)
    orders = [np.random.RandomState(e).permutation(n_train) for e in range(epochs)]
    hist = []
    for use_graph in (False, True):
        eng = Engine('zinb-conddisp', G, G, hs, True, 0.0, ops=ops, hidden_dropout=0.3, input_dropout=0.2, dropout_seed=3)
        eng.set_params(p)
        eng.load_data(X, Y, sf)
        h = fit_engine(eng, n_train, n - n_train, n_train, n - n_train, 0, epochs=epochs, batch_size=B,
                       shuffle_rng=FixedOrders(orders), reduce_lr=1, early_stop=0, use_graph=use_graph)
        hist.append(h.history)
    np.testing.assert_allclose(hist[0]['loss'], hist[1]['loss'], rtol=1e-6)
    np.testing.assert_allclose(hist[0]['val_loss'], hist[1]['val_loss'], rtol=1e-6)


def test_inference_ignores_dropout(ops):
    C.inference_ignores_dropout(ops)
