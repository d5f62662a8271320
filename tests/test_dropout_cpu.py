"""Dropout (dca/network.py:98-99, 137-138): the generator restated in oracle/net_np.py against the
published Random123 known-answer vectors, the oracle's backward through the masks against finite
differences, and the engine (oracle ops) against the oracle."""
import numpy as np
import pytest

from helpers import make_problem, oracle_net
from oracle import net_np as N
from oracle.cpu_ops import CpuRefOps
import _dropout_cases as C


def test_philox_known_answers():
    """Random123 kat_vectors, philox4x32 10 rounds."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kat:
        got = tuple(int(x) for x in N.philox4x32_10(*ctr, *key))
        assert got == want


def test_keep_mask_properties():
    k = N.dropout_keep(7, 3, 1, 0, 2000, 64, 0.3)
    assert abs(k.mean() - 0.7) < 0.01
    # per unit over rows, per row over units: no stuck columns / rows
    assert np.abs(k.mean(axis=0) - 0.7).max() < 0.06 and np.abs(k.mean(axis=1) - 0.7).max() < 0.25
    assert not np.array_equal(k, N.dropout_keep(7, 4, 1, 0, 2000, 64, 0.3))       # step
    assert not np.array_equal(k, N.dropout_keep(7, 3, 2, 0, 2000, 64, 0.3))       # layer
    assert not np.array_equal(k, N.dropout_keep(8, 3, 1, 0, 2000, 64, 0.3))       # seed
    assert not np.array_equal(k, N.dropout_keep(7 + 2 ** 32, 3, 1, 0, 2000, 64, 0.3))   # high key word
    # a rank that starts at global batch row 5 draws rows 5.. of the single-process mask
    np.testing.assert_array_equal(N.dropout_keep(7, 3, 1, 5, 11, 64, 0.3), k[5:16])
    # ragged width: h = 6 uses two Philox blocks per row
    k6 = N.dropout_keep(1, 0, 0, 0, 50, 6, 0.5)
    assert k6.shape == (50, 6)
    # nesting: a unit kept at rate r is kept at every smaller rate (same uniform)
    assert np.all(N.dropout_keep(7, 3, 1, 0, 100, 64, 0.1)[k[:100]])
    assert N.dropout_keep(7, 3, 1, 0, 10, 8, 0.0).all()


@pytest.mark.parametrize('ae,bn', [('zinb-conddisp', True), ('nb', False)])
def test_oracle_backward_through_masks_matches_finite_differences(ae, bn):
    n, G, hs = 12, 9, (6, 4, 6)
    X, Y, sf, p = make_problem(n, G, hs, ae, bn, seed=2)
    net = oracle_net(ae, p, hs, bn, hidden_dropout=[0.3, 0.2, 0.4], input_dropout=0.25, dropout_seed=5)
    net.step = 3
    loss, g = net.loss_and_grads(X, Y, sf)
    rng = np.random.RandomState(0)
    for name in ('W0', 'W1', 'beta0' if bn else 'b0', 'W_mean', 'b_mean'):
        for _ in range(3):
            idx = tuple(rng.randint(0, s) for s in net.p[name].shape)
            eps = 1e-6
            old = net.p[name][idx]
            vals = []
            for d in (eps, -eps):
                net.p[name][idx] = old + d
                net.step = 3                                    # same masks
                saved = {k: net.p[k].copy() for k in net.p if k.startswith(('mm', 'mv'))}
                vals.append(net.loss_and_grads(X, Y, sf)[0])
                net.p.update(saved)
            net.p[name][idx] = old
            fd = (vals[0] - vals[1]) / (2 * eps)
            assert abs(fd - g[name][idx]) < 1e-5 * max(1.0, abs(fd)) + 1e-8, (name, idx, fd, g[name][idx])


def test_engine_steps_match_oracle():
    C.step_parity(CpuRefOps())


def test_engine_steps_match_oracle_no_batchnorm_const_disp():
    C.step_parity(CpuRefOps(), ae='zinb', bn=False)


def test_fit_matches_oracle():
    h = C.fit_parity(CpuRefOps())
    assert np.isfinite(h['loss']).all()


def test_inference_ignores_dropout():
    C.inference_ignores_dropout(CpuRefOps())


def test_rates_validated():
    from dca_amd.engine import Engine
    with pytest.raises(AssertionError):
        Engine('zinb', 10, hidden_size=(4, 2, 4), ops=CpuRefOps(), hidden_dropout=1.0)
    with pytest.raises(AssertionError):
        Engine('zinb', 10, hidden_size=(4, 2, 4), ops=CpuRefOps(), hidden_dropout=[0.1, 0.1])


def test_api_accepts_dropout():
    """dca(adata, hidden_dropout=..) / network_kwds input_dropout reach the engine (the reference builds
    Dropout layers from the same arguments, api.py:25,78-85)."""
    import pandas as pd
    from conftest import synth_counts
    from dca_amd.api import dca
    from dca_amd._anndata import AnnData
    from dca_amd.network import override_ops
    n, G = 80, 30
    ad = AnnData(synth_counts(n, G, 3).astype(np.float32), obs=pd.DataFrame(index=['c%d' % i for i in range(n)]),
                 var=pd.DataFrame(index=['g%d' % i for i in range(G)]))
    with override_ops(CpuRefOps):
        out, net = dca(ad, ae_type='zinb-conddisp', hidden_size=(8, 4, 8), hidden_dropout=0.2, epochs=2, batch_size=16,
                       network_kwds={'input_dropout': 0.1}, return_model=True, copy=True, verbose=False)
    assert net.engine.drop == [0.2, 0.2, 0.2] and net.engine.in_drop == 0.1
    assert int(net.engine.drop_iter.item()) > 0
    assert np.isfinite(out.X).all()
