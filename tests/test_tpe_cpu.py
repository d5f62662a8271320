"""dca_amd/tpe.py: hyperopt's TPE (the algo of dca/hyper.py:97-104) restated -- its building blocks against hand-computed
values of the published algorithm, and the search behaviour on a synthetic objective over the reference's space."""
import numpy as np

from dca_amd import hyper as H
from dca_amd import tpe


def test_linear_forgetting_and_parzen_estimator():
    assert tpe.linear_forgetting_weights(0).size == 0
    np.testing.assert_array_equal(tpe.linear_forgetting_weights(7), np.ones(7))
    w = tpe.linear_forgetting_weights(30)                       # 5 old observations ramp 1/30 .. 1, the newest 25 count fully
    np.testing.assert_allclose(w[:5], np.linspace(1 / 30, 1, 5)); assert (w[5:] == 1).all()
    # no observation: the prior alone
    w, m, s = tpe.adaptive_parzen_normal([], 0.5, 1.0)
    assert w.tolist() == [1.0] and m.tolist() == [0.5] and s.tolist() == [1.0]
    # one observation: half the prior's width beside the prior
    w, m, s = tpe.adaptive_parzen_normal([0.2], 0.5, 1.0)
    assert w.tolist() == [0.5, 0.5] and m.tolist() == [0.2, 0.5] and s.tolist() == [0.5, 1.0]
    # several: sorted, each width = the larger gap to a neighbour, clipped to [prior / min(100, 1 + n), prior]
    w, m, s = tpe.adaptive_parzen_normal([0.2, 0.9, 0.4], 0.5, 1.0)
    assert m.tolist() == [0.2, 0.4, 0.5, 0.9] and w.tolist() == [0.25] * 4
    np.testing.assert_allclose(s, [0.2, 0.2, 1.0, 0.4])
    w, m, s = tpe.adaptive_parzen_normal(np.linspace(0.40, 0.41, 200), 0.5, 1.0)
    assert s.min() == 1.0 / 100 and s.max() == 1.0              # dense observations: the floor of the width
    np.testing.assert_allclose(w.sum(), 1.0)
    assert w[np.argmax(s)] == w.max() and w.min() < 0.01 * w.max()          # the prior counts like a recent observation; the oldest fade
    # the truncated mixture integrates to one
    x = np.linspace(0, 1, 20001)
    p = np.exp(tpe.gmm_lpdf(x, *tpe.adaptive_parzen_normal([0.2, 0.9, 0.4], 0.5, 1.0), 0.0, 1.0))
    np.testing.assert_allclose(np.trapezoid(p, x), 1.0, rtol=1e-4)
    rng = np.random.RandomState(0)
    d = tpe.gmm_sample(rng, *tpe.adaptive_parzen_normal([0.2, 0.21, 0.22], 0.5, 1.0), 0.0, 1.0, 4000)
    assert d.min() >= 0 and d.max() <= 1 and 0.15 < np.median(d) < 0.45
    # categorical: observation counts (with forgetting weights) + one pseudo-count per option
    np.testing.assert_allclose(tpe.categorical_posterior([], 4), [0.25] * 4)
    np.testing.assert_allclose(tpe.categorical_posterior([1, 1, 3], 4), np.array([1, 3, 1, 2]) / 7.0)


def test_split_takes_the_best_quarter_root_in_trial_order():
    t = tpe.TPE({'x': ('uniform', 0, 1)})
    hist = [({'x': i / 40.0}, float((i * 7) % 40)) for i in range(40)] + [({'x': 0.5}, None), ({'x': 0.6}, float('nan'))]
    below, above = t.split(hist)
    assert len(below) == int(np.ceil(0.25 * np.sqrt(40))) == 2 and len(above) == 38      # failed trials do not count
    assert [p['x'] for p in below] == [0.0, 23 / 40.0]                                   # losses 0 and 1, in trial order


def test_search_over_the_reference_space():
    """20 random start-up proposals (identical to a pure random search with the same seed), then proposals that move
    towards the optimum of a synthetic objective in every kind of dimension -- log-uniform, uniform, choice."""
    def loss(v):
        p = H.to_params(v)['model']
        return (np.log10(p['lr']) + 2.3) ** 2 + 0.05 * (np.log10(p['ridge']) + 5) ** 2 + (p['dropout'] - 0.1) ** 2 \
            + 0.4 * (p['activation'] != 'elu') + 0.3 * (p['hidden_size'] != (32, 16, 32))
    runs = {}
    for name, kw in (('tpe', {}), ('random', {'n_startup': 10 ** 9})):
        t, hist = tpe.TPE(H.SPACE, seed=3, **kw), []
        for _ in range(150):
            v = t.suggest(hist)
            assert set(v) == set(H.SPACE)
            assert 1e-3 <= v['m_lr'] <= 1e-2 and 1e-7 <= v['m_ridge'] <= 1e-1 and 0 <= v['m_do'] <= 0.7 and 0 <= v['m_input_do'] <= 0.8
            assert 0 <= v['m_hiddensize'] < len(H.HIDDEN_SIZES) and isinstance(v['m_activation'], int)
            hist.append((v, loss(v)))
        runs[name] = hist
    assert [h[0] for h in runs['tpe'][:20]] == [h[0] for h in runs['random'][:20]]
    lt = np.array([l for _, l in runs['tpe']]); lr = np.array([l for _, l in runs['random']])
    assert lt[100:].mean() < 0.65 * lr[100:].mean() and lt.min() <= lr.min()
    late = [H.to_params(v)['model'] for v, _ in runs['tpe'][100:]]
    assert np.mean([p['activation'] == 'elu' for p in late]) >= 0.4             # 1 of 6 options under random search
    assert np.mean([p['hidden_size'] == (32, 16, 32) for p in late]) >= 0.3     # 1 of 9
    # a failed trial is skipped, not fatal
    hist = runs['tpe'][:60] + [(runs['tpe'][60][0], None)]
    assert set(tpe.TPE(H.SPACE, seed=5).suggest(hist)) == set(H.SPACE)


def test_choice_indices_follow_hyperopt_labels():
    v = {'d_norm_log': 0, 'd_norm_zeromean': 1, 'd_norm_sf': 0, 'm_lr': 2e-3, 'm_ridge': 1e-4, 'm_l1_enc_coef': 1e-5,
         'm_hiddensize': 1, 'm_activation': 3, 'm_aetype': 1, 'm_batchnorm': 1, 'm_do': 0.3, 'm_input_do': 0.1}
    p = H.to_params(v)
    assert p['data'] == {'norm_input_log': True, 'norm_input_zeromean': False, 'norm_input_sf': True}     # hp.choice(.., (True, False))
    assert p['model']['hidden_size'] == (32, 16, 32) and p['model']['activation'] == 'PReLU' and p['model']['aetype'] == 'zinb-conddisp'
    assert p['model']['batchnorm'] is False and p['model']['dropout'] == 0.3
