"""Host logic of the engine (buffer layout, step orchestration, fit loop) on CPU, with the
oracle-backed ops injected, against the straight-line oracle (oracle/net_np.py)."""
import numpy as np
import pytest

from oracle import net_np as N
from oracle.cpu_ops import CpuRefOps
from helpers import make_problem, oracle_net, make_engine, assert_grads_close, run_single_step


@pytest.mark.parametrize('ae_type', N.AE_TYPES)
@pytest.mark.parametrize('batchnorm', [True, False])
@pytest.mark.parametrize('n,G,hs', [(40, 30, (8, 4, 8)), (20, 6, (1,)), (70, 45, (16, 5))])
def test_single_step_matches_oracle(ae_type, batchnorm, n, G, hs):
    if ae_type in N.FORK_HEADS and len(hs) - 1 <= len(hs) // 2:
        pytest.skip('fork networks need a hidden layer behind the centre')
    ridge = 0.03 if ae_type.startswith('zinb') else 0.0
    X, Y, sf, p = make_problem(n, G, hs, ae_type, batchnorm, seed=n)
    rows = np.random.RandomState(1).permutation(n)[:min(n - 3, 33)]
    ref = oracle_net(ae_type, p, hs, batchnorm, ridge)
    rl, rg = ref.loss_and_grads(X[rows].astype(np.float64), Y[rows].astype(np.float64),
                                sf[rows].astype(np.float64))
    ms = {}
    N.rmsprop_step(ref.p, rg, ms, 1e-3)
    eng = make_engine(CpuRefOps(), ae_type, G, hs, batchnorm, ridge, p, X, Y, sf)
    loss, g, newp = run_single_step(eng, rows)
    assert abs(loss - rl) < 2e-6 * abs(rl)
    assert_grads_close(g, rg, rtol=1e-4, atol_scale=1e-6)
    for k in ref.p:                                   # parameters after clip + RMSprop
        if batchnorm and k[0] == 'b' and k[1:].isdigit():
            # bias in front of BatchNormalization: its gradient is round-off noise, and Keras' RMSprop (epsilon outside
            # the root) moves a parameter by lr g / (0.32 |g| + 1e-7): noise of 1e-8 is a step of 1e-4, and no step
            # exceeds lr / sqrt(1 - rho) = 3.2 lr
            assert np.abs(newp[k] - ref.p[k]).max() < 2 * 3.2e-3, k
            continue
        np.testing.assert_allclose(newp[k], ref.p[k], rtol=2e-4, atol=2e-6, err_msg=k)


@pytest.mark.parametrize('ae_type,hs', [('zinb-conddisp', (128, 16, 70)), ('nb', (130,)), ('zinb-fork', (128, 8, 66))])
def test_wide_network_at_a_throughput_batch_takes_the_transposed_products(ae_type, hs):
    """Decoder wider than 64 units (heads as separate kernels) at B >= 256: the head weights, the last activations and
    (first layer >= 128 units) the minibatch of X are transposed once per step and every product runs with A contiguous
    along the contraction (engine._wide_transposed) -- same numbers as the straight-line oracle."""
    n, G, B = 300, 37, 262
    X, Y, sf, p = make_problem(n, G, hs, ae_type, True, seed=11)
    rows = np.random.RandomState(2).permutation(n)[:B]
    ref = oracle_net(ae_type, p, hs, True, 0.0)
    rl, rg = ref.loss_and_grads(X[rows].astype(np.float64), Y[rows].astype(np.float64), sf[rows].astype(np.float64))
    eng = make_engine(CpuRefOps(), ae_type, G, hs, True, 0.0, p, X, Y, sf)
    loss, g, _ = run_single_step(eng, rows)
    assert eng.WhT is not None and eng.HT is not None and eng.XT is not None and eng.ws_heads is None
    assert abs(loss - rl) < 2e-6 * abs(rl)
    assert_grads_close(g, rg, rtol=1e-4, atol_scale=1e-6)


@pytest.mark.parametrize('ae_type', ['zinb-conddisp', 'nb'])
def test_fit_loop_matches_oracle(ae_type):
    from dca_amd.train import fit_engine
    n, G, hs = 75, 20, (6, 3, 6)
    X, Y, sf, p = make_problem(n, G, hs, ae_type, True, seed=4)
    ref = oracle_net(ae_type, p, hs, True)
    rh = N.fit(ref, X.astype(np.float64), Y.astype(np.float64), sf.astype(np.float64), epochs=4,
               batch_size=16, shuffle_rng=np.random.RandomState(9), reduce_lr=1, early_stop=3)
    eng = make_engine(CpuRefOps(), ae_type, G, hs, True, 0.0, p, X, Y, sf)
    n_train = int(n * 0.9)
    h = fit_engine(eng, n_train, n - n_train, n_train, n - n_train, 0, epochs=4, batch_size=16,
                   shuffle_rng=np.random.RandomState(9), reduce_lr=1, early_stop=3)
    assert len(h.history['loss']) == len(rh['loss'])
    # (fp32 buffers against the fp64 oracle over 4 epochs: Keras' RMSprop -- epsilon outside the root -- takes sign-like
    # steps while an accumulator is small, so fp32 noise in small gradients becomes O(lr) in those parameters: BASELINE's
    # per-epoch bound of 1e-4 is the scale on which an fp32 fit follows the fp64 one; the validation loss of this tiny
    # problem rests on 8 cells: 5e-4, the bound the other sign-normalising optimizers are held to)
    np.testing.assert_allclose(h.history['loss'], rh['loss'], rtol=1e-4)
    np.testing.assert_allclose(h.history['val_loss'], rh['val_loss'], rtol=5e-4)
    np.testing.assert_allclose(h.history['lr'], rh['lr'], rtol=1e-7)
    newp = eng.get_params()
    for k in ref.p:
        if k[0] == 'b' and k[1:].isdigit():
            continue      # bias in front of BatchNorm: gradient = round-off noise, which RMSprop's sign-like early steps amplify
        np.testing.assert_allclose(newp[k], ref.p[k], rtol=5e-3, atol=2e-5, err_msg=k)


def test_debug_flag_checks_every_step_for_non_finite_values():
    """--debug (dca/__main__.py:111-113, dca/loss.py:90-100): the fit loop runs eagerly and checks loss, gradients and
    parameters after every step.  A clean fit is unchanged by it; a parameter poisoned with NaN fails at the first step with
    the tensors named."""
    from dca_amd.train import fit_engine
    n, G, hs = 75, 20, (6, 3, 6)
    X, Y, sf, p = make_problem(n, G, hs, 'zinb-conddisp', True, seed=4)
    n_train = int(n * 0.9)
    hists = []
    for debug in (False, True):
        eng = make_engine(CpuRefOps(), 'zinb-conddisp', G, hs, True, 0.0, p, X, Y, sf)
        hists.append(fit_engine(eng, n_train, n - n_train, n_train, n - n_train, 0, epochs=2, batch_size=16,
                                shuffle_rng=np.random.RandomState(9), debug=debug).history)
    assert hists[0]['loss'] == hists[1]['loss'] and hists[0]['val_loss'] == hists[1]['val_loss']
    eng = make_engine(CpuRefOps(), 'zinb-conddisp', G, hs, True, 0.0, p, X, Y, sf)
    eng.lay.view(eng.w, 'bh')[3] = float('nan')
    with pytest.raises(FloatingPointError, match=r'has inf/nans \(epoch 1, step 1\)') as ei:
        fit_engine(eng, n_train, n - n_train, n_train, n - n_train, 0, epochs=2, batch_size=16,
                   shuffle_rng=np.random.RandomState(9), debug=True)
    assert 'loss' in str(ei.value) and 'bh' in str(ei.value)


def test_predict_matches_oracle():
    n, G, hs = 33, 26, (8, 4, 8)
    for ae_type in N.AE_TYPES:
        X, Y, sf, p = make_problem(n, G, hs, ae_type, True, seed=8)
        p['mm1'] = np.random.RandomState(0).normal(0, .3, 4).astype(np.float32)
        p['mv1'] = np.random.RandomState(1).uniform(.5, 2, 4).astype(np.float32)
        ref = oracle_net(ae_type, p, hs, True).predict(X.astype(np.float64), sf.astype(np.float64))
        eng = make_engine(CpuRefOps(), ae_type, G, hs, True, 0.0, p, X, None, sf)
        eng.reserve(16)
        want = {'mean', 'latent'}
        if 'disp' in eng.lay.heads:
            want.add('dispersion')
        if 'pi' in eng.lay.heads:
            want.add('dropout')
        for s in range(0, n, 16):
            b = min(16, n - s)
            out = eng.predict_chunk(s, b, want)
            for k in want:
                np.testing.assert_allclose(out[k].numpy(), ref[k][s:s + b], rtol=2e-5, atol=1e-6, err_msg=k)
        if eng.lay.const_disp:
            np.testing.assert_allclose(eng.const_dispersion(), ref['dispersion'], rtol=1e-6)


@pytest.mark.parametrize('optimizer', __import__('_opt_cases').OPTIMIZERS)
def test_other_optimizers_fit_matches_oracle(optimizer):
    """train.py:54-57 picks the Keras optimizer by name: host logic (slots, step counter, default
    learning rate) against the oracle's restatement of the tf.keras updates."""
    from _opt_cases import run_fit_parity
    run_fit_parity(CpuRefOps(), optimizer=optimizer)


@pytest.mark.parametrize('reg', __import__('_opt_cases').REG_CASES)
def test_l1_l2_regularisers_fit_matches_oracle(reg):
    """network.py:114-126: l1/l2 (encoder-specific when given) on the Dense kernels: penalty in the
    reported loss and val_loss, sign/2w terms in the gradients before clipvalue."""
    from _opt_cases import run_fit_parity
    run_fit_parity(CpuRefOps(), reg=reg)
    run_fit_parity(CpuRefOps(), optimizer='Adam', reg=reg, ae_type='nb')


@pytest.mark.parametrize('batchnorm', [True, False])
@pytest.mark.parametrize('activation', ['linear', 'tanh', 'sigmoid', 'elu', 'selu', 'softplus', 'softsign', 'LeakyReLU'])
def test_activations_single_step_matches_oracle(activation, batchnorm):
    """Activation(self.activation) / LeakyReLU of dca/network.py:132-135: forward, the slope taken
    from the forward output in the backward kernels, moving statistics."""
    n, G, hs, B = 90, 33, (12, 5, 12), 40
    X, Y, sf, p = make_problem(n, G, hs, 'zinb-conddisp', batchnorm, seed=6)
    rows = np.random.RandomState(1).permutation(n)[:B]
    ref = oracle_net('zinb-conddisp', p, hs, batchnorm, activation=activation)
    rl, rg = ref.loss_and_grads(X[rows].astype(np.float64), Y[rows].astype(np.float64), sf[rows].astype(np.float64))
    eng = make_engine(CpuRefOps(), 'zinb-conddisp', G, hs, batchnorm, 0.0, p, X, Y, sf, activation=activation)
    loss, g, newp = run_single_step(eng, rows)
    assert abs(loss - rl) < 1e-5 * abs(rl), (loss, rl)
    assert_grads_close(g, rg)
    N.rmsprop_step(ref.p, rg, {}, 1e-3)                       # the engine's step included the update
    out_ref = ref.predict(X[:16].astype(np.float64), sf[:16].astype(np.float64))
    out = eng.predict_chunk(0, 16, {'mean', 'latent'})
    for k in ('mean', 'latent'):
        np.testing.assert_allclose(out[k].cpu().numpy(), out_ref[k], rtol=2e-3, atol=2e-4, err_msg=k)


def test_keras_initializers_have_the_keras_distributions():
    """kernel_initializer=self.init (network.py:57,124-126): limits / variances of tf.keras 2.x."""
    from dca_amd.engine import keras_initializer, KERAS_INITIALIZERS
    fi, fo = 400, 300
    rng = np.random.RandomState(0)
    lim = {'glorot_uniform': np.sqrt(6.0 / (fi + fo)), 'he_uniform': np.sqrt(6.0 / fi), 'lecun_uniform': np.sqrt(3.0 / fi),
           'random_uniform': 0.05}
    std = {'glorot_normal': np.sqrt(2.0 / (fi + fo)), 'he_normal': np.sqrt(2.0 / fi), 'lecun_normal': np.sqrt(1.0 / fi),
           'random_normal': 0.05}
    for name in KERAS_INITIALIZERS:
        w = keras_initializer(name, rng, fi, fo)
        assert w.shape == (fi, fo) and w.dtype == np.float32
        if name in lim:
            assert np.abs(w).max() <= lim[name] and np.abs(w).max() > 0.99 * lim[name]
            assert abs(w.std() - lim[name] / np.sqrt(3)) < 0.01 * lim[name]
        if name in std:
            assert abs(w.std() - std[name]) < 0.02 * std[name], name
            if name != 'random_normal':                        # truncated at 2 sigma of the untruncated normal
                assert np.abs(w).max() <= 2.0 * std[name] / .87962566103423978 + 1e-6
    w = keras_initializer('truncated_normal', rng, fi, fo)
    assert np.abs(w).max() <= 0.1 + 1e-7 and abs(w.std() - 0.05 * .87962566103423978) < 1e-3
    q = keras_initializer('orthogonal', rng, fi, fo)
    np.testing.assert_allclose(q.T @ q, np.eye(fo), atol=1e-5)
    q = keras_initializer('orthogonal', rng, fo, fi)
    np.testing.assert_allclose(q @ q.T, np.eye(fo), atol=1e-5)
    assert (keras_initializer('zeros', rng, 3, 4) == 0).all() and (keras_initializer('ones', rng, 3, 4) == 1).all()
    with pytest.raises(NotImplementedError):
        keras_initializer('no_such_init', rng, 3, 4)
    from dca_amd.engine import Engine
    eng = Engine('zinb', 20, hidden_size=(8, 4, 8), ops=CpuRefOps())
    eng.init_params(1, 'he_normal')
    p = eng.get_params()
    assert abs(p['W0'].std() - np.sqrt(2.0 / 20)) < 0.08 and (p['b0'] == 0).all() and (p['theta_w'] == 0).all()


@pytest.mark.parametrize('ae_type', ['nb-shared', 'zinb-shared'])
def test_shared_heads_oracle_gradient_matches_finite_differences(ae_type):
    """Dense(1) dispersion / dropout broadcast over genes (network.py:343-362, 464-491): the oracle's
    analytic gradient of the scalar units against central differences of its own loss (ridge on)."""
    n, G, hs = 10, 7, (5, 3, 5)
    X, Y, sf, p = make_problem(n, G, hs, ae_type, False, seed=3)
    assert p['W_disp'].shape == (5, 1) and p['b_disp'].shape == (1,)
    net = oracle_net(ae_type, p, hs, False, ridge=0.05)
    loss, g = net.loss_and_grads(X, Y, sf)
    names = ['W_disp', 'b_disp', 'W_mean', 'W2'] + (['W_pi', 'b_pi'] if ae_type == 'zinb-shared' else [])
    for name in names:
        for idx in np.ndindex(*net.p[name].shape):
            if idx[0] > 2 or (len(idx) > 1 and idx[1] > 2):
                continue
            old, eps, vals = net.p[name][idx], 1e-6, []
            for d in (eps, -eps):
                net.p[name][idx] = old + d
                vals.append(net.loss_and_grads(X, Y, sf)[0])
            net.p[name][idx] = old
            fd = (vals[0] - vals[1]) / (2 * eps)
            assert abs(fd - g[name][idx]) < 1e-6 * max(1.0, abs(fd)) + 1e-9, (name, idx, fd, g[name][idx])


def test_shared_heads_api_and_outputs(tmp_path):
    import pandas as pd
    from conftest import synth_counts
    from dca_amd.api import dca
    from dca_amd._anndata import AnnData
    from dca_amd.network import override_ops
    n, G = 60, 25
    ad = AnnData(synth_counts(n, G, 5).astype(np.float32), obs=pd.DataFrame(index=['c%d' % i for i in range(n)]),
                 var=pd.DataFrame(index=['g%d' % i for i in range(G)]))
    with override_ops(CpuRefOps):
        for ae in ('nb-shared', 'zinb-shared'):
            out, net = dca(ad, ae_type=ae, hidden_size=(8, 4, 8), epochs=2, batch_size=16, return_info=True,
                           return_model=True, copy=True, verbose=False)
            assert out.obsm['X_dca_dispersion'].shape == (n, 1)
            assert (out.obsm['X_dca_dispersion'] > 0).all() and np.isfinite(out.X).all()
            if ae == 'zinb-shared':
                d = out.obsm['X_dca_dropout']
                assert d.shape == (n, 1) and ((d > 0) & (d < 1)).all()
            net.write(out, str(tmp_path / ae), mode='denoise')
            row = open(str(tmp_path / ae / 'dispersion.tsv')).read().strip().split('\t')
            assert len(row) == n


@pytest.mark.parametrize('ae_type', ['nb-fork', 'zinb-fork'])
def test_fork_oracle_gradient_matches_finite_differences(ae_type):
    """network.py:553-760 restated with explicit branches: each head reads only its own slice of the last layer."""
    n, G, hs = 10, 7, (6, 4, 3, 4, 5)
    X, Y, sf, p = make_problem(n, G, hs, ae_type, True, seed=3)
    nh = len(N.FORK_HEADS[ae_type])
    assert p['W3'].shape == (3, nh * 5) and p['W_mean'].shape == (5, G) and 'W4' not in p     # centre -> branches directly
    net = oracle_net(ae_type, p, hs, True, ridge=0.02)
    loss, g = net.loss_and_grads(X, Y, sf)
    rng = np.random.RandomState(0)
    for name in ('W3', 'beta3', 'W2', 'W_mean', 'W_disp', 'b_disp', 'W0'):
        for _ in range(4):
            idx = tuple(rng.randint(0, s) for s in net.p[name].shape)
            old, eps, vals = net.p[name][idx], 1e-6, []
            for d in (eps, -eps):
                net.p[name][idx] = old + d
                saved = {k: net.p[k].copy() for k in net.p if k.startswith(('mm', 'mv'))}
                vals.append(net.loss_and_grads(X, Y, sf)[0])
                net.p.update(saved)
            net.p[name][idx] = old
            fd = (vals[0] - vals[1]) / (2 * eps)
            assert abs(fd - g[name][idx]) < 2e-6 * max(1.0, abs(fd)) + 1e-9, (name, idx, fd, g[name][idx])
    # a head's weights see nothing of the other branches
    h = net.forward(X, sf, training=False)
    net.p['W3'][:, 5:10] += 1.0                       # disturb the dispersion branch only
    h2 = net.forward(X, sf, training=False)
    np.testing.assert_array_equal(h['a_mean'], h2['a_mean'])
    assert not np.allclose(h['a_disp'], h2['a_disp'])


def test_fork_deep_network_with_dropout_and_regularisers_matches_oracle():
    ae, n, G, hs, B = 'zinb-fork', 64, 40, (16, 8, 4, 8, 12), 32
    drop = dict(hidden_dropout=[0.1, 0.0, 0.2, 0.5, 0.3], input_dropout=0.1, dropout_seed=5)
    reg = (1e-4, 2e-4, 3e-4, 0.)
    X, Y, sf, p = make_problem(n, G, hs, ae, True, seed=6)
    ref = N.OracleAE(ae, {k: np.asarray(v, np.float64).copy() for k, v in p.items()}, hs, True, 0.01, reg, **drop)
    assert ref.hidden_size == (16, 8, 4, 36) and ref.hidden_dropout == [0.1, 0.0, 0.2, 0.3]
    eng = make_engine(CpuRefOps(), ae, G, hs, True, 0.01, p, X, Y, sf, **drop)
    eng.set_regularizers(*reg)
    assert eng.lay.hidden == (16, 8, 4, 36) and eng.drop == [0.1, 0.0, 0.2, 0.3] and eng.center == 2
    rows = np.random.RandomState(1).permutation(n)[:B]
    rl, rg = ref.loss_and_grads(X[rows], Y[rows], sf[rows])
    loss, g, _ = run_single_step(eng, rows)
    assert abs(loss - rl) < 1e-5 * abs(rl)
    assert_grads_close(g, rg)
    out = eng.predict_chunk(0, 8, {'latent'})
    assert out['latent'].shape == (8, 4)


def test_fork_needs_a_decoder_layer_and_api_runs():
    import pandas as pd
    from conftest import synth_counts
    from dca_amd.engine import Engine
    from dca_amd.api import dca
    from dca_amd._anndata import AnnData
    from dca_amd.network import override_ops
    with pytest.raises(ValueError):
        Engine('zinb-fork', 10, hidden_size=(4,), ops=CpuRefOps())
    with pytest.raises(ValueError):
        Engine('nb-fork', 10, hidden_size=(4, 2), ops=CpuRefOps())
    n, G = 60, 25
    ad = AnnData(synth_counts(n, G, 5).astype(np.float32), obs=pd.DataFrame(index=['c%d' % i for i in range(n)]),
                 var=pd.DataFrame(index=['g%d' % i for i in range(G)]))
    with override_ops(CpuRefOps):
        for ae in ('nb-fork', 'zinb-fork'):
            out = dca(ad, ae_type=ae, hidden_size=(8, 4, 8), epochs=2, batch_size=16, return_info=True, copy=True,
                      verbose=False)
            assert out.obsm['X_dca_dispersion'].shape == (n, G) and np.isfinite(out.X).all()


def test_elempi_oracle_gradient_matches_finite_differences():
    """ZINBAutoencoderElemPi (network.py:424-461): mean = MeanAct(-Dense), pi = sigmoid(k * (-Dense) + c)."""
    ae, n, G, hs = 'zinb-elempi', 10, 7, (5, 3, 5)
    X, Y, sf, p = make_problem(n, G, hs, ae, False, seed=3)
    assert p['pi_k'].shape == (G,) and p['pi_c'].shape == (G,) and 'W_pi' not in p
    net = oracle_net(ae, p, hs, False, ridge=0.05)
    c = net.forward(X, sf, training=False)
    np.testing.assert_allclose(c['a_mean'], -(c['H'][-1] @ net.p['W_mean'] + net.p['b_mean']))
    np.testing.assert_allclose(c['a_pi'], c['a_mean'] * net.p['pi_k'] + net.p['pi_c'])
    loss, g = net.loss_and_grads(X, Y, sf)
    for name in ('pi_k', 'pi_c', 'W_mean', 'b_mean', 'W_disp', 'W1'):
        for idx in list(np.ndindex(*net.p[name].shape))[:6]:
            old, eps, vals = net.p[name][idx], 1e-6, []
            for d in (eps, -eps):
                net.p[name][idx] = old + d
                vals.append(net.loss_and_grads(X, Y, sf)[0])
            net.p[name][idx] = old
            fd = (vals[0] - vals[1]) / (2 * eps)
            assert abs(fd - g[name][idx]) < 1e-6 * max(1.0, abs(fd)) + 1e-9, (name, idx, fd, g[name][idx])


def test_elempi_regularised_fit_and_sharedpi():
    from _opt_cases import run_fit_parity
    run_fit_parity(CpuRefOps(), optimizer='RMSprop', reg=(1e-4, 2e-4, 0., 0.), ae_type='zinb-elempi')
    # sharedpi: ElementwiseDense(1) -> one (k, c) pair for all genes: entries stay equal, gradient = the total
    from dca_amd.engine import Engine
    n, G, hs = 40, 12, (6, 3, 6)
    X, Y, sf, p = make_problem(n, G, hs, 'zinb-elempi', True, seed=2)
    p['pi_k'] = np.full(G, 0.3); p['pi_c'] = np.full(G, -0.1)
    ref = oracle_net('zinb-elempi', p, hs, True)
    _, rg = ref.loss_and_grads(X[:16], Y[:16], sf[:16])
    eng = Engine('zinb-elempi', G, G, hs, True, 0.0, ops=CpuRefOps(), sharedpi=True)
    eng.set_params(p)
    eng.load_data(X, Y, sf)
    _, g, newp = run_single_step(eng, np.arange(16))
    np.testing.assert_allclose(g['pi_k'], np.full(G, rg['pi_k'].sum()), rtol=1e-4)
    np.testing.assert_allclose(g['pi_c'], np.full(G, rg['pi_c'].sum()), rtol=1e-4)
    assert np.ptp(newp['pi_k']) == 0 and np.ptp(newp['pi_c']) == 0
    eng.init_params(3)
    assert np.ptp(eng.get_params()['pi_k']) == 0
    with pytest.raises(NotImplementedError):
        eng.set_regularizers(1e-3, 0., 0., 0.)


def test_every_ae_type_of_the_reference_builds_and_trains():
    """AE_types (network.py:763-768): all 11 keys resolve, train and predict through dca()."""
    import pandas as pd
    from conftest import synth_counts
    from dca_amd.api import dca
    from dca_amd._anndata import AnnData
    from dca_amd.network import override_ops, AE_types
    assert set(AE_types) == {'normal', 'poisson', 'nb', 'nb-conddisp', 'nb-shared', 'nb-fork', 'zinb', 'zinb-conddisp',
                             'zinb-shared', 'zinb-fork', 'zinb-elempi'}
    n, G = 48, 20
    ad = AnnData(synth_counts(n, G, 9).astype(np.float32), obs=pd.DataFrame(index=['c%d' % i for i in range(n)]),
                 var=pd.DataFrame(index=['g%d' % i for i in range(G)]))
    with override_ops(CpuRefOps):
        for ae in AE_types:
            out = dca(ad, ae_type=ae, hidden_size=(8, 4, 8), epochs=1, batch_size=16, return_info=True, copy=True,
                      verbose=False)
            assert np.isfinite(out.X).all(), ae
            assert len(out.uns['dca_loss_history']['loss']) == 1


@pytest.mark.parametrize('batchnorm', [True, False])
def test_prelu_oracle_finite_differences_and_engine_parity(batchnorm):
    """activation='PReLU' (network.py:132-133): trainable slope per unit behind every hidden layer."""
    ae, n, G, hs = 'zinb-conddisp', 24, 15, (8, 4, 8)
    X, Y, sf, p = make_problem(n, G, hs, ae, batchnorm, seed=4)
    N.add_prelu_params(p, ae, hs)
    rng = np.random.RandomState(2)
    for i in range(3):
        p['alpha%d' % i] = rng.normal(0.1, 0.3, p['alpha%d' % i].shape)       # both signs
    net = oracle_net(ae, p, hs, batchnorm, activation='PReLU')
    loss, g = net.loss_and_grads(X, Y, sf)
    for name in ('alpha0', 'alpha1', 'alpha2', 'W0', 'W2', 'W_pi'):
        for idx in list(np.ndindex(*net.p[name].shape))[:5]:
            old, eps, vals = net.p[name][idx], 1e-6, []
            for d in (eps, -eps):
                net.p[name][idx] = old + d
                saved = {k: net.p[k].copy() for k in net.p if k.startswith(('mm', 'mv'))}
                vals.append(net.loss_and_grads(X, Y, sf)[0])
                net.p.update(saved)
            net.p[name][idx] = old
            fd = (vals[0] - vals[1]) / (2 * eps)
            assert abs(fd - g[name][idx]) < 2e-6 * max(1.0, abs(fd)) + 1e-9, (name, idx, fd, g[name][idx])
    from _dropout_cases import DROP
    ref = oracle_net(ae, p, hs, batchnorm, activation='PReLU', **DROP)
    eng = make_engine(CpuRefOps(), ae, G, hs, batchnorm, 0.0, p, X, Y, sf, activation='PReLU', **DROP)
    rows = np.arange(16)
    rl, rg = ref.loss_and_grads(X[rows], Y[rows], sf[rows])
    loss, g, _ = run_single_step(eng, rows)
    assert abs(loss - rl) < 1e-5 * abs(rl)
    assert_grads_close(g, rg)
    assert (make_engine(CpuRefOps(), ae, G, hs, batchnorm, 0.0, {}, X, Y, sf, activation='PReLU').get_params()['alpha1'] == 0).all()
    # inference (fresh copies: the step above moved the engine's weights)
    eng2 = make_engine(CpuRefOps(), ae, G, hs, batchnorm, 0.0, p, X, Y, sf, activation='PReLU')
    eng2.reserve(8)
    out = eng2.predict_chunk(0, 8, {'mean', 'latent'})
    want = oracle_net(ae, p, hs, batchnorm, activation='PReLU').predict(X[:8], sf[:8])
    np.testing.assert_allclose(out['mean'].numpy()[:, :G], want['mean'], rtol=2e-4)
    np.testing.assert_allclose(out['latent'].numpy(), want['latent'], rtol=2e-4, atol=1e-5)


def test_reseeded_epoch_harness_on_the_oracle_backed_ops():
    """tests/helpers.py::run_reseeded_epoch (the GPU suite walks a whole C3 epoch with it, every step restarted from the
    fp64 oracle's state) on the oracle-backed ops at a small size: an engine whose arithmetic IS the oracle's fp32 twin must
    pass every per-step statement -- and a step taken from a perturbed state must fail them (the harness has teeth)."""
    import torch
    from helpers import run_reseeded_epoch
    from oracle.torch_ref import TorchAE
    n, G, hs, B = 150, 60, (16, 8, 16), 32
    X, Y, sf, p = make_problem(n, G, hs, 'zinb-conddisp', True, seed=4)
    eng = make_engine(CpuRefOps(), 'zinb-conddisp', G, hs, True, 0.0, p, X, Y, sf)
    tnet = TorchAE('zinb-conddisp', p, hs, True, dtype=torch.float64)
    order = np.random.RandomState(1).permutation(n)
    r = run_reseeded_epoch(eng, tnet, order, B)
    assert len(r['loss_eng']) == 5                                    # 4 x 32 + 22
    assert np.abs(r['loss_eng'] / r['loss_or'] - 1).max() < 2e-6
    assert r['grad_viol'].sum() == 0 and r['grad_err'].max() < 1.0
    assert r['upd_err'].max() < 1.0 and r['ms_err'].max() < 1.0 and r['bn_err'].max() < 1.0
    # teeth: the same oracle against an engine that applies a different learning rate / sees other weights
    tnet2 = TorchAE('zinb-conddisp', p, hs, True, dtype=torch.float64)
    eng2 = make_engine(CpuRefOps(), 'zinb-conddisp', G, hs, True, 0.0, p, X, Y, sf)
    real_step = eng2.train_step

    def bad_step(b, **kw):
        eng2.w[5] += 0.05                                            # one first-layer weight off before the step
        return real_step(b, **kw)
    eng2.train_step = bad_step
    r2 = run_reseeded_epoch(eng2, tnet2, order, B)
    assert r2['grad_viol'].sum() > 0 or np.abs(r2['loss_eng'] / r2['loss_or'] - 1).max() > 2e-6
