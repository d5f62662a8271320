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
        np.testing.assert_allclose(newp[k], ref.p[k], rtol=2e-4, atol=2e-6, err_msg=k)


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
    np.testing.assert_allclose(h.history['loss'], rh['loss'], rtol=2e-5)
    np.testing.assert_allclose(h.history['val_loss'], rh['val_loss'], rtol=2e-5)
    np.testing.assert_allclose(h.history['lr'], rh['lr'], rtol=1e-7)
    newp = eng.get_params()
    for k in ref.p:
        np.testing.assert_allclose(newp[k], ref.p[k], rtol=5e-3, atol=2e-5, err_msg=k)


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
