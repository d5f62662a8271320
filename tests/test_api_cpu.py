"""The drop-in surface: dca_amd.api.dca with the assertions of the reference's own test
(dca/test.py:6-59) re-hosted on a synthetic AnnData, plus BASELINE config 1 (biochemists,
plumbing).  Runs on CPU with the oracle-backed ops injected explicitly."""
import numpy as np
import pandas as pd
import pytest

from conftest import synth_counts
from dca_amd.api import dca
from dca_amd._anndata import AnnData
from dca_amd.network import override_ops, AE_types
from oracle.cpu_ops import CpuRefOps


def _adata(n=120, G=50, seed=0):
    return AnnData(synth_counts(n, G, seed).astype(np.float32),
                   obs=pd.DataFrame(index=['c%d' % i for i in range(n)]),
                   var=pd.DataFrame(index=['g%d' % i for i in range(G)]))


def test_api_contract_like_reference_test():
    adata = _adata()
    epochs = 1
    with override_ops(CpuRefOps):
        ret = dca(adata, mode='denoise', copy=True, epochs=epochs, verbose=True)
        assert not np.allclose(ret.X[:10], adata.X[:10])

        ret, model = dca(adata, mode='denoise', ae_type='nb-conddisp', copy=True, epochs=epochs,
                         return_model=True, return_info=True)
        assert not np.allclose(ret.X[:10], adata.X[:10])
        assert 'X_dca_dispersion' in ret.obsm_keys()
        assert model is not None

        ret = dca(adata, mode='denoise', ae_type='nb', copy=True, epochs=epochs,
                  return_model=False, return_info=True)
        assert not np.allclose(ret.X[:10], adata.X[:10])
        assert 'X_dca_dispersion' in ret.var_keys()

        ret = dca(adata, mode='denoise', ae_type='zinb', copy=True, epochs=epochs,
                  return_model=False, return_info=True)
        assert not np.allclose(ret.X[:10], adata.X[:10])
        assert 'X_dca_dropout' in ret.obsm_keys()
        assert 'dca_loss_history' in ret.uns_keys()

        ret = dca(adata, mode='denoise', ae_type='zinb-conddisp', copy=True, epochs=epochs,
                  return_info=True)
        assert {'X_dca_dropout', 'X_dca_dispersion'} <= set(ret.obsm_keys())
        assert set(ret.uns['dca_loss_history']) == {'loss', 'val_loss', 'lr'}
        assert ret.obsm['X_dca_dropout'].shape == adata.shape
        assert ((ret.obsm['X_dca_dropout'] >= 0) & (ret.obsm['X_dca_dropout'] <= 1)).all()
        assert (ret.X > 0).all()                              # mean * size factor

        # simple tests for latent
        hid_size = (10, 2, 10)
        for t in ('nb-conddisp', 'nb', 'zinb', 'zinb-conddisp'):
            ret = dca(adata, mode='latent', ae_type=t, hidden_size=hid_size, copy=True, epochs=epochs)
            assert 'X_dca' in ret.obsm_keys()
            assert ret.obsm['X_dca'].shape[1] == hid_size[1]
            np.testing.assert_array_equal(ret.X, adata.X)     # latent mode restores raw counts

        # in-place contract: returns None, mutates adata, keeps raw counts in .raw
        ad2 = _adata()
        raw = ad2.X.copy()
        assert dca(ad2, ae_type='zinb-conddisp', epochs=epochs) is None
        np.testing.assert_array_equal(ad2.raw.X, raw)
        assert not np.allclose(ad2.X, raw)
        assert {'size_factors', 'n_counts', 'dca_split'} <= set(ad2.obs.columns)


def test_api_errors():
    adata = _adata()
    with pytest.raises(AssertionError, match='AnnData'):
        dca(np.zeros((3, 3)))
    with pytest.raises(AssertionError, match='not a valid mode'):
        dca(adata, mode='full')
    z = _adata()
    z.X[:, 3] = 0
    with pytest.raises(AssertionError, match='all-zero genes'):
        dca(z)
    assert set(AE_types) == {'normal', 'poisson', 'nb', 'nb-conddisp', 'nb-shared', 'nb-fork', 'zinb',
                             'zinb-conddisp', 'zinb-shared', 'zinb-fork', 'zinb-elempi'}
    with override_ops(CpuRefOps):
        with pytest.raises(NotImplementedError):
            dca(_adata(), activation='no_such_activation', epochs=1)
        with pytest.raises(NotImplementedError):
            dca(_adata(), init='no_such_init', epochs=1)
        with pytest.raises(NotImplementedError):
            dca(_adata(), optimizer='Ftrl', epochs=1)


def test_product_build_refuses_to_run_without_gpu():
    """No silent CPU path: without the test hook, building a network needs the GPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        dca(_adata(), ae_type='zinb-conddisp', epochs=1)


def test_biochemists_zinb_ae_plumbing(biochemists):
    """BASELINE configs[0]: data/test-biochemists-zinb-ae.py -- zinb-conddisp, hidden (1,),
    3 epochs on the 915 x 6 table (which holds a non-integer column: check_counts=False)."""
    tab = biochemists['table'].astype(np.float32)
    ad = AnnData(tab, var=pd.DataFrame(index=[str(c) for c in biochemists['columns']]))
    with override_ops(CpuRefOps):
        ret = dca(ad, ae_type='zinb-conddisp', hidden_size=(1,), epochs=3, check_counts=False,
                  copy=True, return_info=True, random_state=1)
    h = ret.uns['dca_loss_history']
    assert len(h['loss']) == 3 and len(h['val_loss']) == 3 and np.isfinite(h['loss']).all()
    assert ret.X.shape == (915, 6)


def _prepared(seed=0):
    from dca_amd import io
    ad = io.read_dataset(_adata(90, 30, seed))
    return io.normalize(ad, filter_min_counts=False, device=False)


def test_checkpoint_best_weights_and_resume(tmp_path):
    """train(save_weights=True) keeps the weights of the best validation epoch (Keras
    ModelCheckpoint(save_best_only=True), train.py:64-69); checkpoint / resume (an extension)
    continues a run bit-for-bit: 4 epochs == 2 epochs + resume to 4."""
    from dca_amd.train import train
    out_a, out_b = str(tmp_path / 'a'), str(tmp_path / 'b')
    with override_ops(CpuRefOps):
        def run(outdir, epochs, resume):
            np.random.seed(3)
            ad = _prepared()
            net = AE_types['zinb-conddisp'](input_size=ad.n_vars, hidden_size=(8, 3, 8), file_path=outdir)
            net.seed = 0
            net.build()
            h = train(ad, net, output_dir=outdir, epochs=epochs, batch_size=16, save_weights=True,
                      verbose=False, checkpoint=True, resume=resume, early_stop=0, reduce_lr=2)
            return h, net
        h4, net4 = run(out_a, 4, False)
        run(out_b, 2, False)
        h22, net22 = run(out_b, 4, True)
    assert h22.history == h4.history
    p4, p22 = net4.engine.get_params(), net22.engine.get_params()
    for k in p4:
        np.testing.assert_array_equal(p4[k], p22[k])
    # the weights file is the best-val_loss epoch, not necessarily the last one
    best = int(np.argmin(h4.history['val_loss']))
    z = np.load(str(tmp_path / 'a' / 'weights.npz'))
    if best == 3:
        for k in p4:
            np.testing.assert_array_equal(z[k], p4[k])
    assert set(z.files) == set(p4)
