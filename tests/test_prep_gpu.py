"""K-PREP on the MI355X: dcahip_prep_* through the C ABI against the host restatement of
dca/io.py:88-111; then the whole dca() call with device preprocessing against the same call with
host preprocessing."""
import numpy as np
import pandas as pd
import pytest
import torch

from conftest import synth_counts
from dca_amd import io, prep
from dca_amd._anndata import AnnData

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from dca_amd.ops import HipOps
    return HipOps()


def _adata(y):
    n, G = y.shape
    return AnnData(y.astype(np.float32), obs=pd.DataFrame(index=['c%d' % i for i in range(n)]),
                   var=pd.DataFrame(index=['g%d' % i for i in range(G)]))


@pytest.mark.parametrize('n,G', [(60, 40), (300, 203), (5, 6), (1000, 1001)])
def test_normalize_device_equals_host(ops, n, G):
    y = synth_counts(n, G, 3)
    if n > 50:
        y[:, [3, 17]] = 0
        y[[5, 44], :] = 0
    a, dd = prep.normalize_device(io.read_dataset(_adata(y)), ops=ops)
    b = io.normalize(io.read_dataset(_adata(y)), device=False)
    assert a.shape == b.shape
    assert list(a.obs.index) == list(b.obs.index) and list(a.var.index) == list(b.var.index)   # bit-exact filters
    np.testing.assert_array_equal(a.raw.X, b.raw.X)
    np.testing.assert_array_equal(a.obs['n_counts'].values, b.obs['n_counts'].values)
    np.testing.assert_array_equal(a.var['n_counts'].values, b.var['n_counts'].values)
    np.testing.assert_array_equal(np.asarray(a.obs['size_factors'].values, np.float32),
                                  np.asarray(b.obs['size_factors'].values, np.float32))
    # log1p differs by <= 1-2 ulp between libm and the device library; the z-score divides by std
    np.testing.assert_allclose(a.X, b.X, rtol=2e-5, atol=2e-6)
    n2, G2 = a.shape
    np.testing.assert_array_equal(dd.Y[:, :G2].cpu().numpy(), a.raw.X)
    assert (dd.X[:, G2:] == 0).all() and (dd.Y[:, G2:] == 0).all()


def test_prep_switches(ops):
    y = synth_counts(120, 77, 9)
    for kw in (dict(size_factors=False), dict(normalize_input=False), dict(logtrans_input=False),
               dict(size_factors=False, normalize_input=False, logtrans_input=False)):
        a, dd = prep.normalize_device(io.read_dataset(_adata(y)), ops=ops, filter_min_counts=False, **kw)
        b = io.normalize(io.read_dataset(_adata(y)), filter_min_counts=False, device=False, **kw)
        np.testing.assert_allclose(a.X, b.X, rtol=2e-5, atol=2e-6)


def test_dca_call_device_prep_equals_host_prep(ops, monkeypatch):
    """api.dca() end to end: K-PREP + device-resident hand-over to train() vs host preprocessing
    + upload.  Same seeds, same kernels afterwards: the denoised output agrees to the accuracy of
    the input difference (1-2 ulp of log1p)."""
    from dca_amd.api import dca
    y = synth_counts(400, 120, 4)
    outs = []
    for dev in ('1', '0'):
        monkeypatch.setenv('DCA_AMD_DEVICE_PREP', dev)
        ad = _adata(y)
        dca(ad, ae_type='zinb-conddisp', epochs=3, batch_size=32, random_state=0, return_info=True)
        outs.append(ad)
    a, b = outs
    assert getattr(a, '_dca_device', None) is not None and getattr(b, '_dca_device', None) is None
    np.testing.assert_allclose(a.X, b.X, rtol=5e-3, atol=1e-4)
    np.testing.assert_allclose(a.uns['dca_loss_history']['loss'], b.uns['dca_loss_history']['loss'], rtol=1e-4)
    np.testing.assert_array_equal(a.raw.X, b.raw.X)
