"""K-PREP host driver (dca_amd/prep.py) against the host restatement of dca/io.py:88-111
(dca_amd.io.normalize): same AnnData afterwards -- filter indices and counts bit-exact, floats to
fp32 round-off.  Runs the driver on CPU tensors through the oracle-backed ops object."""
import numpy as np
import pandas as pd
import pytest

from conftest import synth_counts
from dca_amd import io, prep
from dca_amd._anndata import AnnData
from oracle.cpu_ops import CpuRefOps


def _adata(y):
    n, G = y.shape
    return AnnData(y.astype(np.float32), obs=pd.DataFrame(index=['c%d' % i for i in range(n)]),
                   var=pd.DataFrame(index=['g%d' % i for i in range(G)]))


def _compare(a, b, dd):
    assert a.shape == b.shape
    assert list(a.obs.index) == list(b.obs.index) and list(a.var.index) == list(b.var.index)
    np.testing.assert_array_equal(a.raw.X, b.raw.X)
    np.testing.assert_array_equal(a.obs['n_counts'].values, b.obs['n_counts'].values)
    np.testing.assert_array_equal(np.asarray(a.obs['size_factors'].values, np.float32),
                                  np.asarray(b.obs['size_factors'].values, np.float32))
    np.testing.assert_allclose(a.X, b.X, rtol=1e-5, atol=1e-6)
    n, G = a.shape
    assert dd.n == n and dd.G == G
    np.testing.assert_array_equal(dd.Y[:, :G].numpy(), a.raw.X)
    np.testing.assert_array_equal(dd.X[:, :G].numpy(), a.X)
    np.testing.assert_array_equal(dd.sf.numpy(), np.asarray(a.obs['size_factors'].values, np.float32))


@pytest.mark.parametrize('n,G', [(60, 40), (131, 203), (5, 6)])
def test_normalize_device_equals_host(n, G):
    y = synth_counts(n, G, 3)
    a, dd = prep.normalize_device(io.read_dataset(_adata(y)), ops=CpuRefOps())
    b = io.normalize(io.read_dataset(_adata(y)))
    _compare(a, b, dd)


def test_filters_and_switches():
    y = synth_counts(50, 30, 1)
    y[:, [3, 17]] = 0
    y[[5, 44], :] = 0
    a, dd = prep.normalize_device(io.read_dataset(_adata(y)), ops=CpuRefOps())
    b = io.normalize(io.read_dataset(_adata(y)))
    assert a.shape == (48, 28)
    _compare(a, b, dd)
    np.testing.assert_array_equal(a.var['n_counts'].values, b.var['n_counts'].values)
    for kw in (dict(size_factors=False), dict(normalize_input=False), dict(logtrans_input=False),
               dict(size_factors=False, normalize_input=False, logtrans_input=False)):
        a, dd = prep.normalize_device(io.read_dataset(_adata(y)), ops=CpuRefOps(), **kw)
        b = io.normalize(io.read_dataset(_adata(y)), **kw)
        np.testing.assert_allclose(a.X, b.X, rtol=1e-5, atol=1e-6)
        np.testing.assert_array_equal(np.asarray(a.obs['size_factors'].values, np.float32),
                                      np.asarray(b.obs['size_factors'].values, np.float32))


def test_resident_tensors_are_dropped_when_any_element_of_adata_X_changes():
    """The reference always feeds the CURRENT adata.X (dca/network.py:188-211, dca/train.py:83-89).  normalize() leaves
    tensors on the device and train() / predict() reuse them only while the host matrix still is the one they were made
    from: an in-place edit of ONE element of ANY row (round 3's 64-row sample missed every other row) must be seen."""
    y = synth_counts(1000, 40, 3)
    with_ops = CpuRefOps()
    ad, dd = prep.normalize_device(_adata(y), ops=with_ops, device='cpu')
    assert dd.matches(ad.X)
    for row in (1, 7, 501, 999):                        # rows the old sample (every 15th row + the last) did not contain, and one it did
        x = ad.X
        old = x[row, 3]
        x[row, 3] = old + np.float32(1e-3)
        assert not dd.matches(ad.X), row
        x[row, 3] = old
        assert dd.matches(ad.X)
    # a different object with the same content still matches (anndata's setters copy), a different dtype does not
    assert dd.matches(ad.X.copy())
    assert not dd.matches(ad.X.astype(np.float64))
