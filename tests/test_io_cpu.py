"""IO / preprocessing surface (dca_amd/io.py == dca/io.py): index outputs bit-exact, floats
against an independent restatement."""
import os

import numpy as np
import pandas as pd
import pytest

from conftest import synth_counts
from dca_amd import io
from dca_amd._anndata import AnnData
from oracle import preproc_np as P


def _adata(n=60, G=40, seed=0, dtype=np.float32):
    y = synth_counts(n, G, seed).astype(dtype)
    return AnnData(y, obs=pd.DataFrame(index=['c%d' % i for i in range(n)]),
                   var=pd.DataFrame(index=['g%d' % i for i in range(G)]))


def test_zscore_of_the_oracle_matches_sklearn_standard_scaler():
    """The last step of the preprocessing (scanpy's pp.scale: zero mean, unit UNBIASED variance per gene, constant genes
    left at 0) against scikit-learn's StandardScaler, which divides by the biased standard deviation: the oracle's
    columns are StandardScaler's times sqrt((n - 1) / n).  scanpy itself is not installable here."""
    from sklearn.preprocessing import StandardScaler
    y = synth_counts(80, 25, 4).astype(np.float64)
    y[:, 7] = 0.0                                   # a gene that is constant after the normalisation too
    x, sf, n_counts = P.normalize(y, size_factors=True, logtrans=True, zscore=True)
    lib = y.sum(1)
    assert np.array_equal(n_counts, lib) and np.allclose(sf, lib / np.median(lib), rtol=1e-15)
    pre = np.log1p(y / sf[:, None])
    want = StandardScaler().fit_transform(pre) * np.sqrt((len(y) - 1) / len(y))
    np.testing.assert_allclose(x, want, rtol=1e-10, atol=1e-12)
    assert (x[:, 7] == 0).all()


def test_filter_masks_bit_exact():
    rng = np.random.RandomState(0)
    y = synth_counts(50, 30, 1)
    y[:, [3, 17]] = 0          # all-zero genes
    y[[5, 44], :] = 0          # all-zero cells
    keep_g, num = io.filter_genes_mask(y, 1)
    assert (keep_g == P.gene_keep_mask(y)).all() and keep_g.sum() == 28
    keep_c, _ = io.filter_cells_mask(y, 1)
    assert (keep_c == P.cell_keep_mask(y)).all() and keep_c.sum() == 48
    ad = AnnData(y.astype(np.float32))
    ad = io.read_dataset(ad, check_counts=True)
    ad = io.normalize(ad, filter_min_counts=True)
    # genes filtered first, then cells (io.py:91-92); raw keeps the filtered counts
    assert ad.shape == (48, 28) and ad.raw.X.shape == (48, 28)
    np.testing.assert_array_equal(ad.raw.X, y[keep_c][:, keep_g].astype(np.float32))
    assert list(ad.var.index) == [str(j) for j in np.where(keep_g)[0]]
    assert list(ad.obs.index) == [str(i) for i in np.where(keep_c)[0]]


def test_normalize_matches_independent_restatement():
    ad = _adata(80, 50, 3)
    y = ad.X.copy()
    ad = io.read_dataset(ad)
    ad = io.normalize(ad, filter_min_counts=False)
    x, sf, nc = P.normalize(y)
    np.testing.assert_array_equal(ad.obs['n_counts'].values, nc.astype(np.float32))
    np.testing.assert_allclose(ad.obs['size_factors'].values, sf, rtol=1e-6)
    np.testing.assert_allclose(ad.X, x, rtol=2e-4, atol=2e-5)
    np.testing.assert_array_equal(ad.raw.X, y)
    assert (ad.obs['dca_split'] == 'train').all() and str(ad.obs['dca_split'].dtype) == 'category'
    # switches
    ad2 = io.normalize(io.read_dataset(_adata(80, 50, 3)), filter_min_counts=False,
                       size_factors=False, normalize_input=False, logtrans_input=True)
    np.testing.assert_allclose(ad2.X, np.log1p(y), rtol=1e-6)
    assert (ad2.obs['size_factors'] == 1.0).all()


def test_test_split_indices_are_sklearns():
    from sklearn.model_selection import train_test_split
    ad = io.read_dataset(_adata(101, 10, 2), test_split=True)
    tr, te = train_test_split(np.arange(101), test_size=0.1, random_state=42)
    got_te = np.where(ad.obs['dca_split'].values == 'test')[0]
    assert sorted(te.tolist()) == got_te.tolist()
    assert (ad.obs['dca_split'].values == 'train').sum() == len(tr)


def test_check_counts_and_errors():
    ad = _adata(20, 8, 1)
    ad.X[3, 2] = 0.5
    with pytest.raises(AssertionError, match='unnormalized count data'):
        io.read_dataset(ad)
    io.read_dataset(ad, check_counts=False)
    with pytest.raises(NotImplementedError):
        io.read_dataset(12345)


def test_read_text_transpose_and_write_text_matrix(tmp_path):
    y = synth_counts(6, 4, 0)
    df = pd.DataFrame(y.T, index=['g%d' % i for i in range(4)], columns=['c%d' % i for i in range(6)])
    f = str(tmp_path / 'counts.tsv')
    df.to_csv(f, sep='\t')
    ad = io.read_dataset(f, transpose=True)             # gene x cell file -> cells x genes
    assert ad.shape == (6, 4) and list(ad.var_names) == ['g0', 'g1', 'g2', 'g3']
    np.testing.assert_array_equal(ad.X, y.astype(np.float32))
    out = str(tmp_path / 'm.tsv')
    io.write_text_matrix(np.array([[1.0, 2.5], [3.25, 4.1234567]]), out, rownames=['r0', 'r1'],
                         colnames=['a', 'b'], transpose=True)
    txt = open(out).read().splitlines()
    assert txt[0] == '\tr0\tr1' and txt[1] == 'a\t1.000000\t3.250000' and txt[2] == 'b\t2.500000\t4.123457'
    g = str(tmp_path / 'genes.txt')
    open(g, 'w').write('g1\ng3\ng1\n')
    assert sorted(io.read_genelist(g)) == ['g1', 'g3']


def test_identity_selection_shares_the_matrices_and_partial_selection_copies():
    """dca() takes adata[adata.obs.dca_split == 'train'] (dca/api.py:203); without a test split that selects every
    cell, and anndata's own views do not copy there either."""
    import pandas as pd
    from dca_amd._anndata import MiniAnnData
    y = synth_counts(30, 12, 1).astype(np.float32)
    ad = MiniAnnData(y, obs=pd.DataFrame({'dca_split': ['train'] * 30}, index=['c%d' % i for i in range(30)]))
    ad.raw = ad.copy()
    everything = ad[ad.obs.dca_split == 'train']
    assert np.shares_memory(everything.X, ad.X) and everything.raw.X is ad.raw.X
    # a write through the subset copies first (as a write through an anndata view does) and never reaches the parent
    before = float(ad.X[0, 0])
    everything.X[0, 0] = before + 5.0
    assert float(everything.X[0, 0]) == before + 5.0 and float(ad.X[0, 0]) == before
    assert not np.shares_memory(everything.X, ad.X)
    again = ad[ad.obs.dca_split == 'train']
    with pytest.raises(ValueError):           # other in-place operations find the shared matrix read-only
        again.X += 1.0
    ad.X[0, 0] = y[0, 0]                       # the parent itself stays writeable
    assert list(everything.obs.index) == list(ad.obs.index)
    part = ad[np.arange(30) % 2 == 0]
    assert part.X is not ad.X and part.shape == (15, 12)
    np.testing.assert_array_equal(part.X, y[::2])


def test_device_cache_mark_follows_the_host_matrix():
    """prep.DeviceData.matches: K-PREP's resident tensors are reused only while adata.X still is the matrix they were
    made from (the reference always feeds the current adata.X, dca/network.py:188-211)."""
    from dca_amd import prep
    x = np.random.RandomState(0).rand(200, 17).astype(np.float32)
    dd = prep.DeviceData(None, None, None, 200, 17, host_x=x)
    assert dd.matches(x) and dd.matches(x.copy())
    x2 = x.copy(); x2[199, 3] += 1.0                       # the last row is always part of the mark
    assert not dd.matches(x2)
    assert not dd.matches(x[:100])
    assert prep.DeviceData(None, None, None, 200, 17).matches(x2)      # no mark taken (to_host=False): always reused


def test_parallel_host_copy():
    from dca_amd import hostlib
    a = np.random.RandomState(1).rand(3000, 1001).astype(np.float32)
    b = np.zeros((5000, 1001), np.float32)
    hostlib.parallel_copy(b[1000:4000], a, threads=5)
    np.testing.assert_array_equal(b[1000:4000], a)
    assert (b[:1000] == 0).all() and (b[4000:] == 0).all()
    small = np.arange(10, dtype=np.float32); out = np.empty_like(small)
    hostlib.parallel_copy(out, small)
    np.testing.assert_array_equal(out, small)


def test_copy_on_write_subset_keeps_every_write_through_one_reference():
    """A select-everything subset shares its parent's matrix until written to; TWO writes through one held reference both land
    in the subset's own copy (the second used to rebuild the copy from the parent and lose the first), the parent stays."""
    import numpy as np
    import pandas as pd
    from dca_amd._anndata import AnnData
    a = AnnData(np.arange(12, dtype=np.float32).reshape(3, 4), obs=pd.DataFrame(index=list('abc')),
                var=pd.DataFrame(index=list('wxyz')))
    sub = a[np.ones(3, bool)]
    x = sub.X
    x[0, 0] = 100.0
    x[1, 1] = 200.0
    assert sub.X[0, 0] == 100.0 and sub.X[1, 1] == 200.0
    assert a.X[0, 0] == 0.0 and a.X[1, 1] == 5.0
