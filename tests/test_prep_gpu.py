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


@pytest.mark.parametrize('n,G', [(60, 40), (300, 203), (1000, 1001)])
@pytest.mark.parametrize('switches', [(True, True, True), (False, True, True), (True, False, True), (True, True, False),
                                      (False, False, False)])
def test_prep_kernels_match_the_oracle(ops, n, G, switches):
    """dcahip_prep_* (through the C ABI, prep.py only sequences the calls) against oracle/preproc_np.py -- the
    independent fp64 restatement of what dca/io.py:88-111 asks of scanpy: library sizes (exact), size factors,
    log1p, per-gene z-score (ddof = 1), and the gene / cell keep masks bit-exact.  No product host path in between."""
    from oracle import preproc_np as P
    size_factors, logtrans, zscore = switches
    y = synth_counts(n, G, 5)
    y[:, [1, 7]] = 0                      # empty genes: filtered by the masks, z-score must not divide by 0 on them
    y[3, :] = 0                           # an empty cell
    dev = torch.device('cuda')
    Y = prep._upload(y.astype(np.float32), dev)
    # masks: bit-exact (integer arithmetic on both sides)
    gc = prep.gene_counts(ops, Y, n, G).cpu().numpy()
    cc = prep.cell_counts(ops, Y, n, G).cpu().numpy()
    np.testing.assert_array_equal(gc >= 1, P.gene_keep_mask(y))
    np.testing.assert_array_equal(cc >= 1, P.cell_keep_mask(y))
    np.testing.assert_array_equal(gc, y.sum(axis=0))                 # exact integer sums
    np.testing.assert_array_equal(cc, y.sum(axis=1))
    keep_g, keep_c = gc >= 1, cc >= 1
    yk = y[keep_c][:, keep_g]
    nk, Gk = yk.shape
    Yk = prep._upload(yk.astype(np.float32), dev)
    want_x, want_sf, want_counts = P.normalize(yk, size_factors, logtrans, zscore)
    counts = prep.cell_counts(ops, Yk, nk, Gk)
    np.testing.assert_array_equal(counts.cpu().numpy(), want_counts)
    fac = None
    if size_factors:
        sf = counts.cpu().numpy() / np.median(counts.cpu().numpy())
        np.testing.assert_allclose(sf, want_sf, rtol=1e-7)
        fac = torch.as_tensor(sf.astype(np.float32)).to(dev)
    X = prep.transform(ops, Yk, nk, Gk, fac, logtrans, zscore) if (size_factors or logtrans or zscore) else Yk
    got = X[:, :Gk].cpu().numpy().astype(np.float64)
    # fp32 log1p / z-score against the fp64 oracle: 2e-5 relative + 2e-6 absolute (the z-score divides by an fp32 std)
    np.testing.assert_allclose(got, want_x, rtol=2e-5, atol=2e-6)
    assert (X[:, Gk:] == 0).all()


def test_prep_exact_sums_at_benchmark_size(ops):
    """68 579 x 20 000 (BASELINE configs[2]): library sizes and gene totals from K-PREP equal torch's int64 sums of the
    same resident matrix exactly -- the property the bit-exact gene / cell filtering of dca/io.py:90-92 rests on --
    and the z-scored matrix has per-gene mean 0 / unbiased variance 1 to fp32 accuracy."""
    from dca_amd import synth
    n, G = 68579, 20000
    dev = torch.device('cuda')
    Y = synth.generate_counts(n, G, device=dev)
    cc = prep.cell_counts(ops, Y, n, G)
    gc = prep.gene_counts(ops, Y, n, G)
    want_c = torch.zeros(n, dtype=torch.int64, device=dev)
    want_g = torch.zeros(G, dtype=torch.int64, device=dev)
    for s in range(0, n, 8192):
        blk = Y[s:s + 8192, :G].to(torch.int64)
        want_c[s:s + 8192] = blk.sum(dim=1)
        want_g += blk.sum(dim=0)
    assert want_c.max().item() < 2 ** 24 and want_g.max().item() < 2 ** 24      # representable in fp32: equality is exact
    assert torch.equal(cc.to(torch.int64), want_c)
    assert torch.equal(gc.to(torch.int64), want_g)
    sf = cc / cc.median()
    X = prep.transform(ops, Y, n, G, sf, True, True)
    m = torch.zeros(G, dtype=torch.float64, device=dev)
    q = torch.zeros(G, dtype=torch.float64, device=dev)
    for s in range(0, n, 8192):
        blk = X[s:s + 8192, :G].double()
        m += blk.sum(dim=0); q += (blk * blk).sum(dim=0)
    mean = m / n
    var = (q - n * mean * mean) / (n - 1)
    assert mean.abs().max().item() < 2e-5
    assert (var - 1).abs().max().item() < 2e-4
