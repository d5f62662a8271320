"""The drop-in surface on the MI355X with the real kernels (HipOps): checkpoint -> resume is bit-for-bit, best-epoch
weights file, dca() end to end.  (tests/test_api_cpu.py covers the same host logic on CPU with oracle-backed ops.)"""
import numpy as np
import pandas as pd
import pytest

from conftest import synth_counts
from dca_amd._anndata import AnnData
from dca_amd.network import AE_types

pytestmark = pytest.mark.gpu


def _prepared(n=300, G=120, seed=0):
    from dca_amd import io
    ad = AnnData(synth_counts(n, G, seed).astype(np.float32),
                 obs=pd.DataFrame(index=['c%d' % i for i in range(n)]),
                 var=pd.DataFrame(index=['g%d' % i for i in range(G)]))
    ad = io.read_dataset(ad, transpose=False, test_split=False, copy=False)
    return io.normalize(ad, size_factors=True, logtrans_input=True, normalize_input=True)


@pytest.mark.parametrize('ae_type,optimizer', [('zinb-conddisp', 'RMSprop'), ('nb', 'Adam')])
def test_checkpoint_resume_is_bit_exact_on_the_gpu(tmp_path, ae_type, optimizer):
    """train(checkpoint=True) writes the full training state (parameters, optimizer slots, BN moving statistics,
    step / dropout counters, lr, callback counters, history) after every epoch; resume=True continues from it.
    With the HIP kernels (deterministic: fixed summation orders, no atomics) 5 epochs == 2 epochs + resume to 5,
    bit for bit: history, every parameter, the optimizer slots.  Reference: dca/train.py:64-69 keeps only
    best-val_loss weights (ModelCheckpoint); that file is checked as well."""
    from dca_amd.train import train
    out_a, out_b = str(tmp_path / 'a'), str(tmp_path / 'b')

    def run(outdir, epochs, resume):
        np.random.seed(3)
        ad = _prepared()
        net = AE_types[ae_type](input_size=ad.n_vars, hidden_size=(64, 32, 64), hidden_dropout=0.1, file_path=outdir)
        net.seed = 0
        net.build()
        assert type(net.engine.ops).__name__ == 'HipOps'
        h = train(ad, net, output_dir=outdir, optimizer=optimizer, epochs=epochs, batch_size=32, save_weights=True,
                  verbose=False, checkpoint=True, resume=resume, early_stop=0, reduce_lr=2)
        return h, net

    h5, net5 = run(out_a, 5, False)
    run(out_b, 2, False)
    h25, net25 = run(out_b, 5, True)
    assert h25.history == h5.history
    p5, p25 = net5.engine.get_params(), net25.engine.get_params()
    for k in p5:
        np.testing.assert_array_equal(p5[k], p25[k], err_msg=k)
    np.testing.assert_array_equal(net5.engine.ms.cpu().numpy(), net25.engine.ms.cpu().numpy())
    if net5.engine.slot2 is not None:
        np.testing.assert_array_equal(net5.engine.slot2.cpu().numpy(), net25.engine.slot2.cpu().numpy())
    best = int(np.argmin(h5.history['val_loss']))
    z = np.load(str(tmp_path / 'a' / 'weights.npz'))
    assert set(z.files) == set(p5)
    if best == 4:
        for k in p5:
            np.testing.assert_array_equal(z[k], p5[k])


def test_all_zero_gene_is_refused_before_anything_is_touched():
    """api.py:163-164 ('Please remove all-zero genes before using DCA.') with the per-gene totals taken from the counts
    resident in HBM (io.resident_counts): same assertion, and the caller's AnnData is still the raw counts."""
    from dca_amd.api import dca
    n, G = 200, 60
    Y = synth_counts(n, G, 1).astype(np.float32)
    Y[:, 17] = 0
    ad = AnnData(Y.copy(), obs=pd.DataFrame(index=['c%d' % i for i in range(n)]),
                 var=pd.DataFrame(index=['g%d' % i for i in range(G)]))
    with pytest.raises(AssertionError, match='all-zero genes'):
        dca(ad, epochs=1, verbose=False)
    np.testing.assert_array_equal(ad.X, Y)


def test_staged_upload_and_download_are_exact():
    """prep._upload / _download of a matrix large enough for the page-locked staging path (>= 4 Mi elements; rows not a
    multiple of the chunk, columns not a multiple of 4): bit-identical round trip, padding columns zero."""
    import torch
    from dca_amd import prep
    n, G = 4500, 1003
    X = np.random.RandomState(0).standard_normal((n, G)).astype(np.float32)
    d = prep._upload(X, torch.device('cuda'))
    assert tuple(d.shape) == (n, 1004) and float(d[:, G:].abs().max()) == 0.0
    np.testing.assert_array_equal(d[:, :G].cpu().numpy(), X)
    np.testing.assert_array_equal(prep._download(d, n, G), X)
    Y, totals = prep.resident_counts(np.abs(np.round(X * 3)))
    np.testing.assert_array_equal(totals, np.abs(np.round(X * 3)).sum(axis=0, dtype=np.float64))


@pytest.mark.parametrize('ae_type', ['zinb-conddisp', 'zinb', 'nb-conddisp', 'nb'])
def test_fused_predict_writer_writes_the_files_of_predict_then_write(tmp_path, ae_type):
    """network.predict_write (the CLI's last step as one streaming pass: hidden stack over all cells, then gene blocks
    through heads GEMM -> inference activations -> transpose on the device -> native writer; no cells x genes matrix on
    the host) writes the files predict(mode='full', return_info=True) + write(mode='full') write (dca/network.py:188-231,
    395-421; dca/io.py:120-129) -- byte for byte."""
    import os
    from dca_amd.train import train
    ad = _prepared(n=333, G=530, seed=4)
    net = AE_types[ae_type](input_size=ad.n_vars, hidden_size=(64, 32, 64), file_path=str(tmp_path))
    net.seed = 0
    net.build()
    train(ad, net, epochs=1, batch_size=32, verbose=False, early_stop=0, reduce_lr=0)
    a, b = str(tmp_path / 'fused'), str(tmp_path / 'plain')
    net.predict_write(ad, a, mode='full', gene_block=140)            # three gene blocks of ~177 genes
    net.predict(ad, mode='full', return_info=True)
    net.write(ad, b, mode='full')
    files = sorted(os.listdir(b))
    assert 'mean.tsv' in files and 'latent.tsv' in files and sorted(os.listdir(a)) == files
    for f in files:
        assert open(os.path.join(a, f), 'rb').read() == open(os.path.join(b, f), 'rb').read(), f


def test_an_edit_of_adata_X_after_normalize_reaches_the_network():
    """normalize() on the GPU leaves X / Y / size factors in HBM for train() and predict() (adata._dca_device).  The
    reference always feeds the current adata.X (dca/network.py:188-211): when the caller edits the host matrix in place
    after normalize() -- one element of one row -- the resident copy must not be used: the engine then holds the edited
    values, and predict() answers for them."""
    import torch
    from dca_amd.train import train
    ad = _prepared(n=400, G=90, seed=5)
    assert getattr(ad, '_dca_device', None) is not None and ad._dca_device.matches(ad.X)
    net = AE_types['zinb-conddisp'](input_size=ad.n_vars, hidden_size=(16, 4, 16))
    net.seed = 0
    net.build()
    ad.X[123, 7] += np.float32(2.5)                      # row 123: not among the rows round 3's fingerprint sampled
    assert not ad._dca_device.matches(ad.X)
    train(ad, net, epochs=1, batch_size=32, verbose=False)
    eng = net.engine
    torch.cuda.synchronize()
    assert float(eng.X[123, 7].item()) == float(ad.X[123, 7])      # the engine trained on the edited matrix
    np.testing.assert_array_equal(eng.X[:, :ad.n_vars].cpu().numpy(), ad.X)
