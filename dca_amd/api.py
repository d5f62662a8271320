"""``dca()`` -- the Python API of the reference (dca/api.py:19-211) on the MI355X path.

Identical signature, defaults, in-place / copy behaviour, result placement and return
conventions.  Differences are confined to what executes the training step (HIP kernels instead
of Keras/TensorFlow CPU kernels) and are listed in DESIGN.md.
"""
import os
import random

import numpy as np

from ._anndata import is_anndata
from .io import read_dataset, normalize, filter_genes_mask, resident_counts
from .train import train
from .network import AE_types


def dca(adata,
        mode='denoise',
        ae_type='nb-conddisp',
        normalize_per_cell=True,
        scale=True,
        log1p=True,
        hidden_size=(64, 32, 64),  # network args
        hidden_dropout=0.,
        batchnorm=True,
        activation='relu',
        init='glorot_uniform',
        network_kwds={},
        epochs=300,               # training args
        reduce_lr=10,
        early_stop=15,
        batch_size=32,
        optimizer='RMSprop',
        learning_rate=None,
        random_state=0,
        threads=None,
        verbose=False,
        training_kwds={},
        return_model=False,
        return_info=False,
        copy=False,
        check_counts=True,
        ):
    """Deep count autoencoder (DCA) API -- see dca/api.py:46-144 of the reference for the full
    parameter documentation; every parameter keeps its meaning.

    ``threads`` sized TensorFlow's CPU pools in the reference (train.py:41-48); here it sizes the host thread pools of the
    native host stages (staging copies, checksums, result writers) -- the training step runs on the GPU.  ``random_state`` seeds python / numpy exactly as api.py:150-153 does
    (the per-epoch shuffles consume the numpy global stream like Keras did) and additionally the
    glorot-uniform weight initialisation (TensorFlow's stream is not reproducible outside TF).
    """
    assert is_anndata(adata), 'adata must be an AnnData instance'
    assert mode in ('denoise', 'latent'), '%s is not a valid mode.' % mode
    _seed_host_generators(random_state)

    # counts -> AnnData with .raw and the train/test column (api.py:156-160); copy=True works on a private copy
    work = read_dataset(adata, transpose=False, test_split=False, copy=copy, check_counts=check_counts)
    # api.py:163-164.  With a GPU the counts are uploaded here, once (normalize() and train() use the resident copy), and
    # the per-gene totals are the device's exact integer sums instead of a pass over the host matrix
    resident = resident_counts(work)
    gene_totals = resident[1] if resident is not None else filter_genes_mask(work.X, min_counts=1)[1]
    assert (np.asarray(gene_totals) >= 1).all(), 'Please remove all-zero genes before using DCA.'
    # cell / gene indices must survive untouched, so no count filtering inside normalize (api.py:166-170)
    work = normalize(work, filter_min_counts=False, size_factors=normalize_per_cell,
                     normalize_input=scale, logtrans_input=log1p,
                     _resident=resident[0] if resident is not None else None)

    net = _make_network(ae_type, work.n_vars, random_state,
                        dict(network_kwds, hidden_size=hidden_size, hidden_dropout=hidden_dropout,
                             batchnorm=batchnorm, activation=activation, init=init))
    fit_kwds = dict(training_kwds, epochs=epochs, reduce_lr=reduce_lr, early_stop=early_stop,
                    batch_size=batch_size, optimizer=optimizer, verbose=verbose, threads=threads,
                    learning_rate=learning_rate)
    history = train(work[work.obs.dca_split == 'train'], net, **fit_kwds)               # api.py:203

    predicted = net.predict(work, mode, return_info, copy)      # in place unless copy (network.py:188-211)
    result = predicted if copy else work
    if return_info:
        result.uns['dca_loss_history'] = history.history
    # api.py:208-211: what comes back depends on (copy, return_model)
    if copy:
        return (result, net) if return_model else result
    return net if return_model else None


def _seed_host_generators(seed):
    """api.py:150-153: python's and numpy's global generators (the per-epoch shuffles draw from numpy's)."""
    random.seed(seed)
    np.random.seed(seed)
    os.environ['PYTHONHASHSEED'] = '0'


def _make_network(ae_type, n_genes, seed, kwds):
    """AE_types lookup, save() before build() as the reference does (api.py:181-188, train.py:170-171)."""
    net = AE_types[ae_type](input_size=n_genes, output_size=n_genes, **kwds)
    net.seed = seed
    net.save()
    net.build()
    return net
