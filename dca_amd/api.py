"""``dca()`` -- the Python API of the reference (dca/api.py:19-211) on the MI355X path.

Identical signature, defaults, in-place / copy behaviour, result placement and return
conventions.  Differences are confined to what executes the training step (HIP kernels instead
of Keras/TensorFlow CPU kernels) and are listed in DESIGN.md.
"""
import os
import random

import numpy as np

from ._anndata import is_anndata
from .io import read_dataset, normalize, filter_genes_mask
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

    ``threads`` sized TensorFlow's CPU pools in the reference (train.py:41-48); it is accepted
    and has no effect here.  ``random_state`` seeds python / numpy exactly as api.py:150-153 does
    (the per-epoch shuffles consume the numpy global stream like Keras did) and additionally the
    glorot-uniform weight initialisation (TensorFlow's stream is not reproducible outside TF).
    """
    assert is_anndata(adata), 'adata must be an AnnData instance'
    assert mode in ('denoise', 'latent'), '%s is not a valid mode.' % mode

    # set seed for reproducibility
    random.seed(random_state)
    np.random.seed(random_state)
    os.environ['PYTHONHASHSEED'] = '0'

    # this creates adata.raw with raw counts and copies adata if copy==True
    adata = read_dataset(adata,
                         transpose=False,
                         test_split=False,
                         copy=copy,
                         check_counts=check_counts)

    # check for zero genes
    nonzero_genes, _ = filter_genes_mask(adata.X, min_counts=1)
    assert nonzero_genes.all(), 'Please remove all-zero genes before using DCA.'

    adata = normalize(adata,
                      filter_min_counts=False,  # no filtering, keep cell and gene idxs same
                      size_factors=normalize_per_cell,
                      normalize_input=scale,
                      logtrans_input=log1p)

    network_kwds = {**network_kwds,
                    'hidden_size': hidden_size,
                    'hidden_dropout': hidden_dropout,
                    'batchnorm': batchnorm,
                    'activation': activation,
                    'init': init
                    }

    input_size = output_size = adata.n_vars
    net = AE_types[ae_type](input_size=input_size,
                            output_size=output_size,
                            **network_kwds)
    net.seed = random_state
    net.save()
    net.build()

    training_kwds = {**training_kwds,
                     'epochs': epochs,
                     'reduce_lr': reduce_lr,
                     'early_stop': early_stop,
                     'batch_size': batch_size,
                     'optimizer': optimizer,
                     'verbose': verbose,
                     'threads': threads,
                     'learning_rate': learning_rate
                     }

    hist = train(adata[adata.obs.dca_split == 'train'], net, **training_kwds)
    res = net.predict(adata, mode, return_info, copy)
    adata = res if copy else adata

    if return_info:
        adata.uns['dca_loss_history'] = hist.history

    if return_model:
        return (adata, net) if copy else net
    else:
        return adata if copy else None
