"""CPU oracle for the DCA ZINB-autoencoder training path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``dca_amd/`` may import this package.
Allowed importers: ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` -- always as the checker / the reported
CPU baseline, never as the thing that is shipped or measured as the product.

What it is: a from-scratch restatement (numpy, fp64 or fp32; plus a torch-CPU
autograd twin in ``torch_ref.py``) of the arithmetic the reference executes for
this path.  Every function cites the reference ``file:line`` it follows
(paths are relative to the upstream repo, theislab/dca v0.3.3).

Pinning status
--------------
* ``zinb_np.py`` (NB / ZINB negative log-likelihood and its gradient,
  reference ``dca/loss.py:60-156``) is PINNED by the reference's own
  R-generated fixtures ``data/biochemists*.tsv``: the summed NLL at the
  published MLE reproduces pscl/MASS log-likelihoods (-1549.99 / -1560.96) and
  the gradient vanishes there (tests/test_oracle_golden.py); element-wise it
  is held to ``scipy.stats.nbinom`` and its zero-inflated mixture over a grid
  of counts, means, dispersions and dropout probabilities (same file).
* everything whose arithmetic lives in the un-vendored Keras/TensorFlow
  (``keras>=2.4,<2.6``, ``tensorflow>=2.0,<2.5``: Dense, BatchNormalization,
  RMSprop, clipvalue, the ``fit`` loop, ReduceLROnPlateau, EarlyStopping) and
  in scanpy (filter/normalize_per_cell/log1p/scale) is restated from the
  libraries' documented behaviour; the reference cannot be imported in the
  build image (tensorflow, keras, scanpy, anndata are absent) and its tests pin
  no numbers there: **parity unpinned** for those pieces with respect to Keras /
  scanpy themselves.  What an independent implementation present in this image
  can pin is pinned (tests/test_oracle_golden.py): the Dense -> BatchNorm ->
  ReLU stack with its backward pass and moving statistics on ``torch.nn.Linear``
  / ``torch.nn.BatchNorm1d`` under autograd; RMSprop (epsilon outside the
  root), Adagrad, Adadelta and SGD on ``torch.optim``; ReduceLROnPlateau on
  torch's scheduler (absolute threshold, patience p - 1); whole-network gradients
  of all autoencoder types on the autograd twin ``torch_ref.py``.
"""
