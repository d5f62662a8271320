"""dca_amd: MI355X-native implementation of the DCA ZINB-autoencoder training path.

Drop-in surface: ``dca_amd.api.dca`` (== ``dca.api.dca``), ``dca_amd.io``, ``dca_amd.train.train``,
``dca_amd.network.AE_types`` and the ``python -m dca_amd`` CLI.
"""
__version__ = '0.1.0'
