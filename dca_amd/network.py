"""Autoencoder classes: the object protocol of dca/network.py on the device engine.

``AE_types`` has the reference's keys (network.py:763-768).  Constructor arguments, ``build()``,
``save()``, ``predict()``, ``write()``, ``load_weights()`` and the attributes used by
``train()`` follow the reference; the Keras graph is replaced by ``self.engine``
(dca_amd.engine.Engine): one flat parameter buffer + HIP kernels.  ``predict`` runs ONE
inference forward per chunk of cells and emits mean*sf, dispersion, dropout and the latent
code together (the reference runs 3-4 separate Keras predict passes, network.py:188-211,
395-405).

Every key of the reference's ``AE_types`` builds and trains on the MI355X path: zinb-conddisp (the north-star
class, network.py:366-421), zinb (:496-550), nb-conddisp (:293-339), nb (:249-290) through the fused K-HEADS kernel;
poisson (:233-246), normal (:143-156), nb-shared / zinb-shared (:343-362, 464-491), nb-fork / zinb-fork (:553-760) and
zinb-elempi (:424-461) through the separate head kernels (engine.AE_HEADS).  Weight files are ``.npz`` archives of
the named parameters (``save_weights`` / ``load_weights``), not Keras HDF5: weights written by the reference cannot
be loaded and vice versa.
"""
import os
import pickle

import numpy as np
import torch

from . import engine as _engine
from .io import write_text_matrix

advanced_activations = ('PReLU', 'LeakyReLU')

_test_ops_factory = None     # tests only: see override_ops()


class override_ops:
    """Context manager used by the CPU test-suite to inject an oracle-backed ops object into
    networks built inside it.  Never used by the product: without it build() instantiates
    HipOps, which raises when the HIP library or the GPU is missing."""

    def __init__(self, factory):
        self.factory = factory

    def __enter__(self):
        global _test_ops_factory
        self.prev, _test_ops_factory = _test_ops_factory, self.factory

    def __exit__(self, *a):
        global _test_ops_factory
        _test_ops_factory = self.prev


_stage_cache = {}


def _staging(want, cols, chunk, pinned):
    """Two staging buffers [chunk, cols] per requested output, page-locked when a GPU is present.  ONE set is kept
    between calls (locking pages costs about as much as copying them): a call with other shapes drops it first."""
    keys = {k: (k, chunk, cols[k], pinned) for k in want}
    if any(key not in _stage_cache for key in keys.values()) or len(_stage_cache) != len(keys):
        _stage_cache.clear()
        for key in keys.values():
            _stage_cache[key] = [torch.empty((chunk, key[2]), dtype=torch.float32, pin_memory=pinned) for _ in range(2)]
    return {k: _stage_cache[key] for k, key in keys.items()}


def _host_copy(dst, src):
    """dst[...] = src on the host-thread pool of libdcahost.so; numpy when that library cannot be loaded (no compiler
    on the host): slower, same result."""
    try:
        from . import hostlib
        hostlib.parallel_copy(dst, src)
    except (OSError, RuntimeError, AttributeError):
        dst[...] = src


def lay_G(eng):
    return eng.lay.G_in


class Autoencoder():
    ae_type = 'normal'

    def __init__(self,
                 input_size,
                 output_size=None,
                 hidden_size=(64, 32, 64),
                 l2_coef=0.,
                 l1_coef=0.,
                 l2_enc_coef=0.,
                 l1_enc_coef=0.,
                 ridge=0.,
                 hidden_dropout=0.,
                 input_dropout=0.,
                 batchnorm=True,
                 activation='relu',
                 init='glorot_uniform',
                 file_path=None,
                 debug=False,
                 comm=None):
        self.input_size = input_size
        self.output_size = output_size
        self.hidden_size = hidden_size
        self.l2_coef = l2_coef
        self.l1_coef = l1_coef
        self.l2_enc_coef = l2_enc_coef
        self.l1_enc_coef = l1_enc_coef
        self.ridge = ridge
        self.hidden_dropout = hidden_dropout
        self.input_dropout = input_dropout
        self.batchnorm = batchnorm
        self.activation = activation
        self.init = init
        self.loss = None
        self.file_path = file_path
        self.extra_models = {}
        self.model = None
        self.encoder = None
        self.decoder = None
        self.debug = debug
        self.engine = None
        self.comm = comm
        self.seed = 0

        if self.output_size is None:
            self.output_size = input_size

        if isinstance(self.hidden_dropout, list):
            assert len(self.hidden_dropout) == len(self.hidden_size)
        else:
            self.hidden_dropout = [self.hidden_dropout] * len(self.hidden_size)

    # ------------------------------------------------------------------ build
    def _check_supported(self):
        unsupported = []
        if self.ae_type not in _engine.AE_HEADS:
            unsupported.append('ae_type=%r' % self.ae_type)
        if self.activation not in _engine.ACT_CODES:
            unsupported.append('activation=%r' % self.activation)
        if not isinstance(self.init, str) or self.init.lower() not in _engine.KERAS_INITIALIZERS:
            unsupported.append('init=%r' % (self.init,))
        if unsupported:
            raise NotImplementedError('not implemented on the MI355X path yet: ' + ', '.join(unsupported))

    def build(self):
        """network.py:92-141 + build_output: allocates the device engine and initialises the
        weights (glorot_uniform kernels, zero biases)."""
        self._check_supported()
        ops = _test_ops_factory() if _test_ops_factory is not None else None
        self.engine = _engine.Engine(self.ae_type, self.input_size, self.output_size,
                                     self.hidden_size, self.batchnorm, self.ridge, ops=ops,
                                     comm=self.comm, activation=self.activation,
                                     hidden_dropout=self.hidden_dropout, input_dropout=self.input_dropout,
                                     dropout_seed=self.seed, sharedpi=getattr(self, 'sharedpi', False))
        self.engine.init_params(self.seed, self.init)
        self.engine.set_regularizers(self.l1_coef, self.l2_coef, self.l1_enc_coef, self.l2_enc_coef)
        self.model = self.engine             # what train() drives (reference: the Keras Model)
        self.encoder = self.engine
        self.loss = self.ae_type

    def __getstate__(self):
        d = dict(self.__dict__)
        for k in ('engine', 'model', 'encoder', 'comm'):
            d[k] = None
        return d

    def save(self):
        """network.py:158-162: pickles the (unbuilt) configuration object."""
        if self.file_path:
            os.makedirs(self.file_path, exist_ok=True)
            with open(os.path.join(self.file_path, 'model.pickle'), 'wb') as f:
                pickle.dump(self, f)

    def save_weights(self, filename):
        np.savez(filename, **self.engine.get_params())

    def load_weights(self, filename):
        """network.py:164-167 (the reference reads Keras HDF5; here: the .npz of save_weights)."""
        with np.load(filename) as z:
            self.engine.set_params({k: z[k] for k in z.files})

    # ------------------------------------------------------------------ inference
    def _wanted(self, mode, return_info):
        want = set()
        if mode in ('latent', 'full'):
            if not self.engine.lay.hidden:
                # hidden_size=(): the reference's get_encoder (network.py:179-186) fails the same way at build time; here
                # the network builds and trains (its own fixture scripts fit it, data/test-biochemists-zinb.py) and only
                # the request for the centre layer fails
                raise ValueError('No such layer: center (hidden_size=() has no latent representation)')
            want.add('latent')
        if mode in ('denoise', 'full'):
            want.add('mean')
        return want

    def _run_predict(self, adata, want, chunk=None):
        """One inference pass over all cells; returns host arrays for the requested outputs."""
        eng = self.engine
        if chunk is None:       # rows per device -> host chunk: 2 page-locked staging buffers of this many rows per output
            chunk = eng.cfg.predict_chunk if hasattr(eng, 'cfg') else 1024
        n = adata.n_obs
        X = adata.X
        sf = np.asarray(adata.obs['size_factors'].values, dtype=np.float32)
        dd = getattr(adata, '_dca_device', None)
        if dd is not None and dd.n == n and dd.G == lay_G(eng) and dd.X.device == eng.dev and dd.matches(X):
            # K-PREP's tensors are still in HBM and still ARE adata.X
            eng.attach_device_data(dd.X, dd.Y, dd.sf, norm=dd.norm, compact=dd.compact)
            dd.compact = eng.cc if eng.cc is not None else (False if getattr(eng, 'cc_verdict', None) is False else None)
        else:
            eng.load_data(X, None, sf)
        chunk = min(chunk, n)
        eng.reserve(chunk)
        lay = eng.lay
        cols = {}
        for k in want:
            cols[k] = lay.hidden[eng.center] if k == 'latent' else lay.G_out
            if (k == 'dispersion' and 'disp' in lay.shared) or (k == 'dropout' and 'pi' in lay.shared):
                cols[k] = 1                               # Dense(1) heads of the *-shared networks
        pinned = eng.dev.type == 'cuda'
        outs = {k: np.empty((n, cols[k]), dtype=np.float32) for k in want}
        # Results leave through two page-locked staging buffers per output (kept for the life of the process: locking
        # pages costs about as much as copying them): the device -> host copy of chunk i runs asynchronously while a
        # pool of host threads moves chunk i-1 into the caller's arrays (include/dcahost.h dcahost_parallel_copy; one
        # thread alone is slower than the PCIe link and takes every first-touch page fault itself).
        stage = _staging(want, cols, chunk, pinned)
        events = [torch.cuda.Event() for _ in range(2)] if pinned else None

        def drain(slot, start, rows):
            if pinned:
                events[slot].synchronize()
            for k in want:
                _host_copy(outs[k][start:start + rows], stage[k][slot][:rows].numpy())

        prev = None
        for ci, s in enumerate(range(0, n, chunk)):
            b = min(chunk, n - s)
            res = eng.predict_chunk(s, b, want)
            for k in want:
                stage[k][ci % 2][:b].copy_(res[k], non_blocking=pinned)
            if pinned:
                events[ci % 2].record()
            if prev is not None:
                drain(*prev)
            prev = (ci % 2, s, b)
        if prev is not None:
            drain(*prev)
        return outs

    def predict(self, adata, mode='denoise', return_info=False, copy=False):
        """network.py:188-211."""
        assert mode in ('denoise', 'latent', 'full'), 'Unknown mode'
        adata = adata.copy() if copy else adata
        want = self._wanted(mode, return_info)
        if mode in ('latent', 'full'):
            print('dca: Calculating low dimensional representations...')
        if mode in ('denoise', 'full'):
            print('dca: Calculating reconstructions...')
        outs = self._run_predict(adata, want)
        self._store(adata, outs, mode, return_info)
        return adata if copy else None

    def _store(self, adata, outs, mode, return_info):
        if 'latent' in outs:
            adata.obsm['X_dca'] = outs['latent']
        if 'dispersion' in outs:
            adata.obsm['X_dca_dispersion'] = outs['dispersion']
        if 'dropout' in outs:
            adata.obsm['X_dca_dropout'] = outs['dropout']
        if mode in ('denoise', 'full'):
            adata.X = outs['mean']
        if mode == 'latent':
            adata.X = adata.raw.X.copy()  # recover normalized expression values (network.py:208-209)

    # ------------------------------------------------------------------ predict + write in one pass (the CLI's last step)
    def _write_extra(self, adata, file_path, colnames):
        """Result files that do not come from per-cell heads (the per-gene dispersion of the constant-dispersion types)."""

    def predict_write(self, adata, file_path, mode='full', colnames=None, gene_block=None):
        """``predict(adata, mode, return_info=True)`` followed by ``write(adata, file_path, mode, colnames)``
        (dca/train.py:176-190 -> dca/network.py:188-231, 395-421) for a caller whose only consumer is the output
        directory: the same files, byte for byte, without a cells x genes result matrix on the host.

        The reference computes cell by cell and writes gene by gene (``write_text_matrix(..., transpose=True)``); here
        the hidden stack runs once over all cells (its 64-unit output stays in HBM), then blocks of genes go through
        heads GEMM -> inference activations -> transpose on the device, arrive gene x cell in page-locked chunks and are
        formatted by the native writer (one thread per result file feeding its own pool) while the next block computes.
        ``adata.X`` / ``adata.obsm`` are NOT filled.  Networks whose heads are not per-gene Dense layers (shared, fork,
        elempi), tiny gene counts, or a host without the native writer take predict() + write()."""
        import queue
        import threading
        eng = self.engine
        lay = eng.lay
        cells = adata.obs_names.values
        genes = adata.var_names.values if colnames is None else colnames
        want = self._wanted(mode, True)
        head_keys = [k for k in ('mean', 'dispersion', 'dropout') if k in want]
        try:
            from . import hostlib
            hostlib.lib()
            native = not hostlib.names_need_quoting(cells) and not hostlib.names_need_quoting(genes)
        except Exception:
            native = False
        fusable = (native and eng.dev.type == 'cuda' and hasattr(eng.ops, 'transpose') and lay.G_out >= 256 and head_keys
                   and not (lay.shared or lay.fork or lay.elempi) and mode in ('denoise', 'full')
                   and len(genes) == lay.G_out and getattr(eng, 'cfg', None) is not None and eng.cfg.fused_write
                   and bool(lay.hidden))
        if not fusable:
            self.predict(adata, mode=mode, return_info=True)
            self.write(adata, file_path, mode=mode, colnames=colnames)
            return
        n, G = adata.n_obs, lay.G_out
        sf = np.asarray(adata.obs['size_factors'].values, dtype=np.float32)
        dd = getattr(adata, '_dca_device', None)
        if dd is not None and dd.n == n and dd.G == lay_G(eng) and dd.X.device == eng.dev and dd.matches(adata.X):
            eng.attach_device_data(dd.X, dd.Y, dd.sf, norm=dd.norm, compact=dd.compact)
            dd.compact = eng.cc if eng.cc is not None else (False if getattr(eng, 'cc_verdict', None) is False else None)
        else:
            eng.load_data(adata.X, None, sf)
        print('dca: Calculating low dimensional representations...')
        print('dca: Calculating reconstructions...')
        print('dca: Saving output(s)...')
        os.makedirs(file_path, exist_ok=True)
        latent = eng.hidden_all()
        gb = gene_block or int(min(1024, max(128, 64e6 // (4 * n))))
        nblk = max(1, G // gb)
        bounds = [G * i // nblk for i in range(nblk + 1)]          # near-equal blocks, none below 128 genes
        gbmax = max(b - a for a, b in zip(bounds[:-1], bounds[1:]))
        ldn = (n + 3) // 4 * 4
        dev_out = [{k: torch.zeros(gbmax, ldn, dtype=torch.float32, device=eng.dev) for k in head_keys} for _ in range(2)]
        stage = [{k: torch.empty((gbmax, n), dtype=torch.float32, pin_memory=True) for k in head_keys} for _ in range(2)]
        files = {'mean': 'mean.tsv', 'dispersion': 'dispersion.tsv', 'dropout': 'dropout.tsv'}
        # (the reference passes the cell names to the mean file only: dispersion.tsv / dropout.tsv have no header line,
        # network.py:413-421)
        streams = {k: hostlib.TsvStream(os.path.join(file_path, files[k]), n, colnames=cells if k == 'mean' else None,
                                        index=True) for k in head_keys}
        free = [{k: threading.Event() for k in head_keys} for _ in range(2)]
        for slot in free:
            for ev in slot.values():
                ev.set()
        jobs = {k: queue.Queue() for k in head_keys}
        errors = []
        threads_per_file = max(1, (os.cpu_count() or 1) // len(head_keys))

        def writer(k):
            try:
                while True:
                    job = jobs[k].get()
                    if job is None:
                        return
                    slot, g0, g1, ev = job
                    ev.synchronize()
                    streams[k].rows(stage[slot][k][:g1 - g0].numpy(), genes[g0:g1], threads=threads_per_file)
                    free[slot][k].set()
            except Exception as e:          # noqa: BLE001 -- surfaced on the main thread below
                errors.append(e)
                for slot in free:
                    slot[k].set()

        workers = [threading.Thread(target=writer, args=(k,), daemon=True) for k in head_keys]
        for t in workers:
            t.start()
        try:
            for i in range(nblk):
                slot, g0, g1 = i % 2, bounds[i], bounds[i + 1]
                for k in head_keys:
                    free[slot][k].wait()
                    free[slot][k].clear()
                if errors:
                    break
                eng.heads_gene_block(g0, g1 - g0, set(head_keys), dev_out[slot])
                for k in head_keys:
                    stage[slot][k][:g1 - g0].copy_(dev_out[slot][k][:g1 - g0, :n], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                for k in head_keys:
                    jobs[k].put((slot, g0, g1, ev))
        finally:
            for k in head_keys:
                jobs[k].put(None)
            for t in workers:
                t.join()
            for st in streams.values():
                st.close()
        if errors:
            raise errors[0]
        if 'latent' in want:
            print('dca: Saving latent representations...')
            write_text_matrix(latent.cpu().numpy(), os.path.join(file_path, 'latent.tsv'), rownames=cells, transpose=False)
        self._write_extra(adata, file_path, genes)

    def write(self, adata, file_path, mode='denoise', colnames=None):
        """network.py:213-231."""
        colnames = adata.var_names.values if colnames is None else colnames
        rownames = adata.obs_names.values
        print('dca: Saving output(s)...')
        os.makedirs(file_path, exist_ok=True)
        if mode in ('denoise', 'full'):
            print('dca: Saving denoised expression...')
            write_text_matrix(adata.X, os.path.join(file_path, 'mean.tsv'),
                              rownames=rownames, colnames=colnames, transpose=True)
        if mode in ('latent', 'full'):
            print('dca: Saving latent representations...')
            write_text_matrix(adata.obsm['X_dca'], os.path.join(file_path, 'latent.tsv'),
                              rownames=rownames, transpose=False)


class PoissonAutoencoder(Autoencoder):
    ae_type = 'poisson'


class NBConstantDispAutoencoder(Autoencoder):
    """network.py:249-290: NB loss, one trainable dispersion per gene."""
    ae_type = 'nb'

    def predict(self, adata, mode='denoise', return_info=False, copy=False):
        res = super().predict(adata, mode, return_info, copy)
        adata = res if copy else adata
        if return_info:
            adata.var['X_dca_dispersion'] = self.engine.const_dispersion()      # network.py:278
        return adata if copy else None

    def _write_extra(self, adata, file_path, colnames):
        adata.var['X_dca_dispersion'] = self.engine.const_dispersion()          # network.py:278
        write_text_matrix(np.asarray(adata.var['X_dca_dispersion']).reshape(1, -1),
                          os.path.join(file_path, 'dispersion.tsv'), colnames=colnames, transpose=True)

    def write(self, adata, file_path, mode='denoise', colnames=None):
        colnames = adata.var_names.values if colnames is None else colnames
        super().write(adata, file_path, mode, colnames=colnames)
        if 'X_dca_dispersion' in adata.var_keys():
            # the reference calls .reshape on a pandas Series here (network.py:288, a bug);
            # the intended [1, G] row is written
            write_text_matrix(np.asarray(adata.var['X_dca_dispersion']).reshape(1, -1),
                              os.path.join(file_path, 'dispersion.tsv'),
                              colnames=colnames, transpose=True)


class NBAutoencoder(Autoencoder):
    """network.py:293-339: NB loss, dispersion conditioned on the cell (a Dense head)."""
    ae_type = 'nb-conddisp'

    def _wanted(self, mode, return_info):
        want = super()._wanted(mode, return_info)
        if return_info:
            want.add('dispersion')
        return want

    def write(self, adata, file_path, mode='denoise', colnames=None):
        colnames = adata.var_names.values if colnames is None else colnames
        super().write(adata, file_path, mode, colnames=colnames)
        if 'X_dca_dispersion' in adata.obsm_keys():
            write_text_matrix(adata.obsm['X_dca_dispersion'],
                              os.path.join(file_path, 'dispersion.tsv'),
                              colnames=colnames, transpose=True)


def _write_shared(adata, file_path):
    """The Dense(1) outputs are [n, 1]: one row of n values per file.  (The reference passes the gene
    names as row names here, network.py:336-339 / 413-421 inherited, which pandas rejects.)"""
    for key, fn in (('X_dca_dispersion', 'dispersion.tsv'), ('X_dca_dropout', 'dropout.tsv')):
        if key in adata.obsm_keys():
            write_text_matrix(adata.obsm[key], os.path.join(file_path, fn), transpose=True)


class NBSharedAutoencoder(NBAutoencoder):
    """network.py:343-362: dispersion = Dense(1), one value per cell, broadcast over the genes in the loss."""
    ae_type = 'nb-shared'

    def write(self, adata, file_path, mode='denoise', colnames=None):
        Autoencoder.write(self, adata, file_path, mode, colnames=colnames)
        _write_shared(adata, file_path)


class ZINBAutoencoder(Autoencoder):
    """network.py:366-421: the north-star class (ae_type 'zinb-conddisp')."""
    ae_type = 'zinb-conddisp'

    def _wanted(self, mode, return_info):
        want = super()._wanted(mode, return_info)
        if return_info:
            want |= {'dispersion', 'dropout'}
        return want

    def predict(self, adata, mode='denoise', return_info=False, copy=False, colnames=None):
        return super().predict(adata, mode, return_info, copy)

    def write(self, adata, file_path, mode='denoise', colnames=None):
        colnames = adata.var_names.values if colnames is None else colnames
        super().write(adata, file_path, mode, colnames=colnames)
        if 'X_dca_dispersion' in adata.obsm_keys():
            write_text_matrix(adata.obsm['X_dca_dispersion'],
                              os.path.join(file_path, 'dispersion.tsv'),
                              colnames=colnames, transpose=True)
        if 'X_dca_dropout' in adata.obsm_keys():
            write_text_matrix(adata.obsm['X_dca_dropout'],
                              os.path.join(file_path, 'dropout.tsv'),
                              colnames=colnames, transpose=True)


class ZINBAutoencoderElemPi(ZINBAutoencoder):
    ae_type = 'zinb-elempi'

    def __init__(self, sharedpi=False, **kwds):
        super().__init__(**kwds)
        self.sharedpi = sharedpi


class ZINBSharedAutoencoder(ZINBAutoencoder):
    """network.py:464-491: dropout and dispersion = Dense(1), one value per cell each."""
    ae_type = 'zinb-shared'

    def write(self, adata, file_path, mode='denoise', colnames=None):
        Autoencoder.write(self, adata, file_path, mode, colnames=colnames)
        _write_shared(adata, file_path)


class ZINBConstantDispAutoencoder(Autoencoder):
    """network.py:496-550: ZINB loss, one trainable dispersion per gene (ae_type 'zinb')."""
    ae_type = 'zinb'

    def _wanted(self, mode, return_info):
        want = super()._wanted(mode, return_info)
        if return_info:
            want.add('dropout')
        return want

    def predict(self, adata, mode='denoise', return_info=False, copy=False):
        res = super().predict(adata, mode, return_info, copy)
        adata = res if copy else adata
        if return_info:
            adata.var['X_dca_dispersion'] = self.engine.const_dispersion()      # network.py:530
        return adata if copy else None

    def _write_extra(self, adata, file_path, colnames):
        adata.var['X_dca_dispersion'] = self.engine.const_dispersion()          # network.py:530
        write_text_matrix(adata.var['X_dca_dispersion'].values.reshape(1, -1),
                          os.path.join(file_path, 'dispersion.tsv'), colnames=colnames, transpose=True)

    def write(self, adata, file_path, mode='denoise', colnames=None):
        colnames = adata.var_names.values if colnames is None else colnames
        super().write(adata, file_path, mode, colnames=colnames)
        if 'X_dca_dispersion' in adata.var_keys():
            write_text_matrix(adata.var['X_dca_dispersion'].values.reshape(1, -1),
                              os.path.join(file_path, 'dispersion.tsv'),
                              colnames=colnames, transpose=True)
        if 'X_dca_dropout' in adata.obsm_keys():
            write_text_matrix(adata.obsm['X_dca_dropout'],
                              os.path.join(file_path, 'dropout.tsv'),
                              colnames=colnames, transpose=True)


class ZINBForkAutoencoder(ZINBAutoencoder):
    ae_type = 'zinb-fork'


class NBForkAutoencoder(NBAutoencoder):
    ae_type = 'nb-fork'


AE_types = {'normal': Autoencoder, 'poisson': PoissonAutoencoder,
            'nb': NBConstantDispAutoencoder, 'nb-conddisp': NBAutoencoder,
            'nb-shared': NBSharedAutoencoder, 'nb-fork': NBForkAutoencoder,
            'zinb': ZINBConstantDispAutoencoder, 'zinb-conddisp': ZINBAutoencoder,
            'zinb-shared': ZINBSharedAutoencoder, 'zinb-fork': ZINBForkAutoencoder,
            'zinb-elempi': ZINBAutoencoderElemPi}
