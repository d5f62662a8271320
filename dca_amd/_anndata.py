"""Minimal AnnData stand-in.

The reference API works on ``anndata.AnnData`` (dca/api.py:146).  anndata / scanpy are not
installed in the build image; when the real package is importable it is used unchanged, and
this class only exists so that the drop-in surface (and its tests) runs without it.  It
implements exactly the members the DCA path touches: X, obs, var, obsm, uns, raw, n_obs,
n_vars, obs_names, var_names, copy(), transpose(), boolean/slice row indexing and the
``*_keys()`` helpers used by dca/test.py.
"""
import numpy as np
import pandas as pd

try:                                    # pragma: no cover - not available in the build image
    import anndata as _real_anndata
except Exception:                       # noqa: BLE001
    _real_anndata = None


def _copy_matrix(X):
    """X.copy(); large dense float matrices on several host threads (one thread copies the 5.5 GB benchmark matrix in
    ~0.6 s, first-touch page faults included)."""
    if hasattr(X, 'toarray') or X.nbytes < (1 << 26) or not X.flags['C_CONTIGUOUS']:
        return X.copy()
    try:
        from . import hostlib
        out = np.empty_like(X)
        hostlib.parallel_copy(out, X)
        return out
    except Exception:           # no native host library: numpy's copy
        return X.copy()


class _CowView(np.ndarray):
    """X of a select-everything subset: shares the parent's matrix until it is written to.  ``sub.X[...] = v`` copies the
    matrix into the subset first (as a write through an anndata view does) and never reaches the parent; other in-place
    operations (``sub.X += 1``) find the array read-only."""
    _owner = None

    def __array_finalize__(self, obj):
        self._owner = None                      # slices / results of arithmetic are plain read-only views

    def __setitem__(self, key, value):
        own = self._owner
        if own is None:
            raise ValueError('assignment destination is a read-only view of the parent AnnData\'s matrix; assign through '
                             'subset.X[...] = value (copies on write) or take subset.copy() first')
        if own._X is not self:
            # this reference already triggered the copy: later writes through it belong to the subset's own matrix
            own._X[key] = value
            return
        fresh = np.array(self, copy=True, subok=False)
        fresh[key] = value
        own._X = fresh
        own.__dict__.pop('_dca_device', None)   # the device-resident tensors belong to the parent's matrix


class _Raw:
    def __init__(self, X, var):
        self.X = X
        self.var = var

    @property
    def var_names(self):
        return self.var.index

    def copy(self):
        return _Raw(_copy_matrix(self.X), self.var.copy())


class MiniAnnData:
    def __init__(self, X, obs=None, var=None, obsm=None, uns=None, raw=None, dtype=None):
        X = X if hasattr(X, 'toarray') else np.asarray(X)
        if dtype is not None and not hasattr(X, 'toarray'):
            X = X.astype(dtype, copy=False)
        self._X = X
        n, g = X.shape
        self.obs = obs.copy() if obs is not None else pd.DataFrame(index=pd.RangeIndex(n).astype(str))
        self.var = var.copy() if var is not None else pd.DataFrame(index=pd.RangeIndex(g).astype(str))
        assert len(self.obs) == n and len(self.var) == g
        self.obsm = dict(obsm) if obsm else {}
        self.uns = dict(uns) if uns else {}
        self._raw = raw

    # -- data
    @property
    def X(self):
        return self._X

    @X.setter
    def X(self, v):
        v = v if hasattr(v, 'toarray') else np.asarray(v)
        assert v.shape == self._X.shape, 'X shape is fixed'
        self._X = v

    @property
    def raw(self):
        return self._raw

    @raw.setter
    def raw(self, ad):
        self._raw = None if ad is None else (ad if isinstance(ad, _Raw) else _Raw(ad.X, ad.var))

    @property
    def n_obs(self):
        return self._X.shape[0]

    @property
    def n_vars(self):
        return self._X.shape[1]

    @property
    def shape(self):
        return self._X.shape

    @property
    def obs_names(self):
        return self.obs.index

    @property
    def var_names(self):
        return self.var.index

    def obsm_keys(self):
        return list(self.obsm.keys())

    def var_keys(self):
        return list(self.var.columns)

    def obs_keys(self):
        return list(self.obs.columns)

    def uns_keys(self):
        return list(self.uns.keys())

    # -- structure
    def copy(self):
        return MiniAnnData(_copy_matrix(self._X), self.obs, self.var,
                           {k: np.array(v, copy=True) for k, v in self.obsm.items()},
                           dict(self.uns), None if self._raw is None else self._raw.copy())

    def transpose(self):
        Xt = self._X.T
        return MiniAnnData(Xt.copy() if not hasattr(Xt, 'toarray') else Xt.tocsr(), self.var, self.obs)

    T = property(transpose)

    def __getitem__(self, idx):
        """Row subsetting (boolean mask / index array / slice), optionally (rows, cols)."""
        cols = slice(None)
        if isinstance(idx, tuple):
            idx, cols = idx
        if isinstance(idx, pd.Series):
            idx = idx.values
        idx = np.arange(self.n_obs)[idx]
        cidx = np.arange(self.n_vars)[cols]
        if len(idx) == self.n_obs and len(cidx) == self.n_vars and (idx == np.arange(self.n_obs)).all() \
                and (cidx == np.arange(self.n_vars)).all():
            # selecting everything: a view-like object sharing the matrices, as anndata's own (lazy) views do -- dca()
            # takes adata[adata.obs.dca_split == 'train'] of a dataset without a test split (api.py:203), and copying
            # two 5.5 GB matrices there cost more than the GPU training
            # (read-only, like a view that has not been written to: a write through the subset must not reach the parent)
            Xv = self._X
            if isinstance(Xv, np.ndarray):
                Xv = Xv.view(_CowView)
                Xv.setflags(write=False)
            out = MiniAnnData(Xv, self.obs, self.var, dict(self.obsm), dict(self.uns), self._raw)
            if isinstance(Xv, _CowView):
                out._X = Xv                     # (the constructor's asarray keeps the subclass; the owner link is set here)
                Xv._owner = out
            dd = getattr(self, '_dca_device', None)
            if dd is not None:
                out._dca_device = dd
            return out
        X = self._X[idx][:, cidx]
        raw = None
        if self._raw is not None:
            raw = _Raw(self._raw.X[idx], self._raw.var)
        out = MiniAnnData(X, self.obs.iloc[idx], self.var.iloc[cidx],
                          {k: np.asarray(v)[idx] for k, v in self.obsm.items()}, dict(self.uns), raw)
        dd = getattr(self, '_dca_device', None)
        if dd is not None and len(idx) == self.n_obs and len(cidx) == self.n_vars and \
                (idx == np.arange(self.n_obs)).all() and (cidx == np.arange(self.n_vars)).all():
            out._dca_device = dd          # identity selection: the device-resident tensors still match
        return out

    def __repr__(self):
        return 'MiniAnnData object with n_obs x n_vars = %d x %d' % self.shape


AnnData = _real_anndata.AnnData if _real_anndata is not None else MiniAnnData


def is_anndata(obj):
    if isinstance(obj, MiniAnnData):
        return True
    return _real_anndata is not None and isinstance(obj, _real_anndata.AnnData)
