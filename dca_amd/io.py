"""IO / preprocessing surface of the reference (dca/io.py:53-131), re-hosted on numpy / pandas.

Same function names, arguments and outputs.  The reference delegates the arithmetic to scanpy
(filter_genes / filter_cells / normalize_per_cell / log1p / scale) and scikit-learn
(train_test_split); scanpy is not installable here, so its documented behaviour is restated
below (citations per function); scikit-learn IS used directly, so the train/test split indices
are bit-identical to the reference's.

Works on real ``anndata.AnnData`` objects when anndata is installed and on the bundled
MiniAnnData otherwise.
"""
import pickle

import numpy as np
import pandas as pd
import scipy.sparse as sp_sparse

from ._anndata import AnnData, MiniAnnData, is_anndata
from . import hostlib


def _dense(X):
    return X.toarray() if sp_sparse.issparse(X) else np.asarray(X)


# ------------------------------------------------------------------ scanpy restatements
def filter_genes_mask(X, min_counts=1):
    """sc.pp.filter_genes(X, min_counts): keep genes whose total count >= min_counts.
    Returns (gene_subset, number_per_gene) like scanpy does for array input (api.py:163)."""
    number = np.asarray(X.sum(axis=0)).reshape(-1)
    return number >= min_counts, number


def filter_cells_mask(X, min_counts=1):
    number = np.asarray(X.sum(axis=1)).reshape(-1)
    return number >= min_counts, number


def _subset(adata, rows=None, cols=None):
    if isinstance(adata, MiniAnnData):
        if rows is not None:
            idx = np.where(rows)[0]
            adata._X = adata._X[idx]
            adata.obs = adata.obs.iloc[idx].copy()
            adata.obsm = {k: np.asarray(v)[idx] for k, v in adata.obsm.items()}
            if adata._raw is not None:      # anndata keeps .raw aligned along obs
                adata._raw = type(adata._raw)(adata._raw.X[idx], adata._raw.var)
        if cols is not None:
            idx = np.where(cols)[0]
            adata._X = adata._X[:, idx]
            adata.var = adata.var.iloc[idx].copy()
    else:                                   # real anndata
        if rows is not None:
            adata._inplace_subset_obs(rows)
        if cols is not None:
            adata._inplace_subset_var(cols)


def filter_genes(adata, min_counts=1):
    keep, number = filter_genes_mask(adata.X, min_counts)
    adata.var['n_counts'] = number
    _subset(adata, cols=keep)


def filter_cells(adata, min_counts=1):
    keep, number = filter_cells_mask(adata.X, min_counts)
    adata.obs['n_counts'] = number
    _subset(adata, rows=keep)


def normalize_per_cell(adata):
    """sc.pp.normalize_per_cell(adata): n_counts = row sums (stored in obs), cells with
    n_counts < 1 are dropped, every row is divided by n_counts / median(n_counts)."""
    X = adata.X
    counts = np.asarray(X.sum(axis=1)).reshape(-1)
    adata.obs['n_counts'] = counts
    keep = counts >= 1
    if not keep.all():
        _subset(adata, rows=keep)
        X = adata.X
        counts = counts[keep]
    after = np.median(counts)
    counts = counts + (counts == 0)
    fac = counts / after
    if sp_sparse.issparse(X):
        X = sp_sparse.diags(1.0 / fac) @ X
        adata.X = X.astype(np.float32)
    else:
        X = np.asarray(X)
        if np.issubdtype(X.dtype, np.integer):
            X = X.astype(np.float32)
        adata.X = (X / fac[:, None].astype(X.dtype)).astype(X.dtype)


def log1p(adata):
    X = adata.X
    adata.X = X.log1p() if sp_sparse.issparse(X) else np.log1p(X)


def scale(adata):
    """sc.pp.scale(adata): per-gene z-score, zero centred, unbiased variance (ddof=1), no
    clipping; genes with zero standard deviation are divided by 1."""
    X = _dense(adata.X)
    n = X.shape[0]
    mean = X.mean(axis=0, dtype=np.float64)
    mean_sq = np.multiply(X, X).mean(axis=0, dtype=np.float64)
    var = (mean_sq - mean ** 2) * (n / (n - 1.0)) if n > 1 else np.zeros_like(mean)
    std = np.sqrt(np.maximum(var, 0))
    std[std == 0] = 1
    dt = X.dtype if np.issubdtype(X.dtype, np.floating) else np.float32
    adata.X = ((X - mean.astype(dt)) / std.astype(dt)).astype(dt)


# ------------------------------------------------------------------ reference surface
def read_text(filename, first_column_names=True):
    """Stand-in for sc.read(path, first_column_names=True) on TSV/CSV (io.py:59)."""
    sep = ',' if filename.endswith('.csv') else '\t'
    if first_column_names:
        native = _read_text_native(filename, sep)
        if native is not None:
            return native
    df = pd.read_csv(filename, sep=sep, index_col=0 if first_column_names else None)
    return AnnData(df.values.astype(np.float32),
                   obs=pd.DataFrame(index=df.index.astype(str)),
                   var=pd.DataFrame(index=df.columns.astype(str)))


def _read_text_native(filename, sep):
    """The plain numeric matrices DCA is fed, parsed on all host cores (include/dcahost.h dcahost_tsv_*; pandas reads
    them on one).  Returns None -- the caller then takes the pandas route -- for anything whose pandas semantics the
    native reader does not reproduce: quoted fields, ragged or non-numeric lines, duplicated or empty column labels."""
    import io as _pyio
    try:
        from . import hostlib
        got = hostlib.read_tsv(filename, sep)
    except Exception:
        return None
    if got is None:
        return None
    X, rows, cols = got
    if len(set(cols)) != len(cols) or any(c == '' for c in cols):
        return None             # pandas renames these ('x.1', 'Unnamed: 3')
    # the name column goes through pandas' own type inference ('007' is the integer 7 to read_csv, an empty name NaN)
    try:
        idx = pd.read_csv(_pyio.StringIO('\n'.join(rows) + '\n'), sep=sep, header=None, usecols=[0], skip_blank_lines=False,
                          dtype=None)[0] if rows else pd.Series([], dtype=object)
    except Exception:           # e.g. every name empty (EmptyDataError): pandas' own reader decides what the file means
        return None
    if len(idx) != len(rows):
        return None
    return AnnData(X, obs=pd.DataFrame(index=pd.Index(idx.values).astype(str)), var=pd.DataFrame(index=pd.Index(cols).astype(str)))


def read_dataset(adata, transpose=False, test_split=False, copy=False, check_counts=True):
    """dca/io.py:53-85."""
    if is_anndata(adata):
        if copy:
            adata = adata.copy()
    elif isinstance(adata, str):
        if adata.endswith('.h5ad'):
            import anndata as _ad          # ImportError if unavailable, like the reference
            adata = _ad.read_h5ad(adata)
        else:
            adata = read_text(adata, first_column_names=True)
    else:
        raise NotImplementedError

    if check_counts:
        # check if observations are unnormalized using first 10
        X_subset = adata.X[:10]
        norm_error = 'Make sure that the dataset (adata.X) contains unnormalized count data.'
        if sp_sparse.issparse(X_subset):
            assert (X_subset.astype(int) != X_subset).nnz == 0, norm_error
        else:
            assert np.all(X_subset.astype(int) == X_subset), norm_error

    if transpose:
        adata = adata.transpose()

    if test_split:
        from sklearn.model_selection import train_test_split
        train_idx, test_idx = train_test_split(np.arange(adata.n_obs), test_size=0.1, random_state=42)
        spl = pd.Series(['train'] * adata.n_obs)
        spl.iloc[test_idx] = 'test'
        adata.obs['dca_split'] = spl.values
    else:
        adata.obs['dca_split'] = 'train'

    adata.obs['dca_split'] = adata.obs['dca_split'].astype('category')
    print('dca: Successfully preprocessed {} genes and {} cells.'.format(adata.n_vars, adata.n_obs))
    return adata


def _device_prep_available():
    from . import config as _config
    if not _config.current().device_prep:
        return False
    try:
        import torch
        return torch.cuda.is_available()
    except ImportError:
        return False


def resident_counts(adata):
    """With a GPU present: the raw counts of ``adata`` uploaded once ([n, r4(G)] device tensor) together with their exact
    per-gene totals -- (Y, gene_totals) for ``normalize(..., _resident=Y)``; None when the host path is in use."""
    if not _device_prep_available():
        return None
    from . import prep
    return prep.resident_counts(adata.X)


def normalize(adata, filter_min_counts=True, size_factors=True, normalize_input=True,
              logtrans_input=True, device='auto', _resident=None):
    """dca/io.py:88-111.  With a GPU present the arithmetic runs on the device (K-PREP,
    dca_amd/prep.py): one upload of the raw counts, the training tensors stay in HBM and are
    handed to ``train`` through ``adata._dca_device``; the AnnData receives exactly what this
    host restatement would leave behind.  ``device=False`` (or DCA_AMD_DEVICE_PREP=0) forces the
    host path, which is the reference's own (scanpy on the CPU)."""
    if device is True or (device == 'auto' and _device_prep_available()):
        from . import prep
        adata, dd = prep.normalize_device(adata, filter_min_counts, size_factors, normalize_input,
                                          logtrans_input, Y=_resident)
        adata._dca_device = dd
        return adata
    if filter_min_counts:
        filter_genes(adata, min_counts=1)
        filter_cells(adata, min_counts=1)

    if size_factors or normalize_input or logtrans_input:
        adata.raw = adata.copy()
    else:
        adata.raw = adata

    if size_factors:
        normalize_per_cell(adata)
        adata.obs['size_factors'] = adata.obs.n_counts / np.median(adata.obs.n_counts)
    else:
        adata.obs['size_factors'] = 1.0

    if logtrans_input:
        log1p(adata)

    if normalize_input:
        scale(adata)

    return adata


def read_genelist(filename):
    genelist = list(set(open(filename, 'rt').read().strip().split('\n')))
    assert len(genelist) > 0, 'No genes detected in genelist file'
    print('dca: Subset of {} genes will be denoised.'.format(len(genelist)))
    return genelist


def write_text_matrix(matrix, filename, rownames=None, colnames=None, transpose=False):
    """dca/io.py:120-129: TSV with '%.6f' floats; transpose swaps the name vectors too.

    float32 / float64 matrices go through the native multi-threaded writer (include/dcahost.h,
    dca_amd/csrc/dcahost_tsv.cpp), which writes the bytes pandas' to_csv(float_format='%.6f') writes;
    the transposed (gene x cell) files are read through strides, not copied.  Anything else (integer
    matrices, names that need CSV quoting) takes the pandas call of the reference."""
    if transpose:
        matrix = matrix.T
        rownames, colnames = colnames, rownames
    m = matrix if isinstance(matrix, np.ndarray) else None
    if m is not None and m.ndim == 2 and m.dtype in (np.float32, np.float64) \
            and not hostlib.names_need_quoting(rownames) and not hostlib.names_need_quoting(colnames):
        hostlib.write_tsv(filename, m, rownames, colnames)
        return
    pd.DataFrame(matrix, index=rownames, columns=colnames).to_csv(filename,
                                                                  sep='\t',
                                                                  index=(rownames is not None),
                                                                  header=(colnames is not None),
                                                                  float_format='%.6f')


def read_pickle(inputfile):
    return pickle.load(open(inputfile, "rb"))
