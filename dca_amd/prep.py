"""K-PREP driver: dca/io.py:88-111 (``normalize``) on the GPU.

The reference preprocesses on the host with scanpy (filter_genes / filter_cells /
normalize_per_cell / log1p / scale), then Keras copies batches of the result to the device.
Here the raw counts are uploaded ONCE; gene / cell counts, size factors, log1p and the per-gene
z-score are streaming passes over the resident matrix (dcahip_prep_*), and the tensors the
training engine needs (X, Y, size factors) never leave HBM.  The host AnnData still receives
what the reference's ``normalize`` leaves behind -- ``adata.X`` (normalised), ``adata.raw``
(counts), ``obs['n_counts', 'size_factors']``, ``var['n_counts']`` -- so the function is a
drop-in for ``dca_amd.io.normalize``.

Index outputs (which genes / cells survive the filters) are bit-exact with the host path: the
sums are exact integer arithmetic.  The median of the library sizes is taken on the host
(``np.median`` over n values), exactly as the reference does.
"""
import numpy as np
import torch


def host_chunk_tensor(a):
    """A host chunk as a CPU tensor for the upload (read only here).  A read-only array (a select-everything subset of the
    AnnData stand-in shares its parent's matrix that way) is wrapped through a writeable VIEW: torch warns about -- and
    would misbehave on writes through -- non-writeable arrays, and nothing is written through this tensor."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    if not a.flags.writeable:
        v = a.view()
        try:
            v.flags.writeable = True
            a = v
        except ValueError:                  # the owner itself is read only (a memory map opened 'r', say): copy the chunk
            a = a.copy()
    return torch.from_numpy(a)



class DeviceData:
    """Preprocessed tensors resident on the device: X [n, ldx] (network input), Y [n, ldy]
    (raw counts, the loss target), sf [n]."""

    def __init__(self, X, Y, sf, n, G, host_x=None, norm=None):
        self.X, self.Y, self.sf, self.n, self.G = X, Y, sf, n, G
        # how X was made from Y (fac, do_log, mean, std: dca/io.py:99-109) -- lets the engine run the first layer on the
        # non-zero counts (Engine.attach_device_data); the compact byte store of Y is built once and kept here
        self.norm = norm
        self.compact = None
        # what adata.X held when these tensors were made: train() / predict() use the resident tensors only while the
        # host matrix still is that matrix (the reference always feeds the CURRENT adata.X, network.py:188-211)
        self.host_mark = fingerprint(host_x) if host_x is not None else None

    def matches(self, host_x):
        return self.host_mark is None or fingerprint(host_x) == self.host_mark


def fingerprint(X):
    """EXACT content mark of a host matrix: shape, dtype and a 64-bit hash of every byte (libdcahost, threaded: ~0.1 s for
    the 5.5 GB benchmark matrix; sparse matrices: of their data / indices / indptr arrays).  An in-place edit of any element
    between normalize() and train() / predict() changes it, and the resident tensors are then not used."""
    def mark(a):
        a = np.ascontiguousarray(a)
        try:
            from . import hostlib
            return hostlib.checksum(a)
        except (OSError, RuntimeError, AttributeError):          # no native host library (no compiler on the host): slower, same guarantee
            import hashlib
            return hashlib.blake2b(a.view(np.uint8).reshape(-1), digest_size=8).hexdigest()
    if hasattr(X, 'toarray'):
        # the arrays of the matrix' OWN format (no conversion; the format name is part of the mark: equal matrices in
        # another format or index order do not match, which only costs a re-upload)
        fmt = getattr(X, 'format', type(X).__name__)
        parts = [np.asarray(getattr(X, k)) for k in ('data', 'indices', 'indptr', 'row', 'col', 'offsets') if hasattr(X, k)]
        if not parts:
            parts = [np.asarray(X.tocsr().data)]
        return (tuple(X.shape), 'sparse', fmt) + tuple((str(p.dtype), mark(p)) for p in parts)
    Xa = np.asarray(X)
    return (tuple(Xa.shape), str(Xa.dtype), mark(Xa))


def _r4(x):
    return (x + 3) // 4 * 4


_STAGE = {}


def _stage_buffers(rows, cols):
    """Two page-locked [rows, cols] float32 buffers, kept for the life of the process (locking pages costs about as
    much as copying them)."""
    key = (rows, cols)
    if key not in _STAGE:
        _STAGE.clear()
        _STAGE[key] = [torch.empty(rows, cols, dtype=torch.float32).pin_memory() for _ in range(2)]
    return _STAGE[key]


def _upload(X, dev, chunk_rows=2048):
    """Host matrix -> [n, r4(G)] device tensor.  On a GPU the rows travel through two page-locked staging buffers:
    host threads fill buffer i + 1 (dcahost_parallel_copy) while buffer i crosses PCIe -- a pageable .to(device)
    of the 5.5 GB benchmark matrix moves ~13 GB/s, this ~45."""
    n, G = X.shape
    out = torch.zeros(n, _r4(G), dtype=torch.float32, device=dev)
    dense = not hasattr(X, 'toarray')
    staged = dev.type == 'cuda' and dense and X.dtype == np.float32 and X.flags['C_CONTIGUOUS'] and n * G >= (1 << 22)
    if not staged:
        for s in range(0, n, 8192):
            e = min(n, s + 8192)
            xs = X[s:e]
            xs = xs.toarray() if hasattr(xs, 'toarray') else np.asarray(xs)
            out[s:e, :G] = host_chunk_tensor(xs).to(dev)
        return out
    from . import hostlib
    chunk_rows = min(chunk_rows, n)
    stage = _stage_buffers(chunk_rows, G)
    events = [torch.cuda.Event() for _ in range(2)]
    for ci, s in enumerate(range(0, n, chunk_rows)):
        e = min(n, s + chunk_rows)
        slot = ci % 2
        if ci >= 2:
            events[slot].synchronize()                 # the copy that last read this buffer is done
        hostlib.parallel_copy(stage[slot][:e - s].numpy(), X[s:e])
        out[s:e, :G].copy_(stage[slot][:e - s], non_blocking=True)
        events[slot].record()
    torch.cuda.current_stream().synchronize()
    return out


def _download(X, n, G, chunk_rows=2048):
    """Device [n, >= G] tensor -> new host array [n, G] (the mirror of _upload)."""
    if X.device.type != 'cuda' or n * G < (1 << 22):
        return X[:n, :G].cpu().numpy()
    from . import hostlib
    out = np.empty((n, G), dtype=np.float32)
    chunk_rows = min(chunk_rows, n)
    stage = _stage_buffers(chunk_rows, G)
    events = [torch.cuda.Event() for _ in range(2)]
    prev = None
    for ci, s in enumerate(range(0, n, chunk_rows)):
        e = min(n, s + chunk_rows)
        slot = ci % 2
        stage[slot][:e - s].copy_(X[s:e, :G], non_blocking=True)
        events[slot].record()
        if prev is not None:
            ps, pe, pslot = prev
            events[pslot].synchronize()
            hostlib.parallel_copy(out[ps:pe], stage[pslot][:pe - ps].numpy())
        prev = (s, e, slot)
    if prev is not None:
        ps, pe, pslot = prev
        events[pslot].synchronize()
        hostlib.parallel_copy(out[ps:pe], stage[pslot][:pe - ps].numpy())
    return out


def gene_counts(ops, Y, n, G):
    dev = Y.device
    R = ops.prep_chunks(n)
    part = torch.zeros(R * 2 * _r4(G), dtype=torch.float64, device=dev)
    sums = torch.zeros(G, dtype=torch.float32, device=dev)
    ops.prep_col_pass(Y, Y.shape[1], n, G, None, False, None, 0, part)
    ops.prep_col_finish(part, R, G, float(n), sums, None, None)
    return sums


def cell_counts(ops, Y, n, G):
    out = torch.zeros(n, dtype=torch.float32, device=Y.device)
    ops.prep_row_sums(Y, Y.shape[1], n, G, out)
    return out


def transform(ops, Y, n, G, fac, logtrans_input, normalize_input, comm=None, return_norm=False):
    """X = scale(log1p(Y / fac)) on the device (each step optional).  With a communicator the
    per-gene statistics are those of all ranks' shards (data-parallel preprocessing).
    return_norm: also the description of the transform, (X, dict(fac, do_log, mean, std))."""
    dev = Y.device
    ld = Y.shape[1]
    X = torch.zeros(n, ld, dtype=torch.float32, device=dev)
    R = ops.prep_chunks(n)
    Gp = _r4(G)
    part = torch.zeros(R * 2 * Gp, dtype=torch.float64, device=dev)
    ops.prep_col_pass(Y, ld, n, G, fac, logtrans_input, X, ld, part)
    mean = std = None
    if normalize_input:
        mean = torch.zeros(Gp, dtype=torch.float32, device=dev)
        std = torch.ones(Gp, dtype=torch.float32, device=dev)
        n_total = float(n)
        if comm is not None and comm.world > 1:
            tot = part.view(R, 2 * Gp).sum(dim=0)
            cnt = torch.tensor([n_total], dtype=torch.float64, device=dev)
            comm.all_reduce_sum(tot); comm.all_reduce_sum(cnt)
            n_total = float(cnt.item())
            ops.prep_col_finish(tot, 1, G, n_total, None, mean, std)
        else:
            ops.prep_col_finish(part, R, G, n_total, None, mean, std)
        ops.prep_scale(X, ld, n, G, mean, std)
    if return_norm:
        return X, dict(fac=fac, do_log=bool(logtrans_input), mean=mean, std=std)
    return X


def resident_counts(X, ops=None, device=None):
    """(Y, gene_totals): the host count matrix on the device and the exact integer total of every gene."""
    if ops is None:
        from .ops import HipOps
        ops = HipOps()
    dev = torch.device(device) if device is not None else (
        torch.device('cuda', torch.cuda.current_device()) if ops.device_type == 'cuda' else torch.device('cpu'))
    n, G = X.shape
    Y = _upload(X, dev)
    return Y, gene_counts(ops, Y, n, G).cpu().numpy()


def normalize_device(adata, filter_min_counts=True, size_factors=True, normalize_input=True,
                     logtrans_input=True, ops=None, device=None, to_host=True, Y=None):
    """``io.normalize`` with the arithmetic on the device.  Returns (adata, DeviceData)."""
    from . import io as _io
    if ops is None:
        from .ops import HipOps
        ops = HipOps()
    dev = torch.device(device) if device is not None else (
        torch.device('cuda', torch.cuda.current_device()) if ops.device_type == 'cuda' else torch.device('cpu'))
    n, G = adata.X.shape
    if Y is None or tuple(Y.shape) != (n, _r4(G)):
        Y = _upload(adata.X, dev)       # (else: resident_counts() uploaded these counts already)

    if filter_min_counts:                                         # io.py:90-92
        gc = gene_counts(ops, Y, n, G).cpu().numpy()
        adata.var['n_counts'] = gc
        keep = gc >= 1
        if not keep.all():
            _io._subset(adata, cols=keep)
            idx = torch.as_tensor(np.nonzero(keep)[0], device=dev)
            G = int(keep.sum())
            Yn = torch.zeros(n, _r4(G), dtype=torch.float32, device=dev)
            Yn[:, :G] = Y.index_select(1, idx)
            Y = Yn
        cc = cell_counts(ops, Y, n, G).cpu().numpy()
        adata.obs['n_counts'] = cc
        keep = cc >= 1
        if not keep.all():
            _io._subset(adata, rows=keep)
            Y = Y.index_select(0, torch.as_tensor(np.nonzero(keep)[0], device=dev)).contiguous()
            n = int(keep.sum())

    if size_factors or normalize_input or logtrans_input:         # io.py:94-97
        adata.raw = adata.copy()
    else:
        adata.raw = adata

    fac_d = None
    if size_factors:                                              # io.py:99-101 (normalize_per_cell)
        counts = cell_counts(ops, Y, n, G).cpu().numpy()
        adata.obs['n_counts'] = counts
        keep = counts >= 1
        if not keep.all():
            _io._subset(adata, rows=keep)
            Y = Y.index_select(0, torch.as_tensor(np.nonzero(keep)[0], device=dev)).contiguous()
            counts = counts[keep]
            n = int(keep.sum())
        after = np.median(counts)
        c2 = counts + (counts == 0)
        fac = (c2 / after).astype(np.float32)
        adata.obs['size_factors'] = adata.obs.n_counts / np.median(adata.obs.n_counts)
        fac_d = torch.as_tensor(fac).to(dev)
        sf_d = torch.as_tensor(np.asarray(adata.obs['size_factors'].values, dtype=np.float32)).to(dev)
    else:
        adata.obs['size_factors'] = 1.0
        sf_d = torch.ones(n, dtype=torch.float32, device=dev)

    if fac_d is not None or logtrans_input or normalize_input:
        X, norm = transform(ops, Y, n, G, fac_d, logtrans_input, normalize_input, return_norm=True)
    else:
        X, norm = Y, dict(fac=None, do_log=False, mean=None, std=None)
    if to_host:
        adata.X = _download(X, n, G)
    return adata, DeviceData(X, Y, sf_d, n, G, host_x=adata.X if to_host else None, norm=norm)
