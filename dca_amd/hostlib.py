"""ctypes binding of include/dcahost.h (dca_amd/csrc/libdcahost.so): host-side result writers.

The library is host-only C++ (g++), so unlike libdcahip.so it is built on first use when missing."""
import ctypes
import os

import numpy as np

from . import build as _build

_lib = None

_cpp = ctypes.POINTER(ctypes.c_char_p)
_SIGNATURES = {
    'dcahost_write_tsv_f32': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_long,
                                             ctypes.c_long, _cpp, _cpp, ctypes.c_int]),
    'dcahost_write_tsv_f64': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_long, ctypes.c_long,
                                             ctypes.c_long, _cpp, _cpp, ctypes.c_int]),
    'dcahost_format_f32': (ctypes.c_long, [ctypes.c_void_p, ctypes.c_long, ctypes.c_char_p, ctypes.c_long]),
    'dcahost_format_f64': (ctypes.c_long, [ctypes.c_void_p, ctypes.c_long, ctypes.c_char_p, ctypes.c_long]),
    'dcahost_parallel_copy': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int]),
    'dcahost_checksum': (ctypes.c_ulonglong, [ctypes.c_void_p, ctypes.c_long, ctypes.c_int]),
    'dcahost_tsv_open': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_char, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p),
                                        ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long),
                                        ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]),
    'dcahost_tsv_read_f32': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_char_p, ctypes.c_long,
                                            ctypes.c_char_p, ctypes.c_long]),
    'dcahost_tsv_close': (None, [ctypes.c_void_p]),
    'dcahost_tsv_stream_open': (ctypes.c_int, [ctypes.c_char_p, ctypes.c_long, _cpp, ctypes.c_int,
                                               ctypes.POINTER(ctypes.c_void_p)]),
    'dcahost_tsv_stream_rows_f32': (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_long, _cpp,
                                                   ctypes.c_int]),
    'dcahost_tsv_stream_close': (ctypes.c_int, [ctypes.c_void_p]),
}


def lib():
    global _lib
    if _lib is None:
        path = _build.build_host(verbose=False)
        L = ctypes.CDLL(path, use_errno=True)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _names(names, n):
    if names is None:
        return None, None
    enc = [str(x).encode('utf-8') for x in names]
    if len(enc) != n:
        raise ValueError('expected %d names, got %d' % (n, len(enc)))
    return (ctypes.c_char_p * n)(*enc), enc


def names_need_quoting(names):
    """csv.QUOTE_MINIMAL (pandas' to_csv) would quote these; the native writer emits names verbatim."""
    if names is None:
        return False
    return any(ch in s for s in map(str, names) for ch in ('\t', '"', '\n', '\r'))


DEFAULT_THREADS = 0          # 0: one per hardware thread (the library's own caps apply)


def set_threads(n):
    """Host threads of every native host stage from here on (result writers, input parser, staging copies, checksums): what
    the reference's ``threads`` argument sized for TensorFlow's CPU pools (dca/train.py:41-48) sizes here.  None / 0: all."""
    global DEFAULT_THREADS
    DEFAULT_THREADS = max(0, int(n or 0))
    return DEFAULT_THREADS


def write_tsv(path, matrix, rownames=None, colnames=None, threads=0):
    """matrix: 2-D float32 / float64 ndarray with ANY strides (a transposed view costs nothing);
    rownames / colnames: sequences or None."""
    m = np.asarray(matrix)
    if m.ndim != 2 or m.dtype not in (np.float32, np.float64):
        raise TypeError('write_tsv needs a 2-D float32 / float64 array')
    item = m.dtype.itemsize
    if m.size and (m.strides[0] % item or m.strides[1] % item or m.strides[0] < 0 or m.strides[1] < 0):
        m = np.ascontiguousarray(m)
    nr, nc = m.shape
    rs, cs = (m.strides[0] // item, m.strides[1] // item) if m.size else (nc, 1)
    rn, _keep_r = _names(rownames, nr)
    cn, _keep_c = _names(colnames, nc)
    fn = lib().dcahost_write_tsv_f32 if m.dtype == np.float32 else lib().dcahost_write_tsv_f64
    rc = fn(os.fsencode(path), m.ctypes.data, nr, nc, rs, cs, rn, cn, int(threads or DEFAULT_THREADS))
    if rc == -2:
        err = ctypes.get_errno()
        raise OSError(err, os.strerror(err), str(path))
    if rc != 0:
        raise ValueError('dcahost_write_tsv: invalid arguments')


class TsvStream:
    """A '%.6f' TSV written row block by row block (dcahost_tsv_stream_*): header at once, then rows(values, names) in
    order.  Same bytes as write_tsv of the whole matrix."""

    def __init__(self, path, ncols, colnames=None, index=True):
        self.h = ctypes.c_void_p()
        self.ncols, self.index = int(ncols), bool(index)
        cn, _keep = _names(colnames, self.ncols)
        rc = lib().dcahost_tsv_stream_open(os.fsencode(path), self.ncols, cn, int(self.index), ctypes.byref(self.h))
        if rc != 0:
            err = ctypes.get_errno()
            raise OSError(err, os.strerror(err), str(path))

    def rows(self, values, rownames=None, threads=0):
        """values: float32 [nrows, >= ncols] with unit column stride (any row stride)."""
        v = np.asarray(values)
        assert v.ndim == 2 and v.dtype == np.float32 and v.shape[1] >= self.ncols
        assert v.shape[0] == 0 or v.strides[1] == 4
        ld = v.strides[0] // 4 if v.shape[0] > 1 else max(self.ncols, v.shape[1])
        rn, _keep = _names(rownames, v.shape[0]) if self.index else (None, None)
        rc = lib().dcahost_tsv_stream_rows_f32(self.h, v.ctypes.data, v.shape[0], ld, rn, int(threads or DEFAULT_THREADS))
        if rc != 0:
            raise OSError(ctypes.get_errno(), 'dcahost_tsv_stream_rows_f32 failed (%d)' % rc)

    def close(self):
        if self.h:
            rc = lib().dcahost_tsv_stream_close(self.h)
            self.h = ctypes.c_void_p()
            if rc != 0:
                raise OSError(ctypes.get_errno(), 'closing the result file failed')

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False


def format_values(values):
    """'%.6f' of each value, tab-joined (bytes) -- the formatting kernel, for tests."""
    v = np.ascontiguousarray(values)
    fn = lib().dcahost_format_f32 if v.dtype == np.float32 else lib().dcahost_format_f64
    cap = 400 * max(1, v.size)
    buf = ctypes.create_string_buffer(cap)
    n = fn(v.ctypes.data, v.size, buf, cap)
    if n < 0:
        raise ValueError('dcahost_format failed')
    return buf.raw[:n]


def parallel_copy(dst, src, threads=0):
    """dst[...] = src for two C-contiguous numpy arrays of equal byte size, on several host threads."""
    assert dst.flags['C_CONTIGUOUS'] and src.flags['C_CONTIGUOUS'] and dst.nbytes == src.nbytes
    rc = lib().dcahost_parallel_copy(dst.ctypes.data, src.ctypes.data, dst.nbytes, int(threads or DEFAULT_THREADS))
    if rc != 0:
        raise RuntimeError('dcahost_parallel_copy failed (%d)' % rc)


def read_tsv(path, sep='\t', threads=0):
    """Numeric text matrix with a header line and a name column -> (float32 [n, g] array, row names, column names), parsed
    on several host threads; None when the file holds something the native reader leaves to pandas (quoted fields,
    ragged lines, text in a numeric column) or cannot be opened."""
    L = lib()
    h = ctypes.c_void_p()
    n, g, rb, cb = ctypes.c_long(), ctypes.c_long(), ctypes.c_long(), ctypes.c_long()
    rc = L.dcahost_tsv_open(os.fsencode(path), sep.encode('ascii'), int(threads or DEFAULT_THREADS), ctypes.byref(h), ctypes.byref(n),
                            ctypes.byref(g), ctypes.byref(rb), ctypes.byref(cb))
    if rc != 0:
        return None
    try:
        out = np.empty((n.value, g.value), dtype=np.float32)
        rbuf = ctypes.create_string_buffer(rb.value + 2)
        cbuf = ctypes.create_string_buffer(cb.value + 2)
        rc = L.dcahost_tsv_read_f32(h, out.ctypes.data, g.value, rbuf, rb.value + 2, cbuf, cb.value + 2)
        if rc != 0:
            return None
        rows = rbuf.value.decode('utf-8').split('\n') if n.value else []
        cols = cbuf.value.decode('utf-8').split('\n')
        if len(rows) != n.value or len(cols) != g.value:
            return None
        return out, rows, cols
    finally:
        L.dcahost_tsv_close(h)


def checksum(a, threads=0):
    """Exact 64-bit content mark of a C-contiguous ndarray (every byte; dcahost_checksum)."""
    a = np.ascontiguousarray(a)
    return int(lib().dcahost_checksum(a.ctypes.data, a.nbytes, int(threads or DEFAULT_THREADS)))
