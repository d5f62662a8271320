"""Native TSV writer (include/dcahost.h) against the reference's own call: pandas
DataFrame.to_csv(sep='\\t', float_format='%.6f') of dca/io.py:120-129 -- byte for byte."""
import ctypes
import os

import numpy as np
import pandas as pd
import pytest

from dca_amd import hostlib
from dca_amd.io import write_text_matrix


def _pandas_bytes(tmp_path, matrix, rownames, colnames, transpose=False):
    if transpose:
        matrix = matrix.T
        rownames, colnames = colnames, rownames
    fn = os.path.join(tmp_path, 'ref.tsv')
    pd.DataFrame(matrix, index=rownames, columns=colnames).to_csv(fn, sep='\t', index=(rownames is not None),
                                                                  header=(colnames is not None), float_format='%.6f')
    return open(fn, 'rb').read()


def _hard_values(dtype):
    rng = np.random.RandomState(5)
    info = np.finfo(dtype)
    v = [0.0, -0.0, 1.0, -1.0, 0.5e-6, 1.5e-6, 2.5e-6, -0.5e-6, 4.9999995e-7, 5.0000005e-7, 1e-7, -1e-9, 0.1234565,
         0.1234575, 1e6, 1e-5, 1e4, 123456.7890125, 999999.9999995, 8388608.5, 16777216.0, 9.2e12, 9.3e12, 1e15, 1e20,
         float(info.max), float(info.tiny), float(info.tiny) / 8, np.inf, -np.inf, np.nan]
    # exact binary ties at the 6th decimal only exist for multiples of 2^-k 5^-6 ... k/2^7 * 1e-6 is not
    # representable, but x.5 micro-units are when x.5e-6 * 2^n is an integer: generate from the integer side
    ties = (np.arange(1, 200, 2, dtype=np.float64) * 0.5) * 15625.0 / 2 ** 20      # (k/2) * 1e-6 * 1e6 * 15625 / 2^20
    v += list(rng.uniform(-3, 3, 500)) + list(np.exp(rng.uniform(-20, 25, 500))) + list(-np.exp(rng.uniform(-20, 5, 200)))
    v += list(rng.randint(0, 2000, 200).astype(np.float64)) + list(ties)
    return np.array(v, dtype=dtype)


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
def test_format_is_python_percent_6f(dtype):
    v = _hard_values(dtype)
    got = hostlib.format_values(v).split(b'\t')
    assert len(got) == v.size
    for x, g in zip(v, got):
        want = b'' if np.isnan(x) else ('%.6f' % float(x)).encode()
        assert g == want, (repr(float(x)), g, want)


def test_format_all_halfway_cases_float32():
    """Every float32 whose exact value ends in ...5 at the 7th decimal (true ties) in a small range:
    round-half-even on the exact binary value."""
    k = np.arange(1, 4096, dtype=np.float64)
    v = (k / 128.0).astype(np.float32)                  # k/128 = exact, 7 decimals ending in 5 for odd k*78125
    got = hostlib.format_values(v).split(b'\t')
    for x, g in zip(v, got):
        assert g == ('%.6f' % float(x)).encode()


def test_format_random_bit_patterns_float32():
    rng = np.random.RandomState(11)
    bits = rng.randint(0, 2 ** 32, 200000, dtype=np.uint64).astype(np.uint32)
    v = bits.view(np.float32)
    v = v[np.isfinite(v)]
    got = hostlib.format_values(v).split(b'\t')
    want = [('%.6f' % float(x)).encode() for x in v]
    assert got == want


def test_format_random_bit_patterns_float64():
    rng = np.random.RandomState(12)
    v = rng.randint(0, 2 ** 63, 50000, dtype=np.int64).view(np.float64)
    v = np.concatenate([v[np.isfinite(v)], np.exp(rng.uniform(-30, 30, 50000)) * rng.choice([-1, 1], 50000)])
    got = hostlib.format_values(v).split(b'\t')
    want = [('%.6f' % float(x)).encode() for x in v]
    assert got == want


@pytest.mark.parametrize('dtype', [np.float32, np.float64])
@pytest.mark.parametrize('transpose', [False, True])
@pytest.mark.parametrize('names', ['both', 'rows', 'cols', 'none'])
def test_write_text_matrix_bytes_equal_pandas(tmp_path, dtype, transpose, names):
    rng = np.random.RandomState(3)
    n, g = 37, 23
    m = (np.exp(rng.uniform(-12, 9, (n, g))) * (rng.uniform(size=(n, g)) > 0.3)).astype(dtype)
    m[3, 4] = np.nan
    m[5, 6] = np.inf
    m[7, 8] = -0.0
    rn = ['cell_%d' % i for i in range(n)] if names in ('both', 'rows') else None
    cn = ['gene-%d' % i for i in range(g)] if names in ('both', 'cols') else None
    fn = os.path.join(tmp_path, 'out.tsv')
    write_text_matrix(m, fn, rownames=rn, colnames=cn, transpose=transpose)
    assert open(fn, 'rb').read() == _pandas_bytes(tmp_path, m, rn, cn, transpose)


def test_many_blocks_many_threads_and_views(tmp_path):
    """More row blocks than threads, ragged last block, non-contiguous views, integer names."""
    rng = np.random.RandomState(4)
    big = rng.gamma(0.3, 5.0, (1500, 700)).astype(np.float32)
    for view, tr in ((big, False), (big, True), (big[::2, 5:400], False), (big[10:1300:3, ::7], True)):
        rn = np.arange(view.shape[0])
        cn = ['g%d' % i for i in range(view.shape[1])]
        fn = os.path.join(tmp_path, 'o.tsv')
        write_text_matrix(view, fn, rownames=rn, colnames=cn, transpose=tr)
        assert open(fn, 'rb').read() == _pandas_bytes(tmp_path, view, rn, cn, tr)
    fn = os.path.join(tmp_path, 'one.tsv')
    hostlib.write_tsv(fn, big, threads=1)
    one = open(fn, 'rb').read()
    hostlib.write_tsv(fn, big, threads=7)
    assert open(fn, 'rb').read() == one


def test_empty_and_fallbacks(tmp_path):
    fn = os.path.join(tmp_path, 'e.tsv')
    m = np.zeros((0, 4), np.float32)
    write_text_matrix(m, fn, colnames=list('abcd'))
    assert open(fn, 'rb').read() == _pandas_bytes(tmp_path, m, None, list('abcd'))
    # names that need quoting and integer matrices take the pandas call itself
    m = np.arange(6, dtype=np.float32).reshape(2, 3)
    write_text_matrix(m, fn, rownames=['a\tb', 'c'], colnames=['x', 'y"', 'z'])
    assert open(fn, 'rb').read() == _pandas_bytes(tmp_path, m, ['a\tb', 'c'], ['x', 'y"', 'z'])
    mi = np.arange(6).reshape(2, 3)
    write_text_matrix(mi, fn, rownames=['r0', 'r1'])
    assert open(fn, 'rb').read() == _pandas_bytes(tmp_path, mi, ['r0', 'r1'], None)


def test_errors(tmp_path):
    with pytest.raises(OSError):
        hostlib.write_tsv(os.path.join(tmp_path, 'no', 'such', 'dir', 'x.tsv'), np.zeros((2, 2), np.float32))
    with pytest.raises(ValueError):
        hostlib.write_tsv(os.path.join(tmp_path, 'x.tsv'), np.zeros((2, 2), np.float32), rownames=['a'])
    with pytest.raises(TypeError):
        hostlib.write_tsv(os.path.join(tmp_path, 'x.tsv'), np.zeros((2, 2), np.int32))


def test_exports():
    hostlib.lib()                                   # builds the library on first use
    L = ctypes.CDLL(hostlib._build.HOST_LIB)
    hdr = open(os.path.join(os.path.dirname(hostlib._build.HERE), 'include', 'dcahost.h')).read()
    import re
    declared = set(re.findall(r'\b(dcahost_[a-z0-9_]+)\s*\(', hdr))
    assert declared == set(hostlib._SIGNATURES)
    for name in declared:
        assert hasattr(L, name)


def test_streamed_writer_writes_the_bytes_of_the_one_shot_writer(tmp_path):
    """dcahost_tsv_stream_* (the fused predict writer hands over gene blocks as they leave the GPU) = dcahost_write_tsv_f32
    of the whole matrix: header, row names, values, with row blocks of any size, padded rows and any thread count."""
    from dca_amd import hostlib
    rng = np.random.RandomState(0)
    m = (rng.randn(37, 211) * 100).astype(np.float32)
    m[3, 5] = np.nan; m[0, 0] = -0.0; m[7, 7] = np.inf
    rn = ['g%d' % i for i in range(37)]; cn = ['c%d' % i for i in range(211)]
    hostlib.write_tsv(str(tmp_path / 'a.tsv'), m, rn, cn)
    padded = np.zeros((37, 300), np.float32); padded[:, :211] = m
    with hostlib.TsvStream(str(tmp_path / 'b.tsv'), 211, colnames=cn, index=True) as st:
        st.rows(padded[:10], rn[:10]); st.rows(padded[10:11], rn[10:11]); st.rows(padded[11:11], []); st.rows(padded[11:], rn[11:], threads=3)
    assert (tmp_path / 'a.tsv').read_bytes() == (tmp_path / 'b.tsv').read_bytes()


def test_threads_argument_sizes_the_host_pools(tmp_path):
    """`threads` of dca() / train() / the CLI (dca/train.py:41-48 sized TensorFlow's CPU pools with it) becomes the default
    thread count of the native host stages; the files are the same bytes whatever the count."""
    import numpy as np
    from dca_amd import hostlib
    m = np.random.RandomState(0).rand(300, 40).astype(np.float32)
    old = hostlib.DEFAULT_THREADS
    try:
        outs = []
        for n in (1, 3, 0):
            assert hostlib.set_threads(n) == n
            p = tmp_path / ('t%d.tsv' % n)
            hostlib.write_tsv(str(p), m)
            outs.append(p.read_bytes())
            assert hostlib.checksum(m) == hostlib.checksum(m, threads=2)
        assert outs[0] == outs[1] == outs[2]
    finally:
        hostlib.set_threads(old)
