"""Native text-matrix reader (include/dcahost.h dcahost_tsv_*, dca_amd/csrc/dcahost_read.cpp) against the pandas call it
stands in for (dca/io.py:59 sc.read(first_column_names=True), restated in dca_amd/io.py::read_text): same matrix (bit for
bit as float32), same row / column names; what it does not implement must fall back, not differ."""
import os

import numpy as np
import pandas as pd
import pytest

from dca_amd import hostlib, io as dio


def _pandas(path, sep):
    df = pd.read_csv(path, sep=sep, index_col=0)
    return df.values.astype(np.float32), list(df.index.astype(str)), list(df.columns.astype(str))


def _same(path, sep='\t'):
    got = hostlib.read_tsv(path, sep)
    assert got is not None
    X, rows, cols = got
    Xp, rp, cp = _pandas(path, sep)
    assert X.shape == Xp.shape and cols == cp
    np.testing.assert_array_equal(X, Xp)                     # NaN == NaN here; -0 == 0 (pandas reads '-0' of an integer column as 0)
    ad = dio.read_text(path)
    assert list(ad.obs_names) == rp and list(ad.var_names) == cp
    np.testing.assert_array_equal(np.isnan(ad.X), np.isnan(Xp))
    return X, rows, cols


def test_integer_count_matrix(tmp_path):
    rng = np.random.RandomState(0)
    m = rng.poisson(0.7, (53, 41)).astype(np.int64)
    m[3, 4] = 123456789
    df = pd.DataFrame(m, index=['cell%d' % i for i in range(53)], columns=['g%d' % j for j in range(41)])
    p = str(tmp_path / 'counts.tsv')
    df.to_csv(p, sep='\t')
    X, rows, cols = _same(p)
    np.testing.assert_array_equal(X, m.astype(np.float32))
    assert rows[0] == 'cell0' and cols[-1] == 'g40'


@pytest.mark.parametrize('crlf,trailing_newline,corner', [(False, True, True), (True, True, False), (False, False, True),
                                                          (True, False, False)])
def test_decimals_exponents_missing_values_and_line_ends(tmp_path, crlf, trailing_newline, corner):
    rng = np.random.RandomState(1)
    vals = rng.standard_normal((37, 9)) * 10.0 ** rng.randint(-12, 12, (37, 9))
    nl = '\r\n' if crlf else '\n'
    lines = [('names' + '\t' if corner else '') + '\t'.join('c%d' % j for j in range(9))]
    spell = ['', 'NA', 'nan', 'NaN', 'inf', '-inf', '+3', '1e5', '1E-3', '.5', '5.', '-0', '007', '1e-400', ' 7 ']
    for i in range(37):
        f = [repr(float(v)) for v in vals[i]]
        f[i % 9] = spell[i % len(spell)]
        lines.append('r%d\t' % i + '\t'.join(f))
    lines.insert(5, '')                                      # a blank line in the middle
    p = str(tmp_path / 'm.tsv')
    with open(p, 'w', newline='') as fh:
        fh.write(nl.join(lines) + (nl if trailing_newline else ''))
    _same(p)


def test_csv_and_numeric_looking_names(tmp_path):
    p = str(tmp_path / 'm.csv')
    with open(p, 'w') as fh:
        fh.write(',a,b,c\n001,1,2,3\n002,4,5,6\n10,7,8,9\n')
    X, rows, cols = _same(p, ',')
    assert list(dio.read_text(p).obs_names) == ['1', '2', '10']          # pandas reads this name column as integers
    np.testing.assert_array_equal(X, np.arange(1, 10, dtype=np.float32).reshape(3, 3))


@pytest.mark.parametrize('body', ['\ta\tb\nr1\t1\t"2"\n',            # quoted field
                                  '\ta\tb\nr1\t1\t2\nr2\t3\n',          # ragged: too few
                                  '\ta\tb\nr1\t1\t2\t3\nr2\t3\t4\t5\nr3\t1\n',   # ragged after a consistent start
                                  '\ta\tb\nr1\t1\tx\n',                 # text in a numeric column
                                  '\ta\ta\nr1\t1\t2\n'])                # duplicated column label (pandas renames it)
def test_what_the_native_reader_leaves_to_pandas(tmp_path, body):
    p = str(tmp_path / 'odd.tsv')
    with open(p, 'w') as fh:
        fh.write(body)
    native = hostlib.read_tsv(p, '\t')
    if native is not None:                                    # only the duplicated labels get this far
        assert len(set(native[2])) != len(native[2])
    try:
        want = pd.read_csv(p, sep='\t', index_col=0)
    except Exception:
        with pytest.raises(Exception):
            dio.read_text(p)
        return
    ad = dio.read_text(p) if want.values.dtype.kind in 'fiu' else None
    if ad is not None:
        assert list(ad.var_names) == list(want.columns.astype(str))
        np.testing.assert_array_equal(np.nan_to_num(ad.X, nan=-1), np.nan_to_num(want.values.astype(np.float32), nan=-1))


def test_large_file_is_parsed_by_several_threads(tmp_path):
    """> 4 MB of text per share: the shares must meet exactly at line boundaries (row order, nothing lost or doubled)."""
    rng = np.random.RandomState(2)
    n, g = 6000, 700
    m = rng.poisson(0.5, (n, g)).astype(np.float32)
    m[rng.rand(n, g) < 0.01] *= 1.5
    p = str(tmp_path / 'big.tsv')
    hostlib.write_tsv(p, m, rownames=['c%05d' % i for i in range(n)], colnames=['g%d' % j for j in range(g)])
    assert os.path.getsize(p) > 3 * (4 << 20)
    X, rows, cols = hostlib.read_tsv(p, '\t', threads=7)
    assert rows == ['c%05d' % i for i in range(n)] and cols == ['g%d' % j for j in range(g)]
    np.testing.assert_array_equal(X, m)                      # '%.6f' of these values parses back exactly
    Xp, rp, cp = _pandas(p, '\t')
    np.testing.assert_array_equal(X, Xp)
