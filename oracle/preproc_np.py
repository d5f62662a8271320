"""Oracle (test infrastructure only): independent restatement of the preprocessing the
reference delegates to scanpy (dca/io.py:88-111, dca/api.py:163) -- written with explicit loops
/ different numpy idioms than dca_amd/io.py so that the two can be checked against each other.
scanpy itself is not installable in the build image (parity unpinned against scanpy; the
behaviour restated is its documented one)."""
import numpy as np


def gene_keep_mask(counts, min_counts=1):
    n, g = counts.shape
    keep = np.zeros(g, dtype=bool)
    for j in range(g):
        tot = 0.0
        for i in range(n):
            tot += float(counts[i, j])
        keep[j] = tot >= min_counts
    return keep


def cell_keep_mask(counts, min_counts=1):
    return np.array([float(np.sum(row.astype(np.float64))) >= min_counts for row in counts])


def normalize(counts, size_factors=True, logtrans=True, zscore=True):
    """Returns (X, size_factors, n_counts) for dense float counts (no filtering)."""
    c = counts.astype(np.float64)
    n_counts = c.sum(axis=1)
    if size_factors:
        sf = n_counts / np.median(n_counts)
        x = c / sf.reshape(-1, 1)
    else:
        sf = np.ones(len(c))
        x = c
    if logtrans:
        x = np.log(1.0 + x)
    if zscore:
        mu = x.mean(axis=0)
        sd = x.std(axis=0, ddof=1)
        sd[sd == 0] = 1.0
        x = (x - mu) / sd
    return x, sf, n_counts
