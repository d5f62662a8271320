"""Compact count storage (one byte per count) and the description of how the network input is made from it.

The reference keeps two dense fp32 matrices per dataset: the normalised input ``adata.X`` and the raw counts
``adata.raw.X`` (dca/io.py:88-111, dca/train.py:83-89).  On the device the counts are stored ONCE as bytes
(``include/dcahip.h``, K-SPARSE): K-HEADS reads its targets from them and the first Dense layer
(dca/network.py:124-126) works on the non-zero counts only, because
``X = (log1p(counts / fac) - mean) / std`` is a per-row / per-gene function of the counts.
"""
import torch


class CompactCounts:
    """Yc [n, ldc] uint8 (255 = escape into the per-row overflow list) + the input normalisation:
    x[c, g] = (f(y / fac[c]) - mean[g]) / std[g], f = log1p if do_log; fac / mean / std may be None (1 / 0 / 1)."""

    def __init__(self, Yc, ldc, ovf_ptr, ovf_col, ovf_val, fac=None, do_log=False, mean=None, std=None):
        self.Yc, self.ldc = Yc, int(ldc)
        self.ovf_ptr, self.ovf_col, self.ovf_val = ovf_ptr, ovf_col, ovf_val
        self.fac, self.do_log, self.mean, self.std = fac, bool(do_log), mean, std
        self.lutp = None        # [n, 128] x 8 bytes: f(k / fac[r]) for the counts k = 0 .. 127 as bf16 pieces (dcahip_enc0_lut)

    def with_input(self, fac, do_log, mean, std, ops=None):
        """The same store with the description of the network input (None when the store is too large for the sparse first
        layer's 32-bit byte offsets).  The per-cell table of the common counts (512 B per cell) is made on the first use by
        a training step's first-layer kernels (ensure_lut): a predict-only run never builds it."""
        if ops is not None and self.Yc.shape[0] * self.ldc >= 2 ** 32:
            if not CompactCounts._warned_32bit:
                import sys
                print('dca_amd: %d x %d counts exceed the 32-bit byte offsets of the sparse first layer: dense first layer'
                      % (self.Yc.shape[0], self.ldc), file=sys.stderr)
                CompactCounts._warned_32bit = True
            return None
        return CompactCounts(self.Yc, self.ldc, self.ovf_ptr, self.ovf_col, self.ovf_val, fac, do_log, mean, std)

    _warned_32bit = False

    def ensure_lut(self, ops):
        """f(k / fac[r]) for the counts k = 0 .. 127 of every cell as bf16 pieces (dcahip_enc0_lut), once per store."""
        if self.lutp is None:
            n = self.Yc.shape[0]
            self.lutp = torch.zeros(n, ops.enc0_lut_entries(), 2, dtype=torch.int32, device=self.Yc.device)
            ops.enc0_lut(self.fac, self.do_log, n, self.lutp)
        return self.lutp


def build(ops, Y, n, G, chunk_rows=16384):
    """fp32 counts on the device [n, ld >= G] -> CompactCounts, or None when the matrix does not hold counts
    (negative / fractional / non-finite values: dca(..., check_counts=False) on arbitrary data)."""
    dev = Y.device
    ldc = ops.counts_compact_ld(G)
    Yc = torch.zeros(n, ldc, dtype=torch.uint8, device=dev)
    status = torch.zeros(2, dtype=torch.int32, device=dev)
    ops.counts_compact(Y, Y.shape[1], n, G, Yc, ldc, status)
    bad, esc = (int(v) for v in status.cpu())
    if bad:
        return None
    ovf_ptr = ovf_col = ovf_val = None
    if esc:
        # plumbing, once per dataset: (row, column, value) of every count >= 255 in row-major order
        rows, cols, vals = [], [], []
        for s in range(0, n, chunk_rows):
            e = min(n, s + chunk_rows)
            idx = (Y[s:e, :G] >= 255).nonzero()
            if idx.numel():
                rows.append(idx[:, 0] + s); cols.append(idx[:, 1]); vals.append(Y[s:e, :G][idx[:, 0], idx[:, 1]])
        rows = torch.cat(rows); cols = torch.cat(cols); vals = torch.cat(vals)
        counts = torch.bincount(rows, minlength=n)
        ovf_ptr = torch.zeros(n + 1, dtype=torch.int32, device=dev)
        ovf_ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
        ovf_col = cols.to(torch.int32).contiguous()
        ovf_val = vals.to(torch.float32).contiguous()
    return CompactCounts(Yc, ldc, ovf_ptr, ovf_col, ovf_val)
