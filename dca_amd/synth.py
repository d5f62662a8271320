"""Synthetic single-cell count matrices generated directly in device memory.

Gamma-Poisson counts with extra dropout in the 68k-PBMC sparsity regime (about 93 % zeros at
G = 20 000, median library about 2 000 counts):
    gene log-mean  m_g ~ N(-3.2, 1.6^2),  cell library factor l_c ~ LogNormal(0, 0.4^2),
    lambda_cg = l_c * exp(m_g) * Gamma(2, 1/2),  y_cg ~ Poisson(lambda_cg) * Bernoulli(0.7),
then every gene and every cell is forced to hold at least one count (dca/api.py:163-164 asserts
it; sc.pp.normalize_per_cell would drop empty cells).

Used by bench.py / smoke() / the full-size property tests: the datasets named in
BASELINE.json are synthetic by definition and too large to be shipped or built on the host
(1M x 25k fp32 = 100 GB).  torch is the allocator and RNG here -- plumbing, not the timed path.
"""

import torch


def generate_counts(n, G, seed=20260925, device='cuda', row_offset=0, chunk=4096, dropout=0.3,
                    ld=None, gene_seed=None):
    """Returns Y [n, ld] fp32 (ld = G rounded up to 4, pad columns zero)."""
    ld = (G + 3) // 4 * 4 if ld is None else ld
    dev = torch.device(device)
    gg = torch.Generator(device=dev)
    gg.manual_seed(seed if gene_seed is None else gene_seed)        # gene means shared by all ranks
    m = torch.randn(G, generator=gg, device=dev) * 1.6 - 3.2
    em = torch.exp(m)
    g = torch.Generator(device=dev)
    g.manual_seed(seed + 7919 * (1 + row_offset))
    Y = torch.zeros(n, ld, dtype=torch.float32, device=dev)
    for s in range(0, n, chunk):
        b = min(chunk, n - s)
        lib = torch.exp(torch.randn(b, 1, generator=g, device=dev) * 0.4)
        # Gamma(k=2, theta=1/2) = -(1/2) * (ln u1 + ln u2)
        u = torch.rand(2, b, G, generator=g, device=dev).clamp_min_(1e-12)
        gam = -0.5 * (torch.log(u[0]) + torch.log(u[1]))
        lam = lib * em * gam
        y = torch.poisson(lam, generator=g)
        y *= (torch.rand(b, G, generator=g, device=dev) >= dropout)
        Y[s:s + b, :G] = y
        del u, gam, lam, y
    # every cell / gene gets >= 1 count
    rs = Y.sum(dim=1)
    empty = torch.nonzero(rs == 0).flatten()
    if empty.numel():
        Y[empty, torch.randint(0, G, (empty.numel(),), generator=g, device=dev)] = 1.0
    cs = Y[:, :G].sum(dim=0)
    emptyg = torch.nonzero(cs == 0).flatten()
    if emptyg.numel():
        Y[torch.randint(0, n, (emptyg.numel(),), generator=g, device=dev), emptyg] += 1.0
    return Y


def normalize_on_device(Y, G, comm=None, chunk=8192):
    """dca/io.py:88-111 on device tensors: size factors = n_counts / median, log1p, per-gene
    z-score (ddof = 1).  Returns (X [n, ld], sf [n]).  With a communicator the statistics are
    global over all ranks' shards."""
    n, ld = Y.shape
    counts = Y.sum(dim=1)
    if comm is not None and comm.world > 1:
        allc = comm.all_gather(torch.nn.functional.pad(counts, (0, 0)))   # equal shard sizes only
        med = allc.flatten().median()
    else:
        med = counts.median()
    sf = counts / med
    X = torch.empty_like(Y)
    s1 = torch.zeros(ld, dtype=torch.float64, device=Y.device)
    s2 = torch.zeros(ld, dtype=torch.float64, device=Y.device)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        x = torch.log1p(Y[s:e] / sf[s:e, None])
        X[s:e] = x
        xd = x.double()
        s1 += xd.sum(dim=0)
        s2 += (xd * xd).sum(dim=0)
    ntot = torch.tensor([float(n)], dtype=torch.float64, device=Y.device)
    if comm is not None and comm.world > 1:
        comm.all_reduce_sum(s1); comm.all_reduce_sum(s2); comm.all_reduce_sum(ntot)
    nt = float(ntot.item())
    mean = s1 / nt
    var = (s2 / nt - mean * mean) * (nt / (nt - 1.0))
    std = torch.sqrt(var.clamp_min(0))
    std[std == 0] = 1.0
    mean32, std32 = mean.float(), std.float()
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        X[s:e] = (X[s:e] - mean32) / std32
    if ld > G:
        X[:, G:] = 0
    return X, sf


# ---------------------------------------------------------------------------------------------------------------
# A second generator whose output is a pure function of (n, G, seed) on ANY device: integer hashing + thresholds
# computed with IEEE basic arithmetic only.  The parity fixtures of the full-size configurations are made with it on a
# CPU (tests/golden/make_c3_epoch_golden.py) and the GPU tests regenerate the identical matrix in HBM
# (tests/test_engine_gpu.py::test_c3_whole_epoch_matches_oracle checks a checksum first).  Same model as above with the
# two continuous factors on grids: cell library factor exp(0.4 z), z on the 16 levels of a Binomial(15, 1/2); gene
# log-mean -3.2 + 1.6 z, z on the 31 levels of a Binomial(30, 1/2); given both, the count is the zero-inflated
# (30 %) Gamma-Poisson = negative binomial (r = 2) drawn by inverse CDF from one 32-bit hash of (row, column).
_M32 = 0xFFFFFFFF


def _exp_basic(x):
    """exp(x) from additions, multiplications and divisions of Python floats only (no libm: bit-identical on every
    IEEE machine): Taylor series of exp(x / 1024), then ten squarings."""
    y = x / 1024.0
    term, s = 1.0, 1.0
    for k in range(1, 14):
        term = term * y / k
        s = s + term
    for _ in range(10):
        s = s * s
    return s


def _binomial_cuts(nbits):
    """Upper boundaries (exclusive, as integers in [0, 2^32]) of the levels 0..nbits of a Binomial(nbits, 1/2) draw
    from a uniform 32-bit integer."""
    c, acc, cuts = 1, 0, []
    for k in range(nbits + 1):
        acc += c
        cuts.append(acc << (32 - nbits))
        c = c * (nbits - k) // (k + 1)
    assert cuts[-1] == 1 << 32
    return cuts


def portable_tables(dropout=0.3, kmax=250):
    """(cell-level cuts, gene-level cuts, thresholds [16, 31, kmax] as Python ints): count = number of thresholds <= u."""
    lib = [_exp_basic(0.4 * (k - 7.5) / (15 ** 0.5 / 2.0)) for k in range(16)]
    gmean = [_exp_basic(-3.2 + 1.6 * (k - 15.0) / (30 ** 0.5 / 2.0)) for k in range(31)]
    T = []
    for l in lib:
        row = []
        for m in gmean:
            mu = l * m
            q = mu / (2.0 + mu)
            p = (2.0 / (2.0 + mu)) * (2.0 / (2.0 + mu))          # P(0) of NB(r = 2, mean mu)
            cdf, th = 0.0, []
            for k in range(kmax):
                cdf = cdf + p
                v = int((dropout + (1.0 - dropout) * cdf) * 4294967296.0)
                th.append(min(v, 1 << 32))
                p = p * (k + 2.0) / (k + 1.0) * q
            row.append(th)
        T.append(row)
    return _binomial_cuts(15), _binomial_cuts(30), T


def _mul32(h, c):
    """(h * c) mod 2^32 on int64 tensors without leaving the positive int64 range (h < 2^32, c < 2^32)."""
    lo = h * (c & 0xFFFF)
    hi = (h * (c >> 16)) & 0xFFFF
    return (lo + (hi << 16)) & _M32


def _mix32(h):
    """A 32-bit finaliser (xor-shift / multiply rounds) on int64 tensors holding values below 2^32."""
    h = h ^ (h >> 16)
    h = _mul32(h, 0x7FEB352D)
    h = h ^ (h >> 15)
    h = _mul32(h, 0x846CA68B)
    h = h ^ (h >> 16)
    return h


def generate_counts_portable(n, G, seed=20260925, device='cuda', row_offset=0, chunk=4096, ld=None, dtype=torch.float32):
    """Y [n, ld] (ld = G rounded up to 4 unless given; pad columns zero): the same matrix on every device and machine.
    Rows are global indices row_offset .. row_offset + n: shards of one matrix are generated independently."""
    ld = (G + 3) // 4 * 4 if ld is None else ld
    dev = torch.device(device)
    cuts_c, cuts_g, T = portable_tables()
    kmax = len(T[0][0])
    i64 = dict(dtype=torch.int64, device=dev)
    s = int(seed) & _M32
    gcol = torch.arange(G, **i64)
    lev_g = torch.bucketize(_mix32((gcol * 2 + 1 + s) & _M32), torch.tensor(cuts_g[:-1], **i64), right=True)
    Tt = torch.tensor(T, **i64)                                   # [16, 31, kmax]
    Y = torch.zeros(n, ld, dtype=dtype, device=dev)
    colkey = _mul32(gcol + 1, 0x9E3779B1)
    for c0 in range(0, n, chunk):
        b = min(chunk, n - c0)
        rows = torch.arange(row_offset + c0, row_offset + c0 + b, **i64)
        lev_c = torch.bucketize(_mix32((rows * 2 + s) & _M32), torch.tensor(cuts_c[:-1], **i64), right=True)
        rowkey = _mix32((_mul32(rows + 1, 0x85EBCA77) + s) & _M32)
        u = _mix32((rowkey[:, None] + colkey[None, :]) & _M32)       # [b, G] uniform 32-bit
        t0 = Tt[:, :, 0][lev_c][:, lev_g]
        y = (u >= t0).to(torch.int64)
        idx = torch.nonzero(y.flatten()).flatten()                   # the ~7 % non-zero elements walk up their CDF
        if idx.numel():
            ui = u.flatten()[idx]
            lc = lev_c[idx // G]
            lg = lev_g[idx % G]
            cnt = torch.ones_like(ui)
            alive = torch.arange(idx.numel(), **i64)
            k = 1
            while alive.numel() and k < kmax:
                up = ui[alive] >= Tt[lc[alive], lg[alive], k]
                alive = alive[up]
                cnt[alive] += 1
                k += 1
            yf = y.flatten()
            yf[idx] = cnt
            y = yf.view(b, G)
        Y[c0:c0 + b, :G] = y.to(dtype)
        del u, y
    # every cell / gene holds at least one count (dca/api.py:163-164)
    rs = Y[:, :G].sum(dim=1)
    empty = torch.nonzero(rs == 0).flatten()
    if empty.numel():
        Y[empty, _mix32((empty + row_offset + 77 + s) & _M32) % G] = 1
    if row_offset == 0:
        cs = Y[:, :G].sum(dim=0)
        emptyg = torch.nonzero(cs == 0).flatten()
        if emptyg.numel():
            Y[_mix32((emptyg + 1234567 + s) & _M32) % n, emptyg] += 1
    return Y


def counts_checksum(Y, G):
    """An exact integer fingerprint of a count matrix (any device): sum over elements of count * (1 + (31 row + 17 col) mod
    65521), mod 2^61 - 1 -- position-sensitive, computed in int64 without overflow chunk by chunk."""
    n = Y.shape[0]
    P = (1 << 61) - 1
    tot = 0
    col = torch.arange(G, dtype=torch.int64, device=Y.device) * 17
    for c0 in range(0, n, 4096):
        blk = Y[c0:c0 + 4096, :G].to(torch.int64)
        rows = torch.arange(c0, c0 + blk.shape[0], dtype=torch.int64, device=Y.device) * 31
        w = (rows[:, None] + col[None, :]) % 65521 + 1
        tot = (tot + int((blk * w).sum().item())) % P
    return tot
