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
import math

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
