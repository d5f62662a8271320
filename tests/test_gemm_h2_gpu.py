"""Products from fp16 x 2 planes (dcahip_absmax_exp + dcahip_split_planes_h2 + dcahip_gemm_h2: two fp16 pieces per operand after a
power-of-two block scale, three products per fp32 product) against fp64 numpy, held to the contract include/dcahip.h states
for that arithmetic: 1e-6 of sum |a b| per output element.  Replaces the same MatMul kernels as dcahip_sgemm
(dca/network.py:124-126, 369-380 and their autodiff) on the wide networks' large products."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from dca_amd.ops import HipOps
    return HipOps()


def r8(x):
    return (x + 7) // 8 * 8


def planes_h2(ops, M, scale_rows=None):
    """fp32 numpy [R, C] -> (planes tensor [2, R, r8(C)] fp16, exponent word) through the max and split kernels."""
    R, C = M.shape
    src = torch.as_tensor(M).cuda().contiguous()
    pl = torch.zeros(2, R, r8(C), dtype=torch.float16, device='cuda')
    e = torch.zeros(2, dtype=torch.int32, device='cuda')
    ops.absmax_exp(src, C, R, C, e)
    ops.split_planes_h2(src, C, R, C, pl, e)
    return pl, e


def test_split_planes_h2_reconstructs_fp32(ops):
    rng = np.random.RandomState(0)
    x = (rng.standard_normal((37, 53)) * np.exp(rng.uniform(-6, 0, (37, 53)))).astype(np.float32)
    x[3, 5] = 0.0; x[4, 6] = -0.0
    pl, e = planes_h2(ops, x)
    ex = int(e[0].item())
    assert 2.0 ** 13 <= np.abs(x).max() * 2.0 ** ex < 2.0 ** 14
    p = pl.double().sum(0).cpu().numpy()                    # fp16 pieces: their fp64 sum is exact
    assert np.all(p[:, 53:] == 0)
    xs = np.ldexp(x.astype(np.float64), ex)
    err = np.abs(p[:, :53] - xs)
    assert np.all(err <= np.maximum(2.0 ** -22 * np.abs(xs), 2.0 ** -25)), float((err / np.maximum(np.abs(xs), 1e-30)).max())
    # gathered rows, a given exponent
    perm = torch.as_tensor(rng.permutation(37)[:20].astype(np.int32)).cuda()
    cur = torch.tensor([3], dtype=torch.int64, device='cuda')
    pg = torch.zeros(2, 17, 56, dtype=torch.float16, device='cuda')
    ops.split_planes_h2(torch.as_tensor(x).cuda(), 53, 17, 53, pg, e, perm=perm, cursor=cur)
    assert torch.equal(pg, pl[:, perm[3:20].long()])
    # an all-zero matrix: exponent 0, zero planes
    z = np.zeros((9, 16), np.float32)
    pz, ez = planes_h2(ops, z)
    assert int(ez[0].item()) == 0 and not pz.any()


# a priori: 2^-22 per operand (two pieces) + the dropped a2 b2 (2^-22) = 3 x 2^-22 = 7.2e-7 of sum |a b|, plus the fp32
# accumulation over K; the worst element of 3 M measured 5.7e-7
TOL = 1e-6

SHAPES = [
    # ta, tb, M, N, K, bias, colsum, split
    (0, 0, 512, 512, 64, False, False, 0),
    (0, 0, 600, 515, 272, True, False, 0),                  # ragged tiles
    (0, 0, 2048, 1500, 512, True, False, 0),                # the heads' forward at test size
    (0, 0, 640, 512, 4112, True, False, 0),                 # split-K by the heuristic (the first layer's forward)
    (1, 0, 512, 1500, 2048, False, True, 0),                # weight gradient + column sums (bias gradient)
    (1, 0, 1500, 512, 640, False, True, 0),                 # the first layer's weight gradient
    (0, 1, 640, 512, 1504, False, False, 0),                # input gradient
    (0, 1, 600, 520, 528, False, False, 2),
    (1, 1, 512, 600, 128, False, False, 0),
    (0, 0, 1024, 5000, 512, True, False, 0),                # a partial last round of tiles: the tail path
]


@pytest.mark.parametrize('ta,tb,M,N,K,bias,colsum,split', SHAPES)
def test_gemm_h2_vs_numpy(ops, ta, tb, M, N, K, bias, colsum, split):
    assert ops.gemm_h2_supported(M, N, K)
    rng = np.random.RandomState(M + N + K)
    # operands with the spread of a training step: activations after ReLU / gradients over three decades
    a = (np.maximum(rng.standard_normal((M, K)), 0) * np.exp(rng.normal(0, 1.0, (M, K)))).astype(np.float32)
    b = (rng.standard_normal((K, N)) * 0.05 * np.exp(rng.normal(0, 1.0, (K, N)))).astype(np.float32)
    A_st = np.ascontiguousarray(a.T) if ta else a
    B_st = np.ascontiguousarray(b.T) if tb else b
    pa, ea = planes_h2(ops, A_st)
    pb, eb = planes_h2(ops, B_st)
    bv = rng.standard_normal(N).astype(np.float32) if bias else None
    alpha = 0.37
    Mo = M + (1 if colsum else 0)
    C = torch.full((Mo, N), 7.0, device='cuda')
    nb = ops.gemm_h2_workspace_bytes(M, N, K, colsum, split)
    ws = torch.full((max(nb // 4, 4),), float('nan'), device='cuda')
    ops.gemm_h2(ta, tb, M, N, K, pa, pb, C, N, exp_a=ea, exp_b=eb, alpha=alpha, bias=None if bv is None else torch.as_tensor(bv).cuda(),
                colsum_row=colsum, split_k=split, ws=ws)
    torch.cuda.synchronize()
    got = C.cpu().numpy().astype(np.float64)
    ref = alpha * (a.astype(np.float64) @ b.astype(np.float64)) + (bv.astype(np.float64) if bias else 0.0)
    mag = alpha * (np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)) + (np.abs(bv) if bias else 0.0)
    err = np.abs(got[:M] - ref)
    assert (err <= TOL * mag + 1e-30).all(), float((err / np.maximum(mag, 1e-300)).max())
    if colsum:
        cs = alpha * b.astype(np.float64).sum(0)
        np.testing.assert_allclose(got[M], cs, rtol=0, atol=2e-6 * alpha * np.abs(b).sum(0).max())
    # deterministic
    C2 = torch.zeros_like(C)
    ops.gemm_h2(ta, tb, M, N, K, pa, pb, C2, N, exp_a=ea, exp_b=eb, alpha=alpha, bias=None if bv is None else torch.as_tensor(bv).cuda(),
                colsum_row=colsum, split_k=split, ws=ws)
    assert torch.equal(C, C2)


def test_gemm_h2_static_exponent_and_unsupported_shapes(ops):
    """A producer's static exponent (the gradient planes of dcahip_zinb_nll_planes_h2: g 2^d_exp) through exp_*_add; shapes the
    256 x 256 kernel does not take are refused (the engine keeps the three-piece planes there)."""
    rng = np.random.RandomState(3)
    M, N, K = 512, 512, 256
    a = rng.standard_normal((M, K)).astype(np.float32)
    g = (rng.standard_normal((K, N)) * np.exp(rng.normal(0, 1.5, (K, N)))).astype(np.float32)
    pa, ea = planes_h2(ops, a)
    gs = torch.as_tensor(g * 4.0).cuda().contiguous()                    # the producer writes g 2^2 ...
    pg = torch.zeros(2, K, N, dtype=torch.float16, device='cuda')
    ops.split_planes_h2(gs, N, K, N, pg, None)                          # ... unscaled by the split (exp NULL: 0)
    C = torch.zeros(M, N, device='cuda')
    ws = torch.zeros(max(ops.gemm_h2_workspace_bytes(M, N, K, False, 0) // 4, 4), device='cuda')
    ops.gemm_h2(0, 0, M, N, K, pa, pg, C, N, exp_a=ea, exp_b_add=2, alpha=1.0 / 1024, ws=ws)
    ref = (a.astype(np.float64) @ g.astype(np.float64)) / 1024
    mag = (np.abs(a).astype(np.float64) @ np.abs(g).astype(np.float64)) / 1024
    assert (np.abs(C.cpu().numpy() - ref) <= TOL * mag).all()
    for shp in ((200, 512, 256), (512, 100, 256), (512, 512, 250)):
        assert not ops.gemm_h2_supported(*shp)
        with pytest.raises(RuntimeError):
            ops.gemm_h2(0, 0, *shp, pa, pg, C, N, exp_a=ea)
