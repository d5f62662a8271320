"""Products from pre-split operands (dcahip_split_planes + dcahip_gemm_p3) against fp64 numpy, held to the contract
include/dcahip.h states for the split-bf16 arithmetic: 5e-7 of sum |a b| per output element.  Replaces the same MatMul
kernels as dcahip_sgemm (dca/network.py:124-126, 369-380 and their autodiff) where an operand enters several products."""
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


def planes_of(ops, M, rows_alloc=None):
    """fp32 numpy [R, C] -> planes tensor [3, R, r8(C)] through the split kernel."""
    R, C = M.shape
    src = torch.as_tensor(M).cuda().contiguous()
    pl = ops.planes_alloc(rows_alloc or R, C, src.device)
    ops.split_planes(src, C, R, C, pl)
    return pl


def test_split_planes_reconstructs_fp32(ops):
    rng = np.random.RandomState(0)
    x = (rng.standard_normal((37, 53)) * np.exp(rng.uniform(-20, 20, (37, 53)))).astype(np.float32)
    x[3, 5] = 0.0; x[4, 6] = -0.0; x[5, 7] = 3.0e38; x[6, 8] = 1e-38
    pl = planes_of(ops, x)
    assert pl.shape == (3, 37, 56)
    p = pl.double().sum(0).cpu().numpy()                    # the pieces are bf16: their fp64 sum is exact
    assert np.all(p[:, 53:] == 0)
    err = np.abs(p[:, :53] - x.astype(np.float64))
    # (pieces below the smallest normal bf16 are flushed: 2e-38 absolute)
    assert np.all(err <= 2.0 ** -23 * np.abs(x) + 2e-38), float((err / np.maximum(np.abs(x), 1e-30)).max())
    # gathered rows
    perm = torch.as_tensor(rng.permutation(37)[:20].astype(np.int32)).cuda()
    cur = torch.tensor([3], dtype=torch.int64, device='cuda')
    pg = ops.planes_alloc(17, 53, 'cuda')
    ops.split_planes(torch.as_tensor(x).cuda(), 53, 17, 53, pg, perm=perm, cursor=cur)
    want = pl[:, perm[3:20].long()]
    assert torch.equal(pg, want)


SHAPES = [
    # ta, tb, M, N, K, gather, bias, colsum, split
    (0, 1, 128, 128, 64, False, False, False, 0),
    (0, 1, 200, 130, 520, True, True, False, 0),
    (0, 1, 257, 100, 4104, False, False, False, 0),        # split-K by the heuristic
    (0, 1, 64, 300, 96, False, True, False, 3),
    (0, 0, 128, 128, 64, False, False, False, 0),
    (0, 0, 150, 515, 264, True, True, False, 0),
    (0, 0, 40, 77, 1000, False, False, True, 0),           # colsum row = column sums of B
    (1, 0, 128, 128, 64, False, False, False, 0),
    (1, 0, 300, 131, 777, True, False, True, 0),           # k gathered (the first layer's weight gradient)
    (1, 0, 70, 260, 2051, False, False, True, 4),
    (1, 1, 130, 70, 96, True, False, False, 0),
    (0, 1, 2048, 512, 1000, True, False, False, 0),
    # large outputs, K % 16 == 0, no gather: the 256 x 256 direct-to-LDS kernel, every layout, edge tiles, split-K
    (0, 1, 600, 520, 1008, False, True, False, 0),
    (0, 0, 512, 1000, 208, False, True, False, 0),
    (0, 0, 700, 515, 4096, False, False, True, 0),
    (1, 0, 520, 777, 2048, False, False, True, 0),
    (1, 0, 512, 512, 16, False, False, False, 0),
    (1, 1, 530, 600, 96, False, False, False, 0),
    (0, 1, 2048, 512, 4800, False, False, False, 0),
    (1, 0, 1024, 3000, 48, False, False, True, 2),
    # more than 256 tiles with a partial last round: its tiles are cut into K slices and summed afterwards
    (1, 0, 512, 38400, 128, False, False, True, 0),        # 300 tiles: 44 tail tiles x 2 slices, column sums from m tile 0
    (0, 0, 768, 25000, 256, False, True, False, 0),        # 294 tiles: 38 tail tiles x 4 slices, bias, ragged N
    (0, 1, 8300, 2100, 4096, False, False, False, 0),      # 33 x 9 = 297 tiles: 41 tail tiles x 6 slices, ragged M and N
]


@pytest.mark.parametrize('ta,tb,M,N,K,gather,bias,colsum,split', SHAPES)
def test_gemm_p3_vs_numpy(ops, ta, tb, M, N, K, gather, bias, colsum, split):
    rng = np.random.RandomState(M + 3 * N + 7 * K)
    ra, ca = (K, M) if ta else (M, K)
    rb, cb = (N, K) if tb else (K, N)
    n_store = ra + 11 if gather else ra
    Ast = (rng.uniform(-1, 1, (n_store, ca)) * np.exp(rng.uniform(-3, 3, (n_store, 1)))).astype(np.float32)
    Bm = rng.uniform(-1, 1, (rb, cb)).astype(np.float32)
    cur = 5
    if gather:
        perm = rng.permutation(n_store)[:ra + cur].astype(np.int32)
        Arows = Ast[perm[cur:cur + ra]]
    else:
        perm, Arows = None, Ast
    A64, B64 = Arows.astype(np.float64), Bm.astype(np.float64)
    opA = A64.T if ta else A64
    opB = B64.T if tb else B64
    ref = opA @ opB
    absref = np.abs(opA) @ np.abs(opB)
    bvec = rng.uniform(-1, 1, N).astype(np.float32) if bias else None
    if bias:
        ref = ref + bvec.astype(np.float64)
    # an operand contiguous along k is read in units of 8 k: K is passed rounded up, the planes' zero columns (and zero
    # rows of an operand whose rows are k) make the padding contribute nothing
    Kp = r8(K) if (not ta or tb) else K
    assert not (gather and ta and Kp != K)
    pa = planes_of(ops, Ast, rows_alloc=max(n_store, Kp) if ta else None)
    pb = planes_of(ops, Bm, rows_alloc=Kp if not tb else None)
    ldc = N + (-N) % 4
    Mo = M + (1 if colsum else 0)
    C = torch.full((Mo + 1, ldc), 123.0, device='cuda')
    wsb = ops.gemm_p3_workspace_bytes(M, N, Kp, colsum, split)
    ws = torch.empty(max(wsb // 4, 1), device='cuda')
    dcur = torch.tensor([cur], dtype=torch.int64, device='cuda') if gather else None
    ops.gemm_p3(ta, tb, M, N, Kp, pa, pb, C, ldc, bias=torch.as_tensor(bvec).cuda() if bias else None,
                perm=torch.as_tensor(perm).cuda() if gather else None, cursor=dcur, colsum_row=colsum, split_k=split, ws=ws)
    torch.cuda.synchronize()
    out = C.cpu().numpy().astype(np.float64)
    err = np.abs(out[:M, :N] - ref)
    tol = 5e-7 * absref + (1.2e-7 * np.abs(ref) if bias else 0) + 1e-30
    assert np.all(err <= tol), (float((err / (absref + 1e-30)).max()))
    assert np.all(out[Mo:, :] == 123.0) and np.all(out[:Mo, N:] == 123.0)
    if colsum:
        cs = B64.sum(axis=0)
        np.testing.assert_allclose(out[M, :N], cs, rtol=0, atol=3e-6 * np.abs(B64).sum(axis=0).max())


def test_gemm_p3_rejects_bad_arguments(ops):
    a = ops.planes_alloc(16, 12, 'cuda'); b = ops.planes_alloc(16, 12, 'cuda')
    C = torch.zeros(16, 16, device='cuda')
    with pytest.raises(RuntimeError):
        ops.gemm_p3(0, 1, 16, 16, 12, a, b, C, 16)             # K % 8 != 0 with k-contiguous operands
    with pytest.raises(RuntimeError):
        ops.gemm_p3(1, 0, 16, 16, 16, a, b, C, 8)                                 # ldc < N

