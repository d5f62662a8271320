"""Per-kernel parity: every HIP entry point against the numpy oracle on seeded inputs.

All calls go through the C ABI (dca_amd.hip / dca_amd.ops).  Tolerances are stated against
the fp64 oracle; the kernels compute in fp32.
"""
import numpy as np
import pytest
import torch

from conftest import synth_counts
from oracle import zinb_np as Z

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from dca_amd.ops import HipOps
    return HipOps()


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


def pad_cols(a, ld):
    out = np.zeros((a.shape[0], ld), a.dtype)
    out[:, :a.shape[1]] = a
    return out


# ------------------------------------------------------------------------------- K-ZINB
def _heads(B, G, seed, edge):
    rng = np.random.RandomState(seed)
    am = rng.normal(0, 1.5, (B, G)); ad = rng.normal(0, 2, (B, G)); ap = rng.normal(0, 2, (B, G))
    y = synth_counts(B, G, seed)
    sf = rng.lognormal(0, 0.3, B)
    if edge:
        am[0, :4] = [-14., 15., 0., 30.]
        ad[1, :4] = [-12., 9500., 20., -3.]
        ap[2, :4] = [-30., 30., 0., 12.]
        y[3, :4] = [0, 1, 200, 5000]
        am[3, :4] = [1., 1., 5., 8.]
        y[4, :3] = [2.52, 0.5, 17.0]        # non-integer "counts" (check_counts=False inputs)
        am[5, :2] = [95., -120.]            # exp overflow / underflow
        ad[6, :3] = [9.3, 9.21, 100.]       # theta near / at the 1e4 clip
    return [a.astype(np.float32).astype(np.float64) for a in (am, ad, ap, y, sf)]


@pytest.mark.parametrize('flags', [1, 0, 3, 2])
@pytest.mark.parametrize('B,G,edge', [(8, 40, True), (33, 1000, False), (5, 6, False), (16, 203, True)])
def test_zinb_nll_vs_oracle(ops, flags, B, G, edge):
    has_pi, cdisp = bool(flags & 1), bool(flags & 2)
    am, ad, ap, y, sf = _heads(B, G, 11 + B, edge)
    rng = np.random.RandomState(5)
    tw = rng.normal(0, 1.5, G).astype(np.float32).astype(np.float64)
    n_store = B + 7
    perm = rng.permutation(n_store)[:B + 3].astype(np.int32)
    cur = 3
    rows = perm[cur:cur + B]
    Gp = (G + 3) // 4 * 4
    Yst = np.zeros((n_store, Gp)); Yst[rows, :G] = y
    sfst = np.ones(n_store); sfst[rows] = sf
    ridge = 0.05 if has_pi else 0.0
    inv_n = 1.0 / (B * G)
    if has_pi:
        ls, lm, dm, dd, dp = Z.zinb_loss_and_grads(am, None if cdisp else ad, ap, y, sf, ridge,
                                                   theta_w=tw if cdisp else None)
    else:
        ls, lm, dm, dd = Z.nb_loss_and_grads(am, None if cdisp else ad, y, sf,
                                             theta_w=tw if cdisp else None)
        dp = None
    lda = 3 * Gp
    A = np.zeros((B, lda)); A[:, :G] = am; A[:, Gp:Gp + G] = ad; A[:, 2 * Gp:2 * Gp + G] = ap
    dA = dev(A); dD = torch.full((B, lda), 7.0, device='cuda')
    dY, dsf, dperm = dev(Yst), dev(sfst), torch.as_tensor(perm).cuda()
    dcur = torch.tensor([cur], dtype=torch.int64, device='cuda')
    dtw = dev(tw)
    part = torch.zeros(ops.max_partials, dtype=torch.float64, device='cuda')
    a_mean, a_disp, a_pi = dA[:, 0:], dA[:, Gp:], dA[:, 2 * Gp:]
    d_mean, d_disp, d_pi = dD[:, 0:], dD[:, Gp:], dD[:, 2 * Gp:]
    n = ops.zinb_nll(a_mean, None if cdisp else a_disp, a_pi if has_pi else None, lda,
                     dtw if cdisp else None, dY, Gp, dsf, dperm, dcur, B, G, ridge, inv_n, flags,
                     d_mean, d_disp, d_pi if has_pi else None, lda, part)
    loss = torch.zeros(1, device='cuda')
    ops.loss_finalize(part, n, inv_n, loss)
    torch.cuda.synchronize()
    got = loss.item()
    # fp32 kernel vs fp64 oracle: 3e-6 relative on realistic data; the edge set holds y = 5000
    # / 200 next to theta ~ 1e4, where one ulp of an fp32 lgamma/log term is already ~4e-3
    # absolute (the reference's own fp32 TensorFlow evaluation is no better there): 3e-5.
    assert abs(got - lm) <= (3e-5 if edge else 3e-6) * abs(lm), (got, lm)
    D = dD.cpu().numpy().astype(np.float64)

    def close(g, ref, name):
        scale = np.abs(ref).max()
        err = np.abs(g - ref)
        bad = err > (2e-4 * np.abs(ref) + 2e-6 * scale)
        assert not bad.any(), (name, int(bad.sum()), np.argwhere(bad)[:5], g[bad][:5], ref[bad][:5])
    close(D[:, :G], dm, 'd_mean')
    if cdisp:
        # per-element d nll/d theta * inv_n; the chain + column sum is dcahip_colsum_chain
        out = torch.zeros(G, device='cuda')
        ops.colsum_chain(d_disp, lda, B, G, dtw, out)
        torch.cuda.synchronize()
        close(out.cpu().numpy().astype(np.float64), dd, 'd_theta_w')
    else:
        close(D[:, Gp:Gp + G], dd, 'd_disp')
    if has_pi:
        close(D[:, 2 * Gp:2 * Gp + G], dp, 'd_pi')
    # padded quad columns are written as zero, never garbage
    if Gp > G:
        assert (D[:, G:Gp] == 0).all()
    # loss-only mode (validation) gives the same loss and touches no gradient buffer
    part.zero_()
    n2 = ops.zinb_nll(a_mean, None if cdisp else a_disp, a_pi if has_pi else None, lda,
                      dtw if cdisp else None, dY, Gp, dsf, dperm, dcur, B, G, ridge, inv_n, flags,
                      None, None, None, 0, part)
    ops.loss_finalize(part, n2, inv_n, loss)
    torch.cuda.synchronize()
    # (the loss-only instantiation drops the gradient arithmetic, so its sums associate differently: same tolerance
    # class as against the oracle, not bit-equality)
    assert abs(loss.item() - got) <= (3e-5 if edge else 3e-6) * abs(got)


@pytest.mark.parametrize('flags', [1, 3, 0])
@pytest.mark.parametrize('B,G,dense', [(2500, 1000, False), (2500, 1000, True), (4099, 520, False)])
def test_zinb_nll_row_pairs_and_planes(ops, flags, B, G, dense):
    """The training kernel takes two batch rows per iteration and keeps both in registers until the non-zero elements are
    evaluated (zinb_nll_rows_kernel): batches deep enough that a workgroup walks several row pairs, the second row of the
    last pair missing for some workgroups and present for others; `dense` makes EVERY element non-zero (the queue at its
    capacity of 2 x 256 entries per wave).  The plane-output entry point must hold the same numbers as three bf16 pieces."""
    has_pi, cdisp = bool(flags & 1), bool(flags & 2)
    am, ad, ap, y, sf = _heads(B, G, 3 + B, False)
    if dense:
        y = y + 1.0
    rng = np.random.RandomState(9)
    tw = rng.normal(0, 1.5, G).astype(np.float32).astype(np.float64)
    ridge = 0.01 if has_pi else 0.0
    inv_n = 1.0 / (B * G)
    if has_pi:
        ls, lm, dm, dd, dp = Z.zinb_loss_and_grads(am, None if cdisp else ad, ap, y, sf, ridge, theta_w=tw if cdisp else None)
    else:
        ls, lm, dm, dd = Z.nb_loss_and_grads(am, None if cdisp else ad, y, sf, theta_w=tw if cdisp else None)
        dp = None
    Gp = (G + 7) // 8 * 8
    lda = 3 * Gp
    A = np.zeros((B, lda)); A[:, :G] = am; A[:, Gp:Gp + G] = ad; A[:, 2 * Gp:2 * Gp + G] = ap
    dA = dev(A); dD = torch.full((B, lda), 7.0, device='cuda')
    dY, dsf, dtw = dev(pad_cols(y, Gp)), dev(sf), dev(tw)
    part = torch.zeros(ops.max_partials, dtype=torch.float64, device='cuda')
    a_mean, a_disp, a_pi = dA[:, 0:], dA[:, Gp:], dA[:, 2 * Gp:]
    d_mean, d_disp, d_pi = dD[:, 0:], dD[:, Gp:], dD[:, 2 * Gp:]
    n = ops.zinb_nll(a_mean, None if cdisp else a_disp, a_pi if has_pi else None, lda, dtw if cdisp else None, dY, Gp, dsf,
                     None, None, B, G, ridge, inv_n, flags, d_mean, d_disp, d_pi if has_pi else None, lda, part)
    loss = torch.zeros(1, device='cuda')
    ops.loss_finalize(part, n, inv_n, loss)
    torch.cuda.synchronize()
    got = loss.item()
    assert abs(got - lm) <= 3e-6 * abs(lm), (got, lm)
    D = dD.cpu().numpy().astype(np.float64)

    def close(g, ref, name):
        err = np.abs(g - ref)
        bad = err > (2e-4 * np.abs(ref) + 2e-6 * np.abs(ref).max())
        assert not bad.any(), (name, int(bad.sum()), np.argwhere(bad)[:5], g[bad][:5], ref[bad][:5])
    close(D[:, :G], dm, 'd_mean')
    if not cdisp:
        close(D[:, Gp:Gp + G], dd, 'd_disp')
    if has_pi:
        close(D[:, 2 * Gp:2 * Gp + G], dp, 'd_pi')
    # the same launch with plane output: piece 0 + piece 1 + piece 2 = the fp32 value to 2^-22 of it
    P = ops.planes_alloc(B, lda, 'cuda')
    P.fill_(3.0)
    Dth = torch.full((B, Gp), 7.0, device='cuda')
    part.zero_()
    n2 = ops.zinb_nll_planes(a_mean, None if cdisp else a_disp, a_pi if has_pi else None, lda, dtw if cdisp else None, dY, Gp, dsf,
                             None, None, B, G, ridge, inv_n, flags, P, 0, 0 if cdisp else Gp, 2 * Gp if has_pi else 0,
                             Dth if cdisp else None, Gp, part)
    ops.loss_finalize(part, n2, inv_n, loss)
    torch.cuda.synchronize()
    assert loss.item() == got
    S = P.double().sum(0).cpu().numpy()
    heads = [(0, 'mean')] + ([] if cdisp else [(Gp, 'disp')]) + ([(2 * Gp, 'pi')] if has_pi else [])
    for c0, name in heads:
        ref = D[:, c0:c0 + G]
        assert np.abs(S[:, c0:c0 + G] - ref).max() <= 2.0 ** -22 * np.abs(ref).max() + 1e-38, name
    if cdisp:
        assert torch.equal(Dth[:, :G], dD[:, Gp:Gp + G])


@pytest.mark.parametrize('flag,B,G', [(4, 8, 40), (4, 33, 1000), (8, 16, 203), (8, 5, 6)])
def test_poisson_and_mse_vs_oracle(ops, flag, B, G):
    """DCAHIP_NLL_POISSON / DCAHIP_NLL_MSE (ae_types 'poisson', 'normal'): loss and d loss / d a_mean."""
    am, _, _, y, sf = _heads(B, G, 7 + B, edge=(G == 40))
    if flag == 4:
        ls, lm, dm = Z.poisson_loss_and_grads(am, y, sf)
    else:
        ls, lm, dm = Z.mse_loss_and_grads(am, y, sf)
    Gp = (G + 3) // 4 * 4
    dA = dev(pad_cols(am, Gp)); dD = torch.full((B, Gp), 7.0, device='cuda')
    part = torch.zeros(ops.max_partials, dtype=torch.float64, device='cuda')
    inv_n = 1.0 / (B * G)
    n = ops.zinb_nll(dA, None, None, Gp, None, dev(pad_cols(y, Gp)), Gp, dev(sf), None, None, B, G, 0.0,
                     inv_n, flag, dD, None, None, Gp, part)
    loss = torch.zeros(1, device='cuda')
    ops.loss_finalize(part, n, inv_n, loss)
    torch.cuda.synchronize()
    assert abs(loss.item() - lm) <= 3e-6 * abs(lm), (loss.item(), lm)
    D = dD.cpu().numpy().astype(np.float64)[:, :G]
    scale = np.abs(dm).max()
    assert (np.abs(D - dm) <= 2e-4 * np.abs(dm) + 2e-6 * scale).all()
    # inference heads: linear mean for 'normal'
    out = torch.zeros(B, Gp, device='cuda')
    ops.heads_infer(dA, None, None, Gp, dev(sf), B, G, out, None, None, Gp, flag & 8)
    torch.cuda.synchronize()
    ref = (am if flag == 8 else Z.mean_act(am)) * sf[:, None]
    np.testing.assert_allclose(out.cpu().numpy()[:, :G], ref, rtol=2e-6, atol=1e-30)


def test_zinb_nll_unaligned_scalar_path(ops):
    """ld not a multiple of 4 -> the kernel must take its scalar path and still be right."""
    B, G = 9, 37
    am, ad, ap, y, sf = _heads(B, G, 3, False)
    _, lm, dm, dd, dp = Z.zinb_loss_and_grads(am, ad, ap, y, sf, 0.0)
    lda = 3 * G + 1
    A = np.zeros((B, lda)); A[:, :G] = am; A[:, G:2 * G] = ad; A[:, 2 * G:3 * G] = ap
    dA = dev(A); dD = torch.zeros((B, lda), device='cuda')
    part = torch.zeros(ops.max_partials, dtype=torch.float64, device='cuda')
    n = ops.zinb_nll(dA[:, 0:], dA[:, G:], dA[:, 2 * G:], lda, None, dev(y), G, dev(sf), None, None,
                     B, G, 0.0, 1.0 / (B * G), 1, dD[:, 0:], dD[:, G:], dD[:, 2 * G:], lda, part)
    loss = torch.zeros(1, device='cuda')
    ops.loss_finalize(part, n, 1.0 / (B * G), loss)
    torch.cuda.synchronize()
    assert abs(loss.item() - lm) <= 3e-6 * abs(lm)
    D = dD.cpu().numpy()
    np.testing.assert_allclose(D[:, :G], dm, rtol=2e-4, atol=2e-6 * np.abs(dm).max())
    np.testing.assert_allclose(D[:, 2 * G:3 * G], dp, rtol=2e-4, atol=2e-6 * np.abs(dp).max())


def test_loss_nan_maps_to_inf_and_step_end(ops):
    part = torch.tensor([1.0, float('nan')], dtype=torch.float64, device='cuda')
    loss = torch.zeros(1, device='cuda')
    ops.loss_finalize(part, 2, 1.0, loss)
    torch.cuda.synchronize()
    assert np.isinf(loss.item())
    loss.fill_(2.5)
    hist = torch.zeros(8, device='cuda'); acc = torch.zeros(1, dtype=torch.float64, device='cuda')
    cur = torch.tensor([64], dtype=torch.int64, device='cuda')
    ops.step_end(loss, 32.0, hist, 32, acc, cur, 32)
    ops.step_end(loss, 7.0, hist, 32, acc, cur, 7)
    torch.cuda.synchronize()
    assert cur.item() == 103 and acc.item() == 2.5 * 39
    assert hist.cpu().tolist() == [0, 0, 2.5, 2.5, 0, 0, 0, 0]


def test_heads_infer(ops):
    B, G = 7, 50
    am, ad, ap, y, sf = _heads(B, G, 2, True)
    Gp = 52
    A = np.zeros((B, 3 * Gp)); A[:, :G] = am; A[:, Gp:Gp + G] = ad; A[:, 2 * Gp:2 * Gp + G] = ap
    dA = dev(A)
    mu, th, pi = Z.heads_forward(am, ad, ap, sf)
    ops.heads_infer(dA[:, 0:], dA[:, Gp:], dA[:, 2 * Gp:], 3 * Gp, dev(sf), B, G,
                    dA[:, 0:], dA[:, Gp:], dA[:, 2 * Gp:], 3 * Gp)     # in place
    torch.cuda.synchronize()
    O = dA.cpu().numpy()
    np.testing.assert_allclose(O[:, :G], mu, rtol=3e-6)
    np.testing.assert_allclose(O[:, Gp:Gp + G], th, rtol=3e-6)
    np.testing.assert_allclose(O[:, 2 * Gp:2 * Gp + G], pi, rtol=3e-6, atol=1e-30)


# ------------------------------------------------------------------------------- K-GEMM
GEMM_SHAPES = [
    # ta, tb, M, N, K, gather, bias, colsum, split
    (0, 0, 32, 64, 1000, True, True, False, 0),      # enc0 fwd, B=32 (auto split-K)
    (0, 0, 200, 64, 517, True, True, False, 3),      # ragged K, forced split
    (0, 0, 32, 3000, 64, False, True, False, 0),     # heads fwd, small M (cfg 64x128)
    (0, 0, 300, 1500, 64, False, True, False, 0),    # heads fwd, large M (cfg 128x128)
    (0, 0, 45, 6, 3, False, True, False, 0),         # tiny odd (biochemists-like)
    (1, 0, 64, 3000, 32, False, False, True, 0),     # dW heads + bias grad row
    (1, 0, 64, 700, 300, False, False, True, 2),     # dW heads, split-K with colsum row
    (1, 0, 1000, 64, 32, True, False, True, 0),      # dW0 with gathered K rows
    (1, 0, 333, 33, 130, True, False, True, 0),      # odd sizes (scalar tails)
    (1, 0, 150, 200, 260, False, False, True, 0),    # cfg 128x128, TN
    (0, 1, 32, 64, 3000, False, False, False, 0),    # dH (NT), big K
    (0, 1, 130, 32, 64, False, False, False, 0),     # dH mid layer
    (0, 1, 257, 190, 99, False, False, False, 0),    # cfg 128x128, NT, odd
    (0, 0, 17, 5, 9, False, False, False, 0),        # unaligned ld -> scalar loads
    # skinny outputs over many rows (the first layer at throughput batches: the A-direct kernel), every layout
    (0, 0, 2100, 64, 300, True, True, False, 0),     # enc0 fwd, gathered rows, ragged M and K
    (0, 1, 2049, 33, 517, False, False, False, 3),   # NT, odd N, forced split
    (1, 0, 2500, 64, 260, True, False, True, 0),     # dW0: gathered K rows + bias-gradient row
    (1, 0, 2051, 64, 99, True, False, True, 2),      # dW0, ragged, forced split
    # weight gradients of wide networks: A = the transposed copy (k-contiguous), column sums of B in the NN form
    (0, 0, 128, 700, 300, False, False, True, 0),
    (0, 0, 64, 3000, 260, False, False, True, 2),
    (0, 0, 333, 130, 257, False, False, True, 0),
    # mid-size outputs (hidden stack of the wide networks): 64 x 64 tiles, K slices of >= 128
    (0, 0, 2048, 256, 512, False, True, False, 0),
    (0, 0, 2048, 512, 256, False, True, False, 0),   # exactly 2048 x 512: the last size on this plan
    (0, 0, 2049, 512, 256, False, True, False, 0),   # one row more: 128 x 128 tiles
    (1, 0, 512, 256, 2048, False, False, True, 0),
    (1, 0, 130, 250, 2047, False, False, True, 0),   # ragged, deep split
    (0, 1, 2048, 128, 256, False, False, False, 0),
    (0, 1, 1999, 257, 130, False, False, False, 0),
]


@pytest.mark.parametrize('ta,tb,M,N,K,gather,bias,colsum,split', GEMM_SHAPES)
def test_sgemm_vs_numpy(ops, ta, tb, M, N, K, gather, bias, colsum, split):
    rng = np.random.RandomState(M + N + K)
    ra, ca = (K, M) if ta else (M, K)
    rb, cb = (N, K) if tb else (K, N)
    unaligned = (N == 5)
    lda = ca + (1 if unaligned else (-ca) % 4)
    ldb = cb + (1 if unaligned else (-cb) % 4)
    n_store = ra + 9 if gather else ra
    Ast = rng.uniform(-1, 1, (n_store, lda)).astype(np.float32)
    Bm = rng.uniform(-1, 1, (rb, ldb)).astype(np.float32)
    cur = 4
    if gather:
        perm = rng.permutation(n_store)[:ra + cur].astype(np.int32)
        Arows = Ast[perm[cur:cur + ra]]
    else:
        perm = None
        Arows = Ast
    A = Arows[:, :ca].astype(np.float64)
    Bv = Bm[:, :cb].astype(np.float64)
    opA = A.T if ta else A
    opB = Bv.T if tb else Bv
    ref = opA @ opB
    absref = np.abs(opA) @ np.abs(opB)
    bvec = rng.uniform(-1, 1, N).astype(np.float32) if bias else None
    if bias:
        ref = ref + bvec.astype(np.float64)
    ldc = N + (-N) % 4
    Mo = M + (1 if colsum else 0)
    C = torch.full((Mo + 1, ldc), 123.0, device='cuda')
    wsb = ops.sgemm_workspace_bytes(ta, tb, M, N, K, colsum, split)
    ws = torch.empty(max(wsb // 4, 1), device='cuda')
    dcur = torch.tensor([cur], dtype=torch.int64, device='cuda') if gather else None
    ops.sgemm(ta, tb, M, N, K, dev(Ast), lda, dev(Bm), ldb, C, ldc, bias=dev(bvec) if bias else None,
              perm=torch.as_tensor(perm).cuda() if gather else None, cursor=dcur,
              colsum_row=colsum, split_k=split, ws=ws)
    torch.cuda.synchronize()
    out = C.cpu().numpy().astype(np.float64)
    err = np.abs(out[:M, :N] - ref)
    # the shapes dcahip_sgemm runs as split-bf16 products (NT, and NN with N >= 128) are held to the contract
    # include/dcahip.h states for them: 5e-7 of sum|ab| (+ the rounding of the bias add); the exact-fp32 MFMA kernel
    # (an fmaf chain over K, TN and narrow NN) keeps its chain bound
    x3 = (not ta) and (tb or N >= 128) and split >= 0
    mag = absref + (np.abs(bvec.astype(np.float64)) if bias else 0.0)
    tol = (5e-7 * mag + 1e-30) if x3 else (2e-6 * mag + 1e-6)
    assert (err <= tol).all(), (err.max(), float((err / mag).max()), np.argwhere(err > tol)[:5])
    if colsum:
        cs = Bv.sum(axis=0)
        np.testing.assert_allclose(out[M, :N], cs, rtol=0, atol=2e-6 * np.abs(Bv).sum(axis=0).max() + 1e-6)
    # nothing outside the [Mo, N] window is touched
    assert (out[Mo:, :] == 123.0).all() and (out[:, N:] == 123.0).all()


@pytest.mark.parametrize('R,C,gather', [(1, 1, False), (37, 203, True), (300, 64, False), (2048, 1000, True), (130, 203, True), (67, 64, False)])
def test_transpose_rows(ops, R, C, gather):
    rng = np.random.RandomState(R + C)
    n_store = R + 7 if gather else R
    lds = C + (-C) % 4 + 4
    src = rng.uniform(-1, 1, (n_store, lds)).astype(np.float32)
    ldd = R + (-R) % 4
    dst = torch.full((C + 1, ldd), 7.0, device='cuda')
    cur = 3
    perm = rng.permutation(n_store)[:R + cur].astype(np.int32) if gather else None
    ops.transpose(dev(src), lds, R, C, dst, ldd, perm=torch.as_tensor(perm).cuda() if gather else None,
                  cursor=torch.tensor([cur], dtype=torch.int64, device='cuda') if gather else None)
    torch.cuda.synchronize()
    out = dst.cpu().numpy()
    rows = src[perm[cur:cur + R]] if gather else src
    np.testing.assert_array_equal(out[:C, :R], rows[:, :C].T)
    assert (out[C:] == 7.0).all() and (out[:, R:] == 7.0).all()


# ------------------------------------------------------------------------------- small-batch layer kernels
def _np_layer_fwd(Hp, W, b, beta, mm, mv, batchnorm, mom=0.99, eps=1e-3):
    Z = Hp @ W + b if W is not None else Hp
    if not batchnorm:
        return Z, None, np.maximum(Z, 0), None, mm, mv
    mean, var = Z.mean(0), Z.var(0)
    inv = 1.0 / np.sqrt(var + eps)
    xh = (Z - mean) * inv
    return Z, xh, np.maximum(xh + beta, 0), inv, mm - (mm - mean) * (1 - mom), mv - (mv - var) * (1 - mom)


@pytest.mark.parametrize('batchnorm', [True, False])
@pytest.mark.parametrize('B,hs', [(32, (64, 32, 64)), (17, (50, 7, 64)), (64, (64, 64)), (5, (3, 2)), (40, (20, 33, 9, 64)), (1, (4, 4))])
def test_small_batch_layer_kernels_vs_numpy(ops, B, hs, batchnorm):
    """dcahip_hidden_small_chain (the stack behind the first product in one launch), dcahip_bn_relu_train_small +
    dcahip_dense_bn_small (the same, one launch per layer) and the two backward kernels against the numpy formulas of
    Dense -> BatchNormalization(center, no scale, eps 1e-3, momentum 0.99) -> relu (dca/network.py:124-135)."""
    rng = np.random.RandomState(B + sum(hs))
    L = len(hs)
    ld = [h + (-h) % 4 for h in hs]
    Z0 = rng.normal(size=(B, hs[0]))
    W = [None] + [rng.normal(size=(hs[i - 1], hs[i])) * 0.3 for i in range(1, L)]
    b = [np.zeros(hs[0])] + [rng.normal(size=hs[i]) * 0.1 for i in range(1, L)]
    beta = [rng.normal(size=h) * 0.1 for h in hs]
    mm0 = [rng.normal(size=h) * 0.1 for h in hs]
    mv0 = [rng.uniform(0.5, 1.5, size=h) for h in hs]
    # numpy reference, layer after layer
    ref = []
    cur = Z0
    for i in range(L):
        Zi, xh, Hi, inv, mm1, mv1 = _np_layer_fwd(cur, W[i], b[i], beta[i], mm0[i], mv0[i], batchnorm)
        ref.append(dict(Z=Zi, xh=xh, H=Hi, inv=inv, mm=mm1, mv=mv1))
        cur = Hi
    f32 = dict(dtype=torch.float32, device='cuda')

    def buffers():
        d = dict(Z=[torch.zeros(B, ld[i], **f32) for i in range(L)], XH=[torch.zeros(B, ld[i], **f32) for i in range(L)],
                 H=[torch.zeros(B, ld[i], **f32) for i in range(L)], inv=[torch.zeros(hs[i], **f32) for i in range(L)],
                 mm=[dev(mm0[i].astype(np.float32)) for i in range(L)], mv=[dev(mv0[i].astype(np.float32)) for i in range(L)])
        d['Z'][0][:, :hs[0]] = dev(Z0.astype(np.float32))
        return d
    Wd = [None] + [dev(np.ascontiguousarray(W[i]).astype(np.float32)) for i in range(1, L)]
    bd = [None] + [dev(b[i].astype(np.float32)) for i in range(1, L)]
    betad = [dev(beta[i].astype(np.float32)) for i in range(L)]

    def check(d, what):
        for i in range(L):
            tol = dict(rtol=2e-5, atol=2e-5)
            np.testing.assert_allclose(d['H'][i][:, :hs[i]].cpu().numpy(), ref[i]['H'], err_msg='%s H%d' % (what, i), **tol)
            if i > 0:
                np.testing.assert_allclose(d['Z'][i][:, :hs[i]].cpu().numpy(), ref[i]['Z'], err_msg='%s Z%d' % (what, i), **tol)
            if batchnorm:
                if B > 1:
                    np.testing.assert_allclose(d['XH'][i][:, :hs[i]].cpu().numpy(), ref[i]['xh'], err_msg='%s xhat%d' % (what, i), rtol=1e-4, atol=1e-4)
                    np.testing.assert_allclose(d['inv'][i].cpu().numpy(), ref[i]['inv'], rtol=1e-4, err_msg='%s inv%d' % (what, i))
                np.testing.assert_allclose(d['mm'][i].cpu().numpy(), ref[i]['mm'], rtol=1e-5, atol=1e-6, err_msg='%s mm%d' % (what, i))
                np.testing.assert_allclose(d['mv'][i].cpu().numpy(), ref[i]['mv'], rtol=1e-5, atol=1e-6, err_msg='%s mv%d' % (what, i))
            assert (d['H'][i][:, hs[i]:] == 0).all()            # padding columns untouched

    # (a) the whole stack in one launch
    d = buffers()
    entries = []
    for i in range(L):
        e = dict(H=hs[i], beta=betad[i], moving_mean=d['mm'][i], moving_var=d['mv'][i], Z=d['Z'][i], ldz=ld[i], xhat=d['XH'][i],
                 ldx=ld[i], Hout=d['H'][i], ldh=ld[i], inv_std=d['inv'][i])
        if i > 0:
            e.update(W=Wd[i], ldw=hs[i], bias=bd[i], K=hs[i - 1])
        entries.append(e)
    ops.hidden_small_chain(entries, None, 0, B, batchnorm, 0.99, 1e-3, 1)
    torch.cuda.synchronize()
    check(d, 'chain')
    # (b) one launch per layer
    d2 = buffers()
    if batchnorm:
        ops.bn_relu_train_small(d2['Z'][0], ld[0], B, hs[0], betad[0], d2['mm'][0], d2['mv'][0], 0.99, 1e-3, 1, d2['H'][0], ld[0],
                                d2['XH'][0], ld[0], d2['inv'][0])
    else:
        ops.relu_fwd(d2['Z'][0], ld[0], B, hs[0], d2['H'][0], ld[0], 1)
    for i in range(1, L):
        ops.dense_bn_small(d2['H'][i - 1], ld[i - 1], Wd[i], hs[i], bd[i], B, hs[i - 1], hs[i], batchnorm, betad[i], d2['mm'][i],
                           d2['mv'][i], 0.99, 1e-3, 1, d2['Z'][i], ld[i], d2['XH'][i], ld[i], d2['H'][i], ld[i], d2['inv'][i])
    torch.cuda.synchronize()
    check(d2, 'per layer')
    # ---- backward of the last layer (whole-layer kernel) and of layer 0 (batch-norm only)
    if B == 1 and batchnorm:
        return              # one row: xhat = 0, inv_std = eps^-1/2 -- the backward amplifies rounding, nothing to compare
    i = L - 1
    dH = rng.normal(size=(B, hs[i]))
    Hi, xh, inv = ref[i]['H'], ref[i]['xh'], ref[i]['inv']
    dy = dH * (Hi > 0)
    if batchnorm:
        s1, s2 = dy.sum(0), (dy * xh).sum(0)
        dZ = inv * (dy - s1 / B - xh * s2 / B)
    else:
        s1, dZ = None, dy
    Hp = ref[i - 1]['H']
    gW_ref, gb_ref, dHp_ref = Hp.T @ dZ, dZ.sum(0), dZ @ W[i].T
    dHd = torch.zeros(B, ld[i], **f32); dHd[:, :hs[i]] = dev(dH.astype(np.float32))
    gW = torch.zeros(hs[i - 1] + 1, hs[i], **f32); dbeta = torch.zeros(hs[i], **f32)
    dHp = torch.zeros(B, ld[i - 1], **f32)
    ops.dense_bn_bwd_small(dHd, ld[i], d['H'][i], ld[i], d['XH'][i] if batchnorm else None, ld[i], d['inv'][i], d['H'][i - 1], ld[i - 1],
                           Wd[i], hs[i], B, hs[i - 1], hs[i], batchnorm, float(B), 1, gW, hs[i], dbeta if batchnorm else None,
                           dHp, ld[i - 1])
    torch.cuda.synchronize()
    sc = np.abs(gW_ref).max() + 1e-6
    np.testing.assert_allclose(gW[:hs[i - 1]].cpu().numpy(), gW_ref, rtol=1e-3, atol=2e-5 * sc)
    # (with batch norm the bias gradient is a sum that cancels exactly: the tolerance is relative to the summed magnitudes)
    np.testing.assert_allclose(gW[hs[i - 1]].cpu().numpy(), gb_ref, rtol=1e-3, atol=2e-6 * (np.abs(dZ).sum(0).max() + 1e-6))
    np.testing.assert_allclose(dHp[:, :hs[i - 1]].cpu().numpy(), dHp_ref, rtol=1e-3, atol=2e-5 * (np.abs(dHp_ref).max() + 1e-6))
    if batchnorm:
        np.testing.assert_allclose(dbeta.cpu().numpy(), s1, rtol=1e-4, atol=1e-5)
        dH0 = rng.normal(size=(B, hs[0]))
        dy0 = dH0 * (ref[0]['H'] > 0)
        t1, t2 = dy0.sum(0), (dy0 * ref[0]['xh']).sum(0)
        dZ0_ref = ref[0]['inv'] * (dy0 - t1 / B - ref[0]['xh'] * t2 / B)
        dH0d = torch.zeros(B, ld[0], **f32); dH0d[:, :hs[0]] = dev(dH0.astype(np.float32))
        dZ0 = torch.zeros(B, ld[0], **f32); db0 = torch.zeros(hs[0], **f32)
        ops.bn_bwd_small(dH0d, ld[0], d['H'][0], ld[0], d['XH'][0], ld[0], d['inv'][0], float(B), B, hs[0], dZ0, ld[0], db0)
        torch.cuda.synchronize()
        np.testing.assert_allclose(dZ0[:, :hs[0]].cpu().numpy(), dZ0_ref, rtol=1e-3, atol=2e-5 * (np.abs(dZ0_ref).max() + 1e-6))
        np.testing.assert_allclose(db0.cpu().numpy(), t1, rtol=1e-4, atol=1e-5)


# ------------------------------------------------------------------------------- batch norm
@pytest.mark.parametrize('B,H', [(32, 64), (25, 32), (300, 64), (1000, 130), (8, 1)])
def test_bn_forward_backward(ops, B, H):
    rng = np.random.RandomState(B + H)
    Zm = (rng.normal(0.3, 2.0, (B, H)) + rng.normal(0, 3, H)).astype(np.float32)
    beta = rng.normal(0, .5, H).astype(np.float32)
    mm0 = rng.normal(0, 1, H).astype(np.float32); mv0 = rng.uniform(.5, 2, H).astype(np.float32)
    dHm = rng.normal(0, 1, (B, H)).astype(np.float32)
    z = Zm.astype(np.float64)
    mu = z.mean(0); var = ((z - mu) ** 2).mean(0)
    inv = 1 / np.sqrt(var + 1e-3)
    xh = (z - mu) * inv
    yb = xh + beta
    h = np.maximum(yb, 0)
    dy = dHm * (yb > 0)
    dz = inv * (dy - dy.mean(0) - xh * (dy * xh).mean(0))
    ldz = H + (-H) % 4
    dZ_in = dev(pad_cols(Zm, ldz))
    R = ops.col_moments_chunks(B)
    part = torch.zeros(R * 2 * H, device='cuda')
    ops.col_moments(dZ_in, ldz, B, H, part)
    Hout = torch.zeros(B, ldz, device='cuda'); xhat = torch.zeros(B, ldz, device='cuda')
    inv_std = torch.zeros(H, device='cuda')
    mm, mv = dev(mm0), dev(mv0)
    ops.bn_relu_apply(dZ_in, ldz, B, H, part, None, R, dev(beta), mm, mv, 0.99, 1e-3, True,
                      Hout, ldz, xhat, ldz, inv_std)
    torch.cuda.synchronize()
    np.testing.assert_allclose(Hout.cpu().numpy()[:, :H], h, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(xhat.cpu().numpy()[:, :H], xh, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(inv_std.cpu().numpy(), inv, rtol=1e-5)
    np.testing.assert_allclose(mm.cpu().numpy(), mm0 - (mm0 - mu) * 0.01, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mv.cpu().numpy(), mv0 - (mv0 - var) * 0.01, rtol=1e-5, atol=1e-6)
    # backward
    ddH = dev(pad_cols(dHm, ldz))
    bpart = torch.zeros(R * 2 * H, device='cuda')
    ops.bn_bwd_sums(ddH, ldz, Hout, ldz, xhat, ldz, B, H, bpart)
    dZ = torch.zeros(B, ldz, device='cuda'); dbeta = torch.zeros(H, device='cuda')
    ops.bn_bwd_apply(ddH, ldz, Hout, ldz, xhat, ldz, inv_std, bpart, R, float(B), B, H, dZ, ldz, dbeta)
    torch.cuda.synchronize()
    np.testing.assert_allclose(dZ.cpu().numpy()[:, :H], dz, rtol=1e-4, atol=2e-5 * np.abs(dz).max())
    np.testing.assert_allclose(dbeta.cpu().numpy(), dy.sum(0), rtol=1e-4, atol=1e-4)
    # inference mode uses the moving statistics and leaves them untouched
    mm_b, mv_b = mm.clone(), mv.clone()
    ops.bn_relu_apply(dZ_in, ldz, B, H, None, None, 0, dev(beta), mm, mv, 0.99, 1e-3, True,
                      Hout, ldz, None, 0, None)
    torch.cuda.synchronize()
    ref = np.maximum((z - mm_b.cpu().numpy()) / np.sqrt(mv_b.cpu().numpy() + 1e-3) + beta, 0)
    np.testing.assert_allclose(Hout.cpu().numpy()[:, :H], ref, rtol=2e-5, atol=2e-5)
    assert torch.equal(mm, mm_b) and torch.equal(mv, mv_b)
    # relu backward without batch norm
    ops.relu_bwd(ddH, ldz, Hout, ldz, B, H, dZ, ldz)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(dZ.cpu().numpy()[:, :H], dHm * (ref > 0))


def test_moments_combine_matches_global_stats(ops):
    """SyncBN building block: merging per-rank (count, mean, M2) == statistics of the union."""
    rng = np.random.RandomState(0)
    H = 64
    parts = [rng.normal(1 + r, 2, (b, H)).astype(np.float32) for r, b in enumerate([40, 17, 64])]
    ent, cnt = [], []
    for p in parts:
        B = p.shape[0]
        R = ops.col_moments_chunks(B)
        part = torch.zeros(R * 2 * H, device='cuda')
        ops.col_moments(dev(p), H, B, H, part)
        cr = -(-B // R)
        counts = dev(np.array([min(B, (r + 1) * cr) - r * cr for r in range(R)], np.float32))
        out = torch.zeros(2 * H, device='cuda')
        ops.moments_combine(part, counts, R, H, out)
        ent.append(out); cnt.append(float(B))
    allz = np.concatenate(parts).astype(np.float64)
    Zd = dev(allz)
    Hout = torch.zeros_like(Zd); inv_std = torch.zeros(H, device='cuda')
    mm = torch.zeros(H, device='cuda'); mv = torch.ones(H, device='cuda')
    ops.bn_relu_apply(Zd, H, allz.shape[0], H, torch.cat(ent), dev(np.array(cnt, np.float32)), 3,
                      None, mm, mv, 0.99, 1e-3, False, Hout, H, None, 0, inv_std)
    torch.cuda.synchronize()
    mu = allz.mean(0); var = allz.var(0)
    np.testing.assert_allclose(Hout.cpu().numpy(), (allz - mu) / np.sqrt(var + 1e-3), rtol=2e-5, atol=2e-5)


# ------------------------------------------------------------------------------- optimizer
def test_rmsprop_clip(ops):
    rng = np.random.RandomState(1)
    n = 4 * 1000 + 3
    w = rng.normal(0, 1, n).astype(np.float32); g = (rng.normal(0, 4, n)).astype(np.float32)
    ms = rng.uniform(0, 1, n).astype(np.float32)
    ms[5:400] *= 1e-12                                      # accumulators far below epsilon: where its placement matters
    g[5:400] *= 1e-6
    g[:5] = [0, 1e-9, 7.5, -9, 5.0]
    dw, dg, dms = dev(w), dev(g), dev(ms)
    lr = torch.tensor([1e-3], device='cuda')
    ops.rmsprop_clip(dw, dg, dms, n, lr, 0.9, 1e-7, 5.0)
    torch.cuda.synchronize()
    gc = np.clip(g.astype(np.float64), -5, 5)
    ms_ref = 0.9 * ms + 0.1 * gc * gc
    w_ref = w - 1e-3 * gc / (np.sqrt(ms_ref) + 1e-7)        # Keras RMSprop without momentum: epsilon outside the root
    np.testing.assert_allclose(dms.cpu().numpy(), ms_ref, rtol=1e-6)
    np.testing.assert_allclose(dw.cpu().numpy(), w_ref, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize('B,G,ld', [(1, 1, 4), (37, 203, 208), (300, 20000, 20004)])
def test_shared_head_plumbing_kernels(B, G, ld):
    """dcahip_bcast_cols / dcahip_row_sums_strided (Dense(1) heads of the *-shared networks)."""
    from dca_amd.ops import HipOps
    ops = HipOps()
    dev = torch.device('cuda')
    rng = np.random.RandomState(B)
    s = torch.as_tensor(rng.normal(size=(B, 4)).astype(np.float32)).to(dev)
    out = torch.full((B, ld), -3.0, device=dev)
    ops.bcast_cols(s[:, 1:], 4, B, G, out, ld)
    got = out.cpu().numpy()
    np.testing.assert_array_equal(got[:, :G], np.repeat(s.cpu().numpy()[:, 1:2], G, axis=1))
    assert (got[:, G:] == -3.0).all()
    x = rng.normal(size=(B, ld)).astype(np.float32) * 1e-3
    xd = torch.as_tensor(x).to(dev)
    r = torch.full((B, 4), 9.0, device=dev)
    ops.row_sums_strided(xd, ld, B, G, r[:, 2:], 4)
    got = r.cpu().numpy()
    want = x[:, :G].astype(np.float64).sum(axis=1)
    np.testing.assert_allclose(got[:, 2], want, rtol=2e-7, atol=1e-9)
    assert (got[:, [0, 1, 3]] == 9.0).all()
    r2 = torch.zeros(B, 4, device=dev)
    ops.row_sums_strided(xd, ld, B, G, r2[:, 2:], 4)
    assert torch.equal(r2[:, 2], r[:, 2])                       # deterministic


@pytest.mark.parametrize('B,G,ld', [(1, 1, 4), (37, 203, 208), (4096, 2000, 2000)])
def test_elempi_kernels(B, G, ld):
    """dcahip_elempi_fwd / _bwd (zinb-elempi) against numpy fp64."""
    from dca_amd.ops import HipOps
    ops = HipOps()
    dev = torch.device('cuda')
    rng = np.random.RandomState(B + G)
    a = rng.normal(size=(B, ld)).astype(np.float32)
    k = rng.normal(size=G).astype(np.float32)
    c = rng.normal(size=G).astype(np.float32)
    ad, kd, cd = (torch.as_tensor(x).to(dev) for x in (a, k, c))
    pi = torch.full((B, ld), 5.0, device=dev)
    ops.elempi_fwd(ad, ld, kd, cd, B, G, pi, ld)
    m = -a[:, :G]
    np.testing.assert_array_equal(ad.cpu().numpy()[:, :G], m)
    np.testing.assert_allclose(pi.cpu().numpy()[:, :G], k * m + c, rtol=1e-6, atol=1e-6)
    assert (pi.cpu().numpy()[:, G:] == 5.0).all() and (ad.cpu().numpy()[:, G:] == a[:, G:]).all()
    dm = rng.normal(size=(B, ld)).astype(np.float32) * 1e-3
    dp = rng.normal(size=(B, ld)).astype(np.float32) * 1e-3
    dmd, dpd = torch.as_tensor(dm).to(dev), torch.as_tensor(dp).to(dev)
    gk = torch.zeros(G + 3, device=dev); gc = torch.zeros(G + 3, device=dev)
    ws = torch.zeros(ops.elempi_workspace_doubles(G), dtype=torch.float64, device=dev)
    ops.elempi_bwd(ad, ld, dmd, dpd, ld, kd, B, G, gk, gc, ws)
    np.testing.assert_allclose(dmd.cpu().numpy()[:, :G], -(dm[:, :G] + k * dp[:, :G]), rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(gk.cpu().numpy()[:G], (dp[:, :G].astype(np.float64) * m).sum(0), rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(gc.cpu().numpy()[:G], dp[:, :G].astype(np.float64).sum(0), rtol=1e-5, atol=1e-8)
    gk2 = torch.zeros_like(gk); gc2 = torch.zeros_like(gc)
    dmd.copy_(torch.as_tensor(dm).to(dev))
    ops.elempi_bwd(ad, ld, dmd, dpd, ld, kd, B, G, gk2, gc2, ws)
    assert torch.equal(gk, gk2) and torch.equal(gc, gc2)


@pytest.mark.parametrize('B,h,ld', [(1, 1, 4), (37, 64, 64), (4096, 513, 516)])
def test_prelu_kernels(B, h, ld):
    from dca_amd.ops import HipOps
    ops = HipOps()
    dev = torch.device('cuda')
    rng = np.random.RandomState(B + h)
    x = rng.normal(size=(B, ld)).astype(np.float32)
    x[0, 0] = 0.0
    al = rng.normal(0, 0.5, size=h).astype(np.float32)
    xd, ad = torch.as_tensor(x).to(dev), torch.as_tensor(al).to(dev)
    out = torch.full((B, ld), 5.0, device=dev)
    ops.prelu_fwd(xd, ld, ad, B, h, out, ld)
    xv = x[:, :h]
    np.testing.assert_array_equal(out.cpu().numpy()[:, :h], np.where(xv > 0, xv, al * xv).astype(np.float32))
    assert (out.cpu().numpy()[:, h:] == 5.0).all()
    d = rng.normal(size=(B, ld)).astype(np.float32) * 1e-2
    dd = torch.as_tensor(d).to(dev)
    ga = torch.zeros(h + 3, device=dev)
    ws = torch.zeros(ops.prelu_workspace_doubles(h), dtype=torch.float64, device=dev)
    ops.prelu_bwd(dd, ld, xd, ld, ad, B, h, ga, ws)
    np.testing.assert_array_equal(dd.cpu().numpy()[:, :h], np.where(xv > 0, d[:, :h], al * d[:, :h]).astype(np.float32))
    np.testing.assert_allclose(ga.cpu().numpy()[:h], (d[:, :h].astype(np.float64) * np.minimum(xv, 0)).sum(0),
                               rtol=1e-5, atol=1e-8)
