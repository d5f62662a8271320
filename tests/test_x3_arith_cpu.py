"""The split-bf16 arithmetic of the matrix-pipe kernels, restated in numpy (oracle/x3_np.py), against fp64: the bound the
GPU tests hold the kernels to, and why three products instead of six cannot meet it."""
import numpy as np

from oracle import x3_np as X


def test_three_pieces_carry_an_fp32_value():
    rng = np.random.RandomState(0)
    x = (rng.standard_normal(20000) * np.exp(rng.uniform(-20, 20, 20000))).astype(np.float32)
    p0, p1, p2 = X.split3(x)
    for p in (p0, p1, p2):
        assert (p.view(np.uint32) & 0xFFFF == 0).all()                 # bf16 values
    err = np.abs(x.astype(np.float64) - (p0.astype(np.float64) + p1 + p2))
    assert (err <= 2.0 ** -24 * np.abs(x)).all()
    assert np.array_equal(X.bf16_round(np.float32([1.0, 1.00390625, 1.01171875])), np.float32([1.0, 1.0, 1.015625]))  # ties to even


def test_six_products_are_fp32_accurate_and_three_are_not():
    rng = np.random.RandomState(1)
    for M, K, N in ((64, 4096, 64), (32, 20000, 64), (96, 512, 32)):
        a = rng.standard_normal((M, K)).astype(np.float32)
        b = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
        ref = a.astype(np.float64) @ b.astype(np.float64)
        mag = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
        e6 = (np.abs(X.matmul_x3(a, b, 6) - ref) / mag).max()
        e3 = (np.abs(X.matmul_x3(a, b, 3) - ref) / mag).max()
        assert e6 <= 5e-7, (M, K, N, e6)                                # DESIGN section 4.1's contract (dcahip_x3_product_32x32)
        assert e3 >= 8 * e6, (M, K, N, e3, e6)                          # random signs: the dropped terms average out, and still show
    # operands whose second pieces line up: the dropped a2 b2 term is 2^-18 of every product -- systematic, whatever K
    a = np.full((32, 256), 1.0 + 2.0 ** -9, np.float32); b = np.full((256, 32), 1.0 + 2.0 ** -9, np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    e6 = (np.abs(X.matmul_x3(a, b, 6) - ref) / ref).max()
    e3 = (np.abs(X.matmul_x3(a, b, 3) - ref) / ref).max()
    assert e6 <= 1.2e-7 and 3e-6 <= e3 <= 5e-6, (e6, e3)                # 2^-18 = 3.8e-6: beyond every bound the GPU tests use


def test_first_layer_forward_from_count_tables():
    """dcahip_enc0_fwd_lut's algebra: Z = L (W / std) + (b - sum_g (mean / std) W[g, :]) with L looked up per count from
    the cell's table equals the dense X W + b of dca/network.py:124-126 on the input of dca/io.py:88-111."""
    rng = np.random.RandomState(2)
    n, G, H = 48, 700, 64
    y = rng.poisson(0.3, (n, G)).astype(np.float64) * (rng.uniform(size=(n, G)) < 0.5)
    y[3, 5] = 40.0; y[7, 9] = 300.0
    fac = rng.lognormal(0, 0.4, n).astype(np.float32).astype(np.float64)
    L = np.log1p(y / fac[:, None])
    mean = L.mean(0); std = np.maximum(L.std(0, ddof=1), 1e-3)
    W = (rng.standard_normal((G, H)) * 0.1).astype(np.float32); b = (rng.standard_normal(H) * 0.3).astype(np.float32)
    ref = ((L - mean) / std) @ W.astype(np.float64) + b
    table = np.log1p(np.arange(64)[None, :] / fac[:, None]).astype(np.float32)          # dcahip_enc0_lut
    codes = np.minimum(y, 255).astype(np.int64)
    Lk = np.where(codes < 64, np.take_along_axis(table, np.minimum(codes, 63), axis=1),
                  np.log1p(y / fac[:, None]).astype(np.float32)).astype(np.float32)       # beyond the table: the formula
    Wp = (W / std[:, None].astype(np.float32)).astype(np.float32)
    c0 = -((mean / std)[:, None] * W.astype(np.float64)).sum(0)
    Z = X.matmul_x3(Lk, Wp).astype(np.float64) + c0.astype(np.float32) + b
    mag = np.abs(L / std) @ np.abs(W).astype(np.float64) + np.abs(mean / std) @ np.abs(W).astype(np.float64) + np.abs(b)
    assert (np.abs(Z - ref) <= 1e-6 * mag).all(), float((np.abs(Z - ref) / mag).max())


def test_two_fp16_pieces_carry_an_fp32_value_after_block_scaling():
    """x 2^e = h1 + h2 to 2^-22 |x 2^e| where h2 is a normal fp16 (|x 2^e| >= 2^-3), to 2^-25 absolute below (h2 an fp16
    denormal: the matrix pipe of gfx950 preserves them, tools/microbench/mfma_f16_denorm.hip)."""
    rng = np.random.RandomState(3)
    x = (rng.standard_normal(20000) * np.exp(rng.uniform(-12, 0, 20000))).astype(np.float32)
    e = X.block_exp(x)
    assert 2.0 ** 13 <= np.abs(x).max() * 2.0 ** e < 2.0 ** 14
    h1, h2 = X.split2(x, e)
    xs = np.ldexp(x.astype(np.float64), e)
    err = np.abs(xs - (h1.astype(np.float64) + h2))
    assert (err <= np.maximum(2.0 ** -22 * np.abs(xs), 2.0 ** -25)).all()
    assert np.isfinite(h1).all() and np.abs(h1).max() < 65504
    assert X.block_exp(np.zeros(5, np.float32)) == 0


def test_three_fp16_products_are_fp32_accurate_and_two_are_not():
    """K-HEADS' arithmetic from round 6 against fp64: random operands, operands with a wide dynamic range (gradients), and the
    systematic case; a build that drops one cross term is off by 2^-12 of every product."""
    rng = np.random.RandomState(1)
    for M, K, N in ((64, 4096, 64), (32, 20000, 64), (96, 512, 32)):
        a = rng.standard_normal((M, K)).astype(np.float32)
        b = (rng.standard_normal((K, N)) * 0.05).astype(np.float32)
        ref = a.astype(np.float64) @ b.astype(np.float64)
        mag = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
        e3 = (np.abs(X.matmul_h2(a, b, 3) - ref) / mag).max()
        e2 = (np.abs(X.matmul_h2(a, b, 2) - ref) / mag).max()
        e6 = (np.abs(X.matmul_x3(a, b, 6) - ref) / mag).max()
        assert e3 <= 5e-7 and e3 <= 4 * e6 + 1e-8, (M, K, N, e3, e6)   # the contract of dcahip_x3_product_32x32; within 4x of bf16 x 3 / six
        assert e2 >= 20 * e3, (M, K, N, e2, e3)
    # gradients: a log-normal spread of three decades around the typical value, the block scale fixed by the largest
    a = (rng.standard_normal((64, 4096)) * np.exp(rng.normal(0, 1.5, (64, 4096)))).astype(np.float32)
    b = (rng.standard_normal((4096, 64)) * 0.05).astype(np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    mag = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert (np.abs(X.matmul_h2(a, b, 3) - ref) / mag).max() <= 5e-7
    # operands whose second pieces line up: the dropped a2 b2 is 2^-22 of every product -- systematic, whatever K
    a = np.full((32, 256), 1.0 + 2.0 ** -11, np.float32); b = np.full((256, 32), 1.0 + 2.0 ** -11, np.float32)
    ref = a.astype(np.float64) @ b.astype(np.float64)
    e3 = (np.abs(X.matmul_h2(a, b, 3) - ref) / ref).max()
    e2 = (np.abs(X.matmul_h2(a, b, 2) - ref) / ref).max()
    assert e3 <= 3e-7 and 2e-4 <= e2 <= 6e-4, (e3, e2)


def test_fp16_pieces_on_the_operands_of_a_training_step():
    """The three products of K-HEADS on the operands of an actual step (decoder output after ReLU, glorot head weights, the
    ZINB gradient planes of the oracle: a median |g| of 0.16, a largest of several hundred) with the kernel's scales -- H and W
    by their largest magnitude, D = g 2^8 (kDExp0) -- against fp64: inside the bounds of product_tol."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from helpers import make_problem, oracle_net
    from oracle import zinb_np as Z
    n, G, hs = 256, 1500, (64, 32, 64)
    Xd, Y, sf, p = make_problem(n, G, hs, 'zinb-conddisp', True, seed=3)
    net = oracle_net('zinb-conddisp', p, hs, True)
    net.loss_and_grads(Xd.astype(np.float64), Y.astype(np.float64), sf.astype(np.float64))
    c = net.cache
    _, _, dm, dd, dpi = Z.zinb_loss_and_grads(c['a_mean'], c['a_disp'], c['a_pi'], Y.astype(np.float64), c['sf'], 0.0, None, None)
    g = (np.concatenate([dm, dd, dpi], axis=1) * (n * G)).astype(np.float32)          # unscaled gradients
    H = np.maximum(c['H'][-1], 0).astype(np.float32)
    W = np.concatenate([p['W_mean'], p['W_disp'], p['W_pi']], axis=1).astype(np.float32)
    # D = g 2^8 where that fits the fp16 range; a tile holding a larger gradient (here: several hundred, at a large count) is
    # scaled down as a whole by the kernel's slow path -- emulated on the whole matrix (the less favourable case)
    ed = min(8, X.block_exp(g))
    assert ed < 8 and np.abs(g).max() * 2.0 ** ed < 2.0 ** 14

    def err(a, b, **kw):
        ref = a.astype(np.float64) @ b.astype(np.float64)
        mag = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
        return (np.abs(X.matmul_h2(a, b, 3, **kw) - ref) / np.maximum(mag, 1e-300)).max()
    assert err(H, W) <= 5e-7                                                           # F
    assert err(g, W.T.copy(), ea=ed) <= 1e-6                                           # dH
    assert err(H.T.copy(), g, eb=ed) <= 1.5e-6                                         # dW
