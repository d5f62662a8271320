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
