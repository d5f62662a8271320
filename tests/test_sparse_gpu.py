"""K-SPARSE parity (include/dcahip.h): the compact byte store of the counts, the first Dense layer on the non-zero
counts only (forward and weight gradient; dca/io.py:88-111 feeding dca/network.py:124-126) and K-HEADS reading its
targets from the byte store -- against fp64 numpy on the dense matrix / against the fp32-count path, through the C ABI.
"""
import numpy as np
import pytest
import torch

from conftest import synth_counts
from helpers import make_problem, oracle_net, assert_grads_close, run_single_step

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ops():
    from dca_amd.ops import HipOps
    return HipOps()


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


def counts_with_escapes(n, G, seed, big=True):
    y = synth_counts(n, G, seed)
    if big:                                    # counts at and beyond the escape code, several per row, first / last column
        rng = np.random.RandomState(seed)
        y[0, 0] = 255; y[0, G - 1] = 254; y[min(1, n - 1), G // 2] = 70000
        y[2 % n, 1 % G] = 31; y[2 % n, 2 % G] = 32; y[3 % n, 5 % G] = 200; y[n - 1, G - 2] = 63; y[n - 1, G - 1] = 64   # table edges
        for r in rng.randint(0, n, 6):
            y[r, rng.randint(0, G, 3)] = rng.randint(255, 5000, 3)
    return y


def build_compact(ops, Y):
    from dca_amd import compact
    n, G = Y.shape
    Gp = (G + 3) // 4 * 4
    Yd = torch.zeros(n, Gp, device='cuda'); Yd[:, :G] = dev(Y)
    return Yd, compact.build(ops, Yd, n, G)


def decode(cc, n, G):
    out = cc.Yc.cpu().numpy()[:, :G].astype(np.float64)
    if cc.ovf_ptr is not None:
        ptr, col, val = cc.ovf_ptr.cpu().numpy(), cc.ovf_col.cpu().numpy(), cc.ovf_val.cpu().numpy()
        for r in range(n):
            for i in range(ptr[r], ptr[r + 1]):
                assert out[r, col[i]] == 255
                out[r, col[i]] = val[i]
    return out


@pytest.mark.parametrize('n,G', [(37, 50), (64, 1000), (5, 16), (130, 33)])
def test_compact_store_round_trip(ops, n, G):
    Y = counts_with_escapes(n, G, n + G)
    _, cc = build_compact(ops, Y)
    assert cc is not None and cc.ldc % 16 == 0 and cc.ldc >= G
    assert (cc.Yc.cpu().numpy()[:, G:] == 0).all()
    np.testing.assert_array_equal(decode(cc, n, G), Y)
    esc = int((Y >= 255).sum())
    assert (cc.ovf_col is None and esc == 0) or cc.ovf_col.numel() == esc


def test_compact_store_refuses_what_is_not_a_count(ops):
    for bad in (0.5, -1.0, float('nan'), float('inf')):
        Y = synth_counts(20, 40, 3)
        Y[7, 11] = bad
        assert build_compact(ops, Y)[1] is None


def dense_input(Y, fac, do_log, mean, std):
    x = Y / fac[:, None] if fac is not None else Y.copy()
    if do_log:
        x = np.log1p(x)
    if mean is not None:
        x = x - mean[None, :]
    if std is not None:
        x = x / std[None, :]
    return x


CASES = [
    # B, G, H1, gather, fac, log, scale, escapes
    (32, 200, 64, True, True, True, True, False),
    (300, 1000, 64, True, True, True, True, True),
    (1100, 130, 64, False, True, True, True, True),
    (64, 77, 32, True, True, True, False, True),
    (129, 500, 16, True, False, True, True, False),
    (70, 300, 128, False, True, False, True, True),
    (33, 90, 256, True, False, False, False, True),
    (2100, 64, 64, True, True, True, True, False),
    (517, 290, 32, True, True, True, True, False),
    (70, 300, 128, True, True, True, True, False),
    (300, 1000, 64, True, True, True, True, False),
    (1100, 130, 64, False, True, False, True, False),
    (64, 77, 32, True, False, True, False, False),
    (4101, 3000, 64, True, True, True, True, True),       # several macro steps per gene chunk of the matrix-pipe forward
    (600, 2050, 32, False, True, True, True, True),
]


@pytest.mark.parametrize('B,G,H1,gather,use_fac,do_log,scale,esc', CASES)
def test_sparse_first_layer_vs_numpy(ops, B, G, H1, gather, use_fac, do_log, scale, esc):
    rng = np.random.RandomState(B + G + H1)
    n = B + 9
    Y = counts_with_escapes(n, G, B + G, big=esc)
    fac = (rng.lognormal(0, 0.4, n)).astype(np.float32).astype(np.float64) if use_fac else None
    L = dense_input(Y, fac, do_log, None, None)
    mean = L.mean(0).astype(np.float32).astype(np.float64) if scale else None
    std = np.maximum(L.std(0, ddof=1), 1e-3).astype(np.float32).astype(np.float64) if scale else None
    X = dense_input(Y, fac, do_log, mean, std)
    if gather:
        perm = rng.permutation(n)[:B + 3].astype(np.int32); cur = 3
        rows = perm[cur:cur + B]
    else:
        perm = None; cur = 4
        rows = np.arange(cur, cur + B)
    f32 = lambda a: a.astype(np.float32).astype(np.float64)
    W = f32(rng.normal(0, 0.1, (G, H1)))
    b = f32(rng.normal(0, 0.3, H1))
    dZ = f32(rng.normal(0, 1e-3, (B, H1)))
    Z_ref = X[rows] @ W + b
    gW_ref = X[rows].T @ dZ
    Z_abs = np.abs(X[rows]) @ np.abs(W) + np.abs(b)
    gW_abs = np.abs(X[rows]).T @ np.abs(dZ)
    if scale:       # the sparse form sums L / std and the mean correction separately: bound by those magnitudes
        Lr = dense_input(Y, fac, do_log, None, None)[rows] / std[None, :]
        Z_abs = np.abs(Lr) @ np.abs(W) + np.abs(mean / std) @ np.abs(W) + np.abs(b)
        gW_abs = np.abs(Lr).T @ np.abs(dZ) + np.abs(mean / std)[:, None] * np.abs(dZ).sum(0)[None, :] + \
            np.abs(mean / std)[:, None] * np.abs(dZ.sum(0))[None, :]

    _, cc = build_compact(ops, Y)
    cc = cc.with_input(dev(fac) if use_fac else None, do_log, dev(mean) if scale else None, dev(std) if scale else None, ops=ops)
    dperm = torch.as_tensor(perm).cuda() if perm is not None else None
    dcur = torch.tensor([cur if gather else 0], dtype=torch.int64, device='cuda')
    base = 0 if gather else cur
    dW_, db_, ddZ = dev(W), dev(b), dev(dZ)
    # ---- forward over the non-zero counts only: an EXPERIMENT build of the library (-DDCA_EXP_ENC0_SPARSE_FWD; it lost to the
    # dense product) -- checked when such a build is the one loaded (run twice: the arrival counter of the bias correction
    # must be back at zero)
    if ops.has('dcahip_enc0_fwd_sparse'):
        Zd = torch.full((B, H1), 7.0, device='cuda')
        wsf = torch.zeros(ops.enc0_fwd_sparse_workspace_bytes(H1) // 4 + 4, device='cuda')
        for _ in range(2):
            ops.enc0_fwd_sparse(cc, dperm, dcur, base, B, G, H1, dW_, H1, db_, Zd, H1, wsf)
        torch.cuda.synchronize()
        err = np.abs(Zd.cpu().numpy() - Z_ref)
        assert (err <= 1e-6 * Z_abs + 1e-30).all(), float((err / Z_abs).max())
    # ---- the product on the matrix pipe (looked-up operand), where the width is taken; twice, bit for bit
    nb = ops.enc0_fwd_lut_workspace_bytes(B, G, H1)
    assert (nb > 0) == (H1 in (32, 64))
    if nb:
        wsl = torch.full((nb // 4 + 4,), float("nan"), device="cuda")
        Zl = torch.full((B, H1), 7.0, device='cuda'); Zl2 = torch.full((B, H1), 3.0, device='cuda')
        ops.enc0_fwd_lut(cc, dperm, dcur, base, B, G, H1, dW_, H1, db_, Zl, H1, wsl)
        ops.enc0_fwd_lut(cc, dperm, dcur, base, B, G, H1, dW_, H1, db_, Zl2, H1, wsl)
        torch.cuda.synchronize()
        err = np.abs(Zl.cpu().numpy() - Z_ref)
        assert (err <= 1e-6 * Z_abs + 1e-30).all(), float((err / Z_abs).max())
        assert torch.equal(Zl, Zl2)
    # ---- weight + bias gradient
    if not ops.enc0_sparse_supported(H1):
        return
    # (64 units: both kernels at every size through the call's `form` argument -- by the shape the library takes the ring
    # kernel from 1024 rows up)
    for form in ((1, 2, 0) if H1 == 64 else (0,)):
        gWd = torch.full((G + 1, H1), 7.0, device='cuda')
        ws = torch.full((ops.enc0_dw_sparse_workspace_bytes(B, G, H1) // 4 + 4,), float('nan'), device='cuda')
        ops.enc0_dw_sparse(cc, dperm, dcur, base, B, G, H1, ddZ, H1, gWd, H1, ws, form=form)
        torch.cuda.synchronize()
        got = gWd.cpu().numpy()
        err = np.abs(got[:G] - gW_ref)
        assert (err <= 1e-6 * gW_abs + 1e-30).all(), (form, float((err / np.maximum(gW_abs, 1e-30)).max()))
        np.testing.assert_allclose(got[G], dZ.sum(0), rtol=0, atol=2e-6 * np.abs(dZ).sum(0).max())
        # deterministic: a second launch reproduces every bit
        gW2 = torch.zeros_like(gWd)
        ops.enc0_dw_sparse(cc, dperm, dcur, base, B, G, H1, ddZ, H1, gW2, H1, ws, form=form)
        assert torch.equal(gW2, gWd), form


@pytest.mark.parametrize('flags', [1, 3, 0, 2])
@pytest.mark.parametrize('B,G', [(32, 200), (96, 330), (300, 500)])
def test_heads_read_the_byte_store_bit_for_bit(ops, flags, B, G):
    """K-HEADS on the compact counts = K-HEADS on the fp32 counts (same values, same arithmetic), escapes (counts >= 255)
    included: every output bit where the compiler rounds the two template instantiations alike -- which it does for all but
    the per-gene-dispersion NB kernel of round 6, whose instantiations contract one multiply-add of the likelihood
    differently (1 ulp on ~10 % of the gradient elements, measured): held to 2e-6 of the tensor's scale there, and each
    instantiation to itself bit for bit (a second launch)."""
    rng = np.random.RandomState(B + G + flags)
    hL = 64
    Gp = (G + 3) // 4 * 4
    nh = 1 + (0 if flags & 2 else 1) + (1 if flags & 1 else 0)
    NH = nh * Gp
    n = B + 7
    Y = counts_with_escapes(n, G, B + flags)
    Yd, cc = build_compact(ops, Y)
    Hd = dev(np.maximum(rng.normal(0.3, 1.0, (B, hL)), 0))
    Wh = dev(rng.normal(0, 0.25, (hL + 1, NH)))
    tw = dev(rng.normal(0, 1.5, Gp))
    sf = dev(rng.lognormal(0, 0.3, n))
    perm = torch.as_tensor(rng.permutation(n)[:B + 2].astype(np.int32)).cuda()
    cur = torch.tensor([2], dtype=torch.int64, device='cuda')
    outs = []
    for compact in (None, cc):
        runs = []
        for _ in range(2):
            gW = torch.full((hL + 1, NH), 7.0, device='cuda')
            gth = torch.full((Gp,), 7.0, device='cuda')
            dH = torch.full((B, hL), 7.0, device='cuda')
            part = torch.zeros(ops.max_partials, dtype=torch.float64, device='cuda')
            ws = torch.zeros(ops.heads_fused_workspace_bytes(B, hL, G, Gp, flags) // 4, device='cuda')
            loss = torch.zeros(1, device='cuda')
            ops.heads_fused(Hd, hL, Wh, NH, Wh[hL], Gp, tw if flags & 2 else None, None if compact is not None else Yd, Gp, sf,
                            perm, cur, B, hL, G, 0.01, 1.0 / (B * G), flags, gW, NH, gth if flags & 2 else None, dH, hL, part, ws,
                            loss_out=loss, compact=compact)
            torch.cuda.synchronize()
            runs.append((gW, gth, dH, loss))
        for a, b in zip(*runs):
            assert torch.equal(a, b)                       # deterministic: a second launch reproduces every bit
        outs.append(runs[0])
    for a, b in zip(*outs):
        if flags == 2 and B >= 160:
            assert (a - b).abs().max().item() <= 2e-6 * max(a.abs().max().item(), 1e-30)
        else:
            assert torch.equal(a, b)
    assert np.isfinite(outs[0][3].item())


@pytest.mark.parametrize('ae_type', ['zinb-conddisp', 'zinb', 'nb-conddisp', 'nb'])
@pytest.mark.parametrize('B', [32, 300])
def test_step_with_compact_counts_matches_oracle(ops, ae_type, B):
    """One full training step with the counts in the byte store and BOTH first-layer products built from it (at every batch
    size: the thresholds are lowered): loss and every gradient against the fp64 oracle fed the dense input (the statement of
    test_single_step_matches_oracle)."""
    from dca_amd.engine import Engine
    n, G, hs = 320, 700, (64, 32, 64)
    _, Y, _, p = make_problem(n, G, hs, ae_type, True, seed=B)
    # counts that fit the byte codes (no escapes), then the normalisation of make_problem (fac = sf, log1p, z-score
    # with ddof = 1) -- as the engine is told it
    Y = np.minimum(Y, 200.0).astype(np.float32)
    lib = Y.astype(np.float64).sum(1)
    sf = (lib / np.median(lib)).astype(np.float32)
    Ln = np.log1p(Y.astype(np.float64) / sf.astype(np.float64)[:, None])
    mean, std = Ln.mean(0), np.maximum(Ln.std(0, ddof=1), 1e-12)
    X = ((Ln - mean) / std).astype(np.float32)
    rows = np.random.RandomState(1).permutation(n)[:B]
    ref = oracle_net(ae_type, p, hs, True)
    rl, rg = ref.loss_and_grads(X[rows].astype(np.float64), Y[rows].astype(np.float64), sf[rows].astype(np.float64))
    eng = Engine(ae_type, G, G, hs, True, 0.0, ops=ops)
    eng.set_params(p)
    Gp = (G + 3) // 4 * 4
    Xd = torch.zeros(n, Gp, device='cuda'); Xd[:, :G] = dev(X)
    Yd = torch.zeros(n, Gp, device='cuda'); Yd[:, :G] = dev(Y)
    eng.sparse_dw_min = eng.cfg.lut_fwd_min = 1          # both byte-store kernels at every batch size
    eng.reserve(B)
    norm = dict(fac=dev(sf), do_log=True, mean=dev(mean), std=dev(std))
    eng.attach_device_data(Xd, Yd, dev(sf), norm=norm)
    assert eng.cc is not None and eng.cc_in is not None
    if B == 32:
        # counts >= 255 escape into per-row lists.  A few (between one in 1e5 and one in 1e3 counts): the first layer stays
        # dense, K-HEADS still reads the bytes.  Many (read-count data with large counts): K-HEADS keeps the fp32 targets
        # too -- every escaped element costs it a linear scan of its row's list.
        Ye = Yd.clone(); Ye[:100, 0] = 300.0                 # 100 of 224 000 counts
        e3 = Engine(ae_type, G, G, hs, True, 0.0, ops=ops)
        e3.attach_device_data(Xd, Ye, dev(sf), norm=norm)
        assert e3.cc is not None and e3.cc.ovf_ptr is not None and e3.cc_in is None
        Ye = Yd.clone(); Ye[:, :40] = 300.0                  # 12 800 of 224 000
        e4 = Engine(ae_type, G, G, hs, True, 0.0, ops=ops)
        e4.attach_device_data(Xd, Ye, dev(sf), norm=norm)
        assert e4.cc is None and e4.cc_in is None
    loss, g, _ = run_single_step(eng, rows)
    assert eng._lut_fwd(B, True) and eng._sparse_dw(B)
    assert abs(loss - rl) < 1e-5 * abs(rl), (loss, rl)
    # (the biases in front of batch normalisation have an identically zero gradient: column sums of dZ that cancel --
    # the round-off left over depends on the order of the sum and is held to the absolute bound of the zero case)
    zero_b = tuple('b%d' % i for i in range(len(hs)))
    assert_grads_close(g, rg, skip=zero_b)
    gscale = max(float(np.abs(np.asarray(v)).max()) for v in rg.values())
    assert all(np.abs(g[k]).max() < 1e-5 * gscale for k in zero_b)
    # and the same step on the dense fp32 matrices: the two paths agree far inside the oracle tolerance
    eng2 = Engine(ae_type, G, G, hs, True, 0.0, ops=ops)
    eng2.set_params(p)
    eng2.attach_device_data(Xd, Yd, dev(sf))
    eng2.cc = None
    loss2, g2, _ = run_single_step(eng2, rows)
    assert abs(loss - loss2) < 2e-6 * abs(loss2)
    assert_grads_close(g, {k: np.asarray(v, np.float64) for k, v in g2.items()}, rtol=5e-4, atol_scale=5e-6, skip=zero_b)


@pytest.mark.parametrize('B,G,H1,gather,use_fac,do_log,scale,esc', [
    (32, 200, 64, True, True, True, True, False), (32, 20000, 64, True, True, True, True, True), (25, 333, 64, True, True, True, True, True),
    (64, 77, 32, False, True, True, False, True), (1, 50, 64, True, True, True, True, False), (33, 90, 16, True, False, False, False, True),
    (40, 130, 48, False, True, False, True, False)])
def test_small_batch_weight_gradient_kernel_vs_numpy(ops, B, G, H1, gather, use_fac, do_log, scale, esc):
    """(EXPERIMENT build -DDCA_EXP_DW_SMALL only: the kernel measured slower than the GEMM it would replace and is not in the
    product library.)  dcahip_enc0_dw_small: the first layer's weight + bias gradient of a batch of at most 64 rows (the reference's default
    32, its 25-row last batch of C3, a single row) over the non-zero counts of the byte store, against fp64 numpy on the
    dense input: |err| <= 1e-6 of the summed magnitudes, the bias row to 2e-6; a second launch reproduces every bit."""
    if not ops.has('dcahip_enc0_dw_small'):
        pytest.skip('experiment entry point: build the library with -DDCA_EXP_DW_SMALL')
    rng = np.random.RandomState(B + G + H1)
    n = B + 9
    Y = counts_with_escapes(n, G, B + G, big=esc)
    fac = (rng.lognormal(0, 0.4, n)).astype(np.float32).astype(np.float64) if use_fac else None
    L = dense_input(Y, fac, do_log, None, None)
    mean = L.mean(0).astype(np.float32).astype(np.float64) if scale else None
    std = np.maximum(L.std(0, ddof=1), 1e-3).astype(np.float32).astype(np.float64) if scale else None
    X = dense_input(Y, fac, do_log, mean, std)
    if gather:
        perm = rng.permutation(n)[:B + 3].astype(np.int32); cur = 3
        rows = perm[cur:cur + B]
    else:
        perm = None; cur = 4
        rows = np.arange(cur, cur + B)
    dZ = rng.normal(0, 1e-3, (B, H1)).astype(np.float32).astype(np.float64)
    gW_ref = X[rows].T @ dZ
    Lr = L[rows] / (std[None, :] if scale else 1.0)
    gW_abs = np.abs(Lr).T @ np.abs(dZ)
    if scale:
        gW_abs = gW_abs + np.abs(mean / std)[:, None] * np.abs(dZ).sum(0)[None, :]
    _, cc = build_compact(ops, Y)
    cc = cc.with_input(dev(fac) if use_fac else None, do_log, dev(mean) if scale else None, dev(std) if scale else None, ops=ops)
    dperm = torch.as_tensor(perm).cuda() if perm is not None else None
    dcur = torch.tensor([cur if gather else 0], dtype=torch.int64, device='cuda')
    base = 0 if gather else cur
    assert B <= ops.enc0_dw_small_max_rows
    gWd = torch.full((G + 1, H1), 7.0, device='cuda')
    ops.enc0_dw_small(cc, dperm, dcur, base, B, G, H1, dev(dZ), H1, gWd, H1)
    torch.cuda.synchronize()
    got = gWd.cpu().numpy()
    err = np.abs(got[:G] - gW_ref)
    assert (err <= 1e-6 * gW_abs + 1e-12 * np.abs(dZ).max()).all(), float((err / np.maximum(gW_abs, 1e-30)).max())
    np.testing.assert_allclose(got[G], dZ.sum(0), rtol=0, atol=2e-6 * np.abs(dZ).sum(0).max())
    gW2 = torch.zeros_like(gWd)
    ops.enc0_dw_small(cc, dperm, dcur, base, B, G, H1, dev(dZ), H1, gW2, H1)
    assert torch.equal(gW2, gWd)


@pytest.mark.parametrize('hs,B', [((16, 8, 16), 512), ((48, 16, 48), 600), ((16, 8, 16), 32)])
def test_first_layer_widths_the_byte_store_does_not_take_keep_the_dense_products(ops, hs, B):
    """A first layer of 16 or 48 units (the byte-store kernels take 32 / 64 / 128) on K-PREP-resident counts with a known
    normalisation, at a batch above the weight gradient's byte-store threshold: the engine keeps both dense products
    (cc_in stays None: K-HEADS still reads its targets from the byte store) and the step matches the fp64 oracle."""
    from dca_amd import prep
    from dca_amd.engine import Engine
    n, G = 640, 600
    _, Yh, _, p = make_problem(n, G, hs, 'zinb-conddisp', True, seed=17)
    Yh = np.minimum(Yh, 200.0)
    Gp = (G + 3) // 4 * 4
    Y = torch.zeros(n, Gp, device='cuda'); Y[:, :G] = dev(Yh)
    counts = prep.cell_counts(ops, Y, n, G)
    sf = counts / counts.median()
    X, norm = prep.transform(ops, Y, n, G, sf, True, True, return_norm=True)
    rows = np.random.RandomState(5).permutation(n)[:B]
    rt = torch.as_tensor(rows).cuda()
    ref = oracle_net('zinb-conddisp', p, hs, True)
    rl, rg = ref.loss_and_grads(X[rt][:, :G].cpu().numpy().astype(np.float64), Yh[rows].astype(np.float64),
                                sf[rt].cpu().numpy().astype(np.float64))
    eng = Engine('zinb-conddisp', G, G, hs, True, 0.0, ops=ops)
    eng.set_params(p)
    eng.reserve(B)
    eng.attach_device_data(X, Y, sf, norm=norm)
    assert eng.cc is not None and eng.cc_in is None and eng.ws_enc0 is None
    assert not eng._sparse_dw(B) and not eng._lut_fwd(B, True)
    loss, g, _ = run_single_step(eng, rows)
    assert abs(loss - rl) < 1e-5 * abs(rl), (loss, rl)
    zero_b = tuple('b%d' % i for i in range(len(hs)))
    assert_grads_close(g, rg, skip=zero_b)


def test_weight_gradient_kernels_on_random_shapes(ops):
    """Both 64-unit weight-gradient kernels on shapes drawn around their edges: splits of one to a few K steps (fewer than
    the ring kernel's five steps of lead), batches that are not a multiple of 16, gene counts across the 256 / 512 group
    boundaries, rows with many counts beyond the table (the formula path in most steps), every run twice (bit for bit)."""
    rng = np.random.RandomState(20260927)
    shapes = [(1, 1), (15, 17), (16, 512), (17, 513), (33, 255), (100, 1024), (250, 600), (1030, 40), (1500, 530), (2500, 70)]
    shapes += [(int(rng.randint(1, 1800)), int(rng.randint(1, 1400))) for _ in range(6)]
    H1 = 64
    for B, G in shapes:
        n = B + 5
        Y = counts_with_escapes(n, G, B * 7 + G, big=(B > 4 and G > 8))
        hot = rng.randint(0, G, max(1, G // 40))                 # highly expressed genes: counts 64 .. 254 in half of the rows
        for g in hot:
            rows = rng.rand(n) < 0.5
            Y[rows, g] = rng.randint(64, 255, int(rows.sum()))
        fac = rng.lognormal(0, 0.4, n).astype(np.float32).astype(np.float64)
        L = dense_input(Y, fac, True, None, None)
        mean = L.mean(0).astype(np.float32).astype(np.float64)
        std = np.maximum(L.std(0, ddof=1) if n > 1 else np.ones(G), 1e-3).astype(np.float32).astype(np.float64)
        X = dense_input(Y, fac, True, mean, std)
        perm = rng.permutation(n)[:B].astype(np.int32)
        dZ = rng.normal(0, 1e-3, (B, H1)).astype(np.float32).astype(np.float64)
        gW_ref = X[perm].T @ dZ
        Lr = L[perm] / std[None, :]
        gW_abs = np.abs(Lr).T @ np.abs(dZ) + np.abs(mean / std)[:, None] * np.abs(dZ).sum(0)[None, :] + \
            np.abs(mean / std)[:, None] * np.abs(dZ.sum(0))[None, :]
        _, cc = build_compact(ops, Y)
        cc = cc.with_input(dev(fac), True, dev(mean), dev(std), ops=ops)
        dperm = torch.as_tensor(perm).cuda()
        dcur = torch.zeros(1, dtype=torch.int64, device='cuda')
        ddZ = dev(dZ)
        for form in (1, 2):
            ws = torch.full((ops.enc0_dw_sparse_workspace_bytes(B, G, H1) // 4 + 4,), float('nan'), device='cuda')
            gW1 = torch.full((G + 1, H1), 7.0, device='cuda'); gW2 = torch.full((G + 1, H1), 3.0, device='cuda')
            ops.enc0_dw_sparse(cc, dperm, dcur, 0, B, G, H1, ddZ, H1, gW1, H1, ws, form=form)
            ops.enc0_dw_sparse(cc, dperm, dcur, 0, B, G, H1, ddZ, H1, gW2, H1, ws, form=form)
            torch.cuda.synchronize()
            got = gW1.cpu().numpy()
            err = np.abs(got[:G] - gW_ref)
            assert (err <= 1e-6 * gW_abs + 1e-30).all(), (B, G, form, float((err / np.maximum(gW_abs, 1e-30)).max()))
            np.testing.assert_allclose(got[G], dZ.sum(0), rtol=0, atol=2e-6 * np.abs(dZ).sum(0).max() + 1e-30)
            assert torch.equal(gW1, gW2), (B, G, form)
