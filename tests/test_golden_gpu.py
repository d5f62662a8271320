"""The reference's own known answers THROUGH the HIP kernels (KAT-1 / KAT-2).

data/biochemists.tsv with the maximum-likelihood fits R produced for it (data/biochemists.R:16-42:
pscl::zeroinfl -> data/biochemists-zinb-coef.tsv:8 theta = 2.65477, MASS::glm.nb ->
data/biochemists-nb-coef.tsv:8 theta = 2.26439) pins dca/loss.py:72-156 independently of TensorFlow:

    sum of ZINB.loss terms at R's ZINB fit = -logLik = 1549.9909        (KAT-1)
    sum of NB.loss terms at R's NB fit     = -logLik = 1560.9583
    the gradient w.r.t. (beta, gamma, theta) vanishes at the fit         (KAT-2)

tests/test_oracle_golden.py holds the numpy oracle to these numbers; here the same 915 observations go through
the two kernels that evaluate the likelihood in the product -- K-ZINB (dcahip_zinb_nll: pre-activations in, loss and
element gradients out) and K-HEADS (dcahip_heads_fused: design matrix x coefficients on the matrix pipe, likelihood,
weight / bias / dispersion gradients in one launch) -- as one "gene" (art), the 6-column design matrix (intercept
first) as the decoder output H, R's beta / gamma as the head weights, log-link mean (MeanAct, network.py:38), logit-link
zero inflation (sigmoid head, network.py:369) and the dispersion either as the constant-dispersion parameter
(layers.py:17-21: theta = exp(w)) or as a DispAct head (network.py:39: theta = softplus(bias)) with zero weights.
Fixture: tests/golden/biochemists.npz (generator committed beside it).
"""
import numpy as np
import pytest
import torch

from oracle import zinb_np as Z

pytestmark = pytest.mark.gpu

KAT = {'zinb': 1549.9909, 'nb': 1560.9583}


@pytest.fixture(scope='module')
def ops():
    from dca_amd.ops import HipOps
    return HipOps()


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to(dtype).cuda()


def _fit(b, model, scale_mu=1.0):
    """(y, design matrix, beta, gamma, theta) of R's fit; scale_mu moves the intercept of the count model off the MLE."""
    tab = b['table']
    y, Xd = tab[:, 0], np.c_[np.ones(len(tab)), tab[:, 1:]]
    if model == 'zinb':
        beta, gamma, theta = b['zinb_count_coef'].copy(), b['zinb_zero_coef'].copy(), float(b['zinb_theta'])
    else:
        beta, gamma, theta = b['nb_coef'].copy(), None, float(b['nb_theta'])
    beta[0] += np.log(scale_mu)
    return y, Xd, beta, gamma, theta


def _oracle_gradient(y, Xd, beta, gamma, theta):
    """fp64: the gradient w.r.t. (beta, gamma, log theta) and the sum of |terms| of each component (the scale an fp32
    sum of 915 terms is accurate against)."""
    mu = np.exp(Xd @ beta)
    th = np.full_like(mu, theta)
    if gamma is not None:
        pi = Z.sigmoid(Xd @ gamma)
        dmu, dth, dpi = Z.zinb_grads(y, mu, th, pi)
        terms = [(dmu * mu)[:, None] * Xd, (dpi * pi * (1 - pi))[:, None] * Xd, (dth * theta)[:, None]]
    else:
        dmu, dth = Z.nb_grads(y, mu, th)
        terms = [(dmu * mu)[:, None] * Xd, None, (dth * theta)[:, None]]
    return [None if t is None else t.sum(0) for t in terms], [None if t is None else np.abs(t).sum(0) for t in terms]


@pytest.mark.parametrize('model,cond_disp', [('zinb', False), ('zinb', True), ('nb', False), ('nb', True)])
def test_biochemists_through_zinb_nll_kernel(ops, biochemists, model, cond_disp):
    """K-ZINB on the 915 x 1 problem: loss sum to 1e-5 of R's -logLik; element gradients chained to (beta, gamma, theta)
    on the host in fp64 vanish at the fit and do not vanish 30 % away from it."""
    has_pi = model == 'zinb'
    flags = (1 if has_pi else 0) | (0 if cond_disp else 2)
    for scale_mu in (1.0, 1.3):
        y, Xd, beta, gamma, theta = _fit(biochemists, model, scale_mu)
        B, G, Gp = len(y), 1, 4
        lda = 3 * Gp
        A = np.zeros((B, lda))
        A[:, 0] = Xd @ beta
        A[:, Gp] = np.log(np.expm1(theta))                     # DispAct^-1: softplus(a) = theta
        if has_pi:
            A[:, 2 * Gp] = Xd @ gamma
        dA, dD = dev(A), torch.full((B, lda), 7.0, device='cuda')
        Y = np.zeros((B, Gp)); Y[:, 0] = y
        dY, dsf = dev(Y), dev(np.ones(B))
        dcur = torch.zeros(1, dtype=torch.int64, device='cuda')
        dtw = dev(np.r_[np.log(theta), 0, 0, 0])
        part = torch.zeros(ops.max_partials, dtype=torch.float64, device='cuda')
        n = ops.zinb_nll(dA[:, 0:], dA[:, Gp:] if cond_disp else None, dA[:, 2 * Gp:] if has_pi else None, lda,
                         None if cond_disp else dtw, dY, Gp, dsf, None, dcur, B, G, 0.0, 1.0, flags,
                         dD[:, 0:], dD[:, Gp:], dD[:, 2 * Gp:] if has_pi else None, lda, part)
        loss = torch.zeros(1, device='cuda')
        ops.loss_finalize(part, n, 1.0, loss)
        torch.cuda.synchronize()
        D = dD.cpu().numpy().astype(np.float64)
        g = [D[:, 0] @ Xd, D[:, 2 * Gp] @ Xd if has_pi else None]
        if cond_disp:
            g_th = (D[:, Gp] / Z.sigmoid(A[:, Gp])).sum() * theta   # d/d a_disp = d/d theta * sigmoid(a): back to d/d log theta
        else:
            g_th = D[:, Gp].sum() * theta                          # const. dispersion: d/d theta per element (chain in colsum_chain)
        ref, mag = _oracle_gradient(y, Xd, beta, gamma, theta)
        if scale_mu == 1.0:
            assert abs(loss.item() - KAT[model]) <= 1e-5 * KAT[model], (loss.item(), KAT[model])
            for got, r, m in ((g[0], ref[0], mag[0]), (g[1], ref[1], mag[1]), (g_th, ref[2], mag[2])):
                if got is None:
                    continue
                # |R's own residual| < 1e-3 (tests/test_oracle_golden.py) + fp32 element gradients: 2e-6 of sum |terms|
                assert (np.abs(got) <= 1e-3 + 2e-6 * m).all(), (model, cond_disp, got, m)
                assert np.abs(got).max() <= 1e-3 * B                # the verdict's bar, far looser
        else:
            assert np.abs(g[0]).max() > 1.0                          # the test has teeth
            np.testing.assert_allclose(g[0], ref[0], rtol=1e-4, atol=2e-6 * mag[0].max())


@pytest.mark.parametrize('model,cond_disp', [('zinb', False), ('zinb', True), ('nb', False), ('nb', True)])
def test_biochemists_through_heads_fused_kernel(ops, biochemists, model, cond_disp):
    """K-HEADS on the same problem: H = the design matrix (hL = 6), head weights = R's coefficients.  The kernel forms the
    linear predictors itself (split-bf16 products), evaluates the likelihood and returns the weight gradients =
    d(-logLik)/d(beta, gamma) directly: sum to 1e-5, gradient ~ 0 at the fit, > 1 away from it."""
    has_pi = model == 'zinb'
    flags = (1 if has_pi else 0) | (0 if cond_disp else 2)
    heads = ['mean'] + (['disp'] if cond_disp else []) + (['pi'] if has_pi else [])
    for scale_mu in (1.0, 1.3):
        y, Xd, beta, gamma, theta = _fit(biochemists, model, scale_mu)
        B, G, Gp, hL = len(y), 1, 4, Xd.shape[1]
        ldh = (hL + 3) // 4 * 4
        NH = len(heads) * Gp
        Wh = np.zeros((hL + 1, NH))
        for k, h in enumerate(heads):
            if h == 'mean':
                Wh[:hL, k * Gp] = beta
            elif h == 'pi':
                Wh[:hL, k * Gp] = gamma
            else:
                Wh[hL, k * Gp] = np.log(np.expm1(theta))           # zero weights, bias = DispAct^-1(theta)
        Hp = np.zeros((B, ldh)); Hp[:, :hL] = Xd
        Y = np.zeros((B, Gp)); Y[:, 0] = y
        dWh, dHp, dY, dsf = dev(Wh), dev(Hp), dev(Y), dev(np.ones(B))
        dcur = torch.zeros(1, dtype=torch.int64, device='cuda')
        dtw = dev(np.r_[np.log(theta), 0, 0, 0])
        gWd = torch.full((hL + 1, NH), 7.0, device='cuda')
        gth = torch.full((Gp,), 7.0, device='cuda')
        dHd = torch.full((B, ldh), 7.0, device='cuda')
        part = torch.zeros(ops.max_partials, dtype=torch.float64, device='cuda')
        nb = ops.heads_fused_workspace_bytes(B, hL, G, Gp, flags)
        assert nb > 0
        ws = torch.full((nb // 4,), float('nan'), device='cuda')
        n = ops.heads_fused(dHp, ldh, dWh, NH, dWh[hL], Gp, None if cond_disp else dtw, dY, Gp, dsf, None, dcur, B, hL, G,
                            0.0, 1.0, flags, gWd, NH, None if cond_disp else gth, dHd, ldh, part, ws)
        loss = torch.zeros(1, device='cuda')
        ops.loss_finalize(part, n, 1.0, loss)
        torch.cuda.synchronize()
        gW = gWd.cpu().numpy().astype(np.float64)
        col = {h: k * Gp for k, h in enumerate(heads)}
        g_beta = gW[:hL, col['mean']]
        g_gamma = gW[:hL, col['pi']] if has_pi else None
        if cond_disp:
            g_th = gW[hL, col['disp']] / Z.sigmoid(np.log(np.expm1(theta))) * theta     # bias gradient -> d/d log theta
        else:
            g_th = float(gth[0].item())                              # ConstantDispersionLayer chain: already d/d log theta
        ref, mag = _oracle_gradient(y, Xd, beta, gamma, theta)
        if scale_mu == 1.0:
            assert abs(loss.item() - KAT[model]) <= 1e-5 * KAT[model], (loss.item(), KAT[model])
            for got, m in ((g_beta, mag[0]), (g_gamma, mag[1]), (g_th, mag[2])):
                if got is None:
                    continue
                assert (np.abs(got) <= 1e-3 + 2e-6 * m).all(), (model, cond_disp, got, m)
                assert np.abs(got).max() <= 1e-3 * B
            # bias gradient of the mean head == the intercept component (the intercept column of H is all ones)
            assert abs(gW[hL, col['mean']] - g_beta[0]) <= 2e-6 * mag[0][0] + 1e-6
        else:
            assert np.abs(g_beta).max() > 1.0
            np.testing.assert_allclose(g_beta, ref[0], rtol=1e-4, atol=2e-6 * mag[0].max())
