"""Oracle (test infrastructure only): NB / ZINB negative log-likelihood, the
output-head activations and the analytic gradient, restated in numpy.

Follows, line by line, the arithmetic of the reference:

* ``dca/network.py:38-39``   MeanAct = clip(exp(x), 1e-5, 1e6),
                             DispAct = clip(softplus(x), 1e-4, 1e4)
* ``dca/layers.py:85``       ColwiseMultLayer: mean * size_factor[:, None]
* ``dca/layers.py:17-21``    ConstantDispersionLayer: theta = clip(exp(w), 1e-3, 1e4)
* ``dca/loss.py:72-114``     NB.loss   (eps = 1e-10, theta = min(theta, 1e6))
* ``dca/loss.py:122-156``    ZINB.loss (where(y < 1e-8, zero_case, nb_case) + ridge*pi^2,
                             reduce_mean over all B*G elements, nan -> inf)

The gradient is what TensorFlow's autodiff produces for that graph (``where``
routes the gradient to the selected branch only; ``clip_by_value`` passes the
gradient inside the closed clip window and blocks it outside).  It is
cross-checked against torch autograd in tests/test_oracle_golden.py and pinned
by the reference's biochemists fixtures (gradient == 0 at the R MLE).

All functions are dtype-generic: pass float64 arrays for the "truth" used to
state tolerances, float32 arrays for "what an fp32 evaluation of the reference
formula gives".
"""
import numpy as np
from scipy.special import gammaln, digamma

EPS = 1e-10          # dca/loss.py:65
THETA_MAX = 1e6      # dca/loss.py:85
ZERO_THRESH = 1e-8   # dca/loss.py:138

MEAN_MIN, MEAN_MAX = 1e-5, 1e6     # dca/network.py:38
DISP_MIN, DISP_MAX = 1e-4, 1e4     # dca/network.py:39
CDISP_MIN, CDISP_MAX = 1e-3, 1e4   # dca/layers.py:21


def _c(x, like):
    return np.asarray(x, dtype=like.dtype)


def softplus(x):
    # tf.nn.softplus: log(exp(x) + 1), evaluated without overflow
    return np.logaddexp(x, _c(0, x))


def sigmoid(x):
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1 / (1 + np.exp(-x[pos]))
    e = np.exp(x[~pos])
    out[~pos] = e / (1 + e)
    return out


def mean_act(a):
    """dca/network.py:38"""
    with np.errstate(over='ignore'):
        return np.clip(np.exp(a), _c(MEAN_MIN, a), _c(MEAN_MAX, a))


def disp_act(a):
    """dca/network.py:39"""
    return np.clip(softplus(a), _c(DISP_MIN, a), _c(DISP_MAX, a))


def const_disp(w):
    """dca/layers.py:21 -- theta_exp of ConstantDispersionLayer"""
    with np.errstate(over='ignore'):
        return np.clip(np.exp(w), _c(CDISP_MIN, w), _c(CDISP_MAX, w))


def nb_nll(y, mu, theta):
    """Element-wise NB negative log-likelihood, dca/loss.py:85-88 (mean=False)."""
    eps = _c(EPS, mu)
    theta = np.minimum(theta, _c(THETA_MAX, mu))
    t1 = gammaln(theta + eps) + gammaln(y + 1) - gammaln(y + theta + eps)
    t2 = (theta + y) * np.log(1 + mu / (theta + eps)) \
        + y * (np.log(theta + eps) - np.log(mu + eps))
    return t1 + t2


def zinb_nll(y, mu, theta, pi, ridge=0.0):
    """Element-wise ZINB negative log-likelihood, dca/loss.py:130-140 (before the mean)."""
    eps = _c(EPS, mu)
    nb_case = nb_nll(y, mu, theta) - np.log(1 - pi + eps)
    theta = np.minimum(theta, _c(THETA_MAX, mu))
    zero_nb = np.power(theta / (theta + mu + eps), theta)
    zero_case = -np.log(pi + (1 - pi) * zero_nb + eps)
    res = np.where(y < ZERO_THRESH, zero_case, nb_case)
    return res + _c(ridge, mu) * np.square(pi)


def reduce_mean_nan2inf(x):
    """dca/loss.py:146-148: tf.reduce_mean then nan -> inf."""
    m = x.mean(dtype=x.dtype)
    return x.dtype.type(np.inf) if np.isnan(m) else m


# --------------------------------------------------------------------------
# gradients with respect to (mu, theta, pi), then chained to pre-activations
# --------------------------------------------------------------------------

def nb_grads(y, mu, theta):
    """d nb_nll / d(mu, theta), element-wise (autodiff of dca/loss.py:85-88)."""
    eps = _c(EPS, mu)
    thc = np.minimum(theta, _c(THETA_MAX, mu))
    tp = thc + eps
    dmu = (thc + y) / (tp + mu) - y / (mu + eps)
    dth = digamma(tp) - digamma(y + tp) + np.log(1 + mu / tp) \
        - (thc + y) * mu / (tp * (tp + mu)) + y / tp
    dth = np.where(theta > THETA_MAX, _c(0, mu), dth)   # tf.minimum blocks the grad
    return dmu, dth


def zinb_grads(y, mu, theta, pi, ridge=0.0):
    """d zinb_nll / d(mu, theta, pi), element-wise (autodiff of dca/loss.py:130-140)."""
    eps = _c(EPS, mu)
    one = _c(1, mu)
    thc = np.minimum(theta, _c(THETA_MAX, mu))
    # y > 0 branch
    dmu_nb, dth_nb = nb_grads(y, mu, theta)
    dpi_nb = one / (one - pi + eps)
    # y == 0 branch
    den = thc + mu + eps
    q = thc / den
    z = np.power(q, thc)
    D = pi + (one - pi) * z + eps
    dmu_z = (one - pi) * thc * z / (den * D)
    dth_z = -(one - pi) * z * (np.log(q) + one - q) / D
    dth_z = np.where(theta > THETA_MAX, _c(0, mu), dth_z)
    dpi_z = -(one - z) / D
    zero = y < ZERO_THRESH
    dmu = np.where(zero, dmu_z, dmu_nb)
    dth = np.where(zero, dth_z, dth_nb)
    dpi = np.where(zero, dpi_z, dpi_nb) + 2 * _c(ridge, mu) * pi
    return dmu, dth, dpi


def _act_grads(a_mean, a_disp):
    """d MeanAct / da, d DispAct / da (clip blocks the gradient outside the window)."""
    with np.errstate(over='ignore'):
        e = np.exp(a_mean)
    g_mean = np.where((e >= MEAN_MIN) & (e <= MEAN_MAX), e, _c(0, a_mean))
    g_disp = None
    if a_disp is not None:
        sp = softplus(a_disp)
        g_disp = np.where((sp >= DISP_MIN) & (sp <= DISP_MAX), sigmoid(a_disp), _c(0, a_disp))
    return g_mean, g_disp


def heads_forward(a_mean, a_disp, a_pi, sf):
    """network.py:369-381: returns (mean*sf, theta, pi) from the three head pre-activations."""
    mu = mean_act(a_mean) * sf.reshape(-1, 1).astype(a_mean.dtype)
    theta = disp_act(a_disp) if a_disp is not None else None
    pi = sigmoid(a_pi) if a_pi is not None else None
    return mu, theta, pi


def zinb_loss_and_grads(a_mean, a_disp, a_pi, y, sf, ridge=0.0, n_total=None,
                        theta_w=None):
    """Scalar loss (reduce_mean over the batch) and d loss / d pre-activation.

    a_mean, a_disp, a_pi : [B,G] pre-activations of the mean / dispersion / pi heads.
    theta_w              : [G] log-dispersion of ConstantDispersionLayer (ae_type 'zinb');
                           when given, a_disp must be None and the 4th return value is
                           d loss / d theta_w.
    n_total              : number of elements the mean divides by (B_global*G in data
                           parallel runs); defaults to a_mean.size.
    Returns loss_sum (sum of element-wise NLL, same dtype), loss_mean, d_mean, d_disp, d_pi.
    """
    dt = a_mean.dtype
    sfc = sf.reshape(-1, 1).astype(dt)
    mu = mean_act(a_mean) * sfc
    if theta_w is not None:
        assert a_disp is None
        theta = np.broadcast_to(const_disp(theta_w.astype(dt)).reshape(1, -1), a_mean.shape)
    else:
        theta = disp_act(a_disp)
    pi = sigmoid(a_pi)
    el = zinb_nll(y.astype(dt), mu, theta, pi, ridge)
    n = el.size if n_total is None else n_total
    loss_sum = el.sum(dtype=dt)
    dmu, dth, dpi = zinb_grads(y.astype(dt), mu, theta, pi, ridge)
    g_mean, g_disp = _act_grads(a_mean, a_disp)
    inv = dt.type(1.0 / n)
    d_mean = dmu * sfc * g_mean * inv
    d_pi = dpi * pi * (1 - pi) * inv
    if theta_w is not None:
        with np.errstate(over='ignore'):
            e = np.exp(theta_w.astype(dt))
        gw = np.where((e >= CDISP_MIN) & (e <= CDISP_MAX), e, _c(0, e))
        d_disp = (dth * inv).sum(axis=0, dtype=dt) * gw
    else:
        d_disp = dth * g_disp * inv
    return loss_sum, loss_sum / dt.type(n), d_mean, d_disp, d_pi


def nb_loss_and_grads(a_mean, a_disp, y, sf, n_total=None, theta_w=None):
    """NB twin of zinb_loss_and_grads (ae_types 'nb-conddisp' and 'nb'), loss.py:72-114."""
    dt = a_mean.dtype
    sfc = sf.reshape(-1, 1).astype(dt)
    mu = mean_act(a_mean) * sfc
    if theta_w is not None:
        assert a_disp is None
        theta = np.broadcast_to(const_disp(theta_w.astype(dt)).reshape(1, -1), a_mean.shape)
    else:
        theta = disp_act(a_disp)
    el = nb_nll(y.astype(dt), mu, theta)
    n = el.size if n_total is None else n_total
    loss_sum = el.sum(dtype=dt)
    dmu, dth = nb_grads(y.astype(dt), mu, theta)
    g_mean, g_disp = _act_grads(a_mean, a_disp)
    inv = dt.type(1.0 / n)
    d_mean = dmu * sfc * g_mean * inv
    if theta_w is not None:
        with np.errstate(over='ignore'):
            e = np.exp(theta_w.astype(dt))
        gw = np.where((e >= CDISP_MIN) & (e <= CDISP_MAX), e, _c(0, e))
        d_disp = (dth * inv).sum(axis=0, dtype=dt) * gw
    else:
        d_disp = dth * g_disp * inv
    return loss_sum, loss_sum / dt.type(n), d_mean, d_disp


def rows_in_parallel(fn, row_args, threads, **kw):
    """Evaluates a *_loss_and_grads function over row chunks on a thread pool (numpy / scipy ufuncs
    release the GIL).  The likelihood is element-wise and its reductions are plain sums, so the result
    is the single-call result up to the order of the loss / per-gene sums.  Used only to make
    benchmark-size comparisons (4 096 x 20 000 and larger) finish in seconds.

    row_args: positional arguments of fn that are [B, ...] arrays or None (split along axis 0);
    kw: the remaining keyword arguments; n_total must be given (the chunks must not normalise by
    their own size)."""
    from concurrent.futures import ThreadPoolExecutor
    assert kw.get('n_total') is not None
    B = next(a for a in row_args if a is not None).shape[0]
    nch = max(1, min(int(threads), B // 8 if B >= 8 else 1))
    bounds = np.linspace(0, B, nch + 1).astype(int)
    def piece(i):
        s, e = bounds[i], bounds[i + 1]
        return fn(*[None if a is None else a[s:e] for a in row_args], **kw)
    if nch == 1:
        return piece(0)
    with ThreadPoolExecutor(nch) as ex:
        parts = list(ex.map(piece, range(nch)))
    ls = sum(p[0] for p in parts)
    n = kw['n_total']
    out = [ls, ls / parts[0][0].dtype.type(n)]
    for k in range(2, len(parts[0])):
        vals = [p[k] for p in parts]
        if vals[0] is None:
            out.append(None)
        elif vals[0].ndim == 1:                    # per-gene sums (constant dispersion): add
            out.append(sum(vals))
        else:
            out.append(np.concatenate(vals, axis=0))
    return tuple(out)


# ------------------------------------------------------------------ Poisson / squared error
def poisson_nll(y, mu):
    """dca/loss.py:36-55: y_pred - y*log(y_pred + 1e-10) + lgamma(y + 1)."""
    from scipy.special import gammaln
    return mu - y * np.log(mu + _c(1e-10, mu)) + gammaln(y + 1).astype(mu.dtype)


def poisson_loss_and_grads(a_mean, y, sf, n_total=None):
    """ae_type 'poisson' (dca/network.py:233-246): MeanAct head * size factors, poisson_loss."""
    dt = a_mean.dtype
    sfc = sf.reshape(-1, 1).astype(dt)
    mu = mean_act(a_mean) * sfc
    yy = y.astype(dt)
    el = poisson_nll(yy, mu)
    n = el.size if n_total is None else n_total
    g_mean, _ = _act_grads(a_mean, None)
    d_mean = (1 - yy / (mu + _c(1e-10, mu))) * sfc * g_mean * dt.type(1.0 / n)
    ls = el.sum(dtype=dt)
    return ls, ls / dt.type(n), d_mean


def mse_loss_and_grads(a_mean, y, sf, n_total=None):
    """ae_type 'normal' (dca/network.py:143-156): linear mean head * size factors, mse_loss
    (dca/loss.py:24-27)."""
    dt = a_mean.dtype
    sfc = sf.reshape(-1, 1).astype(dt)
    diff = a_mean * sfc - y.astype(dt)
    el = np.square(diff)
    n = el.size if n_total is None else n_total
    ls = el.sum(dtype=dt)
    return ls, ls / dt.type(n), 2 * diff * sfc * dt.type(1.0 / n)
