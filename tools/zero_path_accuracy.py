"""fp32 emulation (numpy) of the trimmed y = 0 element path of K-HEADS (zinb_math.hpp: zinb_zero_elem) against the
fp64 oracle: loss and the three pre-activation gradients.  Hardware exp2 / log2 / rcp are modelled as correctly
rounded fp32 results of the fp64 function (1 ulp class).  Run on CPU:  python tools/zero_path_accuracy.py
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import zinb_np as Z

f32 = np.float32


def fma(a, b, c):
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def exp2(x):
    with np.errstate(over='ignore', under='ignore'):
        return np.exp2(x.astype(np.float64)).astype(f32)


def log2(x):
    return np.log2(x.astype(np.float64)).astype(f32)


def rcp(x):
    with np.errstate(divide='ignore'):
        return (1.0 / x.astype(np.float64)).astype(f32)


L2E = f32(1.44269504088896340736)
LN2 = f32(0.69314718055994531)
EPS = f32(1e-10)


def fexp_nc(x):
    t = (x * L2E).astype(f32)
    r = fma(x, np.full_like(x, L2E), -t)
    e = exp2(t)
    return (e * fma(r, np.full_like(r, LN2), np.ones_like(r))).astype(f32)        # inf / 0 stay inf / 0


def fexp_raw(x):
    return exp2((x * L2E).astype(f32))


def flog_fast(x):
    return (log2(x) * LN2).astype(f32)


def med3(x, lo, hi):
    return np.minimum(np.maximum(x, f32(lo)), f32(hi))


def zero_elem(am, ad, ap, sf, ridge):
    am, ad, ap, sf = (np.asarray(v, f32) for v in (am, ad, ap, sf))
    one = f32(1)
    e = fexp_raw(am)
    ec = med3(e, 1e-5, 1e6)
    mwin = ec == e
    mu = (ec * sf).astype(f32)
    gm = np.where(mwin, (e * sf).astype(f32), f32(0))
    # dispersion: softplus + sigmoid from one exp / one rcp / one log
    ex = fexp_raw(-np.abs(ad))
    u = (one + ex).astype(f32)
    s = rcp(u)
    d = (u - one).astype(f32)
    l1 = fma((ex - d).astype(f32), s, flog_fast(u))
    sp = (np.maximum(ad, f32(0)) + l1).astype(f32)
    theta = med3(sp, 1e-4, 1e4)
    dwin = theta == sp
    gd = np.where(dwin, np.where(ad >= 0, s, (ex * s).astype(f32)), f32(0))
    # dropout probability
    ex2 = fexp_raw(-np.abs(ap))
    s2 = rcp((one + ex2).astype(f32))
    es2 = (ex2 * s2).astype(f32)
    pi = np.where(ap >= 0, s2, es2)
    omp = np.where(ap >= 0, es2, s2)
    # zero case
    mue = (mu + EPS).astype(f32)
    den = (theta + mue).astype(f32)
    rden = rcp(den)
    t = (mue * rcp(theta)).astype(f32)
    u2 = (one + t).astype(f32)
    d2 = (u2 - one).astype(f32)
    q = (theta * rden).astype(f32)                       # 1 / (1 + t)
    logq = -fma((t - d2).astype(f32), q, flog_fast(u2))
    tl = (theta * logq).astype(f32)
    z = fexp_raw(tl)
    D = (fma(omp, z, pi) + EPS).astype(f32)
    nll = -flog_fast(D)
    invD = rcp(D)
    oz = ((omp * z).astype(f32) * invD).astype(f32)
    dmu = ((oz * theta).astype(f32) * rden).astype(f32)
    fs = -(t * t).astype(f32) * (f32(0.5) - t * (f32(2 / 3) - t * (f32(0.75) - t * (f32(0.8) - t * f32(5 / 6)))))
    fl = fma(mue, rden, logq)
    dth = (-oz * np.where(t < f32(0.03125), fs.astype(f32), fl)).astype(f32)
    x = tl
    ser = (x * fma(x, fma(x, np.full_like(x, f32(1 / 6)), np.full_like(x, f32(0.5))), np.ones_like(x))).astype(f32)
    em1 = np.where(x > f32(-0.015625), ser, (z - one).astype(f32))
    dpi = (em1 * invD).astype(f32)
    dpi = fma(np.full_like(pi, f32(2 * ridge)), pi, dpi)
    nll = fma((pi * f32(ridge)).astype(f32), pi, nll)
    return nll, (dmu * gm).astype(f32), (dth * gd).astype(f32), ((dpi * pi).astype(f32) * omp).astype(f32)


def main():
    rng = np.random.RandomState(0)
    n = 400000
    am = rng.normal(-2.0, 2.5, n); ad = rng.normal(0.5, 2.5, n); ap = rng.normal(0.0, 3.0, n)
    sf = rng.lognormal(0, 0.4, n)
    # edge block
    am[:12] = [-120, -12, -11.6, 13.7, 14, 95, 0, 0, 0, 0, 5, -5]
    ad[:12] = [0, 0, 0, 0, 0, 0, -12, -9.3, 9.21, 9500, 20, -3]
    ap[:12] = [0, 1, -1, 2, -2, 0, -30, 30, 0, 12, -12, 5]
    ridge = 0.05
    a = [v.astype(f32).astype(np.float64).reshape(1, -1) for v in (am, ad, ap)]
    sfv = sf.astype(f32).astype(np.float64)
    y = np.zeros((1, n))
    # oracle per element: treat every element as its own "cell" so that sf varies per element
    ls, lm, dm, dd, dp = Z.zinb_loss_and_grads(a[0].T, a[1].T, a[2].T, y.T, sfv, ridge, n_total=1.0)
    nll_ref = Z.zinb_nll(y.T, *Z.heads_forward(a[0].T, a[1].T, a[2].T, sfv), ridge=ridge)[:, 0] if hasattr(Z, 'zinb_nll') else None
    nll, gm_, gd_, gp_ = zero_elem(am, ad, ap, sf, ridge)
    for nm, v in (('nll', nll), ('gm', gm_), ('gd', gd_), ('gp', gp_)):
        if not np.isfinite(v).all():
            print(nm, 'non-finite at', np.where(~np.isfinite(v))[0][:10])
    for name, got, ref in (('d_mean', gm_, dm[:, 0]), ('d_disp', gd_, dd[:, 0]), ('d_pi', gp_, dp[:, 0])):
        err = np.abs(got.astype(np.float64) - ref)
        rel = err / (np.abs(ref) + 1e-30)
        scale = np.abs(ref).max()
        bad = err > (2e-4 * np.abs(ref) + 2e-6 * scale)
        big = np.abs(ref) > 1e-6 * scale
        print('%-7s max rel err (|ref| > 1e-6 max) %.3e   median %.3e   test-tolerance violations %d   max abs/scale %.3e'
              % (name, rel[big].max(), np.median(rel[big]), int(bad.sum()), (err / scale).max()))
    if nll_ref is not None:
        e = np.abs(nll.astype(np.float64) - nll_ref)
        print('nll     sum rel err %.3e   max abs %.3e  max rel %.3e' % (abs(nll.astype(np.float64).sum() - nll_ref.sum()) / nll_ref.sum(),
                                                                           e.max(), (e / np.maximum(nll_ref, 1e-3)).max()))
    else:
        print('nll sum', nll.astype(np.float64).sum(), 'oracle', ls)


if __name__ == '__main__':
    main()
