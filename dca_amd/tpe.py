"""Tree-structured Parzen Estimator proposals for `dca --hyper` (dca/hyper.py:97-104: fmin(..., algo=tpe.suggest)).

The reference delegates the search to hyperopt (through kopt; both un-vendored and absent here: requirement `kopt` ->
`hyperopt`), so this restates hyperopt's published algorithm -- Bergstra et al., "Algorithms for Hyper-Parameter
Optimization" (NIPS 2011) as hyperopt/tpe.py implements it, with that module's defaults: 20 random start-up trials, the
best ceil(0.25 sqrt(N)) (at most 25) trials form the "good" set, adaptive Parzen estimators with a prior component of
weight 1, linear forgetting beyond 25 observations, 24 candidates per proposal drawn from the good density l(x), the one
maximising l(x) / g(x) taken -- independently per hyper-parameter (the space of dca/hyper.py:19-43 is flat: no conditional
branches).  Host logic only (numpy); the random streams are numpy's, not hyperopt's, so individual proposals differ from
the reference's while the algorithm is the same.
"""
import math

import numpy as np

N_STARTUP_JOBS = 20
N_EI_CANDIDATES = 24
GAMMA = 0.25
PRIOR_WEIGHT = 1.0
LINEAR_FORGETTING = 25


def linear_forgetting_weights(n, lf=LINEAR_FORGETTING):
    """Weights of n observations in chronological order: the newest lf count fully, older ones ramp down to 1 / n."""
    if n == 0:
        return np.zeros(0)
    if n < lf:
        return np.ones(n)
    return np.concatenate([np.linspace(1.0 / n, 1.0, n - lf), np.ones(lf)])


def adaptive_parzen_normal(obs, prior_mu, prior_sigma, prior_weight=PRIOR_WEIGHT, lf=LINEAR_FORGETTING):
    """Gaussian mixture over the observations plus the prior: (weights, mus, sigmas), sorted by mu.  Each component's
    width is its larger gap to a neighbour, clipped to [prior_sigma / min(100, 1 + n), prior_sigma]."""
    obs = np.asarray(obs, np.float64)
    n = len(obs)
    if n == 0:
        mus, sigma, pos, order = np.array([prior_mu]), np.array([prior_sigma]), 0, np.zeros(0, int)
    elif n == 1:
        order = np.zeros(1, int)
        if prior_mu < obs[0]:
            pos, mus, sigma = 0, np.array([prior_mu, obs[0]]), np.array([prior_sigma, prior_sigma * 0.5])
        else:
            pos, mus, sigma = 1, np.array([obs[0], prior_mu]), np.array([prior_sigma * 0.5, prior_sigma])
    else:
        order = np.argsort(obs, kind='stable')
        pos = int(np.searchsorted(obs[order], prior_mu))
        mus = np.concatenate([obs[order[:pos]], [prior_mu], obs[order[pos:]]])
        sigma = np.zeros_like(mus)
        sigma[1:-1] = np.maximum(mus[1:-1] - mus[:-2], mus[2:] - mus[1:-1])
        sigma[0] = mus[1] - mus[0]
        sigma[-1] = mus[-1] - mus[-2]
    if lf and lf < n:
        w = linear_forgetting_weights(n, lf)
        weights = np.concatenate([w[order[:pos]], [prior_weight], w[order[pos:]]])
    else:
        weights = np.ones(len(mus))
        weights[pos] = prior_weight
    sigma = np.clip(sigma, prior_sigma / min(100.0, 1.0 + len(mus)), prior_sigma)
    sigma[pos] = prior_sigma
    return weights / weights.sum(), mus, sigma


def _ncdf(x):
    return 0.5 * (1.0 + np.vectorize(math.erf)(np.asarray(x, np.float64) / math.sqrt(2.0)))


def gmm_sample(rng, weights, mus, sigmas, low, high, size):
    """Draws from the mixture truncated to [low, high] (rejection, as hyperopt's GMM1)."""
    out = []
    while len(out) < size:
        k = int(np.argmax(rng.multinomial(1, weights)))
        x = rng.normal(mus[k], sigmas[k])
        if low <= x <= high:
            out.append(x)
    return np.asarray(out)


def gmm_lpdf(x, weights, mus, sigmas, low, high):
    """log density of the truncated mixture at x."""
    x = np.asarray(x, np.float64)[:, None]
    p_accept = np.sum(weights * (_ncdf((high - mus) / sigmas) - _ncdf((low - mus) / sigmas)))
    z = (x - mus) / sigmas
    comp = np.log(weights) - 0.5 * z * z - np.log(np.sqrt(2.0 * np.pi) * sigmas)
    m = comp.max(axis=1, keepdims=True)
    return (m[:, 0] + np.log(np.exp(comp - m).sum(axis=1))) - np.log(p_accept)


def categorical_posterior(obs, upper, prior_weight=PRIOR_WEIGHT, lf=LINEAR_FORGETTING):
    w = linear_forgetting_weights(len(obs), lf)
    counts = np.bincount(np.asarray(obs, int), minlength=upper, weights=w) if len(obs) else np.zeros(upper)
    pseudo = counts + prior_weight
    return pseudo / pseudo.sum()


class TPE:
    """space: ordered {name: ('choice', n) | ('uniform', low, high) | ('loguniform', low, high)} (bounds in natural units).
    A proposal is {name: index | value}; history entries are (proposal, loss) in trial order, failed trials with loss None."""

    def __init__(self, space, seed=42, n_startup=N_STARTUP_JOBS, n_ei=N_EI_CANDIDATES, gamma=GAMMA):
        self.space, self.rng = dict(space), np.random.RandomState(seed)
        self.n_startup, self.n_ei, self.gamma = n_startup, n_ei, gamma

    def random(self):
        out = {}
        for name, spec in self.space.items():
            if spec[0] == 'choice':
                out[name] = int(self.rng.randint(spec[1]))
            elif spec[0] == 'uniform':
                out[name] = float(self.rng.uniform(spec[1], spec[2]))
            else:
                out[name] = float(np.exp(self.rng.uniform(np.log(spec[1]), np.log(spec[2]))))
        return out

    def split(self, history):
        """(good, bad): the proposals of the finished trials, each group in trial order (hyperopt's ap_split_trials)."""
        ok = [(t, p, l) for t, (p, l) in enumerate(history) if l is not None and np.isfinite(l)]
        n_below = min(int(np.ceil(self.gamma * np.sqrt(len(ok)))), LINEAR_FORGETTING)
        by_loss = sorted(ok, key=lambda r: (r[2], r[0]))
        below = sorted(by_loss[:n_below], key=lambda r: r[0])
        above = sorted(by_loss[n_below:], key=lambda r: r[0])
        return [p for _, p, _ in below], [p for _, p, _ in above]

    def suggest(self, history):
        # hyperopt's start-up phase counts every trial on record, failed ones included (tpe.suggest: len(trials.trials) <
        # n_startup_jobs); the posteriors below then use the successful ones -- with none of them yet, stay random
        n_ok = sum(1 for _, l in history if l is not None and np.isfinite(l))
        if len(history) < self.n_startup or n_ok == 0:
            return self.random()
        below, above = self.split(history)
        out = {}
        for name, spec in self.space.items():
            b = [p[name] for p in below]
            a = [p[name] for p in above]
            if spec[0] == 'choice':
                pb, pa = categorical_posterior(b, spec[1]), categorical_posterior(a, spec[1])
                cand = self.rng.choice(spec[1], size=self.n_ei, p=pb)
                score = np.log(pb[cand]) - np.log(pa[cand])
                out[name] = int(cand[int(np.argmax(score))])
                continue
            if spec[0] == 'uniform':
                low, high, f, g = spec[1], spec[2], (lambda v: np.asarray(v, np.float64)), (lambda v: v)
            else:                           # loguniform: the same estimator on log(x); the Jacobian cancels in l / g
                low, high, f, g = np.log(spec[1]), np.log(spec[2]), (lambda v: np.log(np.asarray(v, np.float64))), np.exp
            prior_mu, prior_sigma = 0.5 * (low + high), high - low
            wb, mb, sb = adaptive_parzen_normal(f(b), prior_mu, prior_sigma)
            wa, ma, sa = adaptive_parzen_normal(f(a), prior_mu, prior_sigma)
            cand = gmm_sample(self.rng, wb, mb, sb, low, high, self.n_ei)
            score = gmm_lpdf(cand, wb, mb, sb, low, high) - gmm_lpdf(cand, wa, ma, sa, low, high)
            out[name] = float(g(cand[int(np.argmax(score))]))
        return out
