"""Oracle (test infrastructure only): torch-CPU autograd twin of the training step.

Two jobs:
1. an independent check of the analytic backward in ``net_np.py`` / ``zinb_np.py``
   (the loss graph is written exactly as dca/loss.py:85-88,130-146 composes it and
   differentiated by torch autograd, like TensorFlow differentiates the reference);
2. the timed CPU baseline of ``bench.py`` (``cpu_baseline.kind == "port"``): the same
   step the reference runs through Keras/TF on CPU threads (train.py:41-59,91-98),
   on torch's CPU kernels with all host cores.  The reference itself cannot run in
   this image (tensorflow / keras / scanpy / anndata absent).
"""
import numpy as np
import torch

from .net_np import BN_EPS, BN_MOMENTUM, FORK_HEADS, effective_hidden

EPS = 1e-10


def mean_act(x):      # dca/network.py:38
    return torch.clamp(torch.exp(x), 1e-5, 1e6)


def disp_act(x):      # dca/network.py:39
    return torch.clamp(torch.nn.functional.softplus(x), 1e-4, 1e4)


def nb_nll(y, mu, theta):
    """dca/loss.py:85-88"""
    theta = torch.clamp(theta, max=1e6)
    t1 = torch.lgamma(theta + EPS) + torch.lgamma(y + 1.0) - torch.lgamma(y + theta + EPS)
    t2 = (theta + y) * torch.log(1.0 + (mu / (theta + EPS))) \
        + (y * (torch.log(theta + EPS) - torch.log(mu + EPS)))
    return t1 + t2


def zinb_nll(y, mu, theta, pi, ridge=0.0):
    """dca/loss.py:130-140"""
    nb_case = nb_nll(y, mu, theta) - torch.log(1.0 - pi + EPS)
    theta = torch.clamp(theta, max=1e6)
    zero_nb = torch.pow(theta / (theta + mu + EPS), theta)
    zero_case = -torch.log(pi + ((1.0 - pi) * zero_nb) + EPS)
    res = torch.where(y < 1e-8, zero_case, nb_case)
    return res + ridge * torch.square(pi)


class TorchAE:
    """Same parameter names as net_np.init_params."""

    def __init__(self, ae_type, params, hidden_size, batchnorm=True, ridge=0.0,
                 dtype=torch.float32):
        self.ae_type = ae_type
        self.hidden_size = effective_hidden(ae_type, hidden_size)      # *-fork: W{last} holds the branches side by side
        self.fork = FORK_HEADS.get(ae_type, ())
        self.hfork = int(tuple(hidden_size)[-1]) if self.fork else 0
        self.batchnorm = batchnorm
        self.ridge = ridge
        self.p = {}
        for k, v in params.items():
            t = torch.tensor(np.asarray(v), dtype=dtype)
            if not k.startswith(('mm', 'mv')):
                t.requires_grad_(True)
            self.p[k] = t
        self.ms = {k: torch.zeros_like(v) for k, v in self.p.items() if v.requires_grad}

    def loss(self, X, Y, sf, training=True, n_total=None):
        p = self.p
        H = X
        for i in range(len(self.hidden_size)):
            Zi = H @ p['W%d' % i] + p['b%d' % i]
            if self.batchnorm:
                if training:
                    mu = Zi.mean(0)
                    var = ((Zi - mu) ** 2).mean(0)
                    with torch.no_grad():
                        p['mm%d' % i] -= (p['mm%d' % i] - mu) * (1 - BN_MOMENTUM)
                        p['mv%d' % i] -= (p['mv%d' % i] - var) * (1 - BN_MOMENTUM)
                else:
                    mu, var = p['mm%d' % i], p['mv%d' % i]
                Zi = (Zi - mu) * torch.rsqrt(var + BN_EPS) + p['beta%d' % i]
            H = torch.relu(Zi)
        if self.fork:
            # network.py:587-612, 633-645: every head has its own last layer (a column block of the wide layer,
            # batch-norm and activation are per unit) and its Dense reads only that block
            hf = self.hfork
            br = {h: H[:, j * hf:(j + 1) * hf] for j, h in enumerate(self.fork)}
            mean = mean_act(br['mean'] @ p['W_mean'] + p['b_mean']) * sf.reshape(-1, 1)
            theta = disp_act(br['disp'] @ p['W_disp'] + p['b_disp'])
            if 'pi' in br:
                el = zinb_nll(Y, mean, theta, torch.sigmoid(br['pi'] @ p['W_pi'] + p['b_pi']), self.ridge)
            else:
                el = nb_nll(Y, mean, theta)
            return el.mean() if n_total is None else el.sum() / n_total
        if self.ae_type == 'zinb-elempi':
            # network.py:431-447: mean_no_act = -Dense(decoder); pi = sigmoid(ElementwiseDense(mean_no_act));
            # mean = MeanAct(mean_no_act)
            mean_no_act = -(H @ p['W_mean'] + p['b_mean'])
            pi = torch.sigmoid(mean_no_act * p['pi_k'] + p['pi_c'])
            mean = mean_act(mean_no_act) * sf.reshape(-1, 1)
            theta = disp_act(H @ p['W_disp'] + p['b_disp'])
            el = zinb_nll(Y, mean, theta, pi, self.ridge)
            return el.mean() if n_total is None else el.sum() / n_total
        if self.ae_type == 'normal':         # dca/network.py:146-149 + dca/loss.py:24-27
            el = torch.square((H @ p['W_mean'] + p['b_mean']) * sf.reshape(-1, 1) - Y)
            return el.mean() if n_total is None else el.sum() / n_total
        mean = mean_act(H @ p['W_mean'] + p['b_mean']) * sf.reshape(-1, 1)
        if self.ae_type == 'poisson':        # dca/network.py:236-240 + dca/loss.py:52
            el = mean - Y * torch.log(mean + EPS) + torch.lgamma(Y + 1.0)
            return el.mean() if n_total is None else el.sum() / n_total
        if 'W_disp' in p:
            theta = disp_act(H @ p['W_disp'] + p['b_disp'])
        else:
            theta = torch.clamp(torch.exp(p['theta_w']), 1e-3, 1e4).reshape(1, -1)
        if 'W_pi' in p:
            pi = torch.sigmoid(H @ p['W_pi'] + p['b_pi'])
            el = zinb_nll(Y, mean, theta, pi, self.ridge)
        else:
            el = nb_nll(Y, mean, theta)
        if n_total is None:
            return el.mean()
        return el.sum() / n_total

    def grads(self, X, Y, sf, n_total=None):
        for v in self.p.values():
            v.grad = None
        loss = self.loss(X, Y, sf, True, n_total)
        loss.backward()
        return loss.detach(), {k: v.grad for k, v in self.p.items() if v.requires_grad}

    def train_step(self, X, Y, sf, lr=1e-3, rho=0.9, eps=1e-7, clip=5.0):
        """clipvalue + Keras RMSprop (momentum 0: eps outside the sqrt, see oracle/net_np.py), train.py:54-57."""
        loss, g = self.grads(X, Y, sf)
        with torch.no_grad():
            for k, gk in g.items():
                gk = gk.clamp(-clip, clip)
                self.ms[k].mul_(rho).addcmul_(gk, gk, value=1 - rho)
                self.p[k] -= lr * gk / (torch.sqrt(self.ms[k]) + eps)
        return loss
