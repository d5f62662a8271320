"""Oracle (test infrastructure only): the DCA autoencoder training path in numpy.

Restates, in fp64 (truth) or fp32, everything the reference executes between
``train()`` and ``predict()`` for ae_types zinb-conddisp / zinb / nb-conddisp / nb:

* stack Dense -> BatchNormalization(center=True, scale=False) -> relu per hidden
  layer, centre index floor(L/2)                       dca/network.py:92-141
* output heads + size-factor scaling                   dca/network.py:249-339, 366-393, 496-516
* loss                                                 dca/loss.py (via zinb_np.py)
* clipvalue + RMSprop, Keras ``fit`` loop with validation_split / shuffle,
  ReduceLROnPlateau, EarlyStopping                     dca/train.py:54-98
* predict outputs                                      dca/network.py:188-211, 395-405

Keras / TensorFlow semantics (un-vendored; documented behaviour, parity unpinned --
see oracle/__init__.py):
  Dense:       y = x W + b
  BatchNorm:   momentum .99, eps 1e-3, train = biased batch moments, moving stats updated
               with the biased variance; inference = moving stats
  RMSprop:     ms = .9 ms + .1 g^2 ; w -= lr g / (sqrt(ms) + 1e-7)  (eps OUTSIDE the sqrt: what both Keras stacks the
               reference runs on compute with momentum = 0 -- standalone keras 2.2 / 2.3, optimizers.py:
               `new_p = p - lr * g / (K.sqrt(new_a) + self.epsilon)`; tf.keras OptimizerV2 (keras >= 2.4), rmsprop.py
               _resource_apply_dense without momentum: `var - lr_t * grad / (sqrt(rms_t) + epsilon)`.  Only TF's fused
               ApplyRMSProp kernel, which OptimizerV2 takes with momentum > 0, has the epsilon inside the root.
               Pinned on torch.optim.RMSprop -- the same rule -- in tests/test_oracle_golden.py, with Adagrad /
               Adadelta / SGD; Dense + BatchNorm forward / backward / moving statistics on torch.nn.Linear +
               torch.nn.BatchNorm1d under autograd, same file, and on the torch autograd twin, torch_ref.py.)
  clipvalue:   g = clip(g, -c, c) element-wise before the update
  fit:         validation = last n - int(n*(1-split)) rows; per epoch a fresh arange is
               shuffled with the numpy global RNG; last partial batch kept; epoch loss =
               sample-weighted mean of batch losses; val loss in inference mode.
"""
import numpy as np
from . import zinb_np as Z

BN_MOMENTUM = 0.99
BN_EPS = 1e-3
AE_TYPES = ('zinb-conddisp', 'zinb', 'nb-conddisp', 'nb', 'poisson', 'normal', 'nb-shared', 'zinb-shared',
            'nb-fork', 'zinb-fork', 'zinb-elempi')
# network.py:553-760: behind the centre every head gets its own Dense(h) -> BN -> act -> dropout, all fed by the
# CENTRE output (last_hidden is not advanced there, so only the last decoder layer reaches the heads)
FORK_HEADS = {'nb-fork': ('mean', 'disp'), 'zinb-fork': ('mean', 'disp', 'pi')}


def effective_hidden(ae_type, hidden_size):
    """Layer widths as computed: *-fork = encoder .. centre, then the per-head last layers side by side."""
    hs = tuple(int(h) for h in hidden_size)
    if ae_type not in FORK_HEADS:
        return hs
    center = int(np.floor(len(hs) / 2.0))
    assert len(hs) - 1 > center, 'fork networks need a hidden layer behind the centre'
    return hs[:center + 1] + (len(FORK_HEADS[ae_type]) * hs[-1],)


SHARED_HEADS = {'nb-shared': ('disp',), 'zinb-shared': ('disp', 'pi')}     # Dense(1): network.py:343-362, 464-491


ACT_CODES = {'linear': 0, 'relu': 1, 'tanh': 2, 'sigmoid': 3, 'elu': 4, 'selu': 5, 'softplus': 6,
             'softsign': 7, 'LeakyReLU': 8, 'PReLU': 9}    # 9: keras.layers.PReLU, slopes are parameters alpha{i}
SELU_SCALE, SELU_ALPHA = 1.0507009873554805, 1.6732632423543772


def act_fwd(code, x):
    """keras.activations by name / LeakyReLU(alpha=0.3) (dca/network.py:132-135)."""
    if code == 0: return x
    if code == 1: return np.maximum(x, 0)
    if code == 2: return np.tanh(x)
    if code == 3: return Z.sigmoid(x)
    if code == 4: return np.where(x > 0, x, np.expm1(np.minimum(x, 0)))
    if code == 5: return SELU_SCALE * np.where(x > 0, x, SELU_ALPHA * np.expm1(np.minimum(x, 0)))
    if code == 6: return Z.softplus(x)
    if code == 7: return x / (1 + np.abs(x))
    if code == 8: return np.where(x > 0, x, 0.3 * x)
    raise ValueError(code)


def act_grad(code, x):
    """d act / dx as a function of the pre-activation x."""
    if code == 0: return np.ones_like(x)
    if code == 1: return (x > 0).astype(x.dtype)
    if code == 2: return 1 - np.tanh(x) ** 2
    if code == 3: s = Z.sigmoid(x); return s * (1 - s)
    if code == 4: return np.where(x > 0, 1.0, np.exp(np.minimum(x, 0)))
    if code == 5: return SELU_SCALE * np.where(x > 0, 1.0, SELU_ALPHA * np.exp(np.minimum(x, 0)))
    if code == 6: return Z.sigmoid(x)
    if code == 7: return 1 / (1 + np.abs(x)) ** 2
    if code == 8: return np.where(x > 0, 1.0, 0.3)
    raise ValueError(code)


def act_grad_from_out(code, h):
    """The same derivative expressed through the output h = act(x) (the C ABI's contract)."""
    if code == 0: return np.ones_like(h)
    if code == 1: return (h > 0).astype(h.dtype)
    if code == 2: return 1 - h * h
    if code == 3: return h * (1 - h)
    if code == 4: return np.where(h > 0, 1.0, h + 1)
    if code == 5: return np.where(h > 0, SELU_SCALE, h + SELU_SCALE * SELU_ALPHA)
    if code == 6: return -np.expm1(-h)
    if code == 7: return (1 - np.abs(h)) ** 2
    if code == 8: return np.where(h > 0, 1.0, 0.3)
    raise ValueError(code)


# ------------------------------------------------------------------ dropout generator
def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 of Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3"
    (SC'11) on uint32 arrays (broadcasting); returns the 4 output words.  Pinned on the Random123
    known-answer vectors in tests/test_dropout_cpu.py.  This is the generator of K-DROP
    (include/dcahip.h dcahip_dropout_apply); the reference's own masks come from TF's stateful RNG
    (keras Dropout, dca/network.py:98-99, 137-138) and are not reproducible."""
    c0, c1, c2, c3 = [np.asarray(c, dtype=np.uint64) & np.uint64(0xffffffff) for c in (c0, c1, c2, c3)]
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint64(int(k0) & 0xffffffff)
    k1 = np.uint64(int(k1) & 0xffffffff)
    m32 = np.uint64(0xffffffff)
    sh = np.uint64(32)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0
        p1 = np.uint64(0xCD9E8D57) * c2
        n0 = (p1 >> sh) ^ c1 ^ k0
        n1 = p1 & m32
        n2 = (p0 >> sh) ^ c3 ^ k1
        n3 = p0 & m32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & m32
        k1 = (k1 + np.uint64(0xBB67AE85)) & m32
    return c0.astype(np.uint32), c1.astype(np.uint32), c2.astype(np.uint32), c3.astype(np.uint32)


def dropout_keep(seed, step, layer, row0, B, h, rate):
    """[B, h] boolean keep mask of K-DROP: unit (r, c) uses word c % 4 of the Philox block with
    counter (group lo, group hi, step, layer), group = (row0 + r) * ceil(h/4) + c // 4, key = seed;
    kept when (word >> 8) * 2^-24 >= rate (TF: random_uniform >= rate)."""
    hq = (h + 3) // 4
    grp = (np.uint64(row0) + np.arange(B, dtype=np.uint64))[:, None] * np.uint64(hq) + np.arange(hq, dtype=np.uint64)[None, :]
    words = philox4x32_10(grp & np.uint64(0xffffffff), grp >> np.uint64(32), np.uint64(int(step) & 0xffffffff),
                          np.uint64(layer), int(seed) & 0xffffffff, (int(seed) >> 32) & 0xffffffff)
    w = np.stack(words, axis=-1).reshape(B, hq * 4)[:, :h]
    u = (w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    return u >= np.float32(rate)


def dropout_scale(rate):
    return np.float32(1.0) / (np.float32(1.0) - np.float32(rate))


INPUT_DROPOUT_LAYER = 255          # Philox counter word 3 of the input dropout; hidden layer i uses i


def glorot_uniform(rng, fan_in, fan_out, dtype):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(dtype)


def add_prelu_params(p, ae_type, hidden_size, dtype=np.float64):
    """keras.layers.PReLU behind every hidden layer: alpha of shape (units,), zeros (alpha_initializer)."""
    for i, h in enumerate(effective_hidden(ae_type, hidden_size)):
        p['alpha%d' % i] = np.zeros(h, dtype)
    return p


def init_params(ae_type, input_size, hidden_size, output_size=None, batchnorm=True,
                seed=0, dtype=np.float64):
    """Glorot-uniform kernels, zero biases / beta, moving_mean 0, moving_var 1, theta_w 0."""
    assert ae_type in AE_TYPES
    output_size = input_size if output_size is None else output_size
    rng = np.random.RandomState(seed)
    p = {}
    fan_in = input_size
    eff = effective_hidden(ae_type, hidden_size)
    nfork = len(FORK_HEADS.get(ae_type, ()))
    for i, h in enumerate(eff):
        if nfork and i == len(eff) - 1:
            p['W%d' % i] = np.concatenate([glorot_uniform(rng, fan_in, h // nfork, dtype) for _ in range(nfork)], axis=1)
        else:
            p['W%d' % i] = glorot_uniform(rng, fan_in, h, dtype)
        p['b%d' % i] = np.zeros(h, dtype)
        if batchnorm:
            p['beta%d' % i] = np.zeros(h, dtype)
            p['mm%d' % i] = np.zeros(h, dtype)
            p['mv%d' % i] = np.ones(h, dtype)
        fan_in = h
    heads = ['mean']
    if nfork:
        fan_in = fan_in // nfork                       # every head reads its own branch
    if ae_type in ('zinb-conddisp', 'nb-conddisp', 'nb-shared', 'zinb-shared', 'nb-fork', 'zinb-fork'):
        heads.append('disp')
    if ae_type.startswith('zinb'):
        heads.append('pi')
    if ae_type == 'zinb-elempi':
        heads = ['mean', 'disp']
        # ElementwiseDense (layers.py:50-82): kernel and bias of shape (units,); glorot on a 1-D shape takes
        # fan_in = fan_out = units
        p['pi_k'] = rng.uniform(-np.sqrt(3.0 / output_size), np.sqrt(3.0 / output_size), size=output_size).astype(dtype)
        p['pi_c'] = np.zeros(output_size, dtype)
    for hd in heads:
        width = 1 if hd in SHARED_HEADS.get(ae_type, ()) else output_size
        p['W_' + hd] = glorot_uniform(rng, fan_in, width, dtype)
        p['b_' + hd] = np.zeros(width, dtype)
    if ae_type in ('zinb', 'nb'):
        p['theta_w'] = np.zeros(output_size, dtype)
    return p


STATE_KEYS = ('mm', 'mv')


def is_trainable(name):
    return not name.startswith(STATE_KEYS)


class OracleAE:
    def __init__(self, ae_type, params, hidden_size, batchnorm=True, ridge=0.0, reg=(0., 0., 0., 0.),
                 activation='relu', hidden_dropout=0., input_dropout=0., dropout_seed=0):
        assert ae_type in AE_TYPES
        hd = hidden_dropout if isinstance(hidden_dropout, (list, tuple)) else [hidden_dropout] * len(hidden_size)
        assert len(hd) == len(hidden_size)                  # network.py:87-90
        self.hidden_dropout = [float(x) for x in hd]
        self.input_dropout = float(input_dropout)
        self.dropout_seed = int(dropout_seed)
        self.step = 0                                       # completed training steps (dropout counter)
        self.row0 = 0
        self.act = ACT_CODES[activation]
        self.reg = tuple(float(x) for x in reg)          # l1, l2, l1_enc, l2_enc (network.py:114-126)
        self.ae_type = ae_type
        self.p = params
        self.n_hidden = len(tuple(hidden_size))                    # as configured (regulariser stages, centre)
        self.center = int(np.floor(self.n_hidden / 2.0))           # network.py:102
        self.hidden_size = effective_hidden(ae_type, hidden_size)
        self.fork = FORK_HEADS.get(ae_type, ())
        self.hfork = int(tuple(hidden_size)[-1]) if self.fork else 0
        if self.fork:                                              # the last layer's rate applies to every branch
            self.hidden_dropout = self.hidden_dropout[:self.center + 1] + [self.hidden_dropout[-1]]
        self.batchnorm = batchnorm
        self.ridge = ridge
        self.dtype = params['W0' if 'W0' in params else 'W_mean'].dtype      # (no hidden layer: the heads read the input)
        self.row_threads = 0                                # > 1: likelihood of large batches on a thread pool
        # {layer: boolean [B, h]}: evaluate ReLU layers on a GIVEN linear piece (unit active where True) instead of the sign
        # of this evaluation's own pre-activation.  A test of an fp32 implementation at sizes where some of its millions of
        # pre-activations land within round-off of zero compares on the implementation's piece (and checks separately that
        # the two patterns differ only where the fp64 pre-activation is ~0).
        self.relu_pattern = None
        self.cache = None                                   # the last training forward's intermediates

    # ---------------------------------------------------------------- forward
    def forward(self, X, sf, training):
        p, dt = self.p, self.dtype
        H = X.astype(dt)
        cache = {'H': [H], 'xh': [], 'inv': [], 'Yb': [], 'Z': [], 'keep': {}}
        if training and self.input_dropout > 0.0:            # network.py:98-99
            keep = dropout_keep(self.dropout_seed, self.step, INPUT_DROPOUT_LAYER, self.row0, H.shape[0],
                                H.shape[1], self.input_dropout)
            H = np.where(keep, H * dt.type(dropout_scale(self.input_dropout)), dt.type(0))
            cache['H'][0] = H
        for i in range(len(self.hidden_size)):
            Zi = H @ p['W%d' % i] + p['b%d' % i]
            cache['Z'].append(Zi)
            if self.batchnorm:
                if training:
                    mu = Zi.mean(axis=0)
                    var = np.square(Zi - mu).mean(axis=0)
                    # moving = moving*momentum + batch*(1-momentum), biased variance
                    p['mm%d' % i] = p['mm%d' % i] - (p['mm%d' % i] - mu) * dt.type(1 - BN_MOMENTUM)
                    p['mv%d' % i] = p['mv%d' % i] - (p['mv%d' % i] - var) * dt.type(1 - BN_MOMENTUM)
                else:
                    mu, var = p['mm%d' % i], p['mv%d' % i]
                inv = 1 / np.sqrt(var + dt.type(BN_EPS))
                xh = (Zi - mu) * inv
                Yb = xh + p['beta%d' % i]
                cache['xh'].append(xh)
                cache['inv'].append(inv)
            else:
                Yb = Zi
            cache['Yb'].append(Yb)
            if self.act == 9:
                H = np.maximum(Yb, 0) + p['alpha%d' % i] * np.minimum(Yb, 0)
            else:
                H = act_fwd(self.act, Yb)
            if self.act == 1 and self.relu_pattern is not None and i in self.relu_pattern:
                H = np.where(self.relu_pattern[i], Yb, dt.type(0))
            if training and self.hidden_dropout[i] > 0.0:    # network.py:137-138
                keep = dropout_keep(self.dropout_seed, self.step, i, self.row0, H.shape[0], H.shape[1],
                                    self.hidden_dropout[i])
                cache['keep'][i] = keep
                H = np.where(keep, H * dt.type(dropout_scale(self.hidden_dropout[i])), dt.type(0))
            cache['H'].append(H)
        cache['a_mean'] = self._head_in(H, 'mean') @ p['W_mean'] + p['b_mean']
        if self.ae_type == 'zinb-elempi':        # network.py:438-447: minus, then ElementwiseDense -> sigmoid
            cache['a_mean'] = -cache['a_mean']
            cache['a_pi_elem'] = cache['a_mean'] * p['pi_k'] + p['pi_c']
        cache['a_disp'] = self._head_in(H, 'disp') @ p['W_disp'] + p['b_disp'] if 'W_disp' in p else None
        cache['a_pi'] = self._head_in(H, 'pi') @ p['W_pi'] + p['b_pi'] if 'W_pi' in p else cache.get('a_pi_elem')
        cache['sf'] = sf.astype(dt)
        return cache

    def _head_in(self, H, head):
        """Input of a head's Dense: the decoder output, or the head's own branch of a fork network."""
        if not self.fork:
            return H
        j = self.fork.index(head)
        return H[:, j * self.hfork:(j + 1) * self.hfork]

    def _loss_grads(self, c, Y, n_total):
        tw = self.p.get('theta_w')
        if self.ae_type == 'poisson':
            ls, lm, dm = Z.poisson_loss_and_grads(c['a_mean'], Y, c['sf'], n_total)
            return ls, lm, dm, None, None
        if self.ae_type == 'normal':
            ls, lm, dm = Z.mse_loss_and_grads(c['a_mean'], Y, c['sf'], n_total)
            return ls, lm, dm, None, None
        # shared heads: the [B, 1] Dense(1) outputs broadcast against [B, G] inside the loss
        # (loss.py:85-88, 130-140; the ridge term too); their gradient is the sum over the genes
        shape = c['a_mean'].shape
        a_disp = np.broadcast_to(c['a_disp'], shape) if c['a_disp'] is not None else None
        a_pi = np.broadcast_to(c['a_pi'], shape) if c['a_pi'] is not None else None
        if self.row_threads > 1 and shape[0] * shape[1] >= 1 << 20:
            # benchmark-size batches: the element-wise likelihood over row chunks on a thread pool
            nt = float(shape[0] * shape[1]) if n_total is None else n_total
            if self.ae_type.startswith('zinb'):
                ls, lm, dm, dd, dpi = Z.rows_in_parallel(Z.zinb_loss_and_grads, (c['a_mean'], a_disp, a_pi, Y, c['sf']),
                                                         self.row_threads, ridge=self.ridge, n_total=nt, theta_w=tw)
            else:
                ls, lm, dm, dd = Z.rows_in_parallel(Z.nb_loss_and_grads, (c['a_mean'], a_disp, Y, c['sf']),
                                                    self.row_threads, n_total=nt, theta_w=tw)
                dpi = None
        elif self.ae_type.startswith('zinb'):
            ls, lm, dm, dd, dpi = Z.zinb_loss_and_grads(c['a_mean'], a_disp, a_pi, Y,
                                                        c['sf'], self.ridge, n_total, tw)
        else:
            ls, lm, dm, dd = Z.nb_loss_and_grads(c['a_mean'], a_disp, Y, c['sf'], n_total, tw)
            dpi = None
        if c['a_disp'] is not None and c['a_disp'].shape[1] == 1 and shape[1] != 1:
            dd = dd.sum(axis=1, keepdims=True)
        if c['a_pi'] is not None and c['a_pi'].shape[1] == 1 and shape[1] != 1:
            dpi = dpi.sum(axis=1, keepdims=True)
        return ls, lm, dm, dd, dpi

    def loss_and_grads(self, X, Y, sf, n_total=None):
        """One training-mode forward + backward. Returns (mean loss, grads dict)."""
        p = self.p
        c = self.cache = self.forward(X, sf, training=True)
        _, loss, d_mean, d_disp, d_pi = self._loss_grads(c, Y, n_total)
        g = {}
        pen, greg = (0.0, {})
        if any(self.reg):
            pen, greg = reg_penalty_and_grads({k: v for k, v in p.items() if is_trainable(k)},
                                              self.n_hidden, *self.reg)
            loss = loss + pen
        HL = c['H'][-1]
        if self.ae_type == 'zinb-elempi':
            g['pi_k'] = (d_pi * c['a_mean']).sum(axis=0)
            g['pi_c'] = d_pi.sum(axis=0)
            d_mean = -(d_mean + d_pi * p['pi_k'])           # back through the affine map and the minus
        g['W_mean'] = self._head_in(HL, 'mean').T @ d_mean
        g['b_mean'] = d_mean.sum(axis=0)
        parts = {'mean': d_mean @ p['W_mean'].T}
        if 'W_disp' in p:
            g['W_disp'] = self._head_in(HL, 'disp').T @ d_disp
            g['b_disp'] = d_disp.sum(axis=0)
            parts['disp'] = d_disp @ p['W_disp'].T
        elif 'theta_w' in p:
            g['theta_w'] = d_disp
        if 'W_pi' in p:
            g['W_pi'] = self._head_in(HL, 'pi').T @ d_pi
            g['b_pi'] = d_pi.sum(axis=0)
            parts['pi'] = d_pi @ p['W_pi'].T
        # shared decoder output: the heads' input gradients add up; fork: each lands in its own branch
        dH = np.concatenate([parts[h] for h in self.fork], axis=1) if self.fork else sum(parts.values())
        for i in reversed(range(len(self.hidden_size))):
            if i in c['keep']:
                dH = np.where(c['keep'][i], dH * self.dtype.type(dropout_scale(self.hidden_dropout[i])),
                              self.dtype.type(0))
            if self.act == 9:
                g['alpha%d' % i] = (dH * np.minimum(c['Yb'][i], 0)).sum(axis=0)
                dYb = dH * np.where(c['Yb'][i] > 0, 1.0, p['alpha%d' % i])
            elif self.act == 1 and self.relu_pattern is not None and i in self.relu_pattern:
                dYb = dH * self.relu_pattern[i]
            else:
                dYb = dH * act_grad(self.act, c['Yb'][i])
            if self.batchnorm:
                xh, inv = c['xh'][i], c['inv'][i]
                g['beta%d' % i] = dYb.sum(axis=0)
                dZ = inv * (dYb - dYb.mean(axis=0) - xh * (dYb * xh).mean(axis=0))
            else:
                dZ = dYb
            g['W%d' % i] = c['H'][i].T @ dZ
            g['b%d' % i] = dZ.sum(axis=0)
            if i > 0:
                dH = dZ @ p['W%d' % i].T
        for k, v in greg.items():
            g[k] = g[k] + v
        self.step += 1
        return loss, g

    def reg_penalty(self):
        if not any(self.reg):
            return 0.0
        return reg_penalty_and_grads({k: v for k, v in self.p.items() if is_trainable(k)},
                                     self.n_hidden, *self.reg)[0]

    def eval_loss_sum(self, X, Y, sf):
        """Inference-mode sum of element-wise NLL over the given rows."""
        c = self.forward(X, sf, training=False)
        ls, _, _, _, _ = self._loss_grads(c, Y, None)
        return ls

    def predict(self, X, sf):
        """network.py:188-211, 395-405: mean*sf, theta, pi, latent (centre Dense output)."""
        c = self.forward(X, sf, training=False)
        if self.ae_type == 'normal':                             # linear mean head (network.py:146-149)
            return {'mean': c['a_mean'] * c['sf'].reshape(-1, 1), 'dispersion': None, 'dropout': None,
                    'latent': c['Z'][self.center]}
        mu, theta, pi = Z.heads_forward(c['a_mean'], c['a_disp'], c['a_pi'], c['sf'])
        if 'theta_w' in self.p:
            theta = Z.const_disp(self.p['theta_w'])          # layers.py:21, per gene
        return {'mean': mu, 'dispersion': theta, 'dropout': pi, 'latent': c['Z'][self.center]}


# ------------------------------------------------------------------ optimizer
def rmsprop_step(params, grads, ms, lr, rho=0.9, eps=1e-7, clip=5.0):
    """Keras clipvalue then tf.keras RMSprop (momentum 0, not centered), train.py:54-57."""
    for k, g in grads.items():
        dt = params[k].dtype
        if clip is not None and clip > 0:
            g = np.clip(g, dt.type(-clip), dt.type(clip))
        if k not in ms:
            ms[k] = np.zeros_like(params[k])
        ms[k] = dt.type(rho) * ms[k] + dt.type(1 - rho) * np.square(g)
        params[k] = params[k] - dt.type(lr) * g / (np.sqrt(ms[k]) + dt.type(eps))


KERAS_DEFAULT_LR = {'sgd': 0.01, 'rmsprop': 0.001, 'adagrad': 0.001, 'adadelta': 0.001, 'adam': 0.001,
                    'adamax': 0.001, 'nadam': 0.001}


def nadam_mu(t):
    return 0.9 * (1.0 - 0.5 * 0.96 ** (0.004 * t))


def optimizer_update(kind, w, g, a, b, lr, t, clip=5.0):
    """One tf.keras optimizer update with default hyper-parameters (train.py:54-57 selects the
    class by name); a, b = the optimizer's slots (None where absent); t = 1-based step."""
    if clip is not None and clip > 0:
        g = np.clip(g, -clip, clip)
    if kind == 'sgd':
        return w - lr * g, a, b
    if kind == 'rmsprop':
        a = 0.9 * a + 0.1 * g * g
        return w - lr * g / (np.sqrt(a) + 1e-7), a, b
    if kind == 'adagrad':
        a = a + g * g
        return w - lr * g / (np.sqrt(a) + 1e-7), a, b
    if kind == 'adadelta':
        a = 0.95 * a + 0.05 * g * g
        u = g * np.sqrt(b + 1e-7) / np.sqrt(a + 1e-7)
        b = 0.95 * b + 0.05 * u * u
        return w - lr * u, a, b
    if kind == 'adam':
        a = 0.9 * a + 0.1 * g
        b = 0.999 * b + 0.001 * g * g
        return w - lr * np.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t) * a / (np.sqrt(b) + 1e-7), a, b
    if kind == 'adamax':
        a = 0.9 * a + 0.1 * g
        b = np.maximum(0.999 * b, np.abs(g))
        return w - lr / (1 - 0.9 ** t) * a / (b + 1e-7), a, b
    if kind == 'nadam':
        # tf.keras optimizer_v2/nadam.py: _prepare_local (momentum schedule, cached running product)
        # and _resource_apply_dense
        mu_t, mu_t1 = nadam_mu(t), nadam_mu(t + 1)
        p_new = float(np.prod([nadam_mu(i) for i in range(1, t + 1)]))
        p_next = p_new * mu_t1
        gp = g / (1.0 - p_new)
        a = 0.9 * a + 0.1 * g
        b = 0.999 * b + 0.001 * g * g
        mbar = (1.0 - mu_t) * gp + mu_t1 * a / (1.0 - p_next)
        return w - lr * mbar / (np.sqrt(b / (1.0 - 0.999 ** t)) + 1e-7), a, b
    raise ValueError(kind)


def reg_coefs(name, n_hidden, l1, l2, l1_enc, l2_enc):
    """(l1, l2) of parameter `name` (dca/network.py:101-126: encoder-specific coefficients for the
    encoder and centre Dense kernels when non-zero; heads use l1/l2; biases / beta / theta: none)."""
    if name.startswith('W_') or name == 'pi_k':          # head kernels, ElementwiseDense kernel (network.py:443-445)
        return l1, l2
    if name[0] == 'W' and name[1:].isdigit():
        i = int(name[1:])
        enc = i <= int(np.floor(n_hidden / 2.0))
        return (l1_enc if (enc and l1_enc != 0.) else l1), (l2_enc if (enc and l2_enc != 0.) else l2)
    return 0.0, 0.0


def reg_penalty_and_grads(params, n_hidden, l1, l2, l1_enc, l2_enc):
    pen, g = 0.0, {}
    for k, v in params.items():
        a, b = reg_coefs(k, n_hidden, l1, l2, l1_enc, l2_enc)
        if a == 0. and b == 0.:
            continue
        pen += a * np.abs(v).sum() + b * np.square(v).sum()
        g[k] = a * np.sign(v) + 2 * b * v
    return pen, g


# ------------------------------------------------------------------ callbacks
class ReduceLROnPlateau:
    """keras.callbacks.ReduceLROnPlateau(monitor='val_loss', patience=p): factor .1,
    mode min, min_delta 1e-4, cooldown 0, min_lr 0 (train.py:70-72)."""

    def __init__(self, patience, factor=0.1, min_delta=1e-4, min_lr=0.0):
        self.patience, self.factor, self.min_delta, self.min_lr = patience, factor, min_delta, min_lr
        self.best, self.wait = np.inf, 0

    def on_epoch_end(self, val_loss, lr):
        if val_loss < self.best - self.min_delta:
            self.best, self.wait = val_loss, 0
        else:
            self.wait += 1
            if self.wait >= self.patience:
                if lr > self.min_lr:
                    lr = float(np.float32(max(lr * self.factor, self.min_lr)))
                self.wait = 0
        return lr


class EarlyStopping:
    """keras.callbacks.EarlyStopping(monitor='val_loss', patience=p): min_delta 0,
    restore_best_weights False (train.py:73-75)."""

    def __init__(self, patience):
        self.patience, self.best, self.wait = patience, np.inf, 0

    def on_epoch_end(self, val_loss):
        if val_loss < self.best:
            self.best, self.wait = val_loss, 0
            return False
        self.wait += 1
        return self.wait >= self.patience


def fit(net, X, Y, sf, epochs=300, batch_size=32, validation_split=0.1, learning_rate=None,
        clip_grad=5.0, reduce_lr=10, early_stop=15, shuffle_rng=np.random, val_batch_size=None,
        on_batch=None, optimizer='rmsprop'):
    """Keras Model.fit as train.py:91-98 drives it. Returns history dict (loss, val_loss, lr).

    shuffle_rng: object with .shuffle (the numpy global RNG in the reference).
    """
    n = X.shape[0]
    split_at = int(n * (1.0 - validation_split)) if validation_split else n
    Xt, Yt, sft = X[:split_at], Y[:split_at], sf[:split_at]
    Xv, Yv, sfv = X[split_at:], Y[split_at:], sf[split_at:]
    optimizer = optimizer.lower()
    lr = float(np.float32(KERAS_DEFAULT_LR[optimizer] if learning_rate is None else learning_rate))
    ms = {}
    slots, step = {}, 0
    rl = ReduceLROnPlateau(reduce_lr) if reduce_lr else None
    es = EarlyStopping(early_stop) if early_stop else None
    hist = {'loss': [], 'val_loss': [], 'lr': []}
    G = Y.shape[1]
    vb = batch_size if val_batch_size is None else val_batch_size
    for epoch in range(epochs):
        idx = np.arange(split_at)
        shuffle_rng.shuffle(idx)
        tot = 0.0
        for s in range(0, split_at, batch_size):
            b = idx[s:s + batch_size]
            loss, g = net.loss_and_grads(Xt[b], Yt[b], sft[b])
            if optimizer == 'rmsprop':
                rmsprop_step(net.p, g, ms, lr, clip=clip_grad)
            else:
                step += 1
                for k, gk in g.items():
                    if k not in slots:
                        slots[k] = (np.full_like(net.p[k], 0.1 if optimizer == 'adagrad' else 0.0),
                                    np.zeros_like(net.p[k]))
                    s1, s2 = slots[k]
                    net.p[k], s1, s2 = optimizer_update(optimizer, net.p[k], gk, s1, s2, lr, step, clip_grad)
                    slots[k] = (s1, s2)
            tot += float(loss) * len(b)
            if on_batch is not None:
                on_batch(epoch, s // batch_size, float(loss))
        hist['loss'].append(tot / split_at)
        hist['lr'].append(lr)
        if len(Xv):
            vt = 0.0
            for s in range(0, len(Xv), vb):
                e = min(s + vb, len(Xv))
                # Keras averages per-batch means weighted by batch size == sum / (nv*G)
                vt += float(net.eval_loss_sum(Xv[s:e], Yv[s:e], sfv[s:e])) / G
            val = vt / len(Xv) + net.reg_penalty()
            hist['val_loss'].append(val)
            if rl is not None:
                lr = rl.on_epoch_end(val, lr)
            if es is not None and es.on_epoch_end(val):
                break
    return hist
