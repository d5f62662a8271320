"""Device-resident training / inference engine of the DCA autoencoder (MI355X-first).

Replaces the Keras model + session of the reference (dca/network.py:92-141, 366-393 build the
graph; dca/train.py:54-98 compiles and fits it) with:

* ONE flat fp32 parameter buffer (weights, biases, BN beta, per-gene log-dispersion) plus flat
  gradient and RMSprop-slot buffers of the same layout -- the optimizer is a single launch and
  the data-parallel exchange a single RCCL all-reduce bucket (the batch loss rides in its last
  slot);
* the count matrices resident in HBM ([n, ld] fp32, ld padded to 4): the shuffled minibatch is
  never materialised -- kernels gather rows through a device permutation + device cursor, so a
  step contains no host synchronisation and can be captured in a hipGraph;
* hand-written HIP kernels for everything (ops.HipOps -> libdcahip.so).  No torch.nn, no
  autograd, no CPU fallback.

Layout of the three head matrices: one [h_L, nheads*Gp] weight block ([mean | dispersion | pi],
Gp = G rounded up to 4) so the heads are ONE GEMM forward and TWO backward.
"""
import math

import numpy as np
import torch

from . import prep as _prep

BN_MOMENTUM = 0.99   # keras BatchNormalization defaults (network.py:127-128)
BN_EPS = 1e-3
RMS_RHO = 0.9        # tf.keras RMSprop defaults (train.py:54-57)
RMS_EPS = 1e-7

AE_HEADS = {                      # ae_type -> (heads in the fused block, const dispersion?)
    'zinb-conddisp': (('mean', 'disp', 'pi'), False),   # network.py:366-393
    'zinb': (('mean', 'pi'), True),                      # network.py:496-516
    'nb-conddisp': (('mean', 'disp'), False),            # network.py:293-318
    'nb': (('mean',), True),                             # network.py:249-270
    'poisson': (('mean',), False),                       # network.py:233-246 (poisson_loss)
    'normal': (('mean',), False),                        # network.py:143-156 (mse_loss, linear mean)
    # Dense(1) dispersion (and dropout) broadcast over the genes: network.py:343-362, 464-491
    'nb-shared': (('mean', 'disp'), False, ('disp',)),
    'zinb-shared': (('mean', 'disp', 'pi'), False, ('disp', 'pi')),
    # one private last decoder layer per head: network.py:553-661 (zinb-fork), 664-760 (nb-fork)
    'nb-fork': (('mean', 'disp'), False, ()),
    'zinb-fork': (('mean', 'disp', 'pi'), False, ()),
}
AE_FORK = ('nb-fork', 'zinb-fork')
# network.py:424-461: mean head negated, dropout logit = ElementwiseDense of it (per-gene scale k and offset c)
AE_HEADS['zinb-elempi'] = (('mean', 'disp', 'pi'), False, ())
AE_ELEMPI = ('zinb-elempi',)
ACT_CODES = {'linear': 0, 'relu': 1, 'tanh': 2, 'sigmoid': 3, 'elu': 4, 'selu': 5, 'softplus': 6,
             'softsign': 7, 'LeakyReLU': 8, 'PReLU': 9}      # 9: own element-wise layer with trainable slopes
INPUT_DROPOUT_LAYER = 255     # Philox counter word 3 of the input dropout (hidden layer i uses i)
AE_LOSS_FLAG = {'poisson': 4, 'normal': 8}              # DCAHIP_NLL_POISSON / DCAHIP_NLL_MSE


def _r4(x):
    return (x + 3) // 4 * 4


def _r16(x):
    return (x + 15) // 16 * 16


class _NullSection:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_NULL = _NullSection()


class EventProfiler:
    """Per-kernel timing with HIP events on the launch stream (bench.py's live roofline leg)."""

    class _Sec:
        def __init__(self, prof, name):
            self.prof, self.name = prof, name

        def __enter__(self):
            self.s = torch.cuda.Event(enable_timing=True)
            self.e = torch.cuda.Event(enable_timing=True)
            self.s.record()
            return self

        def __exit__(self, *a):
            self.e.record()
            self.prof.ev.setdefault(self.name, []).append((self.s, self.e))
            return False

    def __init__(self):
        self.ev = {}

    def section(self, name):
        return EventProfiler._Sec(self, name)

    def reset(self):
        self.ev = {}

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for k, lst in self.ev.items():
            ms = [s.elapsed_time(e) for s, e in lst]
            out[k] = {'count': len(ms), 'mean_ms': float(np.mean(ms)), 'median_ms': float(np.median(ms)), 'total_ms': float(np.sum(ms))}
        return out


class SingleProcess:
    """Communication stub for one GPU."""
    world, rank = 1, 0
    dp = False            # no communicator: the single-GPU kernel choices (fused BatchNorm, graphs) apply

    def all_reduce_sum(self, t):
        return t

    def all_gather(self, t):
        return t.unsqueeze(0)

    def all_reduce_sum_async(self, t):
        return None

    def wait(self, work):
        pass


def _truncated_normal(rng, stddev, shape):
    """TF truncated_normal: values beyond two standard deviations are redrawn."""
    out = rng.normal(0.0, 1.0, size=shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.normal(0.0, 1.0, size=int(bad.sum()))
        bad = np.abs(out) > 2.0
    return out * stddev


KERAS_INITIALIZERS = ('glorot_uniform', 'glorot_normal', 'he_uniform', 'he_normal', 'lecun_uniform', 'lecun_normal',
                      'random_uniform', 'random_normal', 'truncated_normal', 'orthogonal', 'zeros', 'ones')


def keras_initializer(name, rng, fan_in, fan_out, shape=None):
    """[fan_in, fan_out] float32 kernel from a Keras initialiser name (tf.keras 2.x defaults:
    VarianceScaling(scale, mode, distribution) with the 0.8796 truncation correction for the *_normal
    family; RandomUniform +-0.05; RandomNormal / TruncatedNormal stddev 0.05; Orthogonal gain 1)."""
    key = name.lower() if isinstance(name, str) else name
    vs = {'glorot_uniform': (1.0, 'avg', 'u'), 'glorot_normal': (1.0, 'avg', 'n'),
          'he_uniform': (2.0, 'in', 'u'), 'he_normal': (2.0, 'in', 'n'),
          'lecun_uniform': (1.0, 'in', 'u'), 'lecun_normal': (1.0, 'in', 'n')}
    shape = (fan_in, fan_out) if shape is None else shape
    if key in vs:
        scale, mode, distr = vs[key]
        n = (fan_in + fan_out) / 2.0 if mode == 'avg' else float(fan_in)
        if distr == 'u':
            lim = math.sqrt(3.0 * scale / n)
            w = rng.uniform(-lim, lim, size=shape)
        else:
            w = _truncated_normal(rng, math.sqrt(scale / n) / .87962566103423978, shape)
    elif key == 'random_uniform':
        w = rng.uniform(-0.05, 0.05, size=shape)
    elif key == 'random_normal':
        w = rng.normal(0.0, 0.05, size=shape)
    elif key == 'truncated_normal':
        w = _truncated_normal(rng, 0.05, shape)
    elif key == 'orthogonal' and len(shape) == 2:
        a = rng.normal(0.0, 1.0, size=(max(shape), min(shape)))
        q, r = np.linalg.qr(a)
        q = q * np.sign(np.diag(r))
        w = q if fan_in >= fan_out else q.T
    elif key == 'zeros':
        w = np.zeros(shape)
    elif key == 'ones':
        w = np.ones(shape)
    else:
        raise NotImplementedError('init=%r is not implemented on the MI355X path (available: %s)'
                                  % (name, ', '.join(KERAS_INITIALIZERS)))
    return np.ascontiguousarray(w, dtype=np.float32)


class ParamLayout:
    """Offsets of every tensor in the flat parameter / gradient / RMSprop buffers."""

    def __init__(self, ae_type, input_size, output_size, hidden_size, batchnorm, prelu=False):
        self.ae_type = ae_type
        self.prelu = bool(prelu)
        self.heads, self.const_disp = AE_HEADS[ae_type][:2]
        self.shared = AE_HEADS[ae_type][2] if len(AE_HEADS[ae_type]) > 2 else ()
        self.elempi = ae_type in AE_ELEMPI
        # heads that are Dense layers off the decoder output / heads with one Dense unit per gene
        self.dense_heads = tuple(h for h in self.heads if not (self.elempi and h == 'pi'))
        self.planes = tuple(h for h in self.dense_heads if h not in self.shared)
        self.G_in, self.G_out = input_size, output_size
        self.Gp = _r4(output_size)
        # columns of the heads' Dense block: the full-width heads, then 4 columns holding the Dense(1)
        # heads; A / D rows additionally carry one broadcast plane per shared head behind them
        self.NH = len(self.planes) * self.Gp + (4 if self.shared else 0)
        self.ldA = self.NH + (len(self.shared) + (1 if self.elempi else 0)) * self.Gp
        self.hidden = tuple(int(h) for h in hidden_size)
        self.center = int(np.floor(len(self.hidden) / 2.0))         # network.py:102
        # *-fork: every layer behind the centre is built once per head FROM THE CENTRE OUTPUT
        # (network.py:587-612 never advances last_hidden there), so only the last decoder layer reaches the
        # heads.  The per-head Dense(h) layers side by side are one Dense(nheads * h) (batch-norm,
        # activation and dropout act per unit); head j reads columns [j h, (j+1) h) of its output.
        self.fork = len(self.heads) if ae_type in AE_FORK else 0
        self.hfork = 0
        if self.fork:
            if len(self.hidden) - 1 <= self.center:
                raise ValueError('%s needs at least one hidden layer behind the centre (the reference fails with '
                                 'AttributeError: last_hidden_mean)' % ae_type)
            self.hfork = self.hidden[-1]
            self.hidden = self.hidden[:self.center + 1] + (self.fork * self.hfork,)
        self.batchnorm = batchnorm
        self.seg = {}
        off = 0

        def add(name, shape, align=False):
            nonlocal off
            if align:
                off = _r4(off)
            self.seg[name] = (off, shape)
            off += int(np.prod(shape))

        fan_in = input_size
        for i, h in enumerate(self.hidden):
            add('W%d' % i, (fan_in, h), align=True)     # Dense kernel [in, out]
            add('b%d' % i, (h,))                        # bias right behind it: [in+1, out] block
            if batchnorm:
                add('beta%d' % i, (h,))
            if self.prelu:
                add('alpha%d' % i, (h,))                # keras.layers.PReLU slopes (zeros)
            fan_in = h
        self.hL = self.hfork if self.fork else fan_in   # rows of the heads' Dense block (inputs per head)
        add('Wh', (self.hL, self.NH), align=True)
        add('bh', (self.NH,))
        if self.const_disp:
            add('theta_w', (self.Gp,), align=True)
        if self.elempi:
            add('pi_k', (self.Gp,), align=True)         # ElementwiseDense kernel / bias (layers.py:58-68)
            add('pi_c', (self.Gp,), align=True)
        self.P = _r4(off)
        self.total = self.P + 4                         # [P] carries the batch loss

    def view(self, flat, name):
        off, shape = self.seg[name]
        return flat[off:off + int(np.prod(shape))].view(*shape)

    def head_cols(self, head):
        if head in self.shared:
            c = len(self.planes) * self.Gp + self.shared.index(head)
            return c, c + 1
        k = self.planes.index(head)
        return k * self.Gp, k * self.Gp + self.G_out

    def plane_offset(self, head):
        """Column of the head's [B, G] plane inside a row of A / D."""
        if head in self.shared:
            return self.NH + self.shared.index(head) * self.Gp
        if head not in self.planes:                     # the element-wise dropout logit of zinb-elempi
            return self.NH
        return self.planes.index(head) * self.Gp


class Engine:
    def __init__(self, ae_type, input_size, output_size=None, hidden_size=(64, 32, 64),
                 batchnorm=True, ridge=0.0, ops=None, comm=None, device=None, activation='relu',
                 hidden_dropout=0., input_dropout=0., dropout_seed=0, sharedpi=False, config=None):
        if ae_type not in AE_HEADS:
            raise NotImplementedError('ae_type %r is not available on the MI355X path yet '
                                      '(supported: %s)' % (ae_type, ', '.join(AE_HEADS)))
        if ops is None:
            from .ops import HipOps
            ops = HipOps()                              # raises without the HIP library / a GPU
        self.ops = ops
        self.comm = comm or SingleProcess()
        from . import config as _config
        self.cfg = config if config is not None else _config.current()      # every path decision below reads this
        self.dev = torch.device(device) if device is not None else (
            torch.device('cuda', torch.cuda.current_device()) if ops.device_type == 'cuda'
            else torch.device('cpu'))
        output_size = input_size if output_size is None else output_size
        self.lay = ParamLayout(ae_type, input_size, output_size, hidden_size, batchnorm, prelu=(activation == 'PReLU'))
        self.ridge = float(ridge)
        if activation not in ACT_CODES:
            raise NotImplementedError('activation %r is not available on the MI355X path (supported: %s)'
                                      % (activation, ', '.join(ACT_CODES)))
        self.act = ACT_CODES[activation]       # Activation(self.activation), network.py:132-135
        self.prelu = activation == 'PReLU'
        if self.prelu:                         # the BN / bias kernels stay linear, PReLU is its own layer behind them
            self.act = 0
        lay = self.lay
        self.has_pi = 'pi' in lay.heads
        self.flags = (1 if self.has_pi else 0) | (2 if lay.const_disp else 0) | AE_LOSS_FLAG.get(ae_type, 0)
        self.center = lay.center
        f32 = dict(dtype=torch.float32, device=self.dev)
        # flat buffers, padded so that every rank's shard of the sharded-optimizer exchange (below) starts 16-byte aligned
        pad = 4 * self.comm.world
        self.flat_len = (lay.total + pad - 1) // pad * pad
        self.w = torch.zeros(self.flat_len, **f32)
        self.g = torch.zeros(self.flat_len, **f32)
        self.ms = torch.zeros(self.flat_len, **f32)
        # data parallel, RMSprop: reduce-scatter of the gradient -> every rank clips and updates ITS shard of the
        # parameters (and keeps only that shard of the RMSprop slots current) -> all-gather of the parameters, instead of
        # all-reducing the whole gradient and updating everything everywhere.  Same bytes on the wire
        # (2 (N-1)/N P either way), 1/N of the optimizer traffic per rank; EngineConfig.dp_sharded_opt switches it on.
        self.sharded_opt = self.comm.dp and self.cfg.dp_sharded_opt
        if self.comm.dp and self.cfg.dp_peer_exchange and batchnorm and hasattr(self.comm, 'enable_peer_exchange') \
                and self.dev.type == 'cuda' and self.lay.hidden:
            # (collective: every rank builds its engine with the same configuration)
            self.comm.enable_peer_exchange(2 * max(self.lay.hidden))
        self.mm = [torch.zeros(h, **f32) for h in lay.hidden] if batchnorm else []
        self.mv = [torch.ones(h, **f32) for h in lay.hidden] if batchnorm else []
        self.lr = torch.full((1,), 1e-3, **f32)
        self.cursor = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.acc = torch.zeros(2, dtype=torch.float64, device=self.dev)   # [train, val] sums
        self.partials = torch.zeros(ops.max_partials, dtype=torch.float64, device=self.dev)
        self.val_loss_tmp = torch.zeros(1, **f32)
        self.clip = 5.0                                   # train.py:37 clip_grad
        self.ldD = lay.ldA + (lay.Gp if lay.const_disp else 0)
        self.Bmax = 0
        self._hl = None             # (tensor, leading dimension) the heads read: the last hidden layer's output, or the input batch
        self.X = self.Y = self.sf = self.perm = None
        self.tile_order = None
        self.hist = None
        self.prof = None            # EventProfiler or None
        # K-HEADS (heads forward + NLL + both backward products in one kernel) whenever the
        # library supports the shape (tests compare the two paths by setting this attribute)
        self.use_fused = not (lay.shared or lay.fork or lay.elempi)
        self.sharedpi = bool(sharedpi) and lay.elempi       # ElementwiseDense(1): one scale / offset for all genes
        self.ws_elempi = torch.zeros(ops.elempi_workspace_doubles(lay.G_out), dtype=torch.float64, device=self.dev) \
            if lay.elempi else None
        self.ws_heads = None
        self._pending = None        # in-flight all-reduce of the heads bucket (data parallel)
        self.W0T = None             # transposed first-layer kernel (throughput batches)
        self.WhT = self.HT = self.XT = self.dZT = None     # wide networks: transposed head weights / last activations / minibatch / dZ0
        self.pl = None              # wide networks: operands as bf16 planes (reserve)
        self.opt_kind = 'rmsprop'   # train.py:54-57 picks the Keras optimizer by name
        self.slot2 = None
        self.m_sched = None         # Nadam: running product of the momentum schedule
        self.opt_iter = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.reg = None             # l1 / l2 kernel regularisers (network.py:114-126)
        self.reg_ws = None
        self._counts_world_key = None
        self._counts_local, self._counts_world = {}, {}
        # Dropout (network.py:98-99, 137-138): rates per hidden layer + input; masks are a function of
        # (seed, step counter in device memory, layer, global batch row, unit) -- K-DROP
        n_conf = len(tuple(hidden_size))                    # rates come per configured layer (network.py:87-90)
        hd = list(hidden_dropout) if isinstance(hidden_dropout, (list, tuple)) else [hidden_dropout] * n_conf
        assert len(hd) == n_conf
        if lay.fork:                                        # the last layer's rate applies to every branch
            hd = hd[:lay.center + 1] + [hd[-1]]
        self.drop = [float(r) for r in hd]
        self.in_drop = float(input_dropout)
        assert all(0.0 <= r < 1.0 for r in self.drop + [self.in_drop]), 'dropout rates must be in [0, 1)'
        self.has_dropout = any(r > 0.0 for r in self.drop) or self.in_drop > 0.0
        self.drop_seed = int(dropout_seed) & (2 ** 64 - 1)
        self.drop_iter = torch.zeros(1, dtype=torch.int64, device=self.dev)
        self.row0 = 0               # this rank's first row inside the global batch
        self.Xb = None
        # compact counts (dca_amd/compact.py): K-HEADS reads its targets from the byte store (cc); with the input
        # normalisation known (cc_in) both first-layer products are built from it on the matrix pipe (K-SPARSE)
        self.cc = self.cc_in = None
        self.ws_enc0 = self.ws_enc0l = None
        self.ws_stack = None
        # K-STACK at throughput batches: 'steps' = one launch per batch-wide dependency (9 launches instead of 22 for
        # the 64-32-64 stack), 'coop' = one cooperative launch per direction with grid barriers, 'off' = one launch per
        # operation.  Measured on the MI355X at 4096 rows (profiles/r03_stack_*): see DESIGN.md.
        self.stack_mode = self.cfg.stack
        # batch rows from which the byte-store kernels replace the dense first-layer GEMMs -- measured on the MI355X at
        # the benchmark shape (profiles/r03*_enc0_*): the weight gradient on the matrix pipe from the byte store ties the
        # dense TN GEMM at 4096 rows on the 68 579-cell matrix (0.149-0.154 vs 0.157 ms) and wins on a cache-resident one
        # (0.112 vs 0.150); the round-2 forward over the non-zero counts only (gathers of W0 rows from L2) lost to the dense
        # NT GEMM at every batch (0.160 vs 0.107 ms at 4096 rows): an experiment build of the library, not an engine path
        self.sparse_dw_min = self.cfg.sparse_dw_min

    def _t(self, name):
        return self.prof.section(name) if self.prof is not None else _NULL

    @property
    def _dhl(self):
        """Where the heads' backward writes the gradient of their input (a sink when the input is the data itself)."""
        return self.dH[-1] if self.lay.hidden else self.dHin0

    # ------------------------------------------------------------------ parameters
    def init_params(self, seed=0, init='glorot_uniform'):
        """Dense kernels from the named Keras initialiser (network.py:57 default glorot_uniform; every
        Dense of network.py:124-126, 369-380 gets kernel_initializer=self.init), zero biases / beta /
        log-dispersion.  The stream is numpy RandomState(seed): TensorFlow's initialiser stream is not
        reproducible outside TensorFlow, the distributions are (tf.keras 2.x VarianceScaling et al.)."""
        rng = np.random.RandomState(seed)
        lay = self.lay
        p = {}
        fan_in = lay.G_in
        for i, h in enumerate(lay.hidden):
            if lay.fork and i == len(lay.hidden) - 1:       # one Dense(hfork) per head, side by side
                p['W%d' % i] = np.concatenate([keras_initializer(init, rng, fan_in, lay.hfork)
                                               for _ in range(lay.fork)], axis=1)
            else:
                p['W%d' % i] = keras_initializer(init, rng, fan_in, h)
            fan_in = h
        for hd in lay.dense_heads:
            p['W_' + hd] = keras_initializer(init, rng, lay.hL, 1 if hd in lay.shared else lay.G_out)
        if lay.elempi:      # 1-D kernel of shape (units,): Keras takes fan_in = fan_out = units
            units = 1 if self.sharedpi else lay.G_out
            k = keras_initializer(init, rng, units, units, shape=(units,))
            p['pi_k'] = np.full(lay.G_out, k[0], np.float32) if self.sharedpi else k
        self.set_params(p)

    def set_params(self, p):
        """Loads named numpy arrays (names as oracle.net_np.init_params); missing ones keep
        their value."""
        lay = self.lay
        w = self.w.cpu()
        for name, (off, shape) in lay.seg.items():
            if name in ('Wh', 'bh', 'theta_w', 'pi_k', 'pi_c'):
                continue
            if name in p:
                w[off:off + int(np.prod(shape))] = torch.as_tensor(
                    np.asarray(p[name], dtype=np.float32).reshape(-1))
        Wh = lay.view(w, 'Wh'); bh = lay.view(w, 'bh')
        for hd in lay.dense_heads:
            c0, c1 = lay.head_cols(hd)
            if 'W_' + hd in p:
                Wh[:, c0:c1] = torch.as_tensor(np.asarray(p['W_' + hd], dtype=np.float32))
            if 'b_' + hd in p:
                bh[c0:c1] = torch.as_tensor(np.asarray(p['b_' + hd], dtype=np.float32))
        for name in ('theta_w', 'pi_k', 'pi_c'):         # per-gene vectors, padded to Gp
            if name in lay.seg and name in p:
                lay.view(w, name)[:lay.G_out] = torch.as_tensor(np.asarray(p[name], np.float32))
        self.w.copy_(w)
        for i in range(len(self.mm)):
            if 'mm%d' % i in p:
                self.mm[i].copy_(torch.as_tensor(np.asarray(p['mm%d' % i], np.float32)))
            if 'mv%d' % i in p:
                self.mv[i].copy_(torch.as_tensor(np.asarray(p['mv%d' % i], np.float32)))

    def _named(self, flat):
        lay = self.lay
        f = flat.detach().cpu()
        out = {}
        for name in lay.seg:
            if name in ('Wh', 'bh', 'theta_w', 'pi_k', 'pi_c'):
                continue
            out[name] = lay.view(f, name).numpy().copy()
        Wh = lay.view(f, 'Wh'); bh = lay.view(f, 'bh')
        for hd in lay.dense_heads:
            c0, c1 = lay.head_cols(hd)
            out['W_' + hd] = Wh[:, c0:c1].numpy().copy()
            out['b_' + hd] = bh[c0:c1].numpy().copy()
        for name in ('theta_w', 'pi_k', 'pi_c'):
            if name in lay.seg:
                out[name] = lay.view(f, name)[:lay.G_out].numpy().copy()
        return out

    def get_params(self):
        out = self._named(self.w)
        for i in range(len(self.mm)):
            out['mm%d' % i] = self.mm[i].cpu().numpy().copy()
            out['mv%d' % i] = self.mv[i].cpu().numpy().copy()
        return out

    def get_grads(self):
        return self._named(self.g)

    def set_optimizer(self, name):
        """Keras optimizer by name with its default hyper-parameters; resets the slots."""
        name = name.lower()
        if name not in ('sgd', 'rmsprop', 'adagrad', 'adadelta', 'adam', 'adamax', 'nadam'):
            raise NotImplementedError('optimizer %r is not implemented on the MI355X path (available: SGD, '
                                      'RMSprop, Adagrad, Adadelta, Adam, Adamax, Nadam)' % name)
        self.opt_kind = name
        self.ms.fill_(0.1 if name == 'adagrad' else 0.0)       # tf.keras initial_accumulator_value
        self.slot2 = torch.zeros_like(self.ms) if name in ('adadelta', 'adam', 'adamax', 'nadam') else None
        self.m_sched = torch.ones(1, dtype=torch.float32, device=self.dev) if name == 'nadam' else None
        self.opt_iter.zero_()

    def set_regularizers(self, l1=0., l2=0., l1_enc=0., l2_enc=0.):
        """kernel_regularizer=l1_l2(..) of every Dense kernel (network.py:114-126, 144-146, 369-380):
        encoder / centre layers take the *_enc coefficients when those are non-zero."""
        lay = self.lay
        center = lay.center
        segs = []
        for i in range(len(lay.hidden)):
            a = l1_enc if (i <= center and l1_enc != 0.) else l1
            b = l2_enc if (i <= center and l2_enc != 0.) else l2
            off, shape = lay.seg['W%d' % i]
            segs.append((off, off + int(np.prod(shape)), a, b))
        off, shape = lay.seg['Wh']
        segs.append((off, off + int(np.prod(shape)), l1, l2))
        if lay.elempi:               # ElementwiseDense kernel_regularizer (network.py:443-445)
            if self.sharedpi and (l1 != 0. or l2 != 0.):
                raise NotImplementedError('l1 / l2 on the shared ElementwiseDense kernel (sharedpi=True)')
            off, _ = lay.seg['pi_k']
            segs.append((off, off + lay.G_out, l1, l2))
        segs = [sg for sg in segs if sg[2] != 0. or sg[3] != 0.]
        self.reg = self.ops.reg_desc(segs) if segs else None
        self.reg_ws = torch.zeros(self.ops.l1l2_workspace_doubles(), dtype=torch.float64, device=self.dev) \
            if segs else None

    def add_val_penalty(self):
        """Keras reports val_loss including the regularisation losses: adds them to acc[1]."""
        if self.reg is not None and self.comm.rank == 0:
            self.val_loss_tmp.zero_()
            self.ops.l1l2_apply(self.reg, self.w, None, self.val_loss_tmp, self.reg_ws)
            self.ops.step_end(self.val_loss_tmp, 1.0, None, 0, self.acc[1:], None, 0)

    def save_state(self, path, fit_state):
        """Full training state after an epoch: parameters, RMSprop slots, BN moving statistics
        and the fit loop's scalars (epoch, lr, callback counters, history)."""
        import json
        T = self.lay.total
        extra = {} if self.slot2 is None else {'slot2': self.slot2[:T].cpu().numpy()}     # (the flat buffers are padded to 4 x world)
        if self.m_sched is not None:
            extra['m_sched'] = self.m_sched.cpu().numpy()
        np.savez(path, w=self.w[:T].cpu().numpy(), ms=self.ms[:T].cpu().numpy(), opt_iter=self.opt_iter.cpu().numpy(),
                 drop_iter=self.drop_iter.cpu().numpy(), **extra,
                 **{'mm%d' % i: t.cpu().numpy() for i, t in enumerate(self.mm)},
                 **{'mv%d' % i: t.cpu().numpy() for i, t in enumerate(self.mv)},
                 fit=np.frombuffer(json.dumps(fit_state).encode(), dtype=np.uint8))

    def gather_optimizer_slots(self):
        """Sharded optimizer (data parallel): every rank keeps only its shard of the RMSprop slots current; this collects
        them on every rank.  A COLLECTIVE: every rank calls it (the fit loop does, before rank 0 alone writes a checkpoint)."""
        if self.comm.dp and self._use_sharded_opt():
            sh = self.flat_len // self.comm.world
            self.ms.copy_(self.comm.all_gather(self.ms[self.comm.rank * sh:(self.comm.rank + 1) * sh]).reshape(-1))

    def load_state(self, path):
        import json
        with np.load(path) as z:
            T = self.lay.total
            assert z['w'].shape[0] == T, 'checkpoint belongs to a different network'
            self.w[:T].copy_(torch.as_tensor(z['w']))
            self.ms[:T].copy_(torch.as_tensor(z['ms']))
            if 'opt_iter' in z.files:
                self.opt_iter.copy_(torch.as_tensor(z['opt_iter']))
            if self.m_sched is not None and 'm_sched' in z.files:
                self.m_sched.copy_(torch.as_tensor(z['m_sched']))
            if 'drop_iter' in z.files:
                self.drop_iter.copy_(torch.as_tensor(z['drop_iter']))
            if self.slot2 is not None and 'slot2' in z.files:
                self.slot2[:T].copy_(torch.as_tensor(z['slot2'][:T]))          # (older checkpoints hold the padded length)
            for i in range(len(self.mm)):
                self.mm[i].copy_(torch.as_tensor(z['mm%d' % i]))
                self.mv[i].copy_(torch.as_tensor(z['mv%d' % i]))
            return json.loads(bytes(z['fit']).decode())

    def set_lr(self, lr):
        self.lr.fill_(float(np.float32(lr)))

    # ------------------------------------------------------------------ data
    def load_data(self, X, Y=None, sf=None, chunk_rows=8192):
        """Uploads host matrices once into padded device buffers ([n, ld], ld % 4 == 0)."""
        lay = self.lay
        n = X.shape[0]
        assert X.shape[1] == lay.G_in
        self.n = n
        self.ldx = _r4(lay.G_in)
        self.X = torch.zeros(n, self.ldx, dtype=torch.float32, device=self.dev)
        if Y is not None:
            assert Y.shape == (n, lay.G_out)
            self.ldy = lay.Gp
            self.Y = torch.zeros(n, self.ldy, dtype=torch.float32, device=self.dev)
        for s in range(0, n, chunk_rows):
            e = min(n, s + chunk_rows)
            xs = X[s:e]
            xs = xs.toarray() if hasattr(xs, 'toarray') else np.asarray(xs)
            self.X[s:e, :lay.G_in] = _prep.host_chunk_tensor(xs).to(self.dev)
            if Y is not None:
                ys = Y[s:e]
                ys = ys.toarray() if hasattr(ys, 'toarray') else np.asarray(ys)
                self.Y[s:e, :lay.G_out] = _prep.host_chunk_tensor(ys).to(self.dev)
        sfv = np.ones(n, np.float32) if sf is None else np.asarray(sf, dtype=np.float32).reshape(-1)
        self.sf = torch.as_tensor(sfv).to(self.dev)
        self._set_tile_order()
        self._data_scales()
        self.attach_compact()               # K-HEADS reads the byte store; the input normalisation of a host X is unknown

    def attach_device_data(self, X, Y, sf, norm=None, compact=None):
        """Uses tensors that already live on the device (K-PREP, synthetic generators, bench).
        norm: how X was made from Y -- dict(fac=[n] or None, do_log=bool, mean=[G] or None, std=[G] or None) with
        x = (f(y / fac) - mean) / std (dca/io.py:99-109); with it (and counts that fit the compact store) the first
        layer takes the sparse kernels.  compact: a CompactCounts of Y built earlier (else built here)."""
        lay = self.lay
        assert X.shape[1] == _r4(lay.G_in) and (Y is None or Y.shape[1] == lay.Gp)
        self.n, self.ldx, self.ldy = X.shape[0], X.shape[1], lay.Gp
        self.X, self.Y, self.sf = X, Y, sf
        self._set_tile_order()
        self._data_scales()
        self.attach_compact(compact, norm)

    def _data_scales(self):
        """What the fp16 x 2 plane products of the wide networks need to know about the resident data, once per dataset:
        the block exponent of the input matrix X (a device word: dcahip_absmax_exp) and the exponent d_exp of the gradient
        planes, from the bound |g| <= max(1e4, 2 y_max + 50) of the likelihood's formulas (include/dcahip.h,
        dcahip_zinb_nll_planes_h2).  d_exp < 0 (counts beyond ~16 000): the wide path keeps the three-piece bf16 planes."""
        self.x_exp = None
        self.d_exp = None
        self.heads_d_exp = self._heads_d_exp()
        ops = self.ops
        if not (self.cfg.wide_h2 and hasattr(ops, 'gemm_h2') and self.lay.hidden and self.lay.hL > 64) or self.X is None:
            return
        if not self.X.is_cuda:
            return
        if self.Y is not None:
            ymax = float(self.Y.max().item()) if self.Y.numel() else 0.0
            bound = max(1e4, 2.0 * ymax + 50.0) + 0.5 * float(self.ridge)
            self.d_exp = int(np.floor(np.log2(65000.0 / bound)))
        if self.lay.hidden[0] >= 128:
            self.x_exp = torch.zeros(2, dtype=torch.int32, device=self.dev)
            ops.absmax_exp(self.X, self.ldx, self.X.shape[0], self.lay.G_in, self.x_exp)

    def _heads_d_exp(self):
        """K-HEADS' starting exponent for its gradient pieces (include/dcahip.h, dcahip_heads_fused_compact: d_exp <= 0), once
        per dataset.  The kernel carries D = g 2^(8 + d_exp) in fp16: |g| <= 117 fits at d_exp = 0, and a 32 x 32 tile with a larger
        gradient repeats its forward product and likelihood pass with the exponent it needs (exact, ~1.6 x the tile's time).
        Counts in the hundreds THROUGHOUT the matrix (full-length protocols) would make every tile repeat: start lower instead.
        From a sample of the counts: the count c that one element in 20 000 exceeds (1 tile in 20 holds such an element), with
        |g| <~ 2 c.  UMI data (the benchmark matrices: c < 58) keeps 0.  Rank-local on purpose (attach is not a collective:
        predict may run on one rank): data-parallel ranks may start at different exponents, which moves their gradients within
        the products' tolerance only."""
        Y = self.Y
        if Y is None or not getattr(Y, 'is_cuda', False) or Y.numel() == 0:
            return 0
        rows = min(Y.shape[0], max(1, (1 << 24) // max(1, Y.shape[1])))
        samp = Y[:: max(1, Y.shape[0] // rows)][:rows, :self.lay.G_out]
        cnt = torch.bincount(torch.nan_to_num(samp, nan=0.0).clamp(0, 65535.0).to(torch.int64).reshape(-1), minlength=2)
        tail = torch.flip(torch.cumsum(torch.flip(cnt, [0]), 0), [0]).to(torch.float64) / float(samp.numel())    # P(y >= c)
        over = torch.nonzero(tail > 5e-5)
        c = float(over.max().item()) if over.numel() else 0.0
        need = 2.0 * c / 117.0
        return 0 if need <= 1.0 else -min(24, int(np.ceil(np.log2(need))))

    def _h2(self, B):
        """The wide networks' plane products on fp16 x 2 planes (half the matrix instructions of the bf16 x 3 planes) at this
        batch: every product a shape the 256 x 256 kernel takes, the gradient planes inside the fp16 range by construction."""
        lay, ops = self.lay, self.ops
        if not (self.pl is not None and self.pl_exp is not None and self._wide_planes(B) and self._planes_nll(B)
                and self.d_exp is not None and self.d_exp >= 0):
            return False
        if B not in self._h2_ok:
            ok = (ops.gemm_h2_supported(B, lay.NH, _r16(lay.hL)) and ops.gemm_h2_supported(lay.hL, lay.NH, B)
                  and ops.gemm_h2_supported(B, lay.hL, _r16(lay.NH)) and B % 16 == 0)
            self._h2_ok[B] = bool(ok)
        return self._h2_ok[B]

    def _h2_enc0(self, B, training):
        lay, ops = self.lay, self.ops
        if not (self._h2(B) and 'X' in self.pl and self.x_exp is not None and self._planes_enc0(B, training)):
            return False
        key = ('enc0', B)
        if key not in self._h2_ok:
            self._h2_ok[key] = bool(ops.gemm_h2_supported(B, lay.hidden[0], _r16(lay.G_in))
                                    and ops.gemm_h2_supported(lay.G_in, lay.hidden[0], B))
        return self._h2_ok[key]

    def attach_compact(self, compact=None, norm=None):
        """Compact byte store of the resident counts (built here unless given) for K-HEADS and -- when the input
        normalisation is known and input genes = output genes -- the sparse first layer."""
        self.cc = self.cc_in = None
        self.cc_verdict = None              # False: these counts do not take the byte store (a full pass over Y found that out)
        ops, lay = self.ops, self.lay
        if self.Y is None or not hasattr(ops, 'counts_compact'):
            return
        if compact is False:
            self.cc_verdict = False         # an earlier attach found these counts unfit for the byte store: the callers keep
            return                          # the verdict (no second full pass over Y on the next attach)
        if compact is None:
            from . import compact as _compact
            compact = _compact.build(ops, self.Y, self.Y.shape[0], lay.G_out)
        if compact is None:
            self.cc_verdict = False
            return                          # not a count matrix (check_counts=False on arbitrary data): fp32 path
        # (counts >= 255 escape into a per-row list that K-HEADS scans linearly per escaped element, and that one workgroup
        # walks for the first layer: read-count data with many large counts -- Smart-seq and the like -- keeps the fp32
        # targets and the dense first layer.  Thresholds: one escape in 1e3 counts for K-HEADS' targets (the scan of a row's
        # list then costs less than the bytes save), one in 1e5 for the sparse first layer.)
        n_esc = 0 if compact.ovf_col is None else int(compact.ovf_col.numel())
        n_el = float(self.Y.shape[0]) * lay.G_out
        if n_esc > 1e-3 * n_el:
            self.cc_verdict = False
            return
        self.cc = compact
        # cc_in is not None <=> the byte-store kernels take this first-layer width (32 / 64 / 128 units): every other width
        # keeps the dense products (_sparse_dw / _lut_fwd test nothing else about the width)
        if norm is not None and lay.hidden and lay.G_in == lay.G_out and n_esc <= 1e-5 * n_el and \
                ops.enc0_sparse_supported(lay.hidden[0]):
            self.cc_in = compact.with_input(norm.get('fac'), norm.get('do_log', False), norm.get('mean'), norm.get('std'), ops=ops)
        self._sparse_workspaces()

    def _sparse_workspaces(self):
        if self.cc_in is None or self.Bmax <= 0:
            return
        lay, ops = self.lay, self.ops
        need = ops.enc0_dw_sparse_workspace_bytes(self.Bmax, lay.G_in, lay.hidden[0])
        if need and (self.ws_enc0 is None or self.ws_enc0.numel() * 4 < need):      # 0 bytes: width not taken (ws_enc0 stays None)
            self.ws_enc0 = torch.zeros(need // 4 + 4, dtype=torch.float32, device=self.dev)
        if self.Bmax >= self.cfg.lut_fwd_min:
            # the matrix-pipe forward from the byte store: its partials grow with rows x gene chunks -- the largest over
            # the batch sizes this engine can be handed
            need = max(ops.enc0_fwd_lut_workspace_bytes(min(b, self.Bmax), lay.G_in, lay.hidden[0])
                       for b in range(256, self.Bmax + 256, 256))
            if need and (self.ws_enc0l is None or self.ws_enc0l.numel() * 4 < need):
                self.ws_enc0l = torch.zeros(need // 4 + 4, dtype=torch.float32, device=self.dev)

    def _lut_fwd(self, B, training):
        # inference takes this form only once the per-cell tables exist (training made them): a predict-only run keeps the
        # dense product instead of building 512 B of tables per cell for no gain at its chunk size
        return (self.cc_in is not None and self.ws_enc0l is not None and B >= self.cfg.lut_fwd_min
                and not (training and self.in_drop > 0.0) and (training or self.cc_in.lutp is not None))

    def _sparse_dw(self, B):
        return (self.cc_in is not None and self.ws_enc0 is not None and B >= self.sparse_dw_min and self.in_drop == 0.0)

    def _set_tile_order(self):
        """K-HEADS: which 32-gene tiles share a workgroup.  A workgroup lasts as long as its slower tile and the
        tiles' cost follows their non-zero counts (the compacted NB pass), so tiles are sorted by the non-zero count
        of their columns (heaviest first) and taken pairwise: one pass over the resident counts per dataset."""
        self.tile_order = None
        if self.Y is None or not self.use_fused or self.ops.device_type != 'cuda':
            return
        lay = self.lay
        n_ord = self.ops.heads_tile_order_len(lay.G_out)
        ntg = (lay.G_out + 31) // 32
        rows = min(self.Y.shape[0], 65536)                       # a sample is enough to rank the tiles
        nz = torch.zeros(ntg * 32, dtype=torch.float32, device=self.dev)
        for s in range(0, rows, 8192):
            nz[:lay.G_out] += (self.Y[s:min(rows, s + 8192), :lay.G_out] != 0).sum(dim=0)
        order = torch.argsort(nz.view(ntg, 32).sum(dim=1), descending=True).to(torch.int32)
        pad = torch.arange(ntg, n_ord, dtype=torch.int32, device=self.dev)
        self.tile_order = torch.cat([order, pad]).contiguous()

    # ------------------------------------------------------------------ work buffers
    def reserve(self, B):
        """Allocates every per-batch buffer for up to B rows (no allocation inside a step)."""
        if B <= self.Bmax:
            return
        lay, ops = self.lay, self.ops
        f32 = dict(dtype=torch.float32, device=self.dev)
        self.Bmax = B
        self.ldh = [_r4(h) for h in lay.hidden]
        self.Z = [torch.zeros(B, l, **f32) for l in self.ldh]
        self.XH = [torch.zeros(B, l, **f32) for l in self.ldh] if lay.batchnorm else []
        self.H = [torch.zeros(B, l, **f32) for l in self.ldh]
        # outputs after dropout (the next layer's input); layers without dropout alias H
        self.HD = [torch.zeros(B, l, **f32) if r > 0.0 else None for l, r in zip(self.ldh, self.drop)]
        self.HP = [torch.zeros(B, l, **f32) for l in self.ldh] if self.prelu else None    # PReLU outputs
        self.ws_prelu = torch.zeros(ops.prelu_workspace_doubles(max(lay.hidden)), dtype=torch.float64, device=self.dev) \
            if self.prelu else None
        self.Hcur = list(self.H)
        self.Xb = None                                  # input-dropout batch, allocated on first use
        # hidden_size=(): the heads are Dense layers on the input itself (the reference's own fixture scripts fit
        # exactly that network, data/test-biochemists-zinb.py:10-19) -- the gathered minibatch and a sink for its gradient
        self.Hin0 = torch.zeros(B, _r4(lay.G_in), **f32) if not lay.hidden else None
        self.dHin0 = torch.zeros(B, _r4(lay.G_in), **f32) if not lay.hidden else None
        self.dH = [torch.zeros(B, l, **f32) for l in self.ldh]
        self.dZ = [torch.zeros(B, l, **f32) for l in self.ldh]
        self.A = torch.zeros(B, lay.ldA, **f32)
        # gradient planes of the heads (+ one d nll/d theta plane behind them for const-disp)
        self.D = torch.zeros(B, self.ldD, **f32)
        self.Dth = self.D[:, lay.ldA:] if lay.const_disp else None
        R = ops.col_moments_chunks(B)
        self.inv_std = [torch.zeros(h, **f32) for h in lay.hidden]
        self.part = [torch.zeros(max(R, self.comm.world) * 2 * h, **f32) for h in lay.hidden]
        self.bpart = [torch.zeros(R * 2 * h, **f32) for h in lay.hidden]
        self.stat_local = [torch.zeros(2 * h, **f32) for h in lay.hidden]
        self.bsum = [torch.zeros(2 * h, **f32) for h in lay.hidden]      # SyncBN backward: [sum dy | sum dy xhat] per layer
        self.counts_world = torch.zeros(self.comm.world, **f32)
        self._counts_world_key = None
        self._counts_local, self._counts_world = {}, {}
        # split-K workspace: the maximum any GEMM of a step can ask for, over every batch size up
        # to B (the split plan is a function of the shape: a smaller last batch may split more)
        cand = sorted({B} | {b for k in range(0, B // 64 + 2) for b in (64 * k, 64 * k + 1) if 1 <= b <= B})
        need = 0
        for b in cand:
            K = lay.G_in
            for i, h in enumerate(lay.hidden):
                need = max(need, ops.sgemm_workspace_bytes(0, 0, b, h, K))
                need = max(need, ops.sgemm_workspace_bytes(1, 0, K, h, b, True))
                if i > 0:
                    need = max(need, ops.sgemm_workspace_bytes(0, 1, b, K, h))
                K = h
            for _c0, nc, _h0 in self._head_blocks():
                need = max(need, ops.sgemm_workspace_bytes(0, 0, b, nc, lay.hL))
                need = max(need, ops.sgemm_workspace_bytes(1, 0, lay.hL, nc, b, True))
                need = max(need, ops.sgemm_workspace_bytes(0, 1, b, lay.hL, nc))
        nb = ops.heads_fused_workspace_bytes(B, lay.hL, lay.G_out, lay.Gp, self.flags) if self.use_fused else 0
        self.ws_heads = torch.zeros(nb // 4, **f32) if nb > 0 else None
        self.W0T = self.WhT = self.HT = self.XT = self.dZT = None
        if hasattr(ops, 'transpose') and B >= 256 and lay.hidden:
            self.W0T = torch.zeros(lay.hidden[0], _r4(lay.G_in), **f32)
            for b in cand:
                need = max(need, ops.sgemm_workspace_bytes(0, 1, b, lay.hidden[0], lay.G_in))
            if self._wide_transposed(B):
                # separate-kernel heads of a wide decoder (> 64 units) and a wide first layer at throughput batches: every
                # product in the form whose operands K-GEMM reads fastest (A contiguous along the contraction) -- the
                # head weights, the last hidden activations and the minibatch of X are transposed once per step
                hin = lay.hidden[-1]
                self.ldb_t = _r4(B)
                self.WhT = torch.zeros(lay.NH, _r4(lay.hL), **f32)
                self.HT = torch.zeros(hin, self.ldb_t, **f32)
                if lay.hidden[0] >= 128:
                    self.XT = torch.zeros(lay.G_in, self.ldb_t, **f32)
                    self.dZT = torch.zeros(lay.hidden[0], self.ldb_t, **f32)
                for b in cand:
                    for _c0, nc, _h0 in self._head_blocks():
                        need = max(need, ops.sgemm_workspace_bytes(0, 1, b, nc, lay.hL))
                        need = max(need, ops.sgemm_workspace_bytes(0, 0, lay.hL, nc, b, True))
                    if self.XT is not None:
                        need = max(need, ops.sgemm_workspace_bytes(0, 1, lay.G_in, lay.hidden[0], b))
        self.pl = None
        self.pl_exp = None
        self._h2_ok = {}
        if self._wide_planes(B):
            def planes(rows, cols):
                return torch.zeros(3, rows, _r16(cols), dtype=torch.bfloat16, device=self.dev)
            self.pl = {'H': planes(B, lay.hL), 'Wh': planes(_r16(lay.hL), lay.NH), 'D': planes(B, lay.NH)}
            if lay.hidden[0] >= 128:          # a wide first layer as well: the minibatch, its kernel, its output gradient
                self.pl.update(X=planes(B, lay.G_in), W0=planes(_r16(lay.G_in), lay.hidden[0]), dZ0=planes(B, lay.hidden[0]))
            for b in cand:
                need = max(need, ops.gemm_p3_workspace_bytes(b, lay.NH, _r16(lay.hL)),
                           ops.gemm_p3_workspace_bytes(lay.hL, lay.NH, b, True),
                           ops.gemm_p3_workspace_bytes(b, lay.hL, _r16(lay.NH)))
                if 'X' in self.pl:
                    need = max(need, ops.gemm_p3_workspace_bytes(b, lay.hidden[0], _r16(lay.G_in)),
                               ops.gemm_p3_workspace_bytes(lay.G_in, lay.hidden[0], b, True))
            # the same buffers hold the TWO fp16 planes of the fp16 x 2 products (the first two planes: 16-bit elements either
            # way); their block exponents are device words (nothing leaves the stream)
            self.pl_exp = None
            self._h2_ok = {}
            if self.cfg.wide_h2 and hasattr(ops, 'gemm_h2'):
                self.pl_exp = {k: torch.zeros(2, dtype=torch.int32, device=self.dev) for k in ('H', 'Wh', 'W0', 'dZ0')}
                for b in cand:
                    need = max(need, ops.gemm_h2_workspace_bytes(b, lay.NH, _r16(lay.hL)),
                               ops.gemm_h2_workspace_bytes(lay.hL, lay.NH, b, True),
                               ops.gemm_h2_workspace_bytes(b, lay.hL, _r16(lay.NH)))
                    if 'X' in self.pl:
                        need = max(need, ops.gemm_h2_workspace_bytes(b, lay.hidden[0], _r16(lay.G_in)),
                                   ops.gemm_h2_workspace_bytes(lay.G_in, lay.hidden[0], b, True))
        self.ws = torch.zeros(max(need // 4, 4), **f32)
        self.ws_stack = None
        if hasattr(ops, 'hidden_stack_fwd') and lay.hidden and max(lay.hidden) <= 64 and len(lay.hidden) <= 8:
            nb = ops.hidden_stack_workspace_bytes(len(lay.hidden), min(B, ops.hidden_stack_max_rows))
            self.ws_stack = torch.zeros(nb // 4 + 4, **f32) if nb > 0 else None
        self._sparse_workspaces()

    # ------------------------------------------------------------------ forward pieces
    def _hidden_forward(self, B, rows_from, training, counts=None):
        """Dense -> BN -> ReLU stack (network.py:101-138).  rows_from: ('perm',) gathers the
        batch through perm/cursor, ('range', r0) reads storage rows r0..r0+B."""
        lay, ops = self.lay, self.ops
        w = self.w
        K = lay.G_in
        if not lay.hidden:
            # no hidden layer: the heads read the minibatch of the input (network.py:98-99 input dropout applies)
            if rows_from[0] == 'perm':
                ops.dropout_apply(self.X, self.ldx, self.perm, self.cursor, B, K, self.in_drop if training else 0.0,
                                  self.drop_seed, self.drop_iter, INPUT_DROPOUT_LAYER, self.row0, self.Hin0, self.ldx)
                self._hl = (self.Hin0, self.ldx)
            else:
                self._hl = (self.X[rows_from[1]:], self.ldx)
            return K
        for i, h in enumerate(lay.hidden):
            Wi = lay.view(w, 'W%d' % i); bi = lay.view(w, 'b%d' % i)
            if i == 0:
                if self._lut_fwd(B, training):
                    # K-SPARSE on the matrix pipe: the input looked up from the byte store (x = (log1p(y / fac) - mean) / std
                    # never read: 1 byte per count instead of 4, no transposed kernel)
                    if self.cc_in.lutp is None:          # the per-cell table of the common counts: first use only, never in a capture
                        self._not_capturing('first use of the byte-store forward')
                        self.cc_in.ensure_lut(ops)
                    gather = rows_from[0] == 'perm'
                    with self._t('gemm_enc0_fwd'):
                        ops.enc0_fwd_lut(self.cc_in, self.perm if gather else None, self.cursor if gather else None,
                                         0 if gather else rows_from[1], B, K, h, Wi, h, bi, self.Z[0], self.ldh[0],
                                         self.ws_enc0l)
                elif self._planes_enc0(B, training):
                    # wide first layer at throughput batches: the minibatch split into planes once (the weight gradient
                    # reads the same planes, contracting over their rows), the kernel split, the product from planes
                    gather = rows_from[0] == 'perm'
                    with self._t('gemm_enc0_fwd'):
                        if self._h2_enc0(B, training):
                            # two fp16 planes per operand, three products: X scaled by its dataset-wide exponent, W0 by its own
                            ops.split_planes_h2(self.X if gather else self.X[rows_from[1]:], self.ldx, B, K, self.pl['X'], self.x_exp,
                                                perm=self.perm if gather else None, cursor=self.cursor if gather else None)
                            ops.absmax_exp(Wi, h, K, h, self.pl_exp['W0'])
                            ops.split_planes_h2(Wi, h, K, h, self.pl['W0'], self.pl_exp['W0'])
                            ops.gemm_h2(0, 0, B, h, _r16(K), self.pl['X'], self.pl['W0'], self.Z[0], self.ldh[0],
                                        exp_a=self.x_exp, exp_b=self.pl_exp['W0'], bias=bi, ws=self.ws)
                        else:
                            ops.split_planes(self.X if gather else self.X[rows_from[1]:], self.ldx, B, K, self.pl['X'],
                                             perm=self.perm if gather else None, cursor=self.cursor if gather else None)
                            ops.split_planes(Wi, h, K, h, self.pl['W0'])
                            ops.gemm_p3(0, 0, B, h, _r16(K), self.pl['X'], self.pl['W0'], self.Z[0], self.ldh[0], bias=bi,
                                        ws=self.ws)
                elif training and self.in_drop > 0.0:
                    assert rows_from[0] == 'perm'
                    if self.Xb is None or self.Xb.shape[0] < self.Bmax:
                        self.Xb = torch.zeros(self.Bmax, self.ldx, dtype=torch.float32, device=self.dev)
                    ops.dropout_apply(self.X, self.ldx, self.perm, self.cursor, B, K, self.in_drop,
                                      self.drop_seed, self.drop_iter, INPUT_DROPOUT_LAYER, self.row0,
                                      self.Xb, self.ldx)
                    with self._t('gemm_enc0_fwd'):
                        ops.sgemm(0, 0, B, h, K, self.Xb, self.ldx, Wi, h, self.Z[0], self.ldh[0],
                                  bias=bi, ws=self.ws)
                elif rows_from[0] == 'perm' and self._enc0_nt(B):
                    # throughput batches: W0 transposed once per step, the forward product in the NT form (both operands
                    # contiguous along the genes: the fast operand path of K-GEMM)
                    with self._t('gemm_enc0_fwd'):
                        ops.transpose(Wi, h, K, h, self.W0T, self.ldx)
                        ops.sgemm(0, 1, B, h, K, self.X, self.ldx, self.W0T, self.ldx, self.Z[0], self.ldh[0],
                                  bias=bi, perm=self.perm, cursor=self.cursor, ws=self.ws)
                elif rows_from[0] == 'perm':
                    with self._t('gemm_enc0_fwd'):
                        ops.sgemm(0, 0, B, h, K, self.X, self.ldx, Wi, h, self.Z[0], self.ldh[0],
                                  bias=bi, perm=self.perm, cursor=self.cursor, ws=self.ws)
                else:
                    ops.sgemm(0, 0, B, h, K, self.X[rows_from[1]:], self.ldx, Wi, h, self.Z[0],
                              self.ldh[0], bias=bi, ws=self.ws)
            elif training and i == 1 and self._stack_small(B):
                break       # the rest of the stack ran inside the chain launch below
            elif training and self._layer_small(B, i):
                # small batches: Dense -> BatchNormalization -> activation of this layer in ONE launch
                ops.dense_bn_small(self.Hcur[i - 1], self.ldh[i - 1], Wi, h, bi, B, K, h, lay.batchnorm,
                                   lay.view(w, 'beta%d' % i) if lay.batchnorm else None,
                                   self.mm[i] if lay.batchnorm else None, self.mv[i] if lay.batchnorm else None,
                                   BN_MOMENTUM, BN_EPS, self.act, self.Z[i], self.ldh[i],
                                   self.XH[i] if lay.batchnorm else None, self.ldh[i], self.H[i], self.ldh[i],
                                   self.inv_std[i])
                self.Hcur[i] = self.H[i]
                K = h
                continue
            else:
                ops.sgemm(0, 0, B, h, K, self.Hcur[i - 1], self.ldh[i - 1], Wi, h, self.Z[i],
                          self.ldh[i], bias=bi, ws=self.ws)
            if training and i == 0 and self._stack_sync(B):
                # data parallel: one K-STACK step per launch, the (mean, M2) of every rank gathered between two launches
                entries = [self._chain_entry(j) for j in range(len(lay.hidden))]
                n, Wd = len(entries), self.comm.world
                ops.hidden_stack_fwd_sync(entries, B, BN_MOMENTUM, BN_EPS, self.act, 0, None, None, 0, self.stat_local[0],
                                          self.ws_stack)
                self.comm.all_gather_into(self.part[0][:Wd * 2 * lay.hidden[0]], self.stat_local[0], name='all_gather_small')
                for st in range(1, n + 1):
                    ops.hidden_stack_fwd_sync(entries, B, BN_MOMENTUM, BN_EPS, self.act, st, self.part[st - 1], counts, Wd,
                                              self.stat_local[st] if st < n else None, self.ws_stack)
                    if st < n:
                        self.comm.all_gather_into(self.part[st][:Wd * 2 * lay.hidden[st]], self.stat_local[st],
                                                  name='all_gather_small')
                for j, hj in enumerate(lay.hidden):
                    self.Hcur[j] = self.H[j]
                    K = hj
                break
            if training and i == 0 and self._stack_coop(B):
                # throughput batches: batch norm + activation of this layer and the whole stack behind it in ONE cooperative
                # launch (K-STACK: workgroups own row blocks, exchange the batch statistics through grid barriers)
                entries = [self._chain_entry(j) for j in range(len(lay.hidden))]
                if self.stack_mode == 'coop' and B <= 256 * 64:
                    ops.hidden_stack_fwd(entries, B, BN_MOMENTUM, BN_EPS, self.act, self.ws_stack, rows_per_wg=64)
                else:           # one step per launch: the kernel boundary is the barrier, small row blocks fill the chip
                    for st in range(len(entries) + 1):
                        ops.hidden_stack_fwd(entries, B, BN_MOMENTUM, BN_EPS, self.act, self.ws_stack,
                                             rows_per_wg=self._stack_rows(B), steps=(st, st))
                for j, hj in enumerate(lay.hidden):
                    self.Hcur[j] = self.H[j]
                    K = hj
                break
            if training and i == 0 and self._stack_small(B):
                # small batches, every layer at most 64 units: batch norm + activation of this layer and the whole stack
                # behind it (Dense -> BatchNormalization -> activation per layer) in ONE launch
                ops.hidden_small_chain([self._chain_entry(j) for j in range(len(lay.hidden))], None, 0, B, True,
                                       BN_MOMENTUM, BN_EPS, self.act)
                for j, hj in enumerate(lay.hidden):
                    self.Hcur[j] = self.H[j]
                    K = hj
                continue
            if lay.batchnorm:
                beta = lay.view(w, 'beta%d' % i)
                if training and self._bn_small(B):
                    # one launch: batch statistics, moving averages, normalisation, activation
                    ops.bn_relu_train_small(self.Z[i], self.ldh[i], B, h, beta, self.mm[i], self.mv[i], BN_MOMENTUM,
                                            BN_EPS, self.act, self.H[i], self.ldh[i], self.XH[i], self.ldh[i],
                                            self.inv_std[i])
                elif training:
                    entries, cnts, E = self._batch_moments(i, B, h, counts)
                    ops.bn_relu_apply(self.Z[i], self.ldh[i], B, h, entries, cnts, E, beta,
                                      self.mm[i], self.mv[i], BN_MOMENTUM, BN_EPS, self.act, self.H[i],
                                      self.ldh[i], self.XH[i], self.ldh[i], self.inv_std[i])
                else:
                    ops.bn_relu_apply(self.Z[i], self.ldh[i], B, h, None, None, 0, beta, self.mm[i],
                                      self.mv[i], BN_MOMENTUM, BN_EPS, self.act, self.H[i], self.ldh[i],
                                      None, 0, None)
            else:
                ops.relu_fwd(self.Z[i], self.ldh[i], B, h, self.H[i], self.ldh[i], self.act)
            cur = self.H[i]
            if self.prelu:
                ops.prelu_fwd(self.H[i], self.ldh[i], lay.view(w, 'alpha%d' % i), B, h, self.HP[i], self.ldh[i])
                cur = self.HP[i]
            if training and self.drop[i] > 0.0:
                ops.dropout_apply(cur, self.ldh[i], None, None, B, h, self.drop[i], self.drop_seed,
                                  self.drop_iter, i, self.row0, self.HD[i], self.ldh[i])
                cur = self.HD[i]
            self.Hcur[i] = cur
            K = h
        self._hl = (self.Hcur[-1], self.ldh[-1])        # what the heads read
        return K

    def _planes_nll(self, B):
        """K-ZINB writes the heads' gradient planes as bf16 pieces itself (NB / ZINB likelihoods on vector-aligned rows)."""
        return (self.pl is not None and self._wide_planes(B) and B <= self.pl['D'].shape[1]
                and not (self.flags & (AE_LOSS_FLAG['poisson'] | AE_LOSS_FLAG['normal'])) and self.ldy % 4 == 0 and self.lay.ldA % 4 == 0
                and hasattr(self.ops, 'zinb_nll_planes'))

    def _planes_enc0(self, B, training):
        return (self.pl is not None and 'X' in self.pl and self._wide_planes(B) and B <= self.pl['X'].shape[1]
                and not (training and self.in_drop > 0.0)
                and not self._lut_fwd(B, training))

    def _enc0_nt(self, B):
        return self.W0T is not None and B >= self.cfg.enc0_nt_min

    def _wide_planes(self, B):
        """Throughput batches of a network whose heads run as separate kernels (decoder wider than 64 units): every large
        product from pre-split bf16 planes (dcahip_gemm_p3) -- the operands that enter several products of a step (the
        heads' gradient planes, the head weights, the minibatch) are split once, and one stored layout serves the products
        that contract over its rows and over its columns, so no transposed copies are kept."""
        lay = self.lay
        return (self.ws_heads is None and B >= 256 and hasattr(self.ops, 'gemm_p3') and not lay.fork and not lay.elempi
                and not lay.shared and self.cfg.wide_planes and bool(lay.hidden))

    def _wide_transposed(self, B):
        """The same networks without the planes path: transposed operand copies for K-GEMM's fast forms (see reserve)."""
        return (self.ws_heads is None and B >= 256 and hasattr(self.ops, 'transpose') and not self._wide_planes(B)
                and bool(self.lay.hidden))

    def _stack_small(self, B):
        """The whole hidden stack behind the first product in one launch (batch-normalised, every layer small)."""
        L = len(self.lay.hidden)
        return (self.lay.batchnorm and 2 <= L <= 4 and hasattr(self.ops, 'hidden_small_chain') and self._bn_small(B)
                and all(self._layer_small(B, i) for i in range(1, L)) and self.lay.hidden[0] <= 64)

    def _stack_chain(self, B):
        """The backward of the whole hidden stack in ONE single-workgroup launch: batches of at most 64 rows (the
        reference's default 32) through small batch-normalised layers."""
        return self._stack_small(B) and B <= 64 and hasattr(self.ops, 'hidden_stack_bwd') and self.cfg.bwd_chain

    def _stack_coop(self, B):
        """The hidden stack in one cooperative launch per direction (K-STACK): one GPU, batch norm on, every layer at
        most 64 units, no dropout / PReLU, batches beyond the single-workgroup kernels."""
        lay = self.lay
        return (not self.comm.dp and lay.batchnorm and not self.prelu and not self.has_dropout
                and hasattr(self.ops, 'hidden_stack_fwd') and self.ws_stack is not None
                and 1 <= len(lay.hidden) <= 8 and max(lay.hidden) <= 64 and not self._bn_small(B)
                and B <= self.ops.hidden_stack_max_rows and self.stack_mode != 'off')

    def _stack_sync(self, B):
        """Data parallel: the hidden stack as K-STACK's one-step launches with the SyncBN exchange BETWEEN two launches (the
        statistics of every rank enter the next step as one entry per rank) instead of one launch per operation.  A rank
        with an empty batch takes the per-operation path, which issues the same collectives in the same order."""
        lay = self.lay
        return (self.comm.dp and B > 0 and lay.batchnorm and not self.prelu and not self.has_dropout
                and hasattr(self.ops, 'hidden_stack_fwd_sync') and self.ws_stack is not None
                and 1 <= len(lay.hidden) <= 8 and max(lay.hidden) <= 64 and B <= self.ops.hidden_stack_max_rows
                and self._stack_rows(B) == 32 and self.stack_mode != 'off' and self.comm.world <= 128)

    def _stack_rows(self, B):
        """Rows per workgroup of the one-step launches."""
        r = 32              # the partition of the step kernels (all reads of a step up front); coarser beyond 32 768 rows
        while (B + r - 1) // r > 1024:
            r *= 2
        return r

    def _chain_entry(self, j):
        lay, w = self.lay, self.w
        d = dict(H=lay.hidden[j], beta=lay.view(w, 'beta%d' % j), moving_mean=self.mm[j], moving_var=self.mv[j],
                 Z=self.Z[j], ldz=self.ldh[j], xhat=self.XH[j], ldx=self.ldh[j], Hout=self.H[j], ldh=self.ldh[j],
                 inv_std=self.inv_std[j])
        if j > 0:
            d.update(W=lay.view(w, 'W%d' % j), ldw=lay.hidden[j], bias=lay.view(w, 'b%d' % j), K=lay.hidden[j - 1])
        return d

    def _layer_small(self, B, i):
        """Hidden layer i >= 1 at a small batch on one GPU: whole-layer kernels (forward and backward)."""
        if i < 1 or not self._bn_small(B) or self.prelu or self.has_dropout or not hasattr(self.ops, 'dense_bn_small'):
            return False
        kmax = self.ops.dense_small_max_k
        return self.lay.hidden[i - 1] <= kmax and self.lay.hidden[i] <= kmax

    def _bn_small(self, B):
        """Small batches on one GPU take the single-launch batch-norm kernels (the reference-default batch of 32 is
        bound by launch gaps, not by kernels)."""
        return not self.comm.dp and 0 < B <= getattr(self.ops, 'bn_fused_max_rows', 0)

    def _batch_moments(self, i, B, h, counts):
        """Batch statistics of layer i as (entries, counts, E) for bn_relu_apply.  One GPU: the
        row-chunk partials directly.  Data parallel (SyncBN): merge local chunks, all-gather
        one (count, mean, M2) triple per rank."""
        ops = self.ops
        if B > 0:
            ops.col_moments(self.Z[i], self.ldh[i], B, h, self.part[i])
        if not self.comm.dp:
            return self.part[i], None, ops.col_moments_chunks(B)
        R = ops.col_moments_chunks(max(B, 1))
        if B > 0:
            cr = -(-B // R)
            cl = self._counts_local.get((B, R))
            if cl is None:
                # one device buffer per batch size, written once and never again: a captured step keeps reading ITS
                # buffer when steps of another batch size run in between (a shared buffer refreshed "when the batch size
                # changes" is refreshed by the host -- which a graph replay does not involve)
                self._not_capturing('row-chunk counts of a new batch size')
                cl = torch.as_tensor([max(0, min(B, (r + 1) * cr) - r * cr) for r in range(R)], dtype=torch.float32).to(self.dev)
                self._counts_local[(B, R)] = cl
            ops.moments_combine(self.part[i], cl, R, h, self.stat_local[i])
        else:
            self.stat_local[i].zero_()
        # [W, 2h] straight into the entries buffer of bn_relu_apply (no staging tensor, no copy launch)
        self.comm.all_gather_into(self.part[i][:self.comm.world * 2 * h], self.stat_local[i], name='all_gather_small')
        return self.part[i], counts, self.comm.world

    def _heads_forward(self, B, K):
        lay, ops = self.lay, self.ops
        Wh, bh = lay.view(self.w, 'Wh'), lay.view(self.w, 'bh')
        with self._t('gemm_heads_fwd'):
            if self.pl is not None and self._wide_planes(B) and B <= self.pl['H'].shape[1] and self._h2(B):
                ops.absmax_exp(self._hl[0], self._hl[1], B, lay.hL, self.pl_exp['H'])
                ops.split_planes_h2(self._hl[0], self._hl[1], B, lay.hL, self.pl['H'], self.pl_exp['H'])
                ops.absmax_exp(Wh, lay.NH, lay.hL, lay.NH, self.pl_exp['Wh'])
                ops.split_planes_h2(Wh, lay.NH, lay.hL, lay.NH, self.pl['Wh'], self.pl_exp['Wh'])
                ops.gemm_h2(0, 0, B, lay.NH, _r16(lay.hL), self.pl['H'], self.pl['Wh'], self.A, lay.ldA,
                            exp_a=self.pl_exp['H'], exp_b=self.pl_exp['Wh'], bias=bh, ws=self.ws)
            elif self.pl is not None and self._wide_planes(B) and B <= self.pl['H'].shape[1]:
                ops.split_planes(self._hl[0], self._hl[1], B, lay.hL, self.pl['H'])
                ops.split_planes(Wh, lay.NH, lay.hL, lay.NH, self.pl['Wh'])
                ops.gemm_p3(0, 0, B, lay.NH, _r16(lay.hL), self.pl['H'], self.pl['Wh'], self.A, lay.ldA, bias=bh, ws=self.ws)
            elif self.WhT is not None and B >= 256:
                ops.transpose(Wh, lay.NH, lay.hL, lay.NH, self.WhT, self.WhT.shape[1])
                for c0, nc, h0 in self._head_blocks():
                    ops.sgemm(0, 1, B, nc, lay.hL, self._hl[0][:, h0:], self._hl[1], self.WhT[c0:], self.WhT.shape[1],
                              self.A[:, c0:], lay.ldA, bias=bh[c0:], ws=self.ws)
            else:
                for c0, nc, h0 in self._head_blocks():
                    ops.sgemm(0, 0, B, nc, lay.hL, self._hl[0][:, h0:], self._hl[1], Wh[:, c0:], lay.NH,
                              self.A[:, c0:], lay.ldA, bias=bh[c0:], ws=self.ws)
        if lay.elempi:               # m = -(Dense output) in place; dropout logit = k m + c
            ops.elempi_fwd(self._plane(self.A, 'mean'), lay.ldA, lay.view(self.w, 'pi_k'), lay.view(self.w, 'pi_c'),
                           B, lay.G_out, self._plane(self.A, 'pi'), lay.ldA)
        for hd in lay.shared:        # Dense(1) pre-activation -> the plane the loss / inference kernels read
            c0, _ = lay.head_cols(hd)
            ops.bcast_cols(self.A[:, c0:], lay.ldA, B, lay.G_out, self._plane(self.A, hd), lay.ldA)

    def _head_blocks(self):
        """(first column, columns, first input unit) of each GEMM of the heads' Dense block: one for the
        whole block, or one per head when every head has its private decoder layer (*-fork)."""
        lay = self.lay
        if not lay.fork:
            return [(0, lay.NH, 0)]
        return [(j * lay.Gp, lay.Gp, j * lay.hfork) for j in range(lay.fork)]

    def _plane(self, buf, head):
        lay = self.lay
        if head not in lay.heads:
            return None
        return buf[:, lay.plane_offset(head):]

    def _nll(self, B, perm, cursor, Y, sf, inv_n, grad):
        lay, ops = self.lay, self.ops
        A, D = self.A, self.D
        tw = lay.view(self.w, 'theta_w') if lay.const_disp else None
        if grad and self._h2(B):
            # wide networks: the gradient planes leave K-ZINB as the two fp16 pieces of g 2^d_exp the backward products read
            return ops.zinb_nll_planes_h2(self._plane(A, 'mean'), self._plane(A, 'disp'), self._plane(A, 'pi'), lay.ldA, tw,
                                          Y, self.ldy, sf, perm, cursor, B, lay.G_out, self.ridge, inv_n, self.flags, self.d_exp,
                                          self.pl['D'], lay.plane_offset('mean'),
                                          0 if lay.const_disp else lay.plane_offset('disp'),
                                          lay.plane_offset('pi') if 'pi' in lay.heads else 0,
                                          self.Dth, self.ldD, self.partials)
        if grad and self._planes_nll(B):
            # ... or as the three bf16 pieces
            return ops.zinb_nll_planes(self._plane(A, 'mean'), self._plane(A, 'disp'), self._plane(A, 'pi'), lay.ldA, tw,
                                       Y, self.ldy, sf, perm, cursor, B, lay.G_out, self.ridge, inv_n, self.flags,
                                       self.pl['D'], lay.plane_offset('mean'),
                                       0 if lay.const_disp else lay.plane_offset('disp'),
                                       lay.plane_offset('pi') if 'pi' in lay.heads else 0,
                                       self.Dth, self.ldD, self.partials)
        d_disp = (self.Dth if lay.const_disp else self._plane(D, 'disp')) if grad else None
        n = ops.zinb_nll(self._plane(A, 'mean'), self._plane(A, 'disp'), self._plane(A, 'pi'),
                         lay.ldA, tw, Y, self.ldy, sf, perm, cursor, B, lay.G_out, self.ridge,
                         inv_n, self.flags, self._plane(D, 'mean') if grad else None, d_disp,
                         self._plane(D, 'pi') if grad else None, self.ldD if grad else 0,
                         self.partials)
        if grad and lay.elempi:
            gk, gc = lay.view(self.g, 'pi_k'), lay.view(self.g, 'pi_c')
            ops.elempi_bwd(self._plane(A, 'mean'), lay.ldA, self._plane(D, 'mean'), self._plane(D, 'pi'), self.ldD,
                           lay.view(self.w, 'pi_k'), B, lay.G_out, gk, gc, self.ws_elempi)
            if self.sharedpi:        # one scalar pair: every (identical) entry receives the total gradient
                gk[:lay.G_out] = gk[:lay.G_out].sum()
                gc[:lay.G_out] = gc[:lay.G_out].sum()
        if grad:
            for hd in lay.shared:    # gradient of the Dense(1) unit = its plane summed over the genes
                c0, _ = lay.head_cols(hd)
                ops.row_sums_strided(self._plane(D, hd), self.ldD, B, lay.G_out, D[:, c0:], self.ldD)
        return n

    def _not_capturing(self, what):
        if self.dev.type == 'cuda' and torch.cuda.is_current_stream_capturing():
            raise RuntimeError('host -> device copy inside a graph capture (%s): run one eager step of this shape first' % what)

    def set_world_counts(self, world_counts):
        """Data parallel: the rows every rank holds in the coming step(s) (SyncBN weights, dropout row offset).  The device
        copy is refreshed only when the counts change -- and never inside a graph capture: a step runner that replays
        captured steps calls this before the replay (train.py::_StepRunner)."""
        key = tuple(world_counts)
        if self._counts_world_key != key:
            cw = self._counts_world.get(key)
            if cw is None:                               # (one buffer per distinct tuple, written once: see _batch_moments)
                self._not_capturing('per-rank row counts not seen before')
                cw = torch.as_tensor(world_counts, dtype=torch.float32).to(self.dev)
                self._counts_world[key] = cw
            self.counts_world = cw
            self._counts_world_key = key
        self.row0 = int(sum(world_counts[:self.comm.rank]))

    # ------------------------------------------------------------------ one training step
    def assert_finite(self, where=''):
        """--debug (dca/__main__.py:111-113): the reference compiles tf.verify_tensor_all_finite on y_pred / t1 / t2 into the
        loss (dca/loss.py:90-100) and fails the step that produces an inf / nan.  Here: the batch loss, every gradient and
        every parameter of the step just taken (ONE host synchronisation per step: a debugging mode).  A non-finite
        y_pred, t1 or t2 of any element makes the batch loss or a head gradient non-finite, so nothing the reference's check
        catches passes this one."""
        lay = self.lay
        bad = []
        if not bool(torch.isfinite(self.g[lay.P]).item()):
            bad.append('loss')
        if not bool(torch.isfinite(self.g[:lay.P]).all().item()) or not bool(torch.isfinite(self.w[:lay.P]).all().item()):
            for name in lay.seg:
                for flat, kind in ((self.g, 'gradient of '), (self.w, '')):
                    if not bool(torch.isfinite(lay.view(flat, name)).all().item()):
                        bad.append(kind + name)
        if bad:
            raise FloatingPointError('dca: %s has inf/nans%s' % (', '.join(bad), (' (' + where + ')') if where else ''))

    def train_step(self, B, B_global=None, world_counts=None, rows_per_slot=None):
        """Forward + backward + clipvalue/RMSprop on the B rows perm[cursor : cursor+B].
        Asynchronous: no host synchronisation (single GPU).  B may be 0 on a rank whose shard
        is exhausted in the last step of a data-parallel epoch."""
        lay, ops, comm = self.lay, self.ops, self.comm
        Bg = B if B_global is None else B_global
        inv_n = 1.0 / (float(Bg) * lay.G_out)
        w, g = self.w, self.g
        if comm.dp:
            self.set_world_counts(world_counts)
        if B > 0:
            self._forward_backward(B, Bg, inv_n)
        else:
            self._empty_step()
        if comm.dp and self._use_sharded_opt():
            # reduce-scatter -> this rank's shard of clip + RMSprop -> all-gather of the updated parameters
            sh = self.flat_len // comm.world
            lo = comm.rank * sh
            comm.all_reduce_sum(g[lay.P:lay.P + 1])                  # the batch loss (its slot sits in the last shard)
            gs = comm.reduce_scatter_sum(g, self._g_shard(sh))
            n_upd = max(0, min(sh, lay.P - lo))                       # parameters of the shard (the tail is padding / loss)
            if n_upd > 0:
                with self._t('rmsprop_clip'):
                    ops.rmsprop_clip(w[lo:lo + sh], gs, self.ms[lo:lo + sh], n_upd, self.lr, RMS_RHO, RMS_EPS, self.clip)
            comm.all_gather_into(w, w[lo:lo + sh])
            if self.has_dropout:
                ops.counter_add(self.drop_iter, 1)
            ops.step_end(g[lay.P:], float(Bg), self.hist, rows_per_slot or max(self.Bmax, 1), self.acc, self.cursor, B)
            return
        if comm.dp:
            # bucket 2: hidden layers; bucket 1 (heads + loss) has been travelling since the heads'
            # backward finished (_launch_heads_bucket)
            comm.all_reduce_sum(g[:lay.seg['Wh'][0]])
            if self._pending is not None:
                comm.wait(self._pending)
                self._pending = None
        if self.reg is not None:           # after the exchange: every rank adds the same terms once
            ops.l1l2_apply(self.reg, w, g, g[lay.P:], self.reg_ws)
        fused_end = self.opt_kind == 'rmsprop' and not self.has_dropout and hasattr(ops, 'rmsprop_clip_end')
        if fused_end:            # optimizer + end-of-step bookkeeping in one launch
            with self._t('rmsprop_clip'):
                ops.rmsprop_clip_end(w, g, self.ms, lay.P, self.lr, RMS_RHO, RMS_EPS, self.clip, g[lay.P:], float(Bg),
                                     self.hist, rows_per_slot or max(self.Bmax, 1), self.acc, self.cursor, B)
            return
        if self.opt_kind == 'rmsprop':
            with self._t('rmsprop_clip'):
                ops.rmsprop_clip(w, g, self.ms, lay.P, self.lr, RMS_RHO, RMS_EPS, self.clip)
        elif self.opt_kind == 'nadam':
            ops.nadam_step(w, g, self.ms, self.slot2, lay.P, self.lr, self.opt_iter, self.m_sched, self.clip)
            ops.counter_add(self.opt_iter, 1)
        else:
            ops.optimizer_step(self.opt_kind, w, g, None if self.opt_kind == 'sgd' else self.ms,
                               self.slot2, lay.P, self.lr, self.opt_iter, self.clip)
            ops.counter_add(self.opt_iter, 1)
        if self.has_dropout:
            ops.counter_add(self.drop_iter, 1)
        ops.step_end(g[lay.P:], float(Bg), self.hist, rows_per_slot or max(self.Bmax, 1), self.acc,
                     self.cursor, B)

    def _heads_compact(self):
        kw = {'compact': self.cc} if self.cc is not None else {}
        if getattr(self, 'heads_d_exp', 0):
            kw['d_exp'] = self.heads_d_exp          # (large counts throughout: _heads_d_exp)
        return kw

    def _use_sharded_opt(self):
        return self.sharded_opt and self.opt_kind == 'rmsprop' and self.reg is None

    def _g_shard(self, sh):
        if getattr(self, '_gshard', None) is None or self._gshard.numel() != sh:
            self._gshard = torch.zeros(sh, dtype=torch.float32, device=self.dev)
        return self._gshard

    def _launch_heads_bucket(self):
        """Data parallel: all-reduce of g[Wh .. P] (head weights, biases, log-dispersion, batch
        loss) starts now, asynchronously."""
        if self.comm.dp and not self._use_sharded_opt():
            lay = self.lay
            self._pending = self.comm.all_reduce_sum_async(self.g[lay.seg['Wh'][0]:lay.P + 1])

    def _empty_step(self):
        lay = self.lay
        self.g.zero_()
        for i, h in enumerate(lay.hidden):
            if lay.batchnorm:      # take part in the statistics exchange, update moving averages
                entries, cnts, E = self._batch_moments(i, 0, h, self.counts_world)
                self.ops.bn_relu_apply(self.Z[i], self.ldh[i], 0, h, entries, cnts, E,
                                       lay.view(self.w, 'beta%d' % i), self.mm[i], self.mv[i],
                                       BN_MOMENTUM, BN_EPS, self.act, self.H[i], self.ldh[i], None, 0, None)
        self._launch_heads_bucket()
        for i in reversed(range(len(lay.hidden))):
            if lay.batchnorm:
                s = torch.zeros(2 * lay.hidden[i], dtype=torch.float32, device=self.dev)
                self.comm.all_reduce_sum(s)

    def heads_fused_launch(self, B, KL, inv_n):
        """K-HEADS on the decoder output the last forward pass left (ops.heads_fused: the H split, the fused kernel, the reduce).
        Idempotent -- it overwrites the heads' gradient block, the input gradient and the loss slot -- so bench.py can replay it
        from a graph to time the operation as the product launches it."""
        lay, w, g = self.lay, self.w, self.g
        self.ops.heads_fused(self._hl[0], self._hl[1], lay.view(w, 'Wh'), lay.NH,
                             lay.view(w, 'bh'), lay.Gp,
                             lay.view(w, 'theta_w') if lay.const_disp else None, self.Y,
                             self.ldy, self.sf, self.perm, self.cursor, B, KL, lay.G_out,
                             self.ridge, inv_n, self.flags, lay.view(g, 'Wh'), lay.NH,
                             lay.view(g, 'theta_w') if lay.const_disp else None,
                             self._dhl, self._hl[1], self.partials, self.ws_heads,
                             tile_order=self.tile_order, loss_out=g[lay.P:], **self._heads_compact())

    def _forward_backward(self, B, Bg, inv_n):
        lay, ops, comm = self.lay, self.ops, self.comm
        w, g = self.w, self.g
        KL = self._hidden_forward(B, ('perm',), True, self.counts_world)
        self._last_heads_args = (B, KL, inv_n)
        if self.ws_heads is not None:
            with self._t('heads_fused'):
                self.heads_fused_launch(B, KL, inv_n)
        else:
            self._heads_backward_unfused(B, KL, inv_n)
        self._launch_heads_bucket()
        # ---- backward: hidden stack
        L = len(lay.hidden)
        chain = self._stack_chain(B)
        sync = self._stack_sync(B)
        coop = chain or sync or self._stack_coop(B)
        if coop:
            layers = []
            for i, h in enumerate(lay.hidden):
                d = dict(H=h, Hact=self.H[i], ldh=self.ldh[i], xhat=self.XH[i], ldx=self.ldh[i], inv_std=self.inv_std[i],
                         dbeta=lay.view(g, 'beta%d' % i), dH=self.dH[i], lddh=self.ldh[i])
                if i > 0:
                    d.update(W=lay.view(w, 'W%d' % i), ldw=h, K=lay.hidden[i - 1], Hprev=self.H[i - 1], ldp=self.ldh[i - 1],
                             gW=lay.view(g, 'W%d' % i), ldg=h)
                layers.append(d)
            if sync:
                # data parallel: the two batch sums of a layer all-reduced between two launches; d beta stays the local share
                ops.hidden_stack_bwd_sync(layers, B, float(Bg), self.act, self.dZ[0], self.ldh[0], 0, None, self.bsum[L - 1],
                                          self.ws_stack)
                comm.all_reduce_sum(self.bsum[L - 1])
                for st in range(1, L + 1):
                    i = L - st
                    ops.hidden_stack_bwd_sync(layers, B, float(Bg), self.act, self.dZ[0], self.ldh[0], st, self.bsum[i],
                                              self.bsum[i - 1] if i > 0 else None, self.ws_stack)
                    if i > 0:
                        comm.all_reduce_sum(self.bsum[i - 1])
                ops.hidden_stack_bwd(layers, B, float(Bg), self.act, self.dZ[0], self.ldh[0], self.ws_stack,
                                     rows_per_wg=32, steps=(L + 1, L + 1))
            elif chain:
                ops.hidden_stack_bwd(layers, B, float(Bg), self.act, self.dZ[0], self.ldh[0], None, rows_per_wg=64)
            elif self.stack_mode == 'coop' and B <= 256 * 64:
                ops.hidden_stack_bwd(layers, B, float(Bg), self.act, self.dZ[0], self.ldh[0], self.ws_stack, rows_per_wg=64)
            else:
                for st in range(L + 2):
                    ops.hidden_stack_bwd(layers, B, float(Bg), self.act, self.dZ[0], self.ldh[0], self.ws_stack,
                                         rows_per_wg=self._stack_rows(B), steps=(st, st))
        for i in reversed(range(L)):
            if coop and i > 0:
                continue                # K-STACK wrote d beta, the weight / bias gradients and (for layer 0) dZ
            h = lay.hidden[i]
            if not coop:        # (K-STACK has produced dZ of the first layer already)
                if self.drop[i] > 0.0:      # gradient through the dropout of this layer's output: same mask
                    ops.dropout_apply(self.dH[i], self.ldh[i], None, None, B, h, self.drop[i], self.drop_seed,
                                      self.drop_iter, i, self.row0, self.dH[i], self.ldh[i])
                if self.prelu:              # dL/d(PReLU out) -> dL/d(its input) in place, slope gradients
                    ops.prelu_bwd(self.dH[i], self.ldh[i], self.H[i], self.ldh[i], lay.view(w, 'alpha%d' % i), B, h,
                                  lay.view(g, 'alpha%d' % i), self.ws_prelu)
                if self._layer_small(B, i):
                    # the layer's whole backward in one launch: d beta, dZ, weight / bias gradient, input gradient
                    Kp = lay.hidden[i - 1]
                    ops.dense_bn_bwd_small(self.dH[i], self.ldh[i], self.H[i], self.ldh[i],
                                           self.XH[i] if lay.batchnorm else None, self.ldh[i], self.inv_std[i],
                                           self.Hcur[i - 1], self.ldh[i - 1], lay.view(w, 'W%d' % i), h, B, Kp, h,
                                           lay.batchnorm, float(Bg), self.act, lay.view(g, 'W%d' % i), h,
                                           lay.view(g, 'beta%d' % i) if lay.batchnorm else None,
                                           self.dH[i - 1], self.ldh[i - 1])
                    continue
                if lay.batchnorm and self._bn_small(B):
                    ops.bn_bwd_small(self.dH[i], self.ldh[i], self.H[i], self.ldh[i], self.XH[i], self.ldh[i],
                                     self.inv_std[i], float(Bg), B, h, self.dZ[i], self.ldh[i],
                                     lay.view(g, 'beta%d' % i), self.act)
                elif lay.batchnorm:
                    ops.bn_bwd_sums(self.dH[i], self.ldh[i], self.H[i], self.ldh[i], self.XH[i],
                                    self.ldh[i], B, h, self.bpart[i], self.act)
                    E = ops.col_moments_chunks(B)
                    sums, dbeta = self.bpart[i], lay.view(g, 'beta%d' % i)
                    if comm.dp:
                        # SyncBN backward: the apply kernel needs the GLOBAL sums; d beta must stay this rank's LOCAL
                        # share (the gradient bucket is summed over the ranks afterwards)
                        sums, E = self._reduce_bwd_sums(i, E, h, dbeta), 1
                        dbeta = None
                    ops.bn_bwd_apply(self.dH[i], self.ldh[i], self.H[i], self.ldh[i], self.XH[i],
                                     self.ldh[i], self.inv_std[i], sums, E, float(Bg), B, h,
                                     self.dZ[i], self.ldh[i], dbeta, self.act)
                else:
                    ops.relu_bwd(self.dH[i], self.ldh[i], self.H[i], self.ldh[i], B, h, self.dZ[i],
                                 self.ldh[i], self.act)
            Kp = lay.G_in if i == 0 else lay.hidden[i - 1]
            gW = lay.view(g, 'W%d' % i)
            if i == 0:
                with self._t('gemm_enc0_dW'):
                    if self._sparse_dw(B):
                        if self.cc_in.lutp is None:              # the per-cell table of the common counts: first use only
                            self._not_capturing('first use of the byte-store weight gradient')
                            self.cc_in.ensure_lut(ops)
                        ops.enc0_dw_sparse(self.cc_in, self.perm, self.cursor, 0, B, Kp, h, self.dZ[0], self.ldh[0], gW, h,
                                           self.ws_enc0)
                    elif self._h2_enc0(B, True):
                        ops.absmax_exp(self.dZ[0], self.ldh[0], B, h, self.pl_exp['dZ0'])
                        ops.split_planes_h2(self.dZ[0], self.ldh[0], B, h, self.pl['dZ0'], self.pl_exp['dZ0'])
                        ops.gemm_h2(1, 0, Kp, h, B, self.pl['X'], self.pl['dZ0'], gW, h, exp_a=self.x_exp,
                                    exp_b=self.pl_exp['dZ0'], colsum_row=True, ws=self.ws)
                    elif self._planes_enc0(B, True):
                        # the forward's planes of the minibatch, contracted over their rows; bias gradient = column sums
                        ops.split_planes(self.dZ[0], self.ldh[0], B, h, self.pl['dZ0'])
                        ops.gemm_p3(1, 0, Kp, h, B, self.pl['X'], self.pl['dZ0'], gW, h, colsum_row=True, ws=self.ws)
                    elif self.in_drop > 0.0:
                        ops.sgemm(1, 0, Kp, h, B, self.Xb, self.ldx, self.dZ[0], self.ldh[0], gW, h,
                                  colsum_row=True, ws=self.ws)
                    elif self.XT is not None and B >= 256:
                        # X^T dZ with BOTH operands contiguous along the batch (the fastest operand path of K-GEMM): the
                        # gathered minibatch and the (small) dZ transposed; the bias gradient = row sums of dZ^T
                        ops.transpose(self.X, self.ldx, B, Kp, self.XT, self.ldb_t, perm=self.perm, cursor=self.cursor)
                        ops.transpose(self.dZ[0], self.ldh[0], B, h, self.dZT, self.ldb_t)
                        ops.sgemm(0, 1, Kp, h, B, self.XT, self.ldb_t, self.dZT, self.ldb_t, gW, h, ws=self.ws)
                        ops.row_sums_strided(self.dZT, self.ldb_t, h, B, lay.view(g, 'b0'), 1)
                    else:
                        ops.sgemm(1, 0, Kp, h, B, self.X, self.ldx, self.dZ[0], self.ldh[0], gW, h,
                                  perm=self.perm, cursor=self.cursor, colsum_row=True, ws=self.ws)
            else:
                ops.sgemm(1, 0, Kp, h, B, self.Hcur[i - 1], self.ldh[i - 1], self.dZ[i], self.ldh[i], gW,
                          h, colsum_row=True, ws=self.ws)
                ops.sgemm(0, 1, B, Kp, h, self.dZ[i], self.ldh[i], lay.view(w, 'W%d' % i), h,
                          self.dH[i - 1], self.ldh[i - 1], ws=self.ws)

    def _heads_backward_unfused(self, B, KL, inv_n):
        """Heads as separate launches: GEMM forward, K-ZINB, weight-gradient GEMM (+ bias column
        sums), per-gene dispersion chain, input-gradient GEMM."""
        lay, ops = self.lay, self.ops
        w, g = self.w, self.g
        self._heads_forward(B, KL)
        with self._t('zinb_nll'):
            n = self._nll(B, self.perm, self.cursor, self.Y, self.sf, inv_n, True)
        ops.loss_finalize(self.partials, n, inv_n, g[lay.P:])
        gWh, Wh = lay.view(g, 'Wh'), lay.view(w, 'Wh')
        if self.pl is not None and self._wide_planes(B) and B <= self.pl['H'].shape[1] and self._h2(B):
            # fp16 x 2 planes: the forward's H and Wh planes, the likelihood's D = g 2^d_exp; 1 / n leaves in the epilogues
            with self._t('gemm_heads_dW'):
                ops.gemm_h2(1, 0, lay.hL, lay.NH, B, self.pl['H'], self.pl['D'], gWh, lay.NH, exp_a=self.pl_exp['H'],
                            exp_b_add=self.d_exp, alpha=inv_n, colsum_row=True, ws=self.ws)
            if lay.const_disp:
                ops.colsum_chain(self.Dth, self.ldD, B, lay.G_out, lay.view(w, 'theta_w'), lay.view(g, 'theta_w'))
            with self._t('gemm_heads_dH'):
                ops.gemm_h2(0, 1, B, lay.hL, _r16(lay.NH), self.pl['D'], self.pl['Wh'], self._dhl, self._hl[1],
                            exp_a_add=self.d_exp, exp_b=self.pl_exp['Wh'], alpha=inv_n, ws=self.ws)
            return
        if self.pl is not None and self._wide_planes(B) and B <= self.pl['H'].shape[1]:
            # the gradient planes split once for both products; pl['H'] and pl['Wh'] are the forward's
            with self._t('gemm_heads_dW'):
                if not self._planes_nll(B):
                    ops.split_planes(self.D, self.ldD, B, lay.NH, self.pl['D'])
                ops.gemm_p3(1, 0, lay.hL, lay.NH, B, self.pl['H'], self.pl['D'], gWh, lay.NH, colsum_row=True, ws=self.ws)
            if lay.const_disp:
                ops.colsum_chain(self.Dth, self.ldD, B, lay.G_out, lay.view(w, 'theta_w'), lay.view(g, 'theta_w'))
            with self._t('gemm_heads_dH'):
                ops.gemm_p3(0, 1, B, lay.hL, _r16(lay.NH), self.pl['D'], self.pl['Wh'], self._dhl, self._hl[1], ws=self.ws)
            return
        with self._t('gemm_heads_dW'):
            if self.HT is not None and B >= 256:
                hin = self.HT.shape[0]
                ops.transpose(self._hl[0], self._hl[1], B, hin, self.HT, self.ldb_t)
                for c0, nc, h0 in self._head_blocks():
                    ops.sgemm(0, 0, lay.hL, nc, B, self.HT[h0:], self.ldb_t, self.D[:, c0:], self.ldD,
                              gWh[:, c0:], lay.NH, colsum_row=True, ws=self.ws)
            else:
                for c0, nc, h0 in self._head_blocks():      # colsum_row: the bias gradient lands in row hL = 'bh'
                    ops.sgemm(1, 0, lay.hL, nc, B, self._hl[0][:, h0:], self._hl[1], self.D[:, c0:], self.ldD,
                              gWh[:, c0:], lay.NH, colsum_row=True, ws=self.ws)
        if lay.const_disp:
            ops.colsum_chain(self.Dth, self.ldD, B, lay.G_out, lay.view(w, 'theta_w'),
                             lay.view(g, 'theta_w'))
        with self._t('gemm_heads_dH'):
            for c0, nc, h0 in self._head_blocks():
                ops.sgemm(0, 1, B, lay.hL, nc, self.D[:, c0:], self.ldD, Wh[:, c0:], lay.NH,
                          self._dhl[:, h0:], self._hl[1], ws=self.ws)

    def _reduce_bwd_sums(self, i, E, h, dbeta_local):
        """SyncBN backward: local chunk sums -> one [2h] vector (its first half = this rank's d beta, stored) -> all-reduce;
        returns the global [sum dy | sum dy xhat]."""
        s = self.bpart[i][:E * 2 * h].view(E, 2 * h).sum(dim=0)
        dbeta_local.copy_(s[:h])
        self.comm.all_reduce_sum(s)
        return s

    # ------------------------------------------------------------------ evaluation / inference
    def eval_loss_sum(self, r0, r1, scale, chunk=None):
        """Adds scale * sum of element-wise NLL over storage rows [r0, r1) (inference mode) to
        acc[1] (Keras validation pass, train.py:97)."""
        lay, ops = self.lay, self.ops
        chunk = chunk or self.Bmax
        for s in range(r0, r1, chunk):
            b = min(chunk, r1 - s)
            KL = self._hidden_forward(b, ('range', s), False)
            self._heads_forward(b, KL)
            n = self._nll(b, None, None, self.Y[s:], self.sf[s:], 1.0, False)
            ops.loss_finalize(self.partials, n, scale, self.val_loss_tmp)
            ops.step_end(self.val_loss_tmp, 1.0, None, 0, self.acc[1:], None, 0)

    def predict_chunk(self, r0, b, want):
        """Inference forward over storage rows [r0, r0+b).  Returns device views (valid until
        the next call): dict with 'mean' (mean*sf), 'dispersion', 'dropout', 'latent'."""
        lay, ops = self.lay, self.ops
        KL = self._hidden_forward(b, ('range', r0), False)
        out = {}
        if 'latent' in want:
            out['latent'] = self.Z[self.center][:b, :lay.hidden[self.center]]
        if want - {'latent'}:
            self._heads_forward(b, KL)
            A = self.A
            m = self._plane(A, 'mean')
            d = self._plane(A, 'disp') if 'dispersion' in want else None
            p = self._plane(A, 'pi') if 'dropout' in want else None
            ops.heads_infer(m, d, p, lay.ldA, self.sf[r0:], b, lay.G_out,
                            m if 'mean' in want else None, d, p, lay.ldA, self.flags & 8)
            if 'mean' in want:
                out['mean'] = m[:b, :lay.G_out]
            if d is not None:        # shared heads: one value per cell ([n, 1], what the Dense(1) sub-model predicts)
                out['dispersion'] = d[:b, :1 if 'disp' in lay.shared else lay.G_out]
            if p is not None:
                out['dropout'] = p[:b, :1 if 'pi' in lay.shared else lay.G_out]
        return out

    # ------------------------------------------------------------------ inference, gene-major (the fused result writer)
    def hidden_all(self, chunk=4096):
        """Inference forward of the hidden stack over ALL storage rows: the decoder output of every cell stays resident
        ([n, ldh]: 17 MB at 68 579 cells), the latent code is returned ([n, h_centre] device tensor)."""
        lay = self.lay
        n = self.n
        self.reserve(min(chunk, n))
        hL = lay.hidden[-1]
        self.HL_all = torch.zeros(n, self.ldh[-1], dtype=torch.float32, device=self.dev)
        latent = torch.zeros(n, lay.hidden[self.center], dtype=torch.float32, device=self.dev)
        for s in range(0, n, self.Bmax):
            b = min(self.Bmax, n - s)
            self._hidden_forward(b, ('range', s), False)
            self.HL_all[s:s + b].copy_(self.Hcur[-1][:b])
            latent[s:s + b].copy_(self.Z[self.center][:b, :lay.hidden[self.center]])
        return latent

    def heads_gene_block(self, g0, gb, want, out):
        """The heads of genes [g0, g0 + gb) for ALL cells, written gene-major: out[k] [gb, >= n] device tensors for k in
        want (mean = mean * size factor, dispersion, dropout) -- a [n, hL] x [hL, gb] product per head, the inference
        activations, one transpose.  Needs hidden_all() first; plain per-gene Dense heads only."""
        lay, ops = self.lay, self.ops
        assert not (lay.shared or lay.fork or lay.elempi)
        n = self.n
        Wh, bh = lay.view(self.w, 'Wh'), lay.view(self.w, 'bh')
        gbp = _r4(gb)
        nh = len(lay.planes)
        if getattr(self, '_Ablk', None) is None or self._Ablk.shape[1] < nh * gbp or self._Ablk.shape[0] != n:
            self._Ablk = torch.zeros(n, nh * gbp, dtype=torch.float32, device=self.dev)
            self._ws_blk = torch.zeros(max(ops.sgemm_workspace_bytes(0, 0, n, gbp, lay.hL) // 4, 4), dtype=torch.float32,
                                       device=self.dev)
        A = self._Ablk
        lda = A.shape[1]
        planes = {}
        for k, hd in enumerate(lay.planes):
            c0 = k * lay.Gp + g0
            ops.sgemm(0, 0, n, gb, lay.hL, self.HL_all, self.ldh[-1], Wh[:, c0:], lay.NH, A[:, k * gbp:], lda,
                      bias=bh[c0:], ws=self._ws_blk)
            planes[hd] = A[:, k * gbp:]
        m = planes.get('mean')
        d = planes.get('disp') if 'dispersion' in want else None
        p = planes.get('pi') if 'dropout' in want else None
        ops.heads_infer(m, d, p, lda, self.sf, n, gb, m if 'mean' in want else None, d, p, lda, self.flags & 8)
        for key, src in (('mean', m), ('dispersion', d), ('dropout', p)):
            if key in want and src is not None:
                ops.transpose(src, lda, n, gb, out[key], out[key].shape[1])

    def const_dispersion(self):
        """layers.py:21: theta = clip(exp(w), 1e-3, 1e4), per gene."""
        tw = self.lay.view(self.w, 'theta_w')[:self.lay.G_out]
        return torch.clamp(torch.exp(tw), 1e-3, 1e4).cpu().numpy()


