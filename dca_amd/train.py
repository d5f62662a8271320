"""Training driver: the reference's ``train()`` (dca/train.py:35-100) on the device engine.

Same signature, same defaults, same return protocol (an object with ``.history`` holding
``loss`` / ``val_loss`` / ``lr`` lists).  What Keras' ``model.fit`` did per batch on the host
(slice index_array, copy the batch, session.run) is replaced by: the whole dataset resident in
HBM, one int32 permutation upload per epoch, and an asynchronous stream of kernel launches --
the host only reads two scalars back per epoch.

Keras semantics restated here (source not vendored in the reference):
validation_split takes the LAST rows before shuffling; each epoch shuffles a fresh arange with
the numpy global RNG; the last partial batch is kept; the epoch loss is the sample-weighted
mean of the batch losses; ReduceLROnPlateau(factor .1, min_delta 1e-4) and EarlyStopping
(min_delta 0) both monitor val_loss.
"""
import math
import os

import numpy as np
import torch

from . import dist as ddist


KERAS_DEFAULT_LR = {'sgd': 0.01, 'rmsprop': 0.001, 'adagrad': 0.001, 'adadelta': 0.001, 'adam': 0.001,
                    'adamax': 0.001, 'nadam': 0.001}


class History:
    """Stand-in for keras.callbacks.History (api.py:205-206 reads ``.history``)."""

    def __init__(self):
        self.history = {'loss': [], 'val_loss': [], 'lr': []}
        self.epoch = []
        self.stopped_epoch = None


class _ReduceLROnPlateau:     # keras defaults as used at train.py:70-72
    def __init__(self, patience, verbose=False):
        self.patience, self.factor, self.min_delta, self.min_lr = patience, 0.1, 1e-4, 0.0
        self.best, self.wait, self.verbose = np.inf, 0, verbose

    def step(self, epoch, val, lr):
        if val < self.best - self.min_delta:
            self.best, self.wait = val, 0
            return lr
        self.wait += 1
        if self.wait >= self.patience:
            if lr > self.min_lr:
                lr = float(np.float32(max(lr * self.factor, self.min_lr)))
                if self.verbose:
                    print('\nEpoch %05d: ReduceLROnPlateau reducing learning rate to %s.' % (epoch + 1, lr))
            self.wait = 0
        return lr


class _EarlyStopping:         # keras defaults as used at train.py:73-75
    def __init__(self, patience):
        self.patience, self.best, self.wait = patience, np.inf, 0

    def step(self, val):
        if val < self.best:
            self.best, self.wait = val, 0
            return False
        self.wait += 1
        return self.wait >= self.patience


def fit_engine(eng, n_train_global, n_val_global, nt_local, nv_local, t0_global, *, epochs=300,
               batch_size=32, learning_rate=None, clip_grad=5.0, reduce_lr=10, early_stop=15,
               verbose=False, shuffle_rng=None, use_graph=None, on_epoch=None, state=None, debug=False):
    """Runs the Keras-equivalent fit loop on an engine whose storage rows are
    [0, nt_local) = this rank's train shard and [nt_local, nt_local+nv_local) = its
    validation shard.  Returns History.

    ``state`` (a dict from a previous call's ``on_epoch`` hook: 'epoch', 'lr', callback
    counters, history) resumes the loop after that epoch: the engine must already hold the
    matching weights / optimizer slots (Engine.load_state).  NOTE: the shuffles consume the numpy
    RNG stream; a resumed run replays them for the skipped epochs so that the order of the
    remaining epochs is the one an uninterrupted run would have seen.

    ``debug`` (dca/__main__.py:111-113 -> dca/loss.py:90-100, the finite checks compiled into the reference's loss): every step
    runs eagerly and is followed by Engine.assert_finite -- the batch loss, every gradient, every parameter -- which raises
    FloatingPointError naming the tensors and the (epoch, step) of the first non-finite value."""
    comm = eng.comm
    W = comm.world
    rng = np.random if shuffle_rng is None else shuffle_rng
    if batch_size % W != 0:
        raise ValueError('batch_size (%d) must be a multiple of the number of GPUs (%d)' % (batch_size, W))
    b_local = batch_size // W
    nt_all = [ddist.shard(n_train_global, W, r)[1] for r in range(W)]
    steps = int(math.ceil(max(nt_all) / float(b_local))) if max(nt_all) > 0 else 0
    counts = [[int(min(max(nt_all[r] - t * b_local, 0), b_local)) for r in range(W)] for t in range(steps)]
    eng.clip = float(clip_grad) if clip_grad else 0.0
    lr = float(np.float32(0.001 if learning_rate is None else learning_rate))
    eng.set_lr(lr)
    eng.reserve(max(b_local, min(1024, max(nv_local, 1))))     # validation runs in big chunks
    dev = eng.dev
    eng.perm = torch.zeros(max(nt_local, 1), dtype=torch.int32, device=dev)
    eng.hist = torch.zeros(steps + 1, dtype=torch.float32, device=dev)
    G = eng.lay.G_out
    val_scale = 1.0 / (float(n_val_global) * G) if n_val_global > 0 else 0.0
    rl = _ReduceLROnPlateau(reduce_lr, verbose) if reduce_lr else None
    es = _EarlyStopping(early_stop) if early_stop else None
    hist = History()
    first_epoch = 0
    if state is not None:
        first_epoch = int(state['epoch']) + 1
        lr = float(state['lr'])
        eng.set_lr(lr)
        if rl is not None:
            rl.best, rl.wait = state['rl_best'], state['rl_wait']
        if es is not None:
            es.best, es.wait = state['es_best'], state['es_wait']
        hist.history = {k: list(v) for k, v in state['history'].items()}
        hist.epoch = list(range(first_epoch))
    runner = _StepRunner(eng, False if debug else use_graph)
    for epoch in range(epochs):
        idx = np.arange(n_train_global)
        rng.shuffle(idx)                                       # numpy global RNG, like Keras
        if epoch < first_epoch:
            continue                                           # resumed: replay the stream only
        order = ddist.local_order(idx, t0_global, nt_local)
        if nt_local > 0:
            eng.perm[:nt_local].copy_(torch.as_tensor(order), non_blocking=False)
        eng.cursor.zero_()
        eng.acc.zero_()
        t = 0
        while t < steps:                # runs of equal steps go to the runner together (it replays several per graph launch)
            n = 1
            while not debug and t + n < steps and counts[t + n] == counts[t]:
                n += 1
            runner.run(counts[t][comm.rank], sum(counts[t]), counts[t], b_local, n)
            if debug:
                eng.assert_finite('epoch %d, step %d' % (epoch + 1, t + 1))
            t += n
        if nv_local > 0:
            eng.eval_loss_sum(nt_local, nt_local + nv_local, val_scale)
        if n_val_global > 0:
            eng.add_val_penalty()
        if W > 1:
            comm.all_reduce_sum(eng.acc[1:])
        acc = eng.acc.cpu().numpy()                            # the one host sync per epoch
        if getattr(comm, 'peer', None) is not None:
            comm.peer.check()                                  # K-PEER: a rank that never arrived at an exchange fails the fit here
        loss = float(acc[0]) / n_train_global
        hist.history['loss'].append(loss)
        hist.history['lr'].append(lr)
        hist.epoch.append(epoch)
        stop = False
        if n_val_global > 0:
            val = float(acc[1])
            hist.history['val_loss'].append(val)
            if rl is not None:
                new_lr = rl.step(epoch, val, lr)
                if new_lr != lr:
                    lr = new_lr
                    eng.set_lr(lr)
            if es is not None and es.step(val):
                stop = True
        if verbose and comm.rank == 0:
            msg = 'Epoch %d/%d - loss: %.4f' % (epoch + 1, epochs, loss)
            if n_val_global > 0:
                msg += ' - val_loss: %.4f' % hist.history['val_loss'][-1]
            print(msg, flush=True)
        if on_epoch is not None:
            on_epoch(epoch, hist, dict(epoch=epoch, lr=lr, rl_best=rl.best if rl else None,
                                       rl_wait=rl.wait if rl else 0, es_best=es.best if es else None,
                                       es_wait=es.wait if es else 0,
                                       history={k: list(v) for k, v in hist.history.items()}))
        if stop:
            hist.stopped_epoch = epoch
            if verbose and comm.rank == 0:
                print('Epoch %05d: early stopping' % (epoch + 1))
            break
    if n_val_global == 0:
        del hist.history['val_loss']
    if rl is None:
        del hist.history['lr']         # Keras logs 'lr' only when ReduceLROnPlateau is among the callbacks (train.py:70-72)
    return hist


class _StepRunner:
    """Launches training steps; the step (10-25 kernels, no host sync) is captured once per distinct batch
    size into a hipGraph and replayed -- the device cursor makes the same graph valid for every batch of every epoch.
    Runs of equal steps are replayed GRAPH_STEPS at a time from a second graph holding that many consecutive steps:
    the device idles ~9 us between two graph launches, 6 % of a batch-32 step (profiles/r02w_batch32_step_trace.txt).
    Data parallel: the RCCL exchanges of the step are captured with it (every rank replays the same sequence of
    collectives; an eager data-parallel step is host-bound: ~45 launches + 8 collectives took 2.17 ms against 1.25 ms of
    kernels, profiles/r03_dp_one_rank.txt).  EngineConfig.dp_graph = False keeps the data-parallel steps eager; a capture that raises
    does the same for the rest of the run (a capture executes nothing, so ranks that did capture stay in step)."""

    def __init__(self, eng, use_graph):
        self.eng = eng
        cfg = getattr(eng, 'cfg', None)
        if cfg is None:
            from . import config as _config
            cfg = _config.current()
        self.GRAPH_STEPS = cfg.graph_steps
        if use_graph is None:
            use_graph = True
        self.use_graph = bool(use_graph) and eng.ops.device_type == 'cuda' and \
            (not eng.comm.dp or (getattr(eng.comm, 'capturable', False) and cfg.dp_graph))
        self.graphs = {}

    def _capture(self, args, k):
        g = torch.cuda.CUDAGraph()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        # thread-local capture mode: RCCL's watchdog thread polls the events of earlier (eager) collectives with
        # hipEventQuery; under the default global mode that call from ANOTHER thread is illegal while this one captures --
        # it invalidates the capture and kills the watchdog (seen once in ~10 runs of the one-rank RCCL test)
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
                for _ in range(k):
                    self.eng.train_step(*args)
        torch.cuda.current_stream().wait_stream(s)
        return g

    def _graph(self, key, args, k):
        """The graph of k consecutive steps of this shape; None (and eager from now on) if the capture raises."""
        if self.graphs.get((key, k)) is None:
            try:
                self.graphs[(key, k)] = self._capture(args, k)
            except Exception as e:                       # noqa: BLE001 -- whatever the runtime refuses, the eager step still runs
                if not self.eng.comm.dp:
                    raise
                import sys
                print('dca_amd: capture of the data-parallel step failed (%s: %s); eager steps from here'
                      % (type(e).__name__, e), file=sys.stderr)
                self.use_graph = False
                return None
        return self.graphs[(key, k)]

    def run(self, b, b_global, world_counts, rows_per_slot, n=1):
        """n consecutive steps of the same shape."""
        eng = self.eng
        args = (b, b_global, world_counts, rows_per_slot)
        dp = eng.comm.dp
        key = (b, b_global, tuple(world_counts)) if dp else b
        if dp and self.use_graph:
            eng.set_world_counts(world_counts)           # (host -> device copy when the counts change: outside the graphs)

        def eager(m):
            for _ in range(m):
                eng.train_step(*args)

        if not self.use_graph or b == 0:
            return eager(n)
        if (key, 1) not in self.graphs:
            # the first step of each shape runs eagerly (loads the code objects outside a capture)
            eng.train_step(*args)
            self.graphs[(key, 1)] = None
            n -= 1
        K = self.GRAPH_STEPS
        if K > 1 and n >= K:
            gk = self._graph(key, args, K)
            if gk is None:
                return eager(n)
            for _ in range(n // K):
                gk.replay()
            n %= K
        if n > 0:
            g1 = self._graph(key, args, 1)
            if g1 is None:
                return eager(n)
            for _ in range(n):
                g1.replay()

    def step(self, b, b_global, world_counts, rows_per_slot):
        self.run(b, b_global, world_counts, rows_per_slot, 1)


def train(adata, network, output_dir=None, optimizer='RMSprop', learning_rate=None,
          epochs=300, reduce_lr=10, output_subset=None, use_raw_as_output=True,
          early_stop=15, batch_size=32, clip_grad=5., save_weights=False,
          validation_split=0.1, tensorboard=False, verbose=True, threads=None,
          **kwds):
    """Signature and defaults of dca/train.py:35-39.  ``threads`` sized TensorFlow's CPU pools in the reference
    (train.py:41-48); here it sizes the host thread pools of the native host stages (staging copies, checksums, result
    writers: dca_amd/hostlib.py) -- the step itself runs on the GPU.  ``tensorboard`` is accepted and ignored; optimizers:
    SGD, RMSprop, Adagrad, Adadelta, Adam, Adamax, Nadam (Keras defaults)."""
    eng = network.engine
    if eng is None:
        raise RuntimeError('network.build() must be called before train()')
    if threads:
        from . import hostlib
        hostlib.set_threads(threads)
    # train.py:54-57: opt.__dict__[optimizer](lr=learning_rate, clipvalue=clip_grad), the optimizer's
    # own default learning rate when none is given
    eng.set_optimizer(optimizer)
    if learning_rate is None:
        learning_rate = KERAS_DEFAULT_LR[optimizer.lower()]
    if output_dir is not None:
        os.makedirs(output_dir, exist_ok=True)

    X = adata.X
    sf = np.asarray(adata.obs['size_factors'].values, dtype=np.float32)       # train.py:83
    if output_subset:                                                        # train.py:85-87
        gene_idx = [np.where(adata.raw.var_names == x)[0][0] for x in output_subset]
        Y = adata.raw.X[:, gene_idx] if use_raw_as_output else adata.X[:, gene_idx]
    else:
        Y = adata.raw.X if use_raw_as_output else adata.X

    n = X.shape[0]
    split_at = int(n * (1.0 - validation_split)) if validation_split and 0. < validation_split < 1. else n
    n_train, n_val = split_at, n - split_at
    comm = eng.comm
    t0, nt = ddist.shard(n_train, comm.world, comm.rank)
    v0, nv = ddist.shard(n_val, comm.world, comm.rank)
    rows = np.r_[np.arange(t0, t0 + nt), split_at + np.arange(v0, v0 + nv)]
    dd = getattr(adata, '_dca_device', None)
    if comm.world == 1 and dd is not None and not output_subset and use_raw_as_output and \
            dd.n == n and dd.G == X.shape[1] == eng.lay.G_in == eng.lay.G_out and \
            dd.X.device == eng.dev and dd.matches(X):
        eng.attach_device_data(dd.X, dd.Y, dd.sf, norm=dd.norm, compact=dd.compact)      # K-PREP left the tensors in HBM
        dd.compact = eng.cc if eng.cc is not None else (False if eng.cc_verdict is False else None)
    elif comm.world == 1:
        eng.load_data(X, Y, sf)
    else:
        eng.load_data(X[rows], Y[rows], sf[rows])
    # Callbacks of train.py:62-69.  ModelCheckpoint(save_weights_only=True, save_best_only=True,
    # monitor val_loss): the weights file holds the BEST epoch so far; the model itself keeps
    # training (Keras does not restore).  Beyond the reference: ``checkpoint=True`` writes the full
    # training state after every epoch and ``resume=True`` continues from it.
    checkpoint = bool(kwds.pop('checkpoint', False))
    resume = bool(kwds.pop('resume', False))
    best = {'val': np.inf}
    state_path = os.path.join(output_dir, 'train_state.npz') if output_dir is not None else None

    def on_epoch(epoch, h, st):
        if checkpoint and output_dir is not None:
            eng.gather_optimizer_slots()           # a collective with the sharded optimizer: every rank, before rank 0 writes
        if comm.rank != 0 or output_dir is None:
            return
        if save_weights:
            cur = h.history['val_loss'][-1] if 'val_loss' in h.history and h.history['val_loss'] else None
            if cur is None:
                # Keras ModelCheckpoint(monitor='val_loss', save_best_only=True) without validation data warns
                # ("Can save best model only with val_loss available, skipping") and writes nothing (train.py:64-69)
                if not best.get('warned'):
                    print('dca: save_weights: no val_loss available (validation_split=0), skipping the weights file')
                    best['warned'] = True
            elif cur < best['val']:
                best['val'] = cur
                network.save_weights(os.path.join(output_dir, 'weights.npz'))
        if checkpoint:
            eng.save_state(state_path, st)

    state = None
    if resume and state_path is not None and os.path.exists(state_path):
        state = eng.load_state(state_path)
        vl = state['history'].get('val_loss') or []
        if vl:
            best['val'] = float(np.min(vl))
    kwds.setdefault('debug', bool(getattr(network, 'debug', False)))       # --debug: finite checks per step (dca/loss.py:90-100)
    hist = fit_engine(eng, n_train, n_val, nt, nv, t0, epochs=epochs, batch_size=batch_size,
                      learning_rate=learning_rate, clip_grad=clip_grad, reduce_lr=reduce_lr,
                      early_stop=early_stop, verbose=verbose, on_epoch=on_epoch, state=state, **kwds)
    return hist


def train_with_args(args):
    """The command-line pipeline (dca/train.py:103-191): fixed seed 42, read -> normalize (with count filtering) ->
    network -> train -> predict(mode='full', return_info=True) -> result files.  ``--hyper`` hands over to the
    hyper-parameter search and returns (train.py:119-122)."""
    import random
    from . import io
    from .network import AE_types

    seed = 42
    random.seed(seed)
    np.random.seed(seed)
    os.environ['PYTHONHASHSEED'] = '0'
    if args.hyper:
        from .hyper import hyper
        return hyper(args)

    # the CLI reads gene x cell tables unless --transpose says otherwise
    adata = io.read_dataset(args.input, transpose=not args.transpose, check_counts=args.checkcounts,
                            test_split=args.testsplit)
    adata = io.normalize(adata, size_factors=args.sizefactors, logtrans_input=args.loginput,
                         normalize_input=args.norminput)

    subset = None                                   # --denoisesubset: the loss sees these output genes only
    if args.denoisesubset:
        subset = list(set(io.read_genelist(args.denoisesubset)))
        unknown = set(subset) - set(adata.var_names.values)
        assert len(unknown) == 0, 'Gene list is not overlapping with genes from the dataset'
    n_out = len(subset) if subset else adata.n_vars

    widths = [int(w) for w in args.hiddensize.split(',')]
    rates = [float(r) for r in args.dropoutrate.split(',')]
    assert args.type in AE_types, 'loss type not supported'
    net = AE_types[args.type](input_size=adata.n_vars, output_size=n_out, hidden_size=widths,
                              hidden_dropout=rates[0] if len(rates) == 1 else rates,
                              input_dropout=args.inputdropout, batchnorm=args.batchnorm,
                              activation=args.activation, init=args.init, ridge=args.ridge,
                              l1_coef=args.l1, l2_coef=args.l2, l1_enc_coef=args.l1enc, l2_enc_coef=args.l2enc,
                              debug=args.debug, file_path=args.outputdir)
    net.seed = seed
    net.save()
    net.build()

    in_train = (adata.obs.dca_split == 'train').values
    train(adata[in_train], net, output_dir=args.outputdir, optimizer=args.optimizer,
          learning_rate=args.learningrate, epochs=args.epochs, batch_size=args.batchsize,
          reduce_lr=args.reducelr, early_stop=args.earlystop, clip_grad=args.gradclip,
          output_subset=subset, save_weights=args.saveweights, tensorboard=args.tensorboard)

    names = adata.var_names
    columns = names[[int(np.where(names == g)[0][0]) for g in subset]] if subset else names
    # predict(mode='full', return_info=True) + write(...) of the reference (train.py:176-190) as one streaming pass: the
    # result files are the only consumer here, so no cells x genes matrix is staged on the host (network.predict_write)
    net.predict_write(adata, args.outputdir, mode='full', colnames=columns)
