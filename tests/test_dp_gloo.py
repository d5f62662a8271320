"""Data-parallel path on CPU: 2 processes over gloo running the engine's DP step (row shards,
SyncBN statistics exchange, flat gradient all-reduce) must reproduce the single-process run on
the same global batches, and both must match the straight-line oracle."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import net_np as N
from oracle.cpu_ops import CpuRefOps
from helpers import make_problem, oracle_net
from dca_amd import dist as ddist


class FixedOrders:
    """shuffle_rng stand-in that replays prescribed index orders (one per epoch)."""

    def __init__(self, orders):
        self.orders = list(orders)
        self.k = 0

    def shuffle(self, idx):
        idx[:] = self.orders[self.k]
        self.k += 1


def dp_equivalent_orders(n_train, W, b_local, epochs, seed):
    """The single-process index order that visits, batch by batch, exactly the samples the W
    ranks visit together (rank-major inside each global batch)."""
    rs = np.random.RandomState(seed)
    out = []
    for _ in range(epochs):
        idx = np.arange(n_train)
        rs.shuffle(idx)
        loc = []
        for r in range(W):
            s, c = ddist.shard(n_train, W, r)
            loc.append(ddist.local_order(idx, s, c) + s)
        steps = int(np.ceil(max(len(l) for l in loc) / b_local))
        order = []
        for t in range(steps):
            for r in range(W):
                order.extend(loc[r][t * b_local:(t + 1) * b_local].tolist())
        out.append(np.array(order))
    return out


def _run(rank, world, port, cfg, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from dca_amd.engine import Engine
        from dca_amd.train import fit_engine
        n, G, hs, ae, bn, B, epochs, seed = cfg[:8]
        optimizer = cfg[8] if len(cfg) > 8 else None
        drop = cfg[9] if len(cfg) > 9 else {}
        if len(cfg) > 10 and cfg[10]:
            os.environ['DCA_AMD_DP_SHARDED_OPT'] = '1'
        X, Y, sf, p = make_problem(n, G, hs, ae, bn, seed=3)
        n_train = int(n * 0.9)
        n_val = n - n_train
        comm = ddist.TorchDistComm()
        t0, nt = ddist.shard(n_train, world, rank)
        v0, nv = ddist.shard(n_val, world, rank)
        rows = np.r_[np.arange(t0, t0 + nt), n_train + np.arange(v0, v0 + nv)]
        eng = Engine(ae, G, G, hs, bn, 0.0, ops=CpuRefOps(), comm=comm, **drop)
        eng.set_params(p)
        if optimizer:
            eng.set_optimizer(optimizer)
            eng.set_lr(0.01)
        eng.load_data(X[rows], Y[rows], sf[rows])
        h = fit_engine(eng, n_train, n_val, nt, nv, t0, epochs=epochs, batch_size=B,
                       shuffle_rng=np.random.RandomState(seed), reduce_lr=1, early_stop=0)
        if rank == 0:
            q.put((h.history, eng.get_params()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]; s.close()
    return p


@pytest.mark.parametrize('ae,bn,n,B,W', [('zinb-conddisp', True, 60, 16, 2), ('zinb-conddisp', True, 38, 16, 2), ('zinb-conddisp', True, 37, 16, 2),
                                         ('nb', True, 50, 8, 2), ('zinb', False, 60, 16, 2),
                                         # 4 ranks, 18 train rows = shards of 5, 5, 4, 4 at 4 rows per rank and step: in the last
                                         # step of every epoch ranks 2 and 3 are EMPTY (they still join every collective);
                                         # 2 validation rows = shards 1, 1, 0, 0
                                         ('zinb-conddisp', True, 20, 16, 4), ('nb', True, 23, 8, 4),
                                         # 8 ranks (the target node): 36 train rows = shards 5, 5, 5, 5, 4, 4, 4, 4 at 2 rows per
                                         # rank and step: FOUR empty ranks in the last step of every epoch; 5 validation rows =
                                         # shards 1, 1, 1, 1, 1, 0, 0, 0
                                         ('zinb-conddisp', True, 41, 16, 8)])
def test_two_rank_dp_equals_single_process_and_oracle(ae, bn, n, B, W):
    G, hs, epochs, seed = 14, (6, 3, 6), 3, 17
    if W == 4 and n == 20:
        shards = [ddist.shard(int(n * 0.9), W, r)[1] for r in range(W)]
        assert shards == [5, 5, 4, 4] and B // W == 4           # => two empty ranks in the last step
    if W == 8:
        assert [ddist.shard(int(n * 0.9), W, r)[1] for r in range(W)] == [5, 5, 5, 5, 4, 4, 4, 4] and B // W == 2
        assert [ddist.shard(n - int(n * 0.9), W, r)[1] for r in range(W)] == [1, 1, 1, 1, 1, 0, 0, 0]
    cfg = (n, G, hs, ae, bn, B, epochs, seed)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, W, port, cfg, q)) for r in range(W)]
    for pr in procs:
        pr.start()
    hist_dp, p_dp = q.get(timeout=240)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0

    # single process, same global batches
    from dca_amd.engine import Engine
    from dca_amd.train import fit_engine
    X, Y, sf, p = make_problem(n, G, hs, ae, bn, seed=3)
    n_train = int(n * 0.9)
    orders = dp_equivalent_orders(n_train, W, B // W, epochs, seed)
    eng = Engine(ae, G, G, hs, bn, 0.0, ops=CpuRefOps())
    eng.set_params(p)
    eng.load_data(X, Y, sf)
    h1 = fit_engine(eng, n_train, n - n_train, n_train, n - n_train, 0, epochs=epochs, batch_size=B,
                    shuffle_rng=FixedOrders(orders), reduce_lr=1, early_stop=0)
    # fp32 buffers on both sides; the only difference is the association order of the BN
    # statistics merge and of the gradient sum across ranks
    np.testing.assert_allclose(hist_dp['loss'], h1.history['loss'], rtol=3e-5)
    np.testing.assert_allclose(hist_dp['val_loss'], h1.history['val_loss'], rtol=3e-5)
    assert hist_dp['lr'] == h1.history['lr']
    p1 = eng.get_params()
    for k in p1:
        if bn and k.startswith('b') and k[1:].isdigit():
            # the bias of a Dense feeding BatchNormalization: its gradient is identically zero, what is computed is
            # round-off, and Keras' RMSprop (epsilon outside the root) turns round-off into sign-like steps -- the
            # bias random-walks by up to ~3 lr per update, differently for every summation order, and no output
            # depends on it (batch norm removes it)
            continue
        np.testing.assert_allclose(p_dp[k], p1[k], rtol=2e-3, atol=2e-3, err_msg=k)  # RMSprop: |step| <= lr/sqrt(1-rho) per update

    # and the oracle on the same order
    ref = oracle_net(ae, p, hs, bn)
    rh = N.fit(ref, X.astype(np.float64), Y.astype(np.float64), sf.astype(np.float64), epochs=epochs,
               batch_size=B, shuffle_rng=FixedOrders(orders), reduce_lr=1, early_stop=0)
    np.testing.assert_allclose(hist_dp['loss'], rh['loss'], rtol=1e-4)
    np.testing.assert_allclose(hist_dp['val_loss'], rh['val_loss'], rtol=1e-4)


def _dp_fit(cfg, W):
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, W, port, cfg, q)) for r in range(W)]
    for pr in procs:
        pr.start()
    out = q.get(timeout=240)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    return out


@pytest.mark.parametrize('ae,bn,n,B,W', [('zinb-conddisp', True, 60, 16, 2), ('nb', True, 23, 8, 4), ('zinb', True, 41, 16, 8)])
def test_sharded_optimizer_equals_the_all_reduce_path(ae, bn, n, B, W):
    """reduce-scatter -> per-rank clip + RMSprop on its shard -> all-gather of the parameters (DCA_AMD_DP_SHARDED_OPT=1,
    SURVEY 5) against the default two-bucket all-reduce: the same fit, uneven shards and empty ranks included.  With
    two ranks the sums are the same sums; with more, gloo's reduction order may differ between the two collectives."""
    G, hs, epochs, seed = 14, (6, 3, 6), 3, 17
    base = (n, G, hs, ae, bn, B, epochs, seed, None, {})
    h0, p0 = _dp_fit(base + (False,), W)
    h1, p1 = _dp_fit(base + (True,), W)
    # (more than two ranks: a different reduction order is fp32 noise in the gradients, which Keras' RMSprop -- epsilon
    # outside the root: sign-like steps while the accumulator is small -- turns into O(lr) differences of the parameters
    # whose gradients are small; the biases in front of a batch norm have NO gradient but round-off and random-walk)
    tol = 0 if W == 2 else 1e-4
    np.testing.assert_allclose(h1['loss'], h0['loss'], rtol=tol)
    np.testing.assert_allclose(h1['val_loss'], h0['val_loss'], rtol=tol)
    for k in p0:
        if W > 2 and bn and k.startswith('b') and k[1:].isdigit():
            continue
        np.testing.assert_allclose(p1[k], p0[k], rtol=0, atol=0 if W == 2 else 5e-3, err_msg=k)


def test_two_rank_dp_gradients_are_summed_once():
    """SGD is linear in the gradient (RMSprop hides a rank-count factor on a parameter's gradient
    behind its scale invariance): every parameter, batch-norm beta included, must follow the
    single-process trajectory tightly."""
    ae, bn, n, B = 'zinb-conddisp', True, 60, 16
    G, hs, epochs, seed, W = 14, (6, 3, 6), 2, 17, 2
    cfg = (n, G, hs, ae, bn, B, epochs, seed, 'sgd')
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, W, port, cfg, q)) for r in range(W)]
    for pr in procs:
        pr.start()
    hist_dp, p_dp = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    from dca_amd.engine import Engine
    from dca_amd.train import fit_engine
    X, Y, sf, p = make_problem(n, G, hs, ae, bn, seed=3)
    n_train = int(n * 0.9)
    orders = dp_equivalent_orders(n_train, W, B // W, epochs, seed)
    eng = Engine(ae, G, G, hs, bn, 0.0, ops=CpuRefOps())
    eng.set_params(p)
    eng.set_optimizer('sgd')
    eng.set_lr(0.01)
    eng.load_data(X, Y, sf)
    h1 = fit_engine(eng, n_train, n - n_train, n_train, n - n_train, 0, epochs=epochs, batch_size=B,
                    shuffle_rng=FixedOrders(orders), reduce_lr=1, early_stop=0)
    np.testing.assert_allclose(hist_dp['loss'], h1.history['loss'], rtol=1e-5)
    p1 = eng.get_params()
    for k in p1:
        if k[0] == 'b' and k[1:].isdigit():
            continue                      # bias in front of batch-norm: gradient is rounding noise
        np.testing.assert_allclose(p_dp[k], p1[k], rtol=1e-4, atol=1e-5, err_msg=k)
        if k.startswith('beta'):
            assert np.abs(p1[k] - p[k]).max() > 1e-4      # the check is not vacuous


def test_two_rank_dp_with_dropout_draws_the_single_process_masks():
    """K-DROP indexes its generator by the row's position in the GLOBAL batch: two ranks reproduce the
    single-process run (and the oracle) with dropout on."""
    ae, bn, n, B = 'zinb-conddisp', True, 60, 16
    G, hs, epochs, seed, W = 14, (6, 3, 6), 2, 17, 2
    drop = dict(hidden_dropout=[0.3, 0.0, 0.2], input_dropout=0.2, dropout_seed=11)
    cfg = (n, G, hs, ae, bn, B, epochs, seed, None, drop)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, W, port, cfg, q)) for r in range(W)]
    for pr in procs:
        pr.start()
    hist_dp, p_dp = q.get(timeout=120)
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    X, Y, sf, p = make_problem(n, G, hs, ae, bn, seed=3)
    n_train = int(n * 0.9)
    orders = dp_equivalent_orders(n_train, W, B // W, epochs, seed)
    ref = N.OracleAE(ae, {k: np.asarray(v, np.float64).copy() for k, v in p.items()}, hs, bn, **drop)
    rh = N.fit(ref, X.astype(np.float64), Y.astype(np.float64), sf.astype(np.float64), epochs=epochs,
               batch_size=B, shuffle_rng=FixedOrders(orders), reduce_lr=1, early_stop=0)
    np.testing.assert_allclose(hist_dp['loss'], rh['loss'], rtol=1e-4)
    np.testing.assert_allclose(hist_dp['val_loss'], rh['val_loss'], rtol=1e-4)


def test_shard_and_local_order():
    assert [ddist.shard(10, 3, r) for r in range(3)] == [(0, 4), (4, 3), (7, 3)]
    idx = np.array([5, 0, 9, 3, 7, 1, 8, 2, 6, 4])
    assert ddist.local_order(idx, 4, 3).tolist() == [1, 2, 0]        # 5, 6, 4 in visiting order
    assert ddist.local_order(idx, 0, 10).tolist() == idx.tolist()    # one rank: the reference order


def _run_train_checkpoint(rank, world, port, outdir, sharded, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ['DCA_AMD_DP_SHARDED_OPT'] = '1' if sharded else '0'
    torch.set_num_threads(1)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import pandas as pd
        from conftest import synth_counts
        from dca_amd import io
        from dca_amd._anndata import AnnData
        from dca_amd.network import AE_types, override_ops
        from dca_amd.train import train
        n, G = 64, 14
        ad = AnnData(synth_counts(n, G, 2).astype(np.float32), obs=pd.DataFrame(index=['c%d' % i for i in range(n)]),
                     var=pd.DataFrame(index=['g%d' % i for i in range(G)]))
        ad = io.normalize(io.read_dataset(ad, test_split=False), device=False)
        np.random.seed(5)
        with override_ops(CpuRefOps):
            net = AE_types['zinb-conddisp'](input_size=G, hidden_size=(6, 3, 6), comm=ddist.TorchDistComm())
            net.seed = 0
            net.build()
            h = train(ad, net, output_dir=outdir, epochs=3, batch_size=16, verbose=False, checkpoint=True,
                      early_stop=0, reduce_lr=0)
        if rank == 0:
            z = np.load(os.path.join(outdir, 'train_state.npz'))
            q.put((h.history, z['ms'].copy(), z['w'].copy()))
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_checkpoints_of_the_sharded_optimizer_do_not_hang_and_hold_every_slot(tmp_path):
    """train(checkpoint=True) on 2 ranks with the sharded optimizer (DCA_AMD_DP_SHARDED_OPT=1): every rank keeps only its
    shard of the RMSprop slots current, so the checkpoint needs a gather -- a COLLECTIVE, which every rank must enter
    (round 3: only rank 0 did, and the run hung at the end of the first epoch).  The checkpoint rank 0 writes then holds
    the slots and parameters of the all-reduce run, bit for bit (two ranks: the same sums)."""
    res = []
    for sharded in (False, True):
        ctx = mp.get_context('spawn')
        q = ctx.Queue()
        port = _free_port()
        out = str(tmp_path / ('s%d' % sharded))
        procs = [ctx.Process(target=_run_train_checkpoint, args=(r, 2, port, out, sharded, q)) for r in range(2)]
        for pr in procs:
            pr.start()
        res.append(q.get(timeout=240))
        for pr in procs:
            pr.join(timeout=60)
            assert pr.exitcode == 0
    (h0, ms0, w0), (h1, ms1, w1) = res
    assert h0 == h1
    np.testing.assert_array_equal(w1, w0)
    np.testing.assert_array_equal(ms1, ms0)
    assert np.abs(ms0).max() > 0
