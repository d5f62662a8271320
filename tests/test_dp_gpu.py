"""Data-parallel path with the REAL kernels: 2 processes share the one GPU of the test box and
exchange over gloo (RCCL needs one GPU per rank; the exchange pattern -- SyncBN statistics, the
two gradient buckets with the asynchronous heads bucket on its own group, loss slot -- is the same
code).  Must reproduce the single-process run on the same global batches."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import make_problem
from test_dp_gloo import FixedOrders, dp_equivalent_orders, _free_port
from dca_amd import dist as ddist

pytestmark = pytest.mark.gpu


def _run(rank, world, port, cfg, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), DCA_AMD_DIST_BACKEND='gloo')
    try:
        from dca_amd.engine import Engine
        from dca_amd.train import fit_engine
        comm = ddist.init_from_env()
        n, G, hs, ae, bn, B, epochs, seed = cfg
        X, Y, sf, p = make_problem(n, G, hs, ae, bn, seed=3)
        n_train = int(n * 0.9)
        n_val = n - n_train
        t0, nt = ddist.shard(n_train, world, rank)
        v0, nv = ddist.shard(n_val, world, rank)
        rows = np.r_[np.arange(t0, t0 + nt), n_train + np.arange(v0, v0 + nv)]
        eng = Engine(ae, G, G, hs, bn, 0.0, comm=comm)
        eng.set_params(p)
        eng.load_data(X[rows], Y[rows], sf[rows])
        h = fit_engine(eng, n_train, n_val, nt, nv, t0, epochs=epochs, batch_size=B,
                       shuffle_rng=np.random.RandomState(seed), reduce_lr=1, early_stop=0)
        torch.cuda.synchronize()
        if rank == 0:
            q.put((h.history, eng.get_params()))
        dist.barrier()
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize('ae,n,B', [('zinb-conddisp', 300, 64), ('zinb', 203, 32)])
def test_two_ranks_on_one_gpu_equal_single_process(ae, n, B):
    G, hs, epochs, seed, W, bn = 150, (64, 32, 64), 2, 17, 2, True
    cfg = (n, G, hs, ae, bn, B, epochs, seed)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run, args=(r, W, port, cfg, q)) for r in range(W)]
    for pr in procs:
        pr.start()
    hist_dp, p_dp = q.get(timeout=300)
    for pr in procs:
        pr.join(timeout=120)
        assert pr.exitcode == 0

    from dca_amd.engine import Engine
    from dca_amd.train import fit_engine
    X, Y, sf, p = make_problem(n, G, hs, ae, bn, seed=3)
    n_train = int(n * 0.9)
    orders = dp_equivalent_orders(n_train, W, B // W, epochs, seed)
    eng = Engine(ae, G, G, hs, bn, 0.0)
    eng.set_params(p)
    eng.load_data(X, Y, sf)
    h1 = fit_engine(eng, n_train, n - n_train, n_train, n - n_train, 0, epochs=epochs, batch_size=B,
                    shuffle_rng=FixedOrders(orders), reduce_lr=1, early_stop=0)
    # same kernels on both sides; the association order of the cross-rank sums differs
    np.testing.assert_allclose(hist_dp['loss'], h1.history['loss'], rtol=3e-5)
    np.testing.assert_allclose(hist_dp['val_loss'], h1.history['val_loss'], rtol=3e-5)
    assert hist_dp['lr'] == h1.history['lr']
    p1 = eng.get_params()
    for k in p1:
        if k[0] == 'b' and k[1:].isdigit():
            continue
        np.testing.assert_allclose(p_dp[k], p1[k], rtol=2e-3, atol=2e-3, err_msg=k)


def test_bench_multi_rank_branch_runs_on_one_gpu():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), with gloo as the
    exchange backend because the test box has one GPU (DCA_AMD_DIST_BACKEND=gloo; RCCL needs a GPU per rank).  Keeps
    the multi-rank branch of the bench alive: sharded generation, global K-PREP statistics (all-gather of the
    library sizes, all-reduce of the gene moments), the eager step with both gradient buckets, the max-over-ranks
    timing, ONE JSON line from rank 0 with whole-job cells/s and weak scaling."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DCA_AMD_DIST_BACKEND='gloo', MASTER_ADDR='127.0.0.1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(_free_port()), os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2',
           '--cells', '6000', '--genes', '2000', '--batch-size', '256']
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['scaling'] == 'weak' and out['steps'] == 3
    assert out['config']['global_batch'] == 512 and out['config']['parallelism'] == 'dp2'
    assert out['config']['launch'] == 'eager'                     # collectives are not captured
    assert np.isfinite(out['value']) and out['value'] > 0
    assert abs(out['value'] - 3 * 512 / (out['ms_per_step'] * 3e-3)) < 1e-6 * out['value']
    assert np.isfinite(out['loss_first']) and np.isfinite(out['loss_last'])
    assert out['cpu_baseline'] is None                            # rank 0 at N = 1 only


def _run_rccl_one_rank(port, cfg, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    os.environ.pop('DCA_AMD_DIST_BACKEND', None)
    try:
        from dca_amd.engine import Engine
        from dca_amd.train import fit_engine
        comm = ddist.init_from_env(force=True)                       # backend nccl = RCCL, one rank
        assert comm.dp and comm.world == 1 and dist.get_backend() == 'nccl'
        # the communicator's own calls on device tensors
        t = torch.arange(10000, dtype=torch.float32, device='cuda')
        assert torch.equal(comm.all_reduce_sum(t.clone()), t)
        w = comm.all_reduce_sum_async(t)
        comm.wait(w)
        assert torch.equal(comm.all_gather(t[:7])[0], t[:7])
        out = torch.zeros(10000, device='cuda')
        comm.reduce_scatter_sum(t, out)
        assert torch.equal(out, t)
        comm.all_gather_into(out, out[:10000])
        n, G, hs, ae, bn, B, epochs, seed, sharded = cfg
        os.environ['DCA_AMD_DP_SHARDED_OPT'] = '1' if sharded else '0'
        X, Y, sf, p = make_problem(n, G, hs, ae, bn, seed=3)
        n_train = int(n * 0.9)
        res = []
        for graph in ('0', '1'):                                     # eager steps, then the same fit from captured steps
            os.environ['DCA_AMD_DP_GRAPH'] = graph
            eng = Engine(ae, G, G, hs, bn, 0.0, comm=comm)
            eng.set_params(p)
            eng.load_data(X, Y, sf)
            comm.timer = {}
            h = fit_engine(eng, n_train, n - n_train, n_train, n - n_train, 0, epochs=epochs, batch_size=B,
                           shuffle_rng=np.random.RandomState(seed), reduce_lr=1, early_stop=0)
            torch.cuda.synchronize()
            spans = comm.timer_summary()
            comm.timer = None
            res.append((h.history, eng.get_params(), {k: v[0] for k, v in spans.items()}))
        q.put(tuple(res))
        dist.barrier()
    except BaseException as e:                                       # the parent must not wait for its timeout
        import traceback
        q.put(('error', '%s: %s\n%s' % (type(e).__name__, e, traceback.format_exc()), None))
        raise
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize('sharded', [False, True])
@pytest.mark.parametrize('ae,n,B,hs', [('zinb-conddisp', 300, 64, (64, 32, 64)), ('zinb-conddisp', 700, 256, (128, 32, 128))])
def test_rccl_communicator_with_one_rank_runs_the_data_parallel_step(ae, n, B, hs, sharded):
    """The data-parallel step over the REAL backend (nccl = RCCL) on the one GPU of the box: a one-rank communicator
    (init_from_env(force=True)) puts every exchange of the step -- the asynchronous heads bucket on its own RCCL
    communicator and stream, the SyncBN all-gathers / all-reduces, the loss slot, and with DCA_AMD_DP_SHARDED_OPT the
    reduce-scatter / all-gather pair -- between the kernels exactly as N ranks would; with one rank every exchange is the
    identity, so the fit must reproduce the single-process engine.  The fit loop captures these steps -- exchanges included --
    into hipGraphs (train.py::_StepRunner) exactly as it would with N ranks."""
    G, epochs, seed, bn = 150, 2, 17, True
    cfg = (n, G, hs, ae, bn, B, epochs, seed, sharded)
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    pr = ctx.Process(target=_run_rccl_one_rank, args=(_free_port(), cfg, q))
    pr.start()
    got = q.get(timeout=150)
    pr.join(timeout=60)
    assert got[0] != 'error', got[1]
    assert pr.exitcode == 0
    (hist_dp, p_dp, calls), (hist_gr, p_gr, calls_gr) = got
    assert any(k.startswith('wait_async') for k in calls) or sharded, calls      # the exchanges did run
    assert sum(calls.values()) > 0
    # captured steps (exchanges inside the graphs) = eager steps, bit for bit; fewer host-side calls (capture only)
    assert hist_gr == hist_dp
    for k in p_dp:
        assert np.array_equal(p_gr[k], p_dp[k]), k
    assert sum(calls_gr.values()) < sum(calls.values())

    from dca_amd.engine import Engine
    from dca_amd.train import fit_engine
    X, Y, sf, p = make_problem(n, G, hs, ae, bn, seed=3)
    n_train = int(n * 0.9)
    eng = Engine(ae, G, G, hs, bn, 0.0)
    eng.set_params(p)
    eng.load_data(X, Y, sf)
    h1 = fit_engine(eng, n_train, n - n_train, n_train, n - n_train, 0, epochs=epochs, batch_size=B,
                    shuffle_rng=np.random.RandomState(seed), reduce_lr=1, early_stop=0)
    # the data-parallel step takes the separate BatchNorm kernels (statistics exchanged between two launches) where the
    # single-process step takes the fused ones: the same numbers to fp32 re-association in the first epoch, and what
    # RMSprop's sign-like early steps make of that in the second (the 5e-4 class of DESIGN.md 7; measured 1.5e-4)
    np.testing.assert_allclose(hist_dp['loss'][:1], h1.history['loss'][:1], rtol=3e-6)
    np.testing.assert_allclose(hist_dp['loss'], h1.history['loss'], rtol=5e-4)
    np.testing.assert_allclose(hist_dp['val_loss'], h1.history['val_loss'], rtol=5e-4)
    p1 = eng.get_params()
    for k in p1:
        if k[0] == 'b' and k[1:].isdigit():
            continue                                                  # biases in front of BatchNorm: zero true gradient
        np.testing.assert_allclose(p_dp[k], p1[k], rtol=2e-3, atol=2e-3, err_msg=k)


def _run_c4_shard_step(port, q, ae='zinb'):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
    os.environ.pop('DCA_AMD_DIST_BACKEND', None)
    try:
        from dca_amd import synth, prep
        from dca_amd.engine import Engine
        from dca_amd.ops import HipOps
        from helpers import assert_grads_close, oracle_net, run_single_step
        from oracle import net_np as N
        comm = ddist.init_from_env(force=True)
        assert comm.dp and comm.world == 1 and dist.get_backend() == 'nccl'
        ops = HipOps()
        n, G, hs, B = 125000, 25000, (64, 32, 64), 4096
        dev = torch.device('cuda')
        Y = synth.generate_counts_portable(n, G, seed=20260925, device=dev, row_offset=0)      # rank 0's rows of the 1M x 25k matrix
        counts = prep.cell_counts(ops, Y, n, G)
        sf = counts / counts.median()
        X, norm = prep.transform(ops, Y, n, G, sf, True, True, return_norm=True)
        p = N.init_params(ae, G, hs, batchnorm=True, seed=4, dtype=np.float64)
        rng = np.random.RandomState(9)
        for k in p:
            if k[0] in 'bt':
                p[k] = rng.normal(0, .1, p[k].shape)
        p = {k: np.asarray(v, np.float32) for k, v in p.items()}
        eng = Engine(ae, G, G, hs, True, 0.0, ops=ops, comm=comm)
        eng.set_params(p)
        eng.attach_device_data(X, Y, sf, norm=norm)
        rows = np.random.RandomState(1).permutation(n)[:B]
        rt = torch.as_tensor(rows).cuda()
        Xr = X[rt][:, :G].cpu().numpy().astype(np.float64)
        Yr = Y[rt][:, :G].cpu().numpy().astype(np.float64)
        sfr = sf[rt].cpu().numpy().astype(np.float64)
        zeros = float((Yr == 0).mean())
        ref = oracle_net(ae, p, hs, True)
        ref.row_threads = max(1, min(64, os.cpu_count() or 1))
        rl, rg = ref.loss_and_grads(Xr, Yr, sfr)
        comm.timer = {}
        loss, g, newp = run_single_step(eng, rows)
        spans = comm.timer_summary()
        comm.timer = None
        assert abs(loss - rl) < 1e-5 * abs(rl), (loss, rl)
        assert_grads_close(g, rg)
        q.put(('ok', dict(loss=loss, oracle=float(rl), zeros=zeros, fused=bool(eng.use_fused), calls={k: v[0] for k, v in spans.items()})))
        dist.barrier()
    except BaseException as e:
        import traceback
        q.put(('error', '%s: %s\n%s' % (type(e).__name__, e, traceback.format_exc())))
        raise
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize('ae', ['zinb-conddisp', 'zinb'])
def test_c4_rank_shard_step_matches_oracle(ae):
    """BASELINE configs[3] (ZINB autoencoder on 1 000 000 x 25 000, data parallel over 8 GPUs) as ONE rank sees it: the
    rank's shard of 125 000 cells x 25 000 genes resident in HBM (the portable generator: rows 0 .. 124 999 of the
    matrix), 4 096 cells of it per step, the data-parallel step -- SyncBN exchanges, gradient buckets over a real RCCL
    communicator of one rank (init_from_env(force=True) = DCA_AMD_DIST_FORCE) -- against the fp64 oracle of the reference
    step on the same gathered rows: loss to 1e-5, every gradient to the tolerances of the single-GPU step tests.  Both
    readings of "ZINB AE": the three-head network SURVEY 8 sizes the configuration by (network.py:366-393, P = 6 479 416) and
    the constant-dispersion sibling (network.py:496-550)."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    pr = ctx.Process(target=_run_c4_shard_step, args=(_free_port(), q, ae))
    pr.start()
    got = q.get(timeout=900)
    pr.join(timeout=60)
    assert got[0] == 'ok', got[1]
    assert pr.exitcode == 0
    info = got[1]
    print('C4 rank shard (125 000 x 25 000, batch 4 096, %s): loss %.8f, fp64 oracle %.8f; zeros %.3f; exchanges %s'
          % (ae, info['loss'], info['oracle'], info['zeros'], info['calls']))
    assert 0.90 < info['zeros'] < 0.96 and info['fused']
    assert sum(info['calls'].values()) > 0


def _run_peer(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank), DCA_AMD_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    try:
        from dca_amd.engine import Engine
        from dca_amd import peer as _peer
        from dca_amd.peer import PeerExchange
        from dca_amd.train import fit_engine
        comm = ddist.init_from_env()
        dev = torch.device('cuda')
        # ---- raw exchanges: gathers and reductions of several lengths, 60 epochs, then the same inside a hipGraph
        px = PeerExchange(rank, world, 128)        # (runs the init-time ping-pong self check: a failure raises PeerUnavailable)
        # the exchange buffers are FINE-GRAINED device memory (csrc/dcahip_peer.hip: what makes a peer's stores over xGMI
        # visible to the owner's spinning loads): as allocated, and as the ROCr runtime reports the allocation
        fine = _peer.HSA_FLAG_FINE_GRAINED | _peer.HSA_FLAG_EXTENDED_SCOPE_FINE_GRAINED
        assert len(px.mem_flags) == 2
        assert all(f is not None and (f & fine) and not (f & _peer.HSA_FLAG_COARSE_GRAINED) for f in px.mem_flags), px.mem_flags
        plain = torch.zeros(1024, device=dev)      # torch's allocator = plain hipMalloc: the query tells the two apart
        pf = _peer.memory_flags(plain.data_ptr())
        assert pf is not None and (pf & _peer.HSA_FLAG_COARSE_GRAINED), pf
        bad = 0
        for e in range(60):
            n = (1, 7, 64, 128)[e % 4]
            local = (torch.arange(n, device=dev, dtype=torch.float32) * 0.25 + 1000.0 * (rank + 1) + e)
            out = torch.full((world * n,), -1.0, device=dev)
            px.gather(out, local)
            want = torch.cat([torch.arange(n, device=dev, dtype=torch.float32) * 0.25 + 1000.0 * (r + 1) + e for r in range(world)])
            bad += int((out != want).sum().item())
            t = local.clone()
            px.reduce(t)
            bad += int((t != want.view(world, n).sum(0)).sum().item())
        px.check()
        g = torch.cuda.CUDAGraph()
        src = torch.zeros(16, device=dev); dst = torch.zeros(world * 16, device=dev); red = torch.zeros(16, device=dev)
        s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            with torch.cuda.graph(g, stream=s, capture_error_mode='thread_local'):
                px.gather(dst, src)
                red.copy_(src)
                px.reduce(red)
        torch.cuda.current_stream().wait_stream(s)
        for k in range(3):
            src.fill_(float(10 * k + rank + 1))
            g.replay()
            torch.cuda.synchronize()
            want = torch.cat([torch.full((16,), float(10 * k + r + 1), device=dev) for r in range(world)])
            bad += int((dst != want).sum().item()) + int((red != want.view(world, 16).sum(0)).sum().item())
        px.check()
        px.close()
        # ---- a peer that never arrives: rank 0 exchanges alone with a 0.2 s timeout -> NaN results + status, check() raises
        # (a fresh exchange object: its epochs are out of step afterwards)
        px2 = PeerExchange(rank, world, 16)
        if rank == 0:
            px2.timeout_us = 200 * 1000
            t = torch.ones(16, device=dev)
            px2.reduce(t)
            torch.cuda.synchronize()
            bad += int((~torch.isnan(t)).sum().item())
            try:
                px2.check()
                bad += 1000
            except RuntimeError as e:
                assert 'did not arrive' in str(e)
        dist.barrier()
        px2.close()
        # ---- the data-parallel fit with the SyncBN exchanges through K-PEER against the library's collectives
        n, G, hs, ae, B, epochs, seed = 300, 150, (64, 32, 64), 'zinb-conddisp', 64, 2, 17
        X, Y, sf, p = make_problem(n, G, hs, ae, True, seed=3)
        n_train = int(n * 0.9); n_val = n - n_train
        t0, nt = ddist.shard(n_train, world, rank); v0, nv = ddist.shard(n_val, world, rank)
        rows = np.r_[np.arange(t0, t0 + nt), n_train + np.arange(v0, v0 + nv)]
        res = []
        for peer in ('0', '1'):
            os.environ['DCA_AMD_DP_PEER'] = peer
            eng = Engine(ae, G, G, hs, True, 0.0, comm=comm)
            assert (comm.peer is not None) == (peer == '1')
            eng.set_params(p)
            eng.load_data(X[rows], Y[rows], sf[rows])
            comm.timer = {}
            h = fit_engine(eng, n_train, n_val, nt, nv, t0, epochs=epochs, batch_size=B,
                           shuffle_rng=np.random.RandomState(seed), reduce_lr=1, early_stop=0)
            calls = {k: v[0] for k, v in comm.timer_summary().items()}
            comm.timer = None
            if comm.peer is not None:
                comm.peer.check()
            res.append((h.history, eng.get_params(), calls))
        if rank == 0:
            q.put(('ok', bad, res))
        dist.barrier()
    except BaseException as e:
        import traceback
        q.put(('error', '%s: %s\n%s' % (type(e).__name__, e, traceback.format_exc()), None))
        raise
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_peer_exchange_two_processes_on_one_gpu():
    """K-PEER (dcahip_peer_exchange: peer stores into IPC-mapped slots + flags, no library call) with two processes sharing this
    GPU: the exchange buffers are fine-grained memory (allocation flag + what the ROCr runtime reports; torch's plain hipMalloc
    memory reports coarse-grained), the init-time self check passes, 60 gathers and 60 reductions of 1 .. 128 floats give exactly
    the expected vectors, eagerly and replayed from a hipGraph; an exchange a peer never joins returns NaN and raises at check();
    then a two-epoch data-parallel fit whose SyncBN exchanges go through K-PEER (EngineConfig.dp_peer_exchange) equals the fit
    over the library's collectives BIT FOR BIT (two ranks: the sum a + b has one order) -- and did use the peer path."""
    W = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_run_peer, args=(r, W, port, q)) for r in range(W)]
    for pr in procs:
        pr.start()
    got = q.get(timeout=600)
    for pr in procs:
        pr.join(timeout=120)
    assert got[0] == 'ok', got[1]
    assert all(pr.exitcode == 0 for pr in procs)
    _, bad, res = got
    assert bad == 0
    (h0, p0, c0), (h1, p1, c1) = res
    assert h0 == h1, (h0, h1)
    for k in p0:
        assert np.array_equal(p0[k], p1[k]), k
    assert c1.get('peer_gather_small', 0) > 0 and c1.get('peer_reduce_small', 0) > 0 and 'all_gather_small' not in c1
    assert c0.get('all_gather_small', 0) > 0 and 'peer_gather_small' not in c0
    print('K-PEER fit == collective fit; exchanges per fit: %s' % c1)
