import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch, torch.multiprocessing as mp


def _run_dbg(rank, world, port, cfg, q):
    import torch.distributed as dist
    from dca_amd import dist as ddist
    from helpers import make_problem
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank), DCA_AMD_DIST_BACKEND='gloo')
    from dca_amd.engine import Engine
    from dca_amd.train import fit_engine
    comm = ddist.init_from_env()
    n, G, hs, ae, bn, B, epochs, seed = cfg
    X, Y, sf, p = make_problem(n, G, hs, ae, bn, seed=3)
    n_train = int(n * 0.9); n_val = n - n_train
    t0, nt = ddist.shard(n_train, world, rank); v0, nv = ddist.shard(n_val, world, rank)
    rows = np.r_[np.arange(t0, t0 + nt), n_train + np.arange(v0, v0 + nv)]
    eng = Engine(ae, G, G, hs, bn, 0.0, comm=comm)
    if os.environ.get('DBG_UNFUSED'): eng.use_fused = False
    eng.set_params(p); eng.load_data(X[rows], Y[rows], sf[rows])
    h = fit_engine(eng, n_train, n_val, nt, nv, t0, epochs=epochs, batch_size=B, shuffle_rng=np.random.RandomState(seed), reduce_lr=1, early_stop=0)
    torch.cuda.synchronize()
    if rank == 0:
        q.put((h.history, eng.get_params(), eng.hist.cpu().numpy()[:5]))
    dist.barrier(); dist.destroy_process_group()


def main():
    import test_dp_gloo as T
    import test_dp_gpu as TG
    from oracle import net_np as N
    from helpers import make_problem, oracle_net
    from dca_amd.engine import Engine
    from dca_amd.train import fit_engine
    ae = sys.argv[1] if len(sys.argv) > 1 else 'zinb-conddisp'
    n, G, hs, epochs, seed, W, bn, B = 300, 150, (64, 32, 64), int(os.environ.get("DBG_EPOCHS", "2")), 17, 2, True, 64
    cfg = (n, G, hs, ae, bn, B, epochs, seed)
    ctx = mp.get_context('spawn'); q = ctx.Queue(); port = T._free_port()
    procs = [ctx.Process(target=_run_dbg, args=(r, W, port, cfg, q)) for r in range(W)]
    [p.start() for p in procs]
    hist_dp, p_dp, steps_dp = q.get(timeout=300)
    [p.join() for p in procs]
    X, Y, sf, p = make_problem(n, G, hs, ae, bn, seed=3)
    n_train = int(n * 0.9)
    orders = T.dp_equivalent_orders(n_train, W, B // W, epochs, seed)
    for fused in (True, False):
        eng = Engine(ae, G, G, hs, bn, 0.0); eng.use_fused = fused; eng.set_params(p); eng.load_data(X, Y, sf)
        h1 = fit_engine(eng, n_train, n - n_train, n_train, n - n_train, 0, epochs=epochs, batch_size=B, shuffle_rng=T.FixedOrders(orders), reduce_lr=1, early_stop=0)
        print('gpu single fused=%s' % fused, h1.history['loss'], h1.history['val_loss'])
        st = eng.hist.cpu().numpy()[:5]
        print('   last-epoch step losses single', st)
        p1 = eng.get_params()
        for k in ('mm0', 'mv0', 'mm2', 'mv2', 'W0', 'W_mean', 'beta1'):
            print('   %-7s max|dp-single| %.3e  scale %.3e' % (k, np.abs(p_dp[k] - p1[k]).max(), np.abs(p1[k]).max()))
    ref = oracle_net(ae, p, hs, bn)
    rh = N.fit(ref, X.astype(np.float64), Y.astype(np.float64), sf.astype(np.float64), epochs=epochs, batch_size=B, shuffle_rng=T.FixedOrders(orders), reduce_lr=1, early_stop=0)
    print('gpu dp    ', hist_dp['loss'], hist_dp['val_loss'])
    print('   last-epoch step losses dp    ', steps_dp)
    print('oracle64  ', rh['loss'], rh['val_loss'])


if __name__ == '__main__':
    main()
