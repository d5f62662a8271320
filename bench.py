#!/usr/bin/env python
"""bench.py -- training throughput of the ZINB-conddisp autoencoder on MI355X.

Metric (BASELINE.json): cells/sec training, ZINB AE 64-32-64 on a synthetic 68 579 x 20 000
count matrix (BASELINE configs[2], the configuration the metric is quoted on; it fits one GPU).
A "step" = one pass of the hot path over one minibatch: forward (Dense/BN/ReLU x3, three
heads), ZINB NLL + gradient, full backward, clipvalue + RMSprop.  Inputs are resident in HBM
before the timed region starts.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torchrun)

Prints ONE JSON line on rank 0 (contract in the repository task statement), with
  roofline      the dominant kernel: HIP events around each of its launches (on the launch stream).  With the step
                replayed as a hipGraph the individual launches are invisible, so the events come from eager steps of
                the same workload run right AFTER the timed region (config.launch says which); the rocprofv3
                kernel trace of the same command (profiles/) reports the same averages
  cpu_baseline  the oracle's torch-CPU port of the same step on the host cores (rank 0, N=1)
  config.batch32   the reference-default batch (dca/train.py:37) measured in the same process after the timed
                   region: ms/step, cells/s, kernel launches per step (N = 1 only)
  config.epoch     SURVEY 8d's end-to-end epoch at the bench batch: 3 timed epochs (after 1 warm-up) over all
                   train rows incl. the last partial batch, device-synchronised at the epoch ends, plus the
                   validation pass over the held-out 10 % (N = 1 only)
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable)
MFMA_F32_PEAK_TFLOPS = 157.3   # dense fp32 MFMA peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak (the headline 5 PF figure includes 2:1 sparsity)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=30)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--cells', type=int, default=68579)
    ap.add_argument('--genes', type=int, default=20000)
    ap.add_argument('--hidden', type=str, default='64,32,64')
    ap.add_argument('--batch-size', type=int, default=4096, help='cells per GPU per step')
    ap.add_argument('--graph', type=str, default='auto', choices=['auto', 'on', 'off'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    return ap.parse_args()


def kernel_model(name, B, G, hidden, nheads=3):
    """Algorithmic work of one launch: bytes for the HBM-bound
    kernel, flops for the GEMMs."""
    Gp = (G + 3) // 4 * 4
    h1, hL = hidden[0], hidden[-1]
    if name == 'zinb_nll':
        return 'hbm', 28.0 * B * G                      # 3 pre-acts + y read, 3 grads written
    if name == 'rmsprop_clip':
        return 'hbm', None
    if name == 'heads_fused':                           # heads forward + dW + dH in one launch
        return 'mfma', 6.0 * B * hL * nheads * Gp
    fl = {'gemm_enc0_fwd': 2.0 * B * G * h1, 'gemm_enc0_dW': 2.0 * B * G * h1,
          'gemm_heads_fwd': 2.0 * B * hL * nheads * Gp, 'gemm_heads_dW': 2.0 * B * hL * nheads * Gp,
          'gemm_heads_dH': 2.0 * B * hL * nheads * Gp}
    return 'mfma', fl.get(name)


def cpu_baseline(Xh, Yh, sfh, params, hidden, B, budget_s):
    """Times the oracle's torch-CPU port of the training step (oracle/torch_ref.py) on the host:
    at the reference's default batch size 32 (train.py:37 -- also the CPU's best throughput)
    and at the batch size of the GPU run, on a bounded sample of the same matrix."""
    from oracle.torch_ref import TorchAE
    # torch's CPU kernels stop scaling (and then collapse) far below the box's core count on
    # this op mix; 16 threads measured best on the 256-core host (tools/cpu_sweep.py)
    threads = max(1, min(16, os.cpu_count() or 1))
    torch.set_num_threads(threads)
    n = Xh.shape[0]
    X, Y, S = torch.as_tensor(Xh), torch.as_tensor(Yh), torch.as_tensor(sfh)

    def run(b, budget, max_steps):
        net = TorchAE('zinb-conddisp', params, hidden, True, dtype=torch.float32)
        nb = max(n // b, 1)
        if b <= 256:
            net.train_step(X[:b], Y[:b], S[:b])               # warm-up
        t0 = time.perf_counter(); steps = 0
        while True:
            s = (steps % nb) * b
            net.train_step(X[s:s + b], Y[s:s + b], S[s:s + b])
            steps += 1
            el = time.perf_counter() - t0
            if el >= budget or steps >= max_steps:
                break
        return steps, el

    s32, e32 = run(32, 0.45 * budget_s, 400)
    sB, eB = run(B, 0.55 * budget_s, 50) if B != 32 else (s32, e32)
    v32, vB = s32 * 32 / e32, sB * B / eB
    return {'value': max(v32, vB), 'unit': 'cells/s', 'cores': threads, 'kind': 'port',
            'sample': 'oracle/torch_ref.py (torch-CPU fp32 autograd port of the step, %d threads) on the '
                      'first %d cells of the same synthetic matrix: batch 32 (reference default): %d '
                      'steps in %.1f s = %.0f cells/s; batch %d (as the GPU run): %d steps in %.1f s = '
                      '%.0f cells/s; value = the better of the two'
                      % (threads, n, s32, e32, v32, B, sB, eB, vB),
            'cells_per_s_batch32': v32, 'cells_per_s_bench_batch': vB, 'host_cores': os.cpu_count()}


def graph_kernel_nodes(graph):
    """Kernel nodes of a captured step (hipGraphGetNodes / hipGraphNodeGetType through libamdhip64)."""
    import ctypes
    try:
        hip = ctypes.CDLL('libamdhip64.so')
        g = ctypes.c_void_p(graph.raw_cuda_graph())
        n = ctypes.c_size_t(0)
        if hip.hipGraphGetNodes(g, None, ctypes.byref(n)) != 0:
            return None
        nodes = (ctypes.c_void_p * n.value)()
        if hip.hipGraphGetNodes(g, nodes, ctypes.byref(n)) != 0:
            return None
        kernels = 0
        for i in range(n.value):
            ty = ctypes.c_int(-1)
            hip.hipGraphNodeGetType(ctypes.c_void_p(nodes[i]), ctypes.byref(ty))
            kernels += ty.value == 0                      # hipGraphNodeTypeKernel
        return int(kernels)
    except Exception:
        return None


def capture_step(eng, b, counts, k=1):
    """k consecutive training steps in one hipGraph (the device cursor advances inside the graph)."""
    try:
        g = torch.cuda.CUDAGraph(keep_graph=True)
    except TypeError:
        g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(k):
                eng.train_step(b, b, counts, b)
    torch.cuda.current_stream().wait_stream(st)
    return g


def _mark(msg):
    if os.environ.get('DCA_BENCH_TRACE'):
        torch.cuda.synchronize()
        print('bench: ' + msg, file=sys.stderr, flush=True)


def after_measurements(eng, args, B, n_train, n_val, G, dev):
    """Same process, same resident matrix, after the timed region (one GPU): the reference-default batch 32 and
    the end-to-end epoch (all train rows incl. the last partial batch + the validation pass)."""
    out = {}
    gen = torch.Generator(device='cpu'); gen.manual_seed(99)
    # captured steps hold the addresses of eng.perm / eng.hist: ONE buffer each for everything below, refilled in place
    b32, k32 = 32, 400
    eng.perm = torch.zeros(max(n_train, (k32 + 8) * b32), dtype=torch.int32, device=dev)
    eng.hist = torch.zeros(max(n_train // b32, k32) + 16, dtype=torch.float32, device=dev)

    def new_order(count):
        eng.perm[:count].copy_(torch.randperm(n_train, generator=gen, dtype=torch.int32)[:count])
        eng.cursor.zero_(); eng.acc.zero_()

    # ---- batch 32 (dca/train.py:37 default): hipGraph replay, 400 timed steps
    new_order((k32 + 8) * b32)
    _mark('batch-32 eager step')
    eng.train_step(b32, b32, [b32], b32)
    _mark('batch-32 capture')
    try:
        g32 = capture_step(eng, b32, [b32])
        launches = graph_kernel_nodes(g32) if hasattr(g32, 'raw_cuda_graph') else None
        ks = 8                               # steps per graph launch, as the fit loop replays them (dca_amd/train.py)
        g32k = capture_step(eng, b32, [b32], ks)
        g32k.replay()
        new_order((k32 + 8) * b32)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k32 // ks):
            g32k.replay()
        torch.cuda.synchronize(); el = time.perf_counter() - t0
        out['batch32'] = {'ms_per_step': 1e3 * el / k32, 'cells_per_s': k32 * b32 / el, 'launches': launches,
                          'steps': k32, 'launch': 'hipGraph replay, %d steps per graph' % ks}
    except Exception as e:
        out['batch32'] = {'error': str(e)}
    _mark('epoch timing')
    # ---- one epoch at the bench batch: train rows in shuffled order, last partial batch included, then validation
    steps_full, b_last = n_train // B, n_train % B
    new_order(n_train)
    eng.train_step(B, B, [B], B)
    gB = capture_step(eng, B, [B])
    times = []
    for ep in range(4):
        new_order(n_train)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(steps_full):
            gB.replay()
        if b_last:
            eng.train_step(b_last, b_last, [b_last], B)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if n_val:
            eng.eval_loss_sum(n_train, n_train + n_val, 1.0 / (float(n_val) * G))
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if ep > 0:
            times.append((t1 - t0, t2 - t1))
    tr = float(np.median([t[0] for t in times])); va = float(np.median([t[1] for t in times]))
    acc = eng.acc.cpu().numpy()
    out['epoch'] = {'train_s': tr, 'validation_s': va, 'train_cells': n_train, 'validation_cells': n_val,
                    'cells_per_s_train_only': n_train / tr, 'cells_per_s_incl_validation': n_train / (tr + va),
                    'timed_epochs': len(times), 'val_loss_last': float(acc[1]), 'loss_last': float(acc[0]) / n_train,
                    'steps_per_epoch': steps_full + (1 if b_last else 0), 'last_batch': b_last,
                    'train_s_each': [round(t[0], 6) for t in times]}
    return out


def main():
    args = parse()
    from dca_amd import dist as ddist, synth
    from dca_amd.engine import Engine, EventProfiler
    comm = ddist.init_from_env()
    W, rank = comm.world, comm.rank
    if args.gpus != W:
        if W == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run' % args.gpus)
    dev = torch.device('cuda', torch.cuda.current_device())
    hidden = tuple(int(x) for x in args.hidden.split(','))
    G, B = args.genes, args.batch_size
    n_train_global = int(args.cells * 0.9)               # validation_split=0.1 tail is not trained on
    t0, n_local = ddist.shard(n_train_global, W, rank)
    n_local = n_train_global // W                         # equal shards (all-gather of stats)
    n_val = args.cells - n_train_global if W == 1 else 0  # one GPU: the held-out rows sit behind the train rows
    n_store = n_local + n_val

    # ---- synthetic data, generated and normalised in HBM (not timed)
    Y = synth.generate_counts(n_store, G, device=dev, row_offset=rank)
    # K-PREP (dca_amd/prep.py): size factors, log1p, per-gene z-score on the resident counts;
    # with N ranks the median library size and the gene statistics are global
    from dca_amd import prep
    from dca_amd.ops import HipOps
    pops = HipOps()
    counts = prep.cell_counts(pops, Y, n_store, G)
    med = (comm.all_gather(counts).flatten() if W > 1 else counts).median()
    sf = counts / med
    X, norm = prep.transform(pops, Y, n_store, G, sf, True, True, comm if W > 1 else None, return_norm=True)
    eng = Engine('zinb-conddisp', G, G, hidden, True, 0.0, comm=comm)
    eng.init_params(0)
    # norm: how X was made from Y -> the engine keeps the counts as bytes and runs the first layer on the non-zero ones
    eng.attach_device_data(X, Y, sf, norm=norm)
    eng.reserve(max(B, 1024) if W == 1 else B)           # validation runs in chunks of up to 1024 rows
    eng.clip = 5.0
    eng.set_lr(1e-3)
    total_steps = args.warmup + 8 + args.steps          # + one untimed replay of the (up to 8-step) graph
    # shuffled row order: as many reshuffles of the shard as the run needs
    gen = torch.Generator(device='cpu'); gen.manual_seed(1234 + rank)
    need = total_steps * B
    perms = []
    while sum(p.numel() for p in perms) < need:
        perms.append(torch.randperm(n_local, generator=gen, dtype=torch.int32) if n_local >= B
                     else torch.randint(0, n_local, (B,), generator=gen, dtype=torch.int32))
    eng.perm = torch.cat(perms)[:need].to(dev)
    eng.hist = torch.zeros(total_steps + 1, dtype=torch.float32, device=dev)
    eng.cursor.zero_(); eng.acc.zero_()
    counts = [B] * W
    # one GPU: the step is replayed as a hipGraph at every batch size -- ~40 launches per 1.8 ms step leave the host
    # little slack (a busy host starves the GPU: one eager run in ~10 measured 2.66 ms/step with unchanged kernel
    # times); replay is GPU-paced (1.784 ms/step, run-to-run identical).  N > 1 stays eager (collectives).
    use_graph = (args.graph == 'on') or (args.graph == 'auto' and W == 1)
    use_graph = use_graph and W == 1

    def barrier():
        if W > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # clock spin-up (not a training step): the first kernels of a process that starts right after another GPU
    # process has exited were measured up to 2.5x slow (bench_heads: 3.5-3.9 ms instead of 1.32); ~0.2 s of
    # throw-away GEMMs on scratch buffers before the W warm-up steps
    sa = torch.randn(2048, 2048, device=dev); sb = torch.randn(2048, 2048, device=dev); sc = torch.empty(2048, 2048, device=dev)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.2:
        for _ in range(10):
            pops.sgemm(0, 0, 2048, 2048, 2048, sa, 2048, sb, 2048, sc, 2048, split_k=1)
        torch.cuda.synchronize()
    del sa, sb, sc

    # steps per graph launch: the fit loop replays 8 consecutive steps per launch (dca_amd/train.py::_StepRunner); here the
    # largest divisor of --steps up to 8, so that exactly --steps steps are timed
    steps_per_graph = max(k for k in range(1, 9) if args.steps % k == 0) if use_graph else 1
    graph = None
    for i in range(args.warmup):
        if use_graph and i == 1:
            try:
                graph = torch.cuda.CUDAGraph()
                s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    with torch.cuda.graph(graph, stream=s):
                        for _ in range(steps_per_graph):
                            eng.train_step(B, B * W, counts, B)
                torch.cuda.current_stream().wait_stream(s)
            except Exception as e:                       # capture refused: time the eager step instead
                print('bench: hipGraph capture failed (%s); running eager' % e, file=sys.stderr)
                graph, use_graph = None, False
                torch.cuda.synchronize()
        eng.train_step(B, B * W, counts, B)
    if use_graph and graph is None:
        use_graph = False
    if graph is not None:
        graph.replay()                                   # untimed: first replay of the graph (steps_per_graph more warmup steps)
    total_steps = args.warmup + (steps_per_graph if graph is not None else 0) + args.steps
    prof = None
    if not use_graph:
        prof = EventProfiler(); eng.prof = prof
    barrier()
    t_start = time.perf_counter()
    if graph is not None:
        for i in range(args.steps // steps_per_graph):
            graph.replay()
    else:
        for i in range(args.steps):
            eng.train_step(B, B * W, counts, B)
    barrier()
    el = time.perf_counter() - t_start
    elt = torch.tensor([el], dtype=torch.float64, device=dev)
    if W > 1:
        torch.distributed.all_reduce(elt, op=torch.distributed.ReduceOp.MAX)
    el = float(elt.item())
    eng.prof = None
    losses = eng.hist[:total_steps].cpu().numpy()

    # ---- per-kernel timing (HIP events on the launch stream)
    ksum = prof.summary() if prof is not None else {}
    if use_graph:
        # graph replay hides individual launches: time the dominant kernels in isolation
        eng.prof = EventProfiler()
        eng.cursor.zero_()
        for i in range(min(args.steps, 20)):
            eng.train_step(B, B * W, counts, B)
        ksum = eng.prof.summary(); eng.prof = None
    kernels = []
    for name, st in ksum.items():
        bound, work = kernel_model(name, B, G, hidden)
        ent = {'kernel': name, 'mean_ms': st['mean_ms'], 'share_of_step': st['total_ms'] / st['count'] / (1e3 * el / args.steps)}
        if work:
            if bound == 'hbm':
                ent.update(bound='hbm', achieved=work / (st['mean_ms'] * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s')
            else:
                ent.update(bound='mfma', achieved=work / (st['mean_ms'] * 1e-3) / 1e12, peak=MFMA_F32_PEAK_TFLOPS, unit='TFLOP/s')
            ent['frac'] = ent['achieved'] / ent['peak']
        kernels.append(ent)
    kernels.sort(key=lambda e: -e['mean_ms'])
    pmc = {}
    try:
        with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as f:
            pmc = json.load(f)
    except (OSError, ValueError):
        pass
    roof = None
    for e in kernels:
        if 'frac' in e:
            roof = {'kernel': e['kernel'], 'bound': e['bound'], 'achieved': e['achieved'], 'peak': e['peak'],
                    'unit': e['unit'], 'frac': e['frac'], 'traffic': None,
                    'traffic_source': None,
                    'timing': 'HIP events around each launch, ' + ('isolated eager steps after the graph-replayed timed region' if use_graph else 'inside the timed region')}
            if e['bound'] == 'mfma':
                # the two honest denominators: `peak` prices the algorithmic fp32 flops against the fp32-MFMA peak (what an
                # fp32 result costs on this chip's matrix pipe); the kernel computes them as six bf16 products per fp32
                # product on the bf16 pipe, whose bound for the SAME result is the dense bf16 peak / 6
                roof['peak_bf16_pipe_over_6'] = MFMA_BF16_PEAK_TFLOPS / 6.0
                roof['frac_of_bf16_pipe_over_6'] = e['achieved'] / (MFMA_BF16_PEAK_TFLOPS / 6.0)
                roof['note'] = ('K-HEADS is bound by the SUM of its matrix and vector instruction cycles per SIMD '
                                '(DESIGN.md 4.1): MfmaUtil 29 %, VALUBusy 52 % (profiles/r02z_sq_counters_per_kernel.csv)')
            m = pmc.get(e['kernel'])
            if m and m['shape']['B'] == B and m['shape']['G'] == G and m['shape']['hL'] == hidden[-1]:
                roof['traffic'] = m['traffic_bytes']
                roof['traffic_source'] = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same kernel and shape '
                                          '(profiles/pmc_traffic.json, profiles/r02z_pmc_traffic/): bytes per launch; '
                                          'algorithmic HBM bytes %.0f' % m['algorithmic_hbm_bytes'])
            break

    extra = {}
    _mark('timed region and kernel timing done')
    if W == 1:
        extra = after_measurements(eng, args, B, n_local, n_val, G, dev)
    _mark('after-measurements done')

    if rank == 0:
        out = {
            'metric': 'cells/sec training (ZINB AE, 68k x 20k)',
            'value': args.steps * B * W / el, 'unit': 'cells/s', 'n_gpus': W, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': 1e3 * el / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'zinb-conddisp autoencoder %s on synthetic %d x %d counts '
                                   '(%s); train rows %d sharded over %d GPU(s)'
                                   % ('-'.join(map(str, hidden)), args.cells, G,
                                      'BASELINE configs[2]' if (args.cells, G, hidden) == (68579, 20000, (64, 32, 64))
                                      else 'not a BASELINE shape: ad-hoc run', n_train_global, W),
                       'batch_per_gpu': B, 'global_batch': B * W, 'hidden': list(hidden),
                       'parallelism': 'dp%d' % W, 'launch': ('hipGraph replay, %d steps per graph' % steps_per_graph) if use_graph else 'eager',
                       'optimizer': 'RMSprop+clipvalue', 'params': int(eng.lay.P),
                       'arithmetic': 'fp32 results: matrix products as three-way bf16 splits, six products, fp32 accumulation '
                                     '(fp32-dot-product accuracy, tests/test_heads_fused_gpu.py::test_x3_products_are_fp32_accurate); '
                                     'likelihood in fp32',
                       **extra},
            'loss_first': float(losses[0]), 'loss_last': float(losses[total_steps - 1]),
            'roofline': roof, 'kernels': kernels,
        }
        if not args.no_cpu_baseline and W == 1:
            nb = min(n_local, max(B, 4 * B))
            p = eng.get_params()
            # host copy of a bounded sample of the same matrix; weights = current device weights
            out['cpu_baseline'] = cpu_baseline(X[:nb, :G].cpu().numpy(), Y[:nb, :G].cpu().numpy(),
                                               sf[:nb].cpu().numpy(), p, hidden, B, args.cpu_seconds)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if W > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
