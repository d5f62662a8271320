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


# Named workloads = BASELINE.json `configs`.  c3 is the configuration the metric is quoted on (and the default at every N:
# "cells/sec training (ZINB AE, 68k x 20k) at 1/2/4/8 MI355X").  c4 / c5 are the two configurations north_star assigns to an
# 8-GPU node: every rank holds ITS EIGHTH of the matrix (125 000 / 162 500 cells) whatever N is, so `--workload c4 --gpus 1`
# measures exactly the per-GPU step an 8-GPU run starts from.
WORKLOADS = {
    'c3': dict(cells=68579, genes=20000, hidden='64,32,64', ae='zinb-conddisp', batch=4096, shard_of=None,
               name='BASELINE configs[2]: ZINB-conddisp AE on 68k-PBMC-shaped synthetic (68 579 x 20 000)'),
    # c4 / c5 run the THREE-head network (ae 'zinb-conddisp', dca/network.py:366-393: the CLI's default --type, and the one
    # SURVEY 8 sizes these configurations by: P = 6 479 416 / 51.6 M parameters); c4z / c5z are the two-head sibling with a
    # per-gene dispersion (ae 'zinb', dca/network.py:496-550) on the same shapes
    'c4': dict(cells=1000000, genes=25000, hidden='64,32,64', ae='zinb-conddisp', batch=4096, shard_of=8,
               name='BASELINE configs[3]: ZINB AE on 1M cells x 25k genes synthetic, data-parallel 8 x MI355X'),
    'c5': dict(cells=1300000, genes=25000, hidden='512,256,128,256,512', ae='zinb-conddisp', batch=2048, shard_of=8,
               name='BASELINE configs[4]: wide 512-256-128-256-512 ZINB AE on 1.3M-cell atlas-shaped synthetic, 8 GPUs'),
    'c4z': dict(cells=1000000, genes=25000, hidden='64,32,64', ae='zinb', batch=4096, shard_of=8,
                name='configs[3] shape, constant-dispersion ZINB (two heads)'),
    'c5z': dict(cells=1300000, genes=25000, hidden='512,256,128,256,512', ae='zinb', batch=2048, shard_of=8,
                name='configs[4] shape, constant-dispersion ZINB (two heads)'),
}


def source_sha():
    """Fingerprint of the kernel sources (dca_amd/csrc): profiles/pmc_traffic.json carries the one it was measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, 'dca_amd', 'csrc')
    for f in sorted(os.listdir(d)):
        if f.endswith(('.hip', '.hpp', '.cpp', '.inc')):
            h.update(f.encode()); h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:16]


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=0, help='timed steps; default: three epochs of the workload')
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--workload', type=str, default='c3', choices=sorted(WORKLOADS), help='named BASELINE configuration')
    ap.add_argument('--cells', type=int, default=0, help='(ad-hoc runs) override the workload')
    ap.add_argument('--genes', type=int, default=0)
    ap.add_argument('--hidden', type=str, default='')
    ap.add_argument('--ae-type', type=str, default='')
    ap.add_argument('--batch-size', type=int, default=0, help='cells per GPU per step')
    ap.add_argument('--graph', type=str, default='auto', choices=['auto', 'on', 'off'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-seconds', type=float, default=20.0)
    return ap.parse_args()


_SPIN = {}


def _park_device(ms):
    """Keeps the current stream busy for about `ms` milliseconds (torch's spin kernel, calibrated once), so that launches
    enqueued meanwhile run back to back afterwards."""
    import torch
    if 'per_ms' not in _SPIN:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(1000); torch.cuda.synchronize()
        e0.record(); torch.cuda._sleep(2000000); e1.record(); torch.cuda.synchronize()
        _SPIN['per_ms'] = 2000000.0 / max(e0.elapsed_time(e1), 1e-3)
    torch.cuda._sleep(int(ms * _SPIN['per_ms']))


def _graph_timed(fn, n=10, reps=5):
    """Milliseconds per call of `fn` (an idempotent sequence of launches on the current stream) when n calls are captured into
    one hipGraph and replayed: the launch mode of the product's steps, no host in between."""
    import torch
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.graph(g, stream=st, capture_error_mode='thread_local'):        # (RCCL's watchdog thread: see capture_step)
        for _ in range(n):
            fn()
    torch.cuda.current_stream().wait_stream(st)
    g.replay(); torch.cuda.synchronize()
    out = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / n)
    return float(np.median(out))


def kernel_model(name, B, G, hidden, nheads=3):
    """Algorithmic work of one launch: bytes for the HBM-bound
    kernel, flops for the GEMMs."""
    Gp = (G + 3) // 4 * 4
    h1, hL = hidden[0], hidden[-1]
    if name == 'zinb_nll':
        # 3 pre-activations + y read, 3 gradient planes written: fp32 (12 B) or, on the wide networks' planes path, as
        # three bf16 pieces each (18 B)
        return 'hbm', (34.0 if hL > 64 else 28.0) * B * G
    if name == 'rmsprop_clip':
        return 'hbm', None
    if name == 'heads_fused':                           # heads forward + dW + dH in one launch
        return 'mfma', 6.0 * B * hL * nheads * Gp
    fl = {'gemm_enc0_fwd': 2.0 * B * G * h1, 'gemm_enc0_dW': 2.0 * B * G * h1,
          'gemm_heads_fwd': 2.0 * B * hL * nheads * Gp, 'gemm_heads_dW': 2.0 * B * hL * nheads * Gp,
          'gemm_heads_dH': 2.0 * B * hL * nheads * Gp}
    return 'mfma', fl.get(name)


def step_roofline(cells_per_s, B, G, hidden, nheads, n_params, ae_type, heads_ms=None, ppp_heads=3, ppp_enc0=6):
    """SURVEY 8d's STEP-level bounds beside the measured figure (the per-kernel `roofline` prices the dominant launch only):
    algorithmic GEMM flops per cell F = G (4 h1 + 6 hL nheads) + 6 sum h_i h_{i+1}; the matrix-pipe bound at the fp32-MFMA peak
    (SURVEY's 'MFMA bound': the ridge of the fp32 design) and at the dense 16-bit peak divided by the 16-bit products THIS
    arithmetic spends per fp32 product (heads: three fp16 products since round 6; first layer: six bf16 products from the byte
    store, three on the wide networks' planes; hidden stack: six); the HBM bound of the unfused design SURVEY prices (72 G bytes
    per cell + 32 P / B parameter traffic).  `frac` = measured / that matrix-pipe bound (<= 1 by construction);
    `frac_of_bf16x6_bound` = measured / the bound at six products everywhere (the denominator of rounds 2 - 5);
    `frac_of_fp32_ridge` = measured / SURVEY's min(fp32-MFMA, HBM).
    When profiles/sq_pipe_counts.json was measured on THIS source tree: the per-pipe floors of K-HEADS from its counted
    instructions -- vector issue (SQ_INSTS_VALU / 1024 SIMDs x 4 cycles) and matrix pipe (SQ_INSTS_MFMA x 32 cycles / 1024) at
    the profiled clock -- next to its measured launch time."""
    h1, hL = hidden[0], hidden[-1]
    F = G * (4.0 * h1 + 6.0 * hL * nheads) + 6.0 * sum(a * b for a, b in zip(hidden[:-1], hidden[1:]))
    by = 72.0 * G + 32.0 * n_params / B
    b_fp32 = MFMA_F32_PEAK_TFLOPS * 1e12 / F
    b_x6 = MFMA_BF16_PEAK_TFLOPS / 6.0 * 1e12 / F
    F16 = G * (4.0 * h1 * ppp_enc0 + 6.0 * hL * nheads * ppp_heads) + 36.0 * sum(a * b for a, b in zip(hidden[:-1], hidden[1:]))
    b_arith = MFMA_BF16_PEAK_TFLOPS * 1e12 / F16
    b_hbm = HBM_PEAK_GBS * 1e9 / by
    out = {'achieved_cells_s': cells_per_s, 'flops_per_cell': F, 'unfused_hbm_bytes_per_cell': by,
           'bound_fp32_mfma_cells_s': b_fp32, 'bound_bf16x6_cells_s': b_x6, 'bound_16bit_products_cells_s': b_arith,
           'products_per_fp32_product': {'heads': ppp_heads, 'first_layer': ppp_enc0, 'hidden_stack': 6},
           'bound_hbm_unfused_cells_s': b_hbm,
           'bound_fp32_ridge': min(b_fp32, b_hbm), 'frac': cells_per_s / b_arith, 'frac_of_bf16x6_bound': cells_per_s / b_x6,
           'frac_of_fp32_ridge': cells_per_s / min(b_fp32, b_hbm),
           'definition': 'SURVEY.md 8d: F = G (4 h1 + 6 hL heads) + 6 sum h_i h_i+1 flops per cell; bounds = peak / F at 157.3 TF/s '
                         '(fp32 MFMA), at 2 500 TF/s over the 16-bit products this arithmetic issues per fp32 product (heads %d, first '
                         'layer %d, hidden stack 6) and at 2 500 / 6 TF/s (rounds 2 - 5); HBM: 72 G + 32 P / B bytes per cell at 8 TB/s'
                         % (ppp_heads, ppp_enc0)}
    try:
        with open(os.path.join(ROOT, 'profiles', 'sq_pipe_counts.json')) as f:
            pc = json.load(f)
        k = pc['kernels'].get('heads_fused')
        if k and pc.get('source_sha') == source_sha() and k.get('clock_GHz'):
            clk = float(k['clock_GHz']) * 1e9
            valu_ms = 1e3 * float(k['SQ_INSTS_VALU']) / 1024.0 * 4.0 / clk
            mfma_ms = 1e3 * float(k['SQ_INSTS_MFMA']) * 32.0 / 1024.0 / clk
            out['heads_fused_pipe_floors'] = {
                'valu_issue_ms': valu_ms, 'matrix_pipe_ms': mfma_ms, 'measured_ms': heads_ms,
                'profiled_ms': float(k['wall_ns']) * 1e-6, 'clock_GHz': float(k['clock_GHz']),
                'SQ_INSTS_VALU': float(k['SQ_INSTS_VALU']), 'SQ_INSTS_MFMA': float(k['SQ_INSTS_MFMA']),
                'MfmaUtil_pct': k.get('MfmaUtil_pct'), 'VALUBusy_pct': k.get('VALUBusy_pct'),
                'source': 'profiles/sq_pipe_counts.json (tools/gpu_pmc_bench.sh on sources %s): counts are means over the full '
                          'and partial batches of the bench sequence, so are the profiled ms' % pc['source_sha']}
        elif k:
            out['heads_fused_pipe_floors'] = 'not attached: profiles/sq_pipe_counts.json was measured on sources %s, this tree is %s' % (
                pc.get('source_sha'), source_sha())
    except (OSError, ValueError, KeyError):
        pass
    return out


def cpu_baseline(Xh, Yh, sfh, params, hidden, B, budget_s, ae_type='zinb-conddisp'):
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
        net = TorchAE(ae_type, params, hidden, True, dtype=torch.float32)
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


def capture_step(eng, b, counts, k=1, rows_per_slot=None, b_global=None):
    """k consecutive training steps in one hipGraph (the device cursor advances inside the graph)."""
    try:
        g = torch.cuda.CUDAGraph(keep_graph=True)
    except TypeError:
        g = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st, capture_error_mode='thread_local'):     # (RCCL's watchdog thread: see train.py::_StepRunner._capture)
            for _ in range(k):
                eng.train_step(b, b if b_global is None else b_global, counts, rows_per_slot or b)
    torch.cuda.current_stream().wait_stream(st)
    return g


def _mark(msg):
    if os.environ.get('DCA_BENCH_TRACE'):
        torch.cuda.synchronize()
        print('bench: ' + msg, file=sys.stderr, flush=True)


class EpochRunner:
    """Training steps in the order of the fit loop (dca_amd/train.py::fit_engine / dca/train.py:91-98): every epoch visits
    the train rows in a fresh shuffled order, full batches first, the partial one last; on one GPU up to 8 consecutive
    full steps are replayed per hipGraph launch (as train.py::_StepRunner does), the partial batch as its own graph.
    Step k of a run is step k % steps_per_epoch of epoch k // steps_per_epoch; the device cursor walks through the
    concatenated shuffles, so a captured graph serves every epoch."""

    def __init__(self, eng, n_train, B, dev, max_epochs, seed=1234, use_graph=True):
        self.eng, self.n, self.B, self.dev, self.use_graph = eng, n_train, B, dev, use_graph
        self.seq = [B] * (n_train // B) + ([n_train % B] if n_train % B else [])
        self.spe = len(self.seq)
        self.gen = torch.Generator(device='cpu'); self.gen.manual_seed(seed)
        self.max_epochs = max_epochs
        eng.perm = torch.zeros(max_epochs * n_train, dtype=torch.int32, device=dev)     # graphs hold this address
        eng.hist = torch.zeros(max_epochs * self.spe + 16, dtype=torch.float32, device=dev)
        self.graphs = {}
        self.k = 0

    def new_run(self, epochs):
        """Fresh shuffles for `epochs` epochs, cursor and accumulators at zero."""
        assert epochs <= self.max_epochs
        for e in range(epochs):
            self.eng.perm[e * self.n:(e + 1) * self.n].copy_(torch.randperm(self.n, generator=self.gen, dtype=torch.int32))
        self.eng.cursor.zero_(); self.eng.acc.zero_()
        self.k = 0

    def _graph(self, b, k):
        if (b, k) not in self.graphs:
            self.graphs[(b, k)] = capture_step(self.eng, b, [b], k, rows_per_slot=self.B)
        return self.graphs[(b, k)]

    def capture_all(self):
        """Every graph an epoch needs (capturing runs nothing: no step is consumed)."""
        if not self.use_graph:
            return
        full = self.n // self.B
        for k in {min(8, full), full % 8} - {0}:
            self._graph(self.B, k)
        if self.n % self.B:
            self._graph(self.n % self.B, 1)

    def plan(self, steps, start=None):
        """[(batch, consecutive steps)] launches of the next `steps` steps."""
        k0 = self.k if start is None else start
        out = []
        while steps > 0:
            pos = k0 % self.spe
            b = self.seq[pos]
            k = min(steps, 8, (self.n // self.B) - pos) if b == self.B else 1    # consecutive full steps left in this epoch
            out.append((b, k))
            k0 += k; steps -= k
        return out

    def run(self, steps, capture_only=False):
        """The next `steps` steps of the sequence; returns the number of cells they held.  capture_only: only make
        sure every graph the steps need exists (nothing runs, no step is consumed)."""
        cells = 0
        for b, k in self.plan(steps):
            if capture_only:
                if self.use_graph:
                    self._graph(b, k)
                continue
            if self.use_graph:
                self._graph(b, k).replay()
            else:
                for _ in range(k):
                    self.eng.train_step(b, b, [b], self.B)
            self.k += k; cells += b * k
        return cells

    def launches_per_epoch(self):
        full = self.n // self.B
        return (full + 7) // 8 + (1 if self.n % self.B else 0)


def _sclk():
    """Current shader clock as rocm-smi prints it (None when the tool is not there)."""
    import re
    import subprocess
    try:
        out = subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True, timeout=20).stdout
        m = re.search(r'sclk clock level[^(]*\((\d+)Mhz\)', out)
        return int(m.group(1)) if m else None
    except Exception:       # noqa: BLE001
        return None


def after_measurements(eng, args, B, n_train, n_val, G, dev, runner):
    """Same process, same resident matrix, after the timed region (one GPU): three individually timed epochs with the
    validation pass (SURVEY 8d's end-to-end figure), then the reference-default batch 32."""
    out = {}
    _mark('epoch timing')
    times = []
    runner.new_run(4)
    for ep in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        runner.run(runner.spe)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        if n_val:
            eng.eval_loss_sum(n_train, n_train + n_val, 1.0 / (float(n_val) * G))
        torch.cuda.synchronize(); t2 = time.perf_counter()
        if ep > 0:
            times.append((t1 - t0, t2 - t1))
        if ep < 3:
            eng.acc.zero_()
    tr = [t[0] for t in times]; va = [t[1] for t in times]
    acc = eng.acc.cpu().numpy()
    out['epoch'] = {'train_s': {'min': min(tr), 'median': float(np.median(tr)), 'max': max(tr)},
                    'validation_s': {'min': min(va), 'median': float(np.median(va)), 'max': max(va)},
                    'train_cells': n_train, 'validation_cells': n_val,
                    'cells_per_s_train_only': {'min': n_train / max(tr), 'median': n_train / float(np.median(tr)), 'max': n_train / min(tr)},
                    'cells_per_s_incl_validation': n_train / (float(np.median(tr)) + float(np.median(va))),
                    'timed_epochs': len(times), 'val_loss_last': float(acc[1]), 'loss_last': float(acc[0]) / n_train,
                    'steps_per_epoch': runner.spe, 'last_batch': n_train % B,
                    'graph_launches_per_epoch': runner.launches_per_epoch()}
    # ---- batch 32 (dca/train.py:37 default): hipGraph replay, 400 timed steps
    gen = torch.Generator(device='cpu'); gen.manual_seed(99)
    b32, k32 = 32, 400
    eng.perm = torch.zeros((k32 + 16) * b32, dtype=torch.int32, device=dev)
    eng.hist = torch.zeros(k32 + 32, dtype=torch.float32, device=dev)

    def new_order():
        eng.perm.copy_(torch.randperm(n_train, generator=gen, dtype=torch.int32)[:eng.perm.numel()])
        eng.cursor.zero_(); eng.acc.zero_()

    new_order()
    _mark('batch-32 eager step')
    eng.train_step(b32, b32, [b32], b32)
    _mark('batch-32 capture')
    try:
        g32 = capture_step(eng, b32, [b32])
        launches = graph_kernel_nodes(g32) if hasattr(g32, 'raw_cuda_graph') else None
        ks = 8                               # steps per graph launch, as the fit loop replays them (dca_amd/train.py)
        g32k = capture_step(eng, b32, [b32], ks)
        g32k.replay()
        # five runs of 400 steps; beside the time: what the host spent enqueueing the launches (the calls return long before the
        # GPU is done: a host that needs as long as the GPU would starve it) and the shader clock before / after -- a box on
        # which this step is slow (the driver measured 0.109 ms in round 2 and 0.180 in round 3; four boxes of round 4:
        # 0.110-0.111, tools/b32_probe.py) then says why
        runs, enq = [], []
        clk0 = _sclk()
        for _ in range(5):
            new_order()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(k32 // ks):
                g32k.replay()
            t1 = time.perf_counter()
            torch.cuda.synchronize(); runs.append(time.perf_counter() - t0); enq.append(t1 - t0)
        el = float(np.median(runs))
        out['batch32'] = {'ms_per_step': 1e3 * el / k32, 'cells_per_s': k32 * b32 / el, 'launches': launches,
                          'steps': k32, 'launch': 'hipGraph replay, %d steps per graph' % ks,
                          'ms_per_step_runs': [1e3 * r / k32 for r in runs],
                          'host_enqueue_ms_per_graph_launch': 1e3 * float(np.median(enq)) / (k32 // ks),
                          'gpu_ms_per_graph_launch': 1e3 * el / (k32 // ks), 'sclk_before_after': [clk0, _sclk()]}
    except Exception as e:
        out['batch32'] = {'error': str(e)}
    return out


def main():
    args = parse()
    from dca_amd import dist as ddist, synth
    from dca_amd.engine import Engine, EventProfiler
    comm = ddist.init_from_env()
    W, rank = comm.world, comm.rank
    # the N-rank branch also runs with ONE rank under DCA_AMD_DIST_FORCE=1 (a one-rank RCCL communicator: every exchange of
    # the data-parallel step in place, on a one-GPU box)
    multi = bool(getattr(comm, 'dp', W > 1))
    if args.gpus != W:
        if W == 1 and args.gpus > 1:
            raise SystemExit('bench.py --gpus %d must be launched with torch.distributed.run' % args.gpus)
    dev = torch.device('cuda', torch.cuda.current_device())
    wl = dict(WORKLOADS[args.workload])
    adhoc = bool(args.cells or args.genes or args.hidden or args.ae_type)
    cells_total = args.cells or wl['cells']
    G = args.genes or wl['genes']
    hidden = tuple(int(x) for x in (args.hidden or wl['hidden']).split(','))
    ae_type = args.ae_type or wl['ae']
    B = args.batch_size or wl['batch']
    shard_of = None if adhoc else wl['shard_of']
    if shard_of:
        # an 8-GPU configuration: this rank's eighth of the matrix, 90 % of it trained on (rank r holds rows r / 8 .. )
        n_train_global = int(cells_total * 0.9)
        n_local = n_train_global // shard_of
        n_val = 0
    else:
        n_train_global = int(cells_total * 0.9)           # validation_split=0.1 tail is not trained on
        n_local = n_train_global // W                     # equal shards (all-gather of stats)
        n_val = cells_total - n_train_global if not multi else 0  # one GPU: the held-out rows sit behind the train rows
    n_store = n_local + n_val

    # ---- synthetic data, generated and normalised in HBM (not timed)
    Y = synth.generate_counts(n_store, G, device=dev, row_offset=rank)
    # K-PREP (dca_amd/prep.py): size factors, log1p, per-gene z-score on the resident counts;
    # with N ranks the median library size and the gene statistics are global
    from dca_amd import prep
    from dca_amd.ops import HipOps
    pops = HipOps()
    counts = prep.cell_counts(pops, Y, n_store, G)
    med = (comm.all_gather(counts).flatten() if multi else counts).median()
    sf = counts / med
    X, norm = prep.transform(pops, Y, n_store, G, sf, True, True, comm if multi else None, return_norm=True)
    eng = Engine(ae_type, G, G, hidden, True, 0.0, comm=comm)
    eng.init_params(0)
    # norm: how X was made from Y -> the engine keeps the counts as bytes and runs the first layer on the non-zero ones
    eng.attach_device_data(X, Y, sf, norm=norm)
    eng.reserve(max(B, 1024) if not multi else B)           # validation runs in chunks of up to 1024 rows
    eng.clip = 5.0
    eng.set_lr(1e-3)
    counts = [B] * W
    # one GPU: the steps are replayed as hipGraphs -- ~30 launches per 1.3 ms step leave the host little slack (a busy
    # host starves the GPU); replay is GPU-paced.  N > 1 stays eager (collectives).
    use_graph = ((args.graph == 'on') or (args.graph == 'auto' and not multi)) and not multi

    def barrier():
        if multi:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # clock spin-up (not a training step): the first kernels of a process that starts right after another GPU
    # process has exited were measured up to 2.5x slow; ~0.2 s of throw-away GEMMs on scratch buffers first
    sa = torch.randn(2048, 2048, device=dev); sb = torch.randn(2048, 2048, device=dev); sc = torch.empty(2048, 2048, device=dev)
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < 0.2:
        for _ in range(10):
            pops.sgemm(0, 0, 2048, 2048, 2048, sa, 2048, sb, 2048, sc, 2048, split_k=1)
        torch.cuda.synchronize()
    del sa, sb, sc

    prof = None
    comm_ms = None
    if not multi:
        # ---- one GPU: the K timed steps are K consecutive steps of the FIT LOOP's sequence -- epoch after epoch over the
        # train rows in shuffled order, full batches then the partial one -- after W eager warm-up steps and one
        # untimed epoch through the graphs.  The default K is three epochs: `value` is then the >= 3-epoch figure.
        spe_guess = (n_local + B - 1) // B
        K = args.steps if args.steps > 0 else 3 * spe_guess
        runner = EpochRunner(eng, n_local, B, dev, max_epochs=(args.warmup + K) // spe_guess + 6, use_graph=use_graph)
        runner.new_run(runner.max_epochs)
        runner.use_graph = False
        runner.run(args.warmup)                            # eager warm-up steps (kernels' first launches, allocator)
        runner.use_graph = use_graph
        try:
            runner.capture_all()
        except Exception as e:                             # capture refused: time the eager steps instead
            print('bench: hipGraph capture failed (%s); running eager' % e, file=sys.stderr)
            runner.use_graph = use_graph = False
            torch.cuda.synchronize()
        runner.run(runner.spe - (runner.k % runner.spe))   # untimed: to the end of the epoch, first replay of every graph
        runner.run(runner.spe)                             # untimed: one whole epoch through the graphs
        runner.run(K, capture_only=True)                   # (a K that ends inside an epoch needs a shorter graph)
        if not use_graph:
            prof = EventProfiler(); eng.prof = prof
        barrier()
        t_start = time.perf_counter()
        cells_timed = runner.run(K)
        barrier()
        el = time.perf_counter() - t_start
        total_steps = runner.k
        steps_timed = K
        launch_desc = ('hipGraph replay, up to 8 consecutive steps per graph launch, the partial last batch of an epoch as '
                       'its own graph (%d launches per %d-step epoch)' % (runner.launches_per_epoch(), runner.spe)) \
            if use_graph else 'eager'
    else:
        # ---- N GPUs: weak scaling, every rank takes B rows of its shard per step.  The steps are replayed as hipGraphs with
        # the RCCL exchanges captured inside (8 consecutive steps per launch, as train.py::_StepRunner does for the fit
        # loop): an eager data-parallel step is host-bound.  --graph off, DCA_AMD_DP_GRAPH=0 or a capture that raises on
        # any rank -> eager steps on every rank.
        runner = None
        K = args.steps if args.steps > 0 else 48
        EXTRA = 8 + 8 + 8 + 24                        # eager steps with the exchange timers, first replays, per-kernel timing
        total_steps = args.warmup + K + EXTRA
        gen = torch.Generator(device='cpu'); gen.manual_seed(1234 + rank)
        need = total_steps * B
        perms = []
        while sum(p.numel() for p in perms) < need:
            perms.append(torch.randperm(n_local, generator=gen, dtype=torch.int32) if n_local >= B
                         else torch.randint(0, n_local, (B,), generator=gen, dtype=torch.int32))
        eng.perm = torch.cat(perms)[:need].to(dev)
        eng.hist = torch.zeros(total_steps + 1, dtype=torch.float32, device=dev)
        eng.cursor.zero_(); eng.acc.zero_()
        for i in range(args.warmup):
            eng.train_step(B, B * W, counts, B)
        # exposed communication of the EAGER step: what the compute stream spent inside (or waiting for) each exchange
        comm.timer = {}
        for i in range(8):
            eng.train_step(B, B * W, counts, B)
        comm_ms = {k: {'calls_per_step': v[0] / 8, 'ms_per_step': v[1] / 8} for k, v in comm.timer_summary().items()}
        comm.timer = None
        use_graph = args.graph != 'off' and eng.cfg.dp_graph and getattr(comm, 'capturable', False)
        graphs = {}
        if use_graph:
            ok = 1.0
            try:
                for k in {min(8, K), K % 8} - {0}:
                    graphs[k] = capture_step(eng, B, counts, k, rows_per_slot=B, b_global=B * W)
            except Exception as e:                    # noqa: BLE001
                print('bench: capture of the data-parallel step failed on rank %d (%s: %s); eager' % (rank, type(e).__name__, e),
                      file=sys.stderr)
                ok = 0.0
            flag = torch.tensor([ok], device=dev)
            torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
            use_graph = bool(flag.item() > 0.5)
        if use_graph:
            for k, g in graphs.items():              # untimed: the first replay of every graph
                g.replay()
            torch.cuda.synchronize()
            if not bool(torch.isfinite(eng.g[eng.lay.P]).item()):
                raise SystemExit('bench: the replayed data-parallel steps left a non-finite loss')
        else:
            prof = EventProfiler(); eng.prof = prof
        barrier()
        t_start = time.perf_counter()
        if use_graph:
            for i in range(K // 8):
                graphs[8].replay()
            if K % 8:
                graphs[K % 8].replay()
        else:
            for i in range(K):
                eng.train_step(B, B * W, counts, B)
        barrier()
        el = time.perf_counter() - t_start
        cells_timed = K * B * W
        steps_timed = K
        launch_desc = ('hipGraph replay with the RCCL exchanges captured, 8 consecutive steps per graph launch'
                       if use_graph else 'eager')
    elt = torch.tensor([el], dtype=torch.float64, device=dev)
    if multi:
        torch.distributed.all_reduce(elt, op=torch.distributed.ReduceOp.MAX)
    el = float(elt.item())
    eng.prof = None
    loss_last = float(eng.g[eng.lay.P].item())
    loss_first = float(eng.hist[0].item())

    # ---- per-kernel timing (HIP events on the launch stream)
    ksum = prof.summary() if prof is not None else {}
    if use_graph:
        # graph replay hides individual launches: time the dominant kernels in isolation (eager full-batch steps)
        eng.prof = EventProfiler()
        eng.cursor.zero_()
        # The events must see a GPU that is never waiting for the host: a Python-level launch costs 10 - 20 us, a scope of three
        # launches between two event records would otherwise include the host's gaps (measured: K-HEADS' scope 0.79 ms against
        # 0.705 ms for its three kernels under rocprofv3).  The device is parked on a spin kernel while the host enqueues all the
        # profiled steps.
        n_prof = min(steps_timed, 16, n_local // B)
        _park_device(30.0)
        for i in range(n_prof):
            eng.train_step(B, B * W, counts, B)
        ksum = eng.prof.summary(); eng.prof = None
    # The dominant operation once more, as the product launches it: ten consecutive launches captured into ONE hipGraph, replayed
    # between two events.  (Events round every scope of an eager step cost a barrier packet each: the scope of K-HEADS' three
    # launches reads 0.76 - 0.79 ms that way against 0.70 ms for the same three kernels under rocprofv3.)
    graph_ms = {}
    last = getattr(eng, '_last_heads_args', None)
    if last is not None and last[0] == B and not multi:       # (one process only: nothing new is captured beside live communicators)
        try:
            if eng.ws_heads is not None:
                graph_ms['heads_fused'] = _graph_timed(lambda: eng.heads_fused_launch(*last))
            else:
                graph_ms['gemm_heads_fwd'] = _graph_timed(lambda: eng._heads_forward(last[0], last[1]))
        except Exception as exc:                           # (never fatal for the bench line: the event figure stays)
            sys.stderr.write('[bench] graph timing of the dominant operation failed: %r\n' % (exc,))
    kernels = []
    # 16-bit matrix products spent per fp32 product: K-HEADS and the wide networks' plane GEMMs run on two fp16 pieces and three
    # products (round 6); the first layer from the byte store (and the three-piece planes) on three bf16 pieces and six
    wide_h2 = bool(getattr(eng, '_h2', None) and eng.pl is not None and eng._h2(B))
    ppp = {'heads_fused': 3}
    for nm in ('gemm_heads_fwd', 'gemm_heads_dW', 'gemm_heads_dH', 'gemm_enc0_fwd', 'gemm_enc0_dW'):
        ppp[nm] = 3 if wide_h2 else 6
    for name, st in ksum.items():
        bound, work = kernel_model(name, B, G, hidden, nheads={'zinb-conddisp': 3, 'zinb': 2, 'nb-conddisp': 2, 'nb': 1}.get(ae_type, 3))
        if name in graph_ms:                               # the event figure stays beside it
            st = dict(st, eager_event_mean_ms=st['mean_ms'], mean_ms=graph_ms[name])
        ent = {'kernel': name, 'mean_ms': st['mean_ms'], 'median_ms': st.get('median_ms'), 'share_of_step': st['mean_ms'] / (1e3 * el / steps_timed)}
        if 'eager_event_mean_ms' in st:
            ent['eager_event_mean_ms'] = st['eager_event_mean_ms']
            ent['timing'] = 'ten consecutive launches of the operation in one hipGraph, replayed between two events (median of 5 replays)'
        if work:
            if bound == 'hbm':
                ent.update(bound='hbm', achieved=work / (st['mean_ms'] * 1e-3) / 1e9, peak=HBM_PEAK_GBS, unit='GB/s')
            else:
                # The products are computed as `n` 16-bit MFMAs per fp32 product (3: two fp16 pieces; 6: three bf16 pieces): the
                # bound of the matrix pipe for THIS result is the dense 16-bit peak / n -- `peak`.  (The fp32-MFMA peak,
                # 157.3 TF/s, prices the same algorithmic flops on an instruction these kernels do not use and can exceed:
                # a secondary field; so is the fraction of peak / 6, the denominator of rounds 2 - 5.)
                n16 = ppp.get(name, 6)
                ent.update(bound='mfma', achieved=work / (st['mean_ms'] * 1e-3) / 1e12, peak=MFMA_BF16_PEAK_TFLOPS / n16,
                           unit='TFLOP/s')
                ent['products_per_fp32_product'] = n16
                ent['frac_of_bf16x6_peak'] = ent['achieved'] / (MFMA_BF16_PEAK_TFLOPS / 6.0)
                ent['peak_fp32_mfma'] = MFMA_F32_PEAK_TFLOPS
                ent['frac_of_fp32_mfma_peak'] = ent['achieved'] / MFMA_F32_PEAK_TFLOPS
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
                    'timing': 'HIP events on the launch stream around the scope of the launches that make up the operation ('
                              + ('eager steps after the graph-replayed timed region, enqueued while the device is parked on a spin '
                                 'kernel so that no host gap falls between two events' if use_graph else 'inside the timed region')
                              + '); heads_fused = heads_split_h + heads_fused_h2 + heads_reduce_both (three launches: their '
                                'rocprofv3 averages add up to this figure), gemm_enc0_* = operand split + product + reduce / finish'}
            if e.get('timing'):
                roof['timing'] = e['timing'] + ('; the operation = heads_split_h + heads_fused_h2 + heads_reduce_both (their rocprofv3 '
                                                'averages add up to this figure)' if e['kernel'] == 'heads_fused' else
                                                '; the operation = operand maxima + plane splits + gemm_h2w (+ tail sum)') + \
                                 '; eager_event_mean_ms = the same launches between two events of an eager step (each event costs a barrier)'
                roof['eager_event_mean_ms'] = e.get('eager_event_mean_ms')
            if e['bound'] == 'mfma':
                roof['peak_note'] = ('dense 16-bit MFMA peak (2 500 TF/s) / %d: the matrix-pipe bound of an fp32-accurate product '
                                     'computed as %d 16-bit products (%s); frac_of_bf16x6_peak = the same launch against peak / 6, '
                                     'the denominator of rounds 2 - 5; the fp32-MFMA peak is a secondary field'
                                     % (e['products_per_fp32_product'], e['products_per_fp32_product'],
                                        'two fp16 pieces per operand' if e['products_per_fp32_product'] == 3 else 'three bf16 pieces per operand'))
                roof['products_per_fp32_product'] = e['products_per_fp32_product']
                roof['frac_of_bf16x6_peak'] = e['frac_of_bf16x6_peak']
                roof['peak_fp32_mfma'] = e['peak_fp32_mfma']
                roof['frac_of_fp32_mfma_peak'] = e['frac_of_fp32_mfma_peak']
            m = pmc.get(e['kernel'])
            if m and m['shape']['B'] == B and m['shape']['G'] == G and m['shape']['hL'] == hidden[-1] and \
                    m['shape'].get('ae_type', 'zinb-conddisp') == ae_type:
                if m.get('source_sha') == source_sha():
                    roof['traffic'] = m['traffic_bytes']
                    roof['traffic_source'] = ('rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of the same kernel and shape on THIS '
                                              'source tree (profiles/pmc_traffic.json, %s; sources %s): bytes per launch; algorithmic '
                                              'HBM bytes %.0f' % (m.get('raw', 'raw files not recorded'), m['source_sha'],
                                                                  m['algorithmic_hbm_bytes']))
                else:
                    roof['traffic_source'] = ('not attached: profiles/pmc_traffic.json was measured on kernel sources %s, this tree is '
                                              '%s (tools/gpu_pmc_traffic.sh re-measures)' % (m.get('source_sha'), source_sha()))
            break

    if roof is not None:
        nheads = {'zinb-conddisp': 3, 'zinb': 2, 'nb-conddisp': 2, 'nb': 1}.get(ae_type, 3)
        hm = [e['mean_ms'] for e in kernels if e['kernel'] == 'heads_fused']
        roof['step'] = step_roofline(cells_timed / el / W, B, G, hidden, nheads, int(eng.lay.P), ae_type, hm[0] if hm else None,
                                     ppp_heads=3 if (hm or wide_h2) else 6, ppp_enc0=3 if wide_h2 else 6)
    if getattr(comm, 'peer', None) is not None:
        comm.peer.check()                                  # K-PEER: an exchange a rank never joined fails the run here
    extra = {}
    _mark('timed region and kernel timing done')
    if not multi:
        extra = after_measurements(eng, args, B, n_local, n_val, G, dev, runner)
    _mark('after-measurements done')

    if rank == 0:
        out = {
            'metric': 'cells/sec training (ZINB AE, 68k x 20k)' if args.workload == 'c3' and not adhoc else
                      'cells/sec training (%s)' % (args.workload if not adhoc else 'ad-hoc shape'),
            'value': cells_timed / el, 'unit': 'cells/s', 'n_gpus': W, 'steps': steps_timed,
            'warmup': args.warmup, 'ms_per_step': 1e3 * el / steps_timed, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': ('%s [--workload %s]; %s autoencoder %s; ' % (wl['name'], args.workload, ae_type, '-'.join(map(str, hidden)))
                                    if not adhoc else 'ad-hoc run (not a BASELINE shape): %s autoencoder %s on synthetic %d x %d; '
                                    % (ae_type, '-'.join(map(str, hidden)), cells_total, G)) +
                                   ('every rank holds its 1/%d of the matrix: %d train cells resident per GPU, %d GPU(s) running'
                                    % (shard_of, n_local, W) if shard_of else
                                    'train rows %d sharded over %d GPU(s)' % (n_train_global, W)),
                       'workload_key': args.workload if not adhoc else 'adhoc',
                       'batch_per_gpu': B, 'global_batch': B * W, 'hidden': list(hidden),
                       'parallelism': 'dp%d' % W, 'launch': launch_desc,
                       'timed_region': ('%d consecutive steps of the fit loop (epochs of %d full batches of %d cells + one of %d), '
                                        '%d cells; ms_per_step averages over full and partial batches'
                                        % (steps_timed, n_local // B, B, n_local % B, cells_timed)) if not multi else
                                       ('%d steps of %d cells per GPU' % (steps_timed, B)),
                       'exposed_comm_ms_per_step': comm_ms,
                       'exposed_comm_source': ('8 eager steps before the timed region: events round each exchange (and each wait for '
                                               'the asynchronous bucket) on the compute stream') if multi else None,
                       'optimizer': 'RMSprop+clipvalue', 'params': int(eng.lay.P),
                       'heads_d_exp': int(getattr(eng, 'heads_d_exp', 0)),
                       'arithmetic': 'fp32 results.  Heads (K-HEADS; the wide networks\' plane products): operands block-scaled by a '
                                     'power of two and split into two fp16 pieces, three products, fp32 accumulation; first layer from '
                                     'the byte store and the hidden stack: three bf16 pieces, six products (fp32-dot-product accuracy: '
                                     'tests/test_heads_fused_gpu.py::test_x3_products_are_fp32_accurate, tests/test_gemm_h2_gpu.py, '
                                     'tests/test_x3_arith_cpu.py); likelihood in fp32',
                       **extra},
            'loss_first': loss_first, 'loss_last': loss_last,
            'roofline': roof, 'kernels': kernels,
            # the reference-default batch (dca/train.py:37) and the end-to-end epoch once more at the top level (records that
            # prune `config` keep them)
            'batch32': extra.get('batch32'), 'epoch': extra.get('epoch'),
        }
        if not args.no_cpu_baseline and not multi:
            nb = min(n_local, max(B, 4 * B))
            p = eng.get_params()
            # host copy of a bounded sample of the same matrix; weights = current device weights
            out['cpu_baseline'] = cpu_baseline(X[:nb, :G].cpu().numpy(), Y[:nb, :G].cpu().numpy(),
                                               sf[:nb].cpu().numpy(), p, hidden, B, args.cpu_seconds, ae_type)
        else:
            out['cpu_baseline'] = None
        print(json.dumps(out), flush=True)
    if multi:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
