"""The training step at the reference's default batch of 32 (dca/train.py:37) on BASELINE configs[2], looked at the way the
round-3 review asked: the driver measured 0.109 ms per step in round 2 and 0.180 in round 3 on different boxes of the pool.
Prints, for this box: the GPU's clocks, ten back-to-back timings of 400 graph-replayed steps (cold: right after start; warm:
after a second of dense GEMMs), and -- when run under `rocprofv3 --kernel-trace --stats` -- leaves the per-kernel durations
in the profiler's output.

    python tools/b32_probe.py [--rounds 10] [--spin 1.0]
"""
import argparse
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def clocks():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--showperflevel'], capture_output=True, text=True, timeout=20).stdout
        keep = [l.strip() for l in out.splitlines() if any(k in l for k in ('sclk', 'mclk', 'fclk', 'Power', 'Performance Level'))]
        return ' | '.join(keep[:8])
    except Exception as e:      # noqa: BLE001
        return 'rocm-smi unavailable (%s)' % e


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rounds', type=int, default=10)
    ap.add_argument('--spin', type=float, default=1.0)
    args = ap.parse_args()
    from bench import capture_step
    from dca_amd import synth, prep
    from dca_amd.engine import Engine
    from dca_amd.ops import HipOps
    dev = torch.device('cuda')
    ops = HipOps()
    n, G, hidden, B = 68579, 20000, (64, 32, 64), 32
    print('clocks at start:', clocks(), flush=True)
    Y = synth.generate_counts(n, G, device=dev)
    counts = prep.cell_counts(ops, Y, n, G)
    sf = counts / counts.median()
    X, norm = prep.transform(ops, Y, n, G, sf, True, True, return_norm=True)
    eng = Engine('zinb-conddisp', G, G, hidden, True, 0.0)
    eng.init_params(0)
    eng.attach_device_data(X, Y, sf, norm=norm)
    eng.reserve(1024)
    eng.clip = 5.0
    eng.set_lr(1e-3)
    k32 = 400
    gen = torch.Generator(device='cpu'); gen.manual_seed(99)
    eng.perm = torch.randperm(n, generator=gen, dtype=torch.int32)[:(k32 + 16) * B].to(dev)
    eng.hist = torch.zeros(k32 + 32, dtype=torch.float32, device=dev)
    eng.cursor.zero_(); eng.acc.zero_()
    eng.train_step(B, B, [B], B)
    g8 = capture_step(eng, B, [B], 8)
    g8.replay(); torch.cuda.synchronize()

    def timed():
        eng.cursor.zero_(); eng.acc.zero_()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k32 // 8):
            g8.replay()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / k32

    cold = [timed() for _ in range(args.rounds)]
    print('cold  (ms per step, %d x %d steps): %s' % (args.rounds, k32, ' '.join('%.4f' % v for v in cold)), flush=True)
    print('clocks after the cold rounds:', clocks(), flush=True)
    a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev); c = torch.empty(4096, 4096, device=dev)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < args.spin:
        for _ in range(20):
            ops.sgemm(0, 0, 4096, 4096, 4096, a, 4096, b, 4096, c, 4096, split_k=1)
        torch.cuda.synchronize()
    warm = [timed() for _ in range(args.rounds)]
    print('warm  (right after %.1f s of dense GEMMs):     %s' % (args.spin, ' '.join('%.4f' % v for v in warm)), flush=True)
    print('clocks after the warm rounds:', clocks(), flush=True)
    # steps per graph launch: what the host spends enqueueing a launch (the call returns before the GPU is done) against what
    # the GPU spends executing it -- a host that needs longer than the GPU starves it
    for K in (8, 16, 32, 64):
        gk = capture_step(eng, B, [B], K)
        gk.replay(); torch.cuda.synchronize()
        reps = max(1, 384 // K)
        eng.cursor.zero_(); eng.acc.zero_()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            gk.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print('  %2d steps per graph: %.4f ms per step; host enqueue %.3f ms per launch (%.1f us per step), GPU %.3f ms per launch'
              % (K, 1e3 * (t2 - t0) / (reps * K), 1e3 * (t1 - t0) / reps, 1e6 * (t1 - t0) / (reps * K), 1e3 * (t2 - t0) / reps), flush=True)
    time.sleep(1.0)
    idle = [timed() for _ in range(args.rounds)]
    print('after 1 s of idling:                          %s' % ' '.join('%.4f' % v for v in idle), flush=True)
    print('summary: cold median %.4f  warm median %.4f  after-idle first %.4f median %.4f' % (np.median(cold), np.median(warm), idle[0], np.median(idle)))


if __name__ == '__main__':
    main()
