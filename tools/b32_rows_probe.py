"""Is the batch-32 step's first-layer product (and K-HEADS' count reads) waiting for COLD, SCATTERED rows?  The same 400
graph-replayed steps of tools/b32_probe.py with three row orders: the shuffled permutation of the fit loop, consecutive rows
(same bytes per step, few pages), and a set of 256 rows visited over and over (always cache-resident).
    python tools/b32_rows_probe.py"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from bench import capture_step
    from dca_amd import synth, prep
    from dca_amd.engine import Engine
    from dca_amd.ops import HipOps
    dev = torch.device('cuda')
    ops = HipOps()
    n, G, hidden, B = 68579, 20000, (64, 32, 64), 32
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
    orders = {
        'shuffled (the fit loop)': torch.randperm(n, generator=gen, dtype=torch.int32)[:(k32 + 16) * B],
        'consecutive rows': torch.arange((k32 + 16) * B, dtype=torch.int32),
        '256 rows over and over': (torch.arange((k32 + 16) * B, dtype=torch.int32) % 256),
    }
    eng.perm = orders['shuffled (the fit loop)'].to(dev)
    eng.hist = torch.zeros(k32 + 32, dtype=torch.float32, device=dev)
    eng.cursor.zero_(); eng.acc.zero_()
    eng.train_step(B, B, [B], B)
    g8 = capture_step(eng, B, [B], 8)
    g8.replay(); torch.cuda.synchronize()
    for name, order in orders.items():
        eng.perm.copy_(order.to(dev))
        out = []
        for _ in range(5):
            eng.cursor.zero_(); eng.acc.zero_()
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(k32 // 8):
                g8.replay()
            torch.cuda.synchronize()
            out.append(1e3 * (time.perf_counter() - t0) / k32)
        print('%-26s ms per step: %s' % (name, ' '.join('%.4f' % v for v in out)), flush=True)


if __name__ == '__main__':
    main()
