"""Per-phase cycle counters of the wave-specialised K-HEADS kernel (heads_fused_ps_kernel, -DDCA_HEADS_TIMING build):
where a producer wave and a consumer wave spend their time, and how long each waits for the other.

    python tools/timing_heads_ps.py [B] [G] [hL] [flags] [iters]        (COMPACT=1: counts from the byte store)
"""
import os, sys, ctypes
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dca_amd import build as b
so = os.path.join(ROOT, 'tools', '_dbg', 'libdcahip_timing.so')
if not os.path.exists(so):
    os.makedirs(os.path.dirname(so), exist_ok=True)
    b.build_hip(defines=('DCA_HEADS_TIMING',), out=so)
b.LIB = so
b.needs_build = lambda: False
from dca_amd import hip
L = hip.lib()
L.dcahip_heads_set_timing.argtypes = [ctypes.c_void_p]
tim = torch.zeros(2048 * 8 * 10, dtype=torch.int64, device='cuda')
L.dcahip_heads_set_timing(tim.data_ptr())
os.environ['ONLY_FUSED'] = '1'
exec(open(os.path.join(ROOT, 'tools', 'bench_heads.py')).read())
t = tim.cpu().numpy().reshape(-1, 10)
t = t[t[:, 9] > 0]
ps3 = os.environ.get('DCAHIP_HEADS_VARIANT', 'ps3') == 'ps3'
for role, nme, fields in ((0, 'PRODUCER', [(0, 'item prologue (W split -> LDS) + barriers'), (1, 'F: 72 MFMA + weight operand reads'),
                                           (2, 'wait: consumer(s) done with the tile'), (3, 'staging stores + dense likelihood'),
                                           (4, 'partial / next-tile loads, queue pass, hand-over')] +
                                          ([(5, 'dH: 72 MFMA, D split, partial stores')] if ps3 else [])),
                          (1, 'CONSUMER', [(0, 'item prologue + barriers (+ dW tree) / stores'), (5, 'operand load issue'),
                                           (6, 'wait: producer tile full'), (7, 'dW (ps3: one head, 24 MFMA) / dH + dW (ps)')])):
    r = t[t[:, 8] == role]
    if not len(r):
        continue
    tot = r[:, 9].mean()
    print('%s waves %d, mean cycles in the kernel %.0f' % (nme, len(r), tot))
    for i, f in fields:
        print('    %-55s %12.0f  %5.1f%%' % (f, r[:, i].mean(), 100 * r[:, i].mean() / tot))
print('waves with role 0 / 1 per workgroup:', np.bincount(t[:, 8].astype(int)))
