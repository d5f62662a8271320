"""K-HEADS against the fp64 oracle over shapes that exercise the work plans of make_heads_plan (uniform plans with 1 .. 4 batch
splits, tail launches with 4 / 8 / 16 splits, ragged last gene tiles, batches that are not multiples of 32), with the
tolerances of tests/test_heads_fused_gpu.py.      python tools/fuzz_heads_plans.py [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import test_heads_fused_gpu as T      # noqa: E402
from dca_amd.ops import HipOps        # noqa: E402

ops = HipOps()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.RandomState(seed)
shapes = [(2048, 17000), (2048, 17500), (2048, 25500), (2048, 26011), (4096, 9000), (4096, 12500), (4096, 13007),
          (1999, 16500), (3000, 9001), (1024, 20000), (700, 5000), (4096, 2000)]
shapes += [(int(rng.randint(160, 3000)), int(rng.randint(600, 20000))) for _ in range(4)]
thr = max(1, min(64, os.cpu_count() or 1))
for i, (B, G) in enumerate(shapes):
    flags = (1, 3, 0, 2)[i % 4]
    t0 = time.time()
    out = T.run_case(ops, flags, B, G, 64 if i % 3 else 50, seed=seed + i, ridge=0.02 if flags & 1 else 0.0, threads=thr)
    T.check(out)
    print('ok  B %5d  G %6d  flags %d  (%.1f s)' % (B, G, flags, time.time() - t0), flush=True)
print('all %d shapes inside the tolerances' % len(shapes))
