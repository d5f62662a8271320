"""Debug build of the fused heads kernel with per-phase cycle counters (s_memtime)."""
import os, sys, ctypes, subprocess
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
src = os.path.join(ROOT, 'dca_amd', 'csrc')
so = os.path.join(ROOT, 'tools', '_dbg', 'libdcahip_timing.so')
if not os.path.exists(so):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-DDCA_HEADS_TIMING',
                           '-I' + os.path.join(ROOT, 'include'), '-o', so] + [os.path.join(src, f) for f in
                           __import__('dca_amd.build', fromlist=['SOURCES']).SOURCES])
from dca_amd import build as b
b.LIB = so
b.needs_build = lambda: False
from dca_amd import hip
from dca_amd.ops import HipOps
ops = HipOps()
L = hip.lib()
L.dcahip_heads_set_timing.argtypes = [ctypes.c_void_p]
tim = torch.zeros(2048 * 8 * 10, dtype=torch.int64, device='cuda')
L.dcahip_heads_set_timing(tim.data_ptr())
sys.argv = [sys.argv[0]] + sys.argv[1:]
exec(open(os.path.join(ROOT, 'tools', 'bench_heads.py')).read())
t = tim.cpu().numpy().reshape(-1, 10)
t = t[t.sum(1) > 0]
names = ['loop-top', 'H->LDS', 'F (96 mfma)', 'staging stores', 'Z dense+sparse', 'dH (96 mfma) + partial store', 'Hd loads + dW (96 mfma)', 'rest', 'PROLOGUE (W -> LDS)', 'EPILOGUE (dW tree + stores)']
tot = t.sum(1).mean()
print('waves', len(t), 'mean cycles per wave (s_memtime @100MHz ticks?)', tot)
for i, nme in enumerate(names):
    print('  %-16s %12.0f  %5.1f%%' % (nme, t[:, i].mean(), 100 * t[:, i].mean() / tot))
