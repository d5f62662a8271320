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
names_f32 = ['loop-top', 'H->LDS', 'F (96 mfma)', 'staging stores', 'Z dense+sparse', 'dH: partial stores + Hd / hv load issue', 'dW (96 mfma)', 'dH: 96 mfma', 'PROLOGUE (W -> LDS)', 'EPILOGUE (dW tree + stores)']
names_x3 = ['loop-top', 'F (72 mfma, weight operands via transposing LDS reads)', '-', 'staging stores', 'Z dense+sparse',
            'dH: partial stores + operand load issue', 'dW (72 mfma + D split)', 'dH (72 mfma + D split)', 'PROLOGUE (W split -> LDS)',
            'EPILOGUE (dW tree + stores)']
names = names_f32 if os.environ.get('DCA_HEADS_F32MFMA') == '1' else names_x3
tot = t.sum(1).mean()
print('waves', len(t), 'mean cycles per wave (s_memtime @100MHz ticks?)', tot)
for i, nme in enumerate(names):
    print('  %-50s %12.0f  %5.1f%%' % (nme, t[:, i].mean(), 100 * t[:, i].mean() / tot))
# spread of the main-loop time inside a workgroup (8 waves): what the barrier in front of the dW tree waits for
full = tim.cpu().numpy().reshape(-1, 10)
nw = 8
g = full[:(len(full) // nw) * nw].reshape(-1, nw, 10)
g = g[g[:, :, :8].sum(axis=(1, 2)) > 0]
loop = g[:, :, :8].sum(axis=2)                      # [workgroup, wave]
print('workgroups', len(g), ' loop cycles per wave: mean %.0f  within-workgroup max-mean %.0f  max-min %.0f' % (
    loop.mean(), (loop.max(axis=1) - loop.mean(axis=1)).mean(), (loop.max(axis=1) - loop.min(axis=1)).mean()))
print('epilogue per wave: mean %.0f  min over the workgroup (= the last arriver) %.0f' % (g[:, :, 9].mean(), g[:, :, 9].min(axis=1).mean()))
# wave = g * WR + r (g = gene tile of the pair, r = row slot): where does the spread come from?
lg = loop.reshape(len(loop), 2, 4)
print('spread between the two gene tiles of a workgroup |mean_g0 - mean_g1|: %.0f   spread over row slots within a gene tile (max-min): %.0f'
      % (np.abs(lg[:, 0].mean(axis=1) - lg[:, 1].mean(axis=1)).mean(), (lg.max(axis=2) - lg.min(axis=2)).mean()))
