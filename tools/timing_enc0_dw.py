"""Where the time of the second-form first-layer weight-gradient kernel goes, per workgroup (-DDCA_DW_TIMING build of
dcahip_sparse.hip, tools/_dbg/libdcahip_DCA_DW_TIMING.so): clock at entry / after the prologue / after the loop / at the end.
    python tools/timing_enc0_dw.py [B=4096]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dca_amd import build as b
b.LIB = os.path.join(ROOT, 'tools', '_dbg', os.environ.get('DCA_DW_LIB', 'libdcahip_DCA_DW_TIMING.so'))
b.needs_build = lambda: False
from dca_amd import hip, synth, prep, compact
from dca_amd.ops import HipOps
ops = HipOps(); L = hip.lib()
L.dcahip_enc0_dw_set_timing.argtypes = [ctypes.c_void_p]
dev = torch.device('cuda')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G, h, n = 20000, 64, 68579
Y = synth.generate_counts(n, G, device=dev)
counts = prep.cell_counts(ops, Y, n, G); sf = counts / counts.median()
X, norm = prep.transform(ops, Y, n, G, sf, True, True, return_norm=True); del X
cc = compact.build(ops, Y, n, G).with_input(norm['fac'], norm['do_log'], norm['mean'], norm['std'], ops=ops)
cc.ensure_lut(ops)
cur = torch.zeros(1, dtype=torch.int64, device=dev)
perm = torch.randperm(n, device=dev, dtype=torch.int32)[:B].contiguous()
dZ = torch.randn(B, h, device=dev) * 1e-3
gW = torch.zeros(G + 1, h, device=dev)
wsd = torch.zeros(ops.enc0_dw_sparse_workspace_bytes(B, G, h) // 4 + 4, device=dev)
for _ in range(3):
    ops.enc0_dw_sparse(cc, perm, cur, 0, B, G, h, dZ, h, gW, h, wsd)
torch.cuda.synchronize()
tim = torch.zeros(4096 * 8 + 8, dtype=torch.int64, device=dev)
L.dcahip_enc0_dw_set_timing(tim.data_ptr())
s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
s.record(); ops.enc0_dw_sparse(cc, perm, cur, 0, B, G, h, dZ, h, gW, h, wsd); e.record(); torch.cuda.synchronize()
L.dcahip_enc0_dw_set_timing(None)
dbg = tim.cpu().numpy()[4096 * 8:]
if dbg.any():
    print('  a visit of the formula path without a place: step %d, counts %016x %016x, lane/wave %d, or %d, any %d' % (int(dbg[0]), int(dbg[1]) & (2**64-1), int(dbg[2]) & (2**64-1), int(dbg[3]), int(dbg[4]), int(dbg[5])))
t = tim.cpu().numpy()[:4096 * 8].reshape(-1, 8)
nwg = int((t[:, 0] != 0).sum())
t = t[:nwg].astype(float)
print('launch (split + kernel + finish) %.4f ms; %d workgroups' % (s.elapsed_time(e), len(t)))
pro, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
for nme, v in (('prologue', pro), ('loop', loop), ('partial sums out', epi)):
    print('  %-18s mean %8.0f  min %8.0f  max %8.0f cycles' % (nme, v.mean(), v.min(), v.max()))
rt0, rt1 = t[:, 4] - t[:, 4].min(), t[:, 5] - t[:, 4].min()
print('  entry (100 MHz clock): first 0, last %.2f us;  end: first %.2f us, last %.2f us' % (rt0.max() / 100, rt1.min() / 100, rt1.max() / 100))
groups = (G + 511) // 512
nsp = nwg // groups
L_ = np.arange(nwg)
q_, r_ = nwg >> 3, nwg & 7
cell = (L_ & 7) * q_ + np.minimum(L_ & 7, r_) + (L_ >> 3)            # xcd_cell() of dcahip_sparse.hip
bx, by = cell % groups, cell // groups
print('  formula path: entered %d times (waves), %d places;  per workgroup mean %.1f, max %d' % (t[:, 6].sum(), t[:, 7].sum(), t[:, 6].mean(), t[:, 6].max()))
print('  loop cycles against formula-path entries of the workgroup: corr %.3f' % np.corrcoef(loop, t[:, 6])[0, 1])
order = np.argsort(loop)
for i in list(order[:4]) + list(order[-6:]):
    print('    group %2d split %d (xcd %d): loop %7.0f cycles, formula path %3d times, %3d places' % (bx[i], by[i], i & 7, loop[i], t[i, 6], t[i, 7]))
print('  mean loop cycles by xcd:', ' '.join('%.0f' % loop[(L_ & 7) == x].mean() for x in range(8)))
print('  mean loop cycles by split:', ' '.join('%.0f' % loop[by == y].mean() for y in range(nsp)))
if os.environ.get('DW_BY_GROUP'):
    full = by < nsp - 1
    print('  by gene group (splits but the last): mean loop cycles / formula-path entries')
    for g in range(groups):
        sel = full & (bx == g)
        print('    group %2d: %7.0f  %5.1f' % (g, loop[sel].mean(), t[sel, 6].mean()))
