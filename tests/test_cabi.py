"""The C-ABI library builds, loads (no GPU needed) and exports every symbol include/dcahip.h declares."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'dcahip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(dcahip_\w+)\s*\(', src)))


def test_library_exports_every_declared_symbol():
    from dca_amd import build, hip
    lib_path = build.build_hip(verbose=False)
    assert os.path.exists(lib_path)
    L = ctypes.CDLL(lib_path)
    names = _declared()
    assert len(names) >= 17
    for n in names:
        assert hasattr(L, n), 'libdcahip.so does not export %s' % n
    # the Python binding table covers the header as well
    bound = set(hip._SIGNATURES)
    assert set(names) <= bound, set(names) - bound
    L.dcahip_version.restype = ctypes.c_int
    assert L.dcahip_version() == 1


def test_product_library_has_no_setters_and_no_experiment_kernels():
    """include/dcahip.h: "keeps no global state ... every choice of kernel is a pure function of the arguments".  The product
    library exports no `*_set_*` switch, and the kernels that were measured and lost (pipelined one-wave-per-SIMD K-HEADS, the
    non-zero-only first-layer forward, the small-batch byte-store weight gradient, the four-wave matrix-pipe forward) are in
    experiment builds (-DDCA_EXP_*) only: neither their entry points nor their device code is in libdcahip.so."""
    import shutil
    import subprocess
    from dca_amd import build
    lib_path = build.build_hip(verbose=False)
    nm = shutil.which('nm') or '/opt/rocm/lib/llvm/bin/llvm-nm'
    syms = subprocess.run([nm, '-D', '--defined-only', lib_path], capture_output=True, text=True, check=True).stdout
    exported = re.findall(r'\b(dcahip_\w+)', syms)
    assert len(exported) >= 70
    assert not [n for n in exported if '_set_' in n], [n for n in exported if '_set_' in n]
    for gone in ('dcahip_enc0_dw_small', 'dcahip_enc0_fwd_sparse', 'dcahip_heads_set_p4_min_tiles'):
        assert gone not in exported
    assert set(exported) == set(_declared()), set(exported) ^ set(_declared())
    blob = open(lib_path, 'rb').read()
    for kernel in (b'heads_fused_p4_kernel', b'heads_fused_x3_kernel', b'enc0_dw_small_kernel', b'enc0_fwd_kernel'):
        assert kernel not in blob, kernel
    assert b'enc0_fwd_lut_kernelILi64ELi2E' not in blob and b'enc0_fwd_lut_kernelILi64ELi1E' in blob


def test_product_path_fails_loudly_without_gpu():
    import pytest
    import torch
    from dca_amd.ops import HipOps
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        HipOps()


def test_heads_kernel_stays_inside_its_scratch_budget():
    """K-HEADS runs at 2 waves/SIMD with 256 VGPRs; builds whose spills pushed the private segment of the main
    variants up showed sporadic 3x slow launches on the MI355X in round 1 (the runtime's scratch handling) and every
    in-loop spill reload waits for all global prefetches in flight, so the budget is part of the contract: the 8-wave
    variants of the byte-store path (the product path) stay <= 128 bytes/lane, the four-wave ones <= 32, LDS <= 160 KiB."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    src = os.path.join(ROOT, 'dca_amd', 'csrc', 'dcahip_heads.hip')
    out = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(ROOT, 'include'),
                          '--cuda-device-only', '-c', src, '-o', os.devnull, '-Rpass-analysis=kernel-resource-usage'],
                         capture_output=True, text=True, check=True).stderr
    names = re.findall(r'Function Name: (\S+)', out)
    scratch = [int(x) for x in re.findall(r'ScratchSize \[bytes/lane\]: (\d+)', out)]
    lds = [int(x) for x in re.findall(r'LDS Size \[bytes/block\]: (\d+)', out)]
    assert len(names) == len(scratch) == len(lds) and len(names) >= 16
    seen = 0
    for n, s, l in zip(names, scratch, lds):
        if 'heads_fused' not in n:
            continue
        assert l <= 163840, (n, l)
        if 'heads_fused_h2_kernel' in n:
            seen += 1
            assert s <= (128 if n.endswith('Lb1EEEvNS_10HeadsArgs2E') else 200), (n, s)      # byte store | fp32 counts
        if 'heads_fused_small_kernel' in n:
            seen += 1
            assert s <= 32, (n, s)
    assert seen == 16         # {zinb, nb} x {conditional, constant dispersion} x {8 waves, 4 waves} x {fp32 counts, byte store}


def _lib():
    from dca_amd import build
    L = ctypes.CDLL(build.build_hip(verbose=False))
    for n in ('dcahip_enc0_fwd_lut_workspace_bytes', 'dcahip_enc0_dw_sparse_workspace_bytes'):
        getattr(L, n).restype = ctypes.c_long
        getattr(L, n).argtypes = [ctypes.c_int] * 3
    return L


def test_first_layer_plans_are_pure_functions_of_the_shape():
    """Host side of K-SPARSE (no launch): the matrix-pipe forward takes 32 / 64 units only and says so with 0 bytes; its
    workspace holds the split weights, the per-tile bias shares and one partial per gene chunk of whole 128-gene super steps;
    the weight gradient cuts the batch into splits of at most 2 048 rows (their storage rows sit in LDS)."""
    L = _lib()
    r16 = lambda x: (x + 15) // 16 * 16
    for H1 in (16, 128, 256, 0):
        assert L.dcahip_enc0_fwd_lut_workspace_bytes(4096, 20000, H1) == 0
    assert L.dcahip_enc0_fwd_lut_workspace_bytes(0, 20000, 64) == 0 and L.dcahip_enc0_fwd_lut_workspace_bytes(64, 0, 64) == 0
    for B, G, H1 in ((4096, 20000, 64), (281, 20000, 64), (1024, 25000, 32), (70000, 130, 64), (300, 64, 32)):
        n_ms = 2 * ((G + 127) // 128)
        rgs = (B + 255) // 256
        nsk = max(1, min(32, 256 // rgs))
        ms_per = 2 * ((n_ms // 2 + nsk - 1) // nsk)
        chunks = (n_ms + ms_per - 1) // ms_per
        tile = 3 * (H1 // 32) * 4 * 2 * 32 * 8 * 2
        want = r16(n_ms * tile) + r16(n_ms * H1 * 8) + r16(chunks * rgs * 256 * H1 * 4)
        assert L.dcahip_enc0_fwd_lut_workspace_bytes(B, G, H1) == want, (B, G, H1)
    # weight gradient: [splits][Gs][H1] partials + column sums ...: the number of splits follows from the size.  The 64-unit
    # gradient has two kernels with different split counts (the `form` argument of dcahip_enc0_dw_sparse); the workspace covers either
    def splits(B, G, H1):
        groups = (G + 255) // 256
        ns = max(1, min(16, 256 // groups, (B + 63) // 64))
        return max(ns, (B + 2047) // 2048)

    def splits2(B, G):                        # the ring form: 512 genes per workgroup, one workgroup per CU
        groups = (G + 511) // 512
        ns = max(1, min(16, 256 // groups, (B + 15) // 16))
        return max(ns, (B + 1023) // 1024)
    for B, G, H1 in ((4096, 20000, 64), (65536, 20000, 64), (40000, 2000, 32), (512, 25000, 128), (300, 77, 64)):
        ns = max(splits(B, G, H1), splits2(B, G)) if H1 == 64 else splits(B, G, H1)
        Gs, steps = (G + 511) // 512 * 512, (B + 15) // 16
        dz = 3 * (H1 // 32) * 2 * 32 * 8
        want = r16((ns * Gs * H1 + ns * H1) * 4) + r16(steps * H1 * 4) + r16(steps * dz * 2) + r16(B * 4)
        assert L.dcahip_enc0_dw_sparse_workspace_bytes(B, G, H1) == want, (B, G, H1)
        assert (B + splits(B, G, H1) - 1) // splits(B, G, H1) <= 2048
        if H1 == 64:
            assert ((B + splits2(B, G) - 1) // splits2(B, G) + 15) // 16 * 16 <= 1024 or B > 16 * 1024
    assert L.dcahip_enc0_dw_sparse_workspace_bytes(4096, 20000, 16) == 0


def test_first_layer_kernels_stay_inside_their_budget():
    """The byte-store kernels of the first layer run one 8-wave workgroup per CU: no scratch, LDS <= 160 KiB."""
    import shutil
    import subprocess
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    src = os.path.join(ROOT, 'dca_amd', 'csrc', 'dcahip_sparse.hip')
    out = subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(ROOT, 'include'),
                          '--cuda-device-only', '-c', src, '-o', os.devnull, '-Rpass-analysis=kernel-resource-usage'],
                         capture_output=True, text=True, check=True).stderr
    names = re.findall(r'Function Name: (\S+)', out)
    scratch = [int(x) for x in re.findall(r'ScratchSize \[bytes/lane\]: (\d+)', out)]
    lds = [int(x) for x in re.findall(r'LDS Size \[bytes/block\]: (\d+)', out)]
    assert len(names) == len(scratch) == len(lds)
    seen = 0
    for n, s, l in zip(names, scratch, lds):
        if 'enc0_dw_kernel' in n or 'enc0_dw2_kernel' in n or 'enc0_fwd_lut_kernel' in n:
            seen += 1
            assert s == 0 and l <= 163840, (n, s, l)
    assert seen == 6          # weight gradient at 32 / 64 / 128 units + the ring form at 64, forward at 32 / 64


def _blocks_with_matrix_instructions(asm, kernel_substr):
    """(kernel name, basic block label, matrix instructions, scratch instructions) of every basic block of the kernels whose
    mangled name contains kernel_substr and that holds a matrix instruction."""
    out, cur_kernel, cur_block, n_m, n_s = [], None, None, 0, 0

    def flush():
        if cur_kernel is not None and n_m:
            out.append((cur_kernel, cur_block, n_m, n_s))
    for ln in asm.split('\n'):
        t = ln.strip()
        m = re.match(r'^(_Z\w+):', ln)
        if m:
            flush()
            cur_kernel = m.group(1) if kernel_substr in m.group(1) else None
            cur_block, n_m, n_s = 'entry', 0, 0
            continue
        if cur_kernel is None:
            continue
        m = re.match(r'^(\.LBB\d+_\d+):', t)
        if m or t.startswith(('s_cbranch', 's_branch', 's_endpgm')):
            flush()
            cur_block, n_m, n_s = (m.group(1) if m else cur_block + "'"), 0, 0
            if t.startswith('s_endpgm'):
                cur_kernel = None
            continue
        if t.startswith('v_mfma'):
            n_m += 1
        elif t.startswith('scratch_'):
            n_s += 1
    flush()
    return out


def test_matrix_loops_of_the_product_kernels_are_spill_free():
    """A spill reload inside a loop of matrix instructions waits for every request in flight (and a lone wave per SIMD has
    nothing to hide it behind): no basic block that issues MFMAs may touch scratch -- the wide networks' plane GEMMs
    (gemm_p3w_kernel / gemm_h2w_kernel: their 308-328 B/lane of scratch sit in the epilogue; gemm_h2m_kernel, the four-wave
    forward form), the byte-store first-layer kernels and the 8-wave
    K-HEADS kernels (the product instantiations over the byte store: at most the 2 scratch instructions of the tile loop the
    round-4 review counted)."""
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    budget = [('dcahip_gemm.hip', 'gemm_p3w_kernel', 0), ('dcahip_gemm.hip', 'gemm_h2w_kernel', 0),
              ('dcahip_gemm.hip', 'gemm_h2m_kernel', 0), ('dcahip_sparse.hip', 'enc0_', 0),
              ('dcahip_heads.hip', 'heads_fused_h2_kernelILb1ELb0ELb1E', 2)]
    isa = {}
    for src, sub, allowed in budget:
        if src not in isa:
            with tempfile.TemporaryDirectory() as td:
                out = os.path.join(td, 'k.s')
                subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(ROOT, 'include'),
                                '--cuda-device-only', '-S', os.path.join(ROOT, 'dca_amd', 'csrc', src), '-o', out],
                               capture_output=True, text=True, check=True)
                isa[src] = open(out).read()
        blocks = _blocks_with_matrix_instructions(isa[src], sub)
        assert blocks, (src, sub)
        assert sum(b[2] for b in blocks) >= 24, (src, sub)
        bad = [(k[-48:], b, m, s) for k, b, m, s in blocks if s > allowed]
        assert not bad, (src, bad[:5])


def test_ring_weight_gradient_loop_has_no_compiler_waits_on_the_ring():
    """enc0_dw2_kernel fills its LDS ring with global_load_lds and retires the requests with COUNTED s_waitcnt vmcnt
    statements of its own.  The compiler cannot tell which LDS bytes such a load writes: in front of any LDS read it generates
    itself it puts s_waitcnt vmcnt(0), which stalls the ring for a memory round trip per step (measured).  So in every basic
    block of the kernel that issues matrix instructions: all LDS reads and all vmcnt waits stand inside instruction statements
    (between #ASMSTART / #ASMEND), and there is no scalar memory load (it would break the counted lgkmcnt waits)."""
    import shutil
    import subprocess
    import tempfile
    hipcc = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, 'k.s')
        subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-I' + os.path.join(ROOT, 'include'),
                        '--cuda-device-only', '-S', os.path.join(ROOT, 'dca_amd', 'csrc', 'dcahip_sparse.hip'), '-o', out],
                       check=True, capture_output=True)
        asm = open(out).read()
    m = re.search(r'^(_Z\w*enc0_dw2_kernel\w*):', asm, re.M)
    assert m
    body = asm[m.end():asm.index('s_endpgm', m.end())]
    blocks, cur = [], []
    for ln in body.split('\n'):
        t = ln.strip()
        if re.match(r'^\.LBB\d+_\d+:', t) or t.startswith(('s_cbranch', 's_branch')):
            blocks.append(cur); cur = []
            continue
        cur.append(t)
    blocks.append(cur)
    seen = 0
    for b in blocks:
        if sum(1 for t in b if t.startswith('v_mfma')) < 12:
            continue
        seen += 1
        inside = False
        for t in b:
            if t.startswith(';;#ASMSTART'):
                inside = True
            elif t.startswith(';;#ASMEND'):
                inside = False
            elif not inside:
                assert not t.startswith(('ds_read', 'ds_load')), t
                assert not t.startswith('s_waitcnt vmcnt'), t
                assert not t.startswith(('s_load', 's_buffer_load')), t
                assert not t.startswith('scratch_'), t
    assert seen >= 4          # two phases of twelve matrix instructions per step, two steps per trip of the loop


def test_library_counts_as_stale_when_it_was_built_from_other_source_contents():
    """Modification times do not say which sources a library was built from (a checkout makes a source OLDER than a library
    built from its edited state; a copied tree has fresh times everywhere): build.needs_build() also compares the fingerprint
    of the source contents stored beside the library."""
    from dca_amd import build
    side = build.LIB + '.src'
    if not os.path.exists(build.LIB):
        pytest.skip('library not built')
    assert not build.needs_build()
    keep = open(side).read()
    try:
        with open(side, 'w') as f:
            f.write('0' * 64 + '\n')
        assert build.needs_build()
        os.remove(side)
        assert build.needs_build()
    finally:
        with open(side, 'w') as f:
            f.write(keep)
    assert not build.needs_build()
