"""Builds dca_amd/csrc/libdcahip.so for gfx950 with hipcc (cross-compiles without a GPU) and the
host-only dca_amd/csrc/libdcahost.so (include/dcahost.h: result writers) with g++.

    python -m dca_amd.build            # rebuild if any source is newer than the library
    python -m dca_amd.build --force

The library is a plain C-ABI shared object (include/dcahip.h); no torch headers, no JIT cache:
it lives in-tree so it travels with the repository snapshot to the GPU box.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(CSRC, 'libdcahip.so')
SOURCES = ['dcahip_zinb.hip', 'dcahip_gemm.hip', 'dcahip_layers.hip', 'dcahip_heads.hip', 'dcahip_prep.hip', 'dcahip_opt.hip',
           'dcahip_dropout.hip', 'dcahip_sparse.hip', 'dcahip_peer.hip']
HEADERS = ['zinb_math.hpp', 'h2_math.hpp']
ARCH = 'gfx950'
HOST_LIB = os.path.join(CSRC, 'libdcahost.so')
HOST_SOURCES = ['dcahost_tsv.cpp', 'dcahost_read.cpp']


def _hipcc():
    for c in (shutil.which('hipcc'), '/opt/rocm/bin/hipcc'):
        if c and os.path.exists(c):
            return c
    raise RuntimeError('hipcc not found: cannot build libdcahip.so')


def _deps():
    return [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.join(ROOT, 'include', 'dcahip.h')]


def source_fingerprint():
    """sha256 over the contents of everything libdcahip.so is compiled from."""
    import hashlib
    h = hashlib.sha256()
    for d in _deps():
        h.update(os.path.basename(d).encode())
        if os.path.exists(d):
            with open(d, 'rb') as f:
                h.update(f.read())
    return h.hexdigest()


def needs_build():
    """True when the library is absent, older than a source, or was built from OTHER source contents (the fingerprint beside
    it, LIB + '.src': modification times do not survive every copy of the tree, and a checkout can make a source older than a
    library built from its edited state)."""
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    if any(os.path.exists(d) and os.path.getmtime(d) > t for d in _deps()):
        return True
    try:
        with open(LIB + '.src') as f:
            return f.read().strip() != source_fingerprint()
    except OSError:
        return True


def _obj_path(src, tag):
    return os.path.join(CSRC, '_obj', os.path.basename(src) + ('.' + tag if tag else '') + '.o')


def _obj_fingerprint(src, hdrs, defines):
    """sha256 over what one object is compiled from: its source, every header, the -D list."""
    import hashlib
    h = hashlib.sha256()
    for d in [src] + list(hdrs):
        h.update(os.path.basename(d).encode())
        with open(d, 'rb') as f:
            h.update(f.read())
    h.update('\0'.join(defines).encode())
    return h.hexdigest()


def _obj_is_fresh(obj, fp):
    try:
        with open(obj + '.src') as f:
            return os.path.exists(obj) and f.read().strip() == fp
    except OSError:
        return False


def build_hip(force=False, verbose=True, defines=(), out=None):
    """One object per source (compiled side by side), then the link.  An object is rebuilt when the CONTENTS it was compiled
    from differ (the fingerprint beside it, obj + '.src'), not by modification time: a checkout can make an edited source
    older than its stale object, and a stale object relinked under a fresh library fingerprint would stay stale for good.
    ``defines`` / ``out``: experiment builds next to the product library (tools/)."""
    out = out or LIB
    if not force and not defines and out == LIB and not needs_build():
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(os.path.join(CSRC, '_obj'), exist_ok=True)
    tag = '_'.join(d.replace('=', '-') for d in defines)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.join(ROOT, 'include', 'dcahip.h')]
    hdrs = [h for h in hdrs if os.path.exists(h)]
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs, stamps = [], []
    for src in srcs:
        obj = _obj_path(src, tag)
        fp = _obj_fingerprint(src, hdrs, defines)
        if force or not _obj_is_fresh(obj, fp):
            if os.path.exists(obj + '.src'):
                os.remove(obj + '.src')              # no stamp survives a failed compile
            jobs.append([_hipcc(), '--offload-arch=' + ARCH, '-O3', '-std=c++17', '-fPIC', '-c',
                         '-I' + os.path.join(ROOT, 'include')] + ['-D' + d for d in defines] + ['-o', obj, src])
            stamps.append((obj, fp))
    if verbose:
        for j in jobs:
            print(' '.join(j), flush=True)
    with ThreadPoolExecutor(max_workers=max(1, min(len(jobs), os.cpu_count() or 1))) as ex:
        list(ex.map(subprocess.check_call, jobs))
    for obj, fp in stamps:
        with open(obj + '.src', 'w') as f:
            f.write(fp + '\n')
    cmd = [_hipcc(), '--offload-arch=' + ARCH, '-shared', '-fPIC', '-o', out] + [_obj_path(s, tag) for s in srcs]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    if not defines and out == LIB:
        with open(LIB + '.src', 'w') as f:
            f.write(source_fingerprint() + '\n')
    return out


def host_needs_build():
    if not os.path.exists(HOST_LIB):
        return True
    t = os.path.getmtime(HOST_LIB)
    deps = [os.path.join(CSRC, s) for s in HOST_SOURCES] + [os.path.join(ROOT, 'include', 'dcahost.h')]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build_host(force=False, verbose=True):
    if not force and not host_needs_build():
        return HOST_LIB
    cxx = shutil.which('g++') or shutil.which('c++')
    if not cxx:
        raise RuntimeError('g++ not found: cannot build libdcahost.so')
    cmd = [cxx, '-O3', '-std=c++17', '-fPIC', '-shared', '-pthread', '-I' + os.path.join(ROOT, 'include'),
           '-o', HOST_LIB] + [os.path.join(CSRC, s) for s in HOST_SOURCES]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return HOST_LIB


if __name__ == '__main__':
    print(build_hip(force='--force' in sys.argv))
    print(build_host(force='--force' in sys.argv))
