"""The C-ABI library builds, loads (no GPU needed) and exports every symbol include/dcahip.h declares."""
import ctypes
import os
import re

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


def test_product_path_fails_loudly_without_gpu():
    import pytest
    import torch
    from dca_amd.ops import HipOps
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        HipOps()
