"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatement of the arithmetic the HIP kernels use for an
fp32-accurate matrix product on the bf16 matrix pipe -- every fp32 operand as three bf16 pieces (round-to-nearest-even
residuals, `split_pair` in dca_amd/csrc/dcahip_sparse.hip / dcahip_heads.hip), SIX piece products a1b1 a1b2 a2b1 a1b3 a2b2
a3b1 (`MFMA_X3`), each exact in fp32 (8 x 8 mantissa bits), accumulated in fp32.  Replaces nothing of the reference: it
states what "matrix products: fp32 results as split-bf16 products"  means, so that the bounds the GPU
parity tests hold the kernels to (tests/test_heads_fused_gpu.py::product_tol, tests/test_sparse_gpu.py) can be checked on
the CPU against fp64, together with the reason a three-product build must fail them.
"""
import numpy as np


def bf16_round(x):
    """fp32 -> nearest bf16 (ties to even), returned as fp32 (v_cvt_pk_bf16_f32)."""
    u = np.asarray(x, np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split3(x):
    """x = p0 + p1 + p2 up to 2^-24 |x|: three bf16 pieces of an fp32 array."""
    x = np.asarray(x, np.float32)
    p0 = bf16_round(x)
    r = (x - p0).astype(np.float32)
    p1 = bf16_round(r)
    p2 = bf16_round((r - p1).astype(np.float32))
    return p0, p1, p2


def matmul_x3(a, b, products=6):
    """a [M, K] @ b [K, N] as the kernels compute it: piece products accumulated in fp32, small terms first."""
    A, B = split3(a), split3(b)
    terms = [(2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)]          # (piece of a, piece of b): MFMA_X3's order
    if products == 3:
        terms = [(1, 0), (0, 1), (0, 0)]
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for i, j in terms:
        # a bf16 x bf16 product is exact in fp32; the sum over K runs in fp32 (the MFMA's accumulator)
        acc = (acc + (A[i].astype(np.float32) @ B[j].astype(np.float32)).astype(np.float32)).astype(np.float32)
    return acc
