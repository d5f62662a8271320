"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatement of the arithmetic the HIP kernels use for an
fp32-accurate matrix product on the 16-bit matrix pipe.  Two forms:

  bf16 x 3 (the first-layer kernels and the plane GEMMs; K-HEADS until round 5): every fp32 operand as three bf16 pieces
  (round-to-nearest-even residuals, `split_pair` in dca_amd/csrc/dcahip_sparse.hip / dcahip_gemm.hip), SIX piece products
  a1b1 a1b2 a2b1 a1b3 a2b2 a3b1 (`MFMA_X3`), each exact in fp32 (8 x 8 mantissa bits), accumulated in fp32;

  fp16 x 2 (K-HEADS from round 6, dca_amd/csrc/dcahip_heads.hip): every operand BLOCK scaled by the power of two that brings
  its largest magnitude into [2^13, 2^14), then two fp16 pieces x 2^e = h1 + h2 (round to nearest; 2^-22 relative, or 2^-25
  absolute where h2 is an fp16 denormal, which the matrix pipe preserves), THREE piece products a1b1 + a1b2 + a2b1 (`MFMA_H3`;
  the dropped a2b2 is 2^-22 of a product), accumulated in fp32, the scales taken out at the end (exact).

Replaces nothing of the reference: it states what "matrix products: fp32 results as split 16-bit products" (DESIGN.md
section 7) means, so that the bounds the GPU parity tests hold the kernels to (tests/test_heads_fused_gpu.py::product_tol,
tests/test_sparse_gpu.py) can be checked on the CPU against fp64, together with the reason narrower builds must fail them.
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


# ---------------------------------------------------------------------------------------------------------- fp16 x 2
def block_exp(x, top=13):
    """The exponent e with max|x| 2^e in [2^top, 2^(top + 1)); 0 for an all-zero block (block_exp in dcahip_heads.hip)."""
    m = float(np.abs(np.asarray(x, np.float32)).max()) if np.size(x) else 0.0
    if not (m > 0.0) or not np.isfinite(m):
        return 0
    return int(np.clip(top + 1 - np.frexp(m)[1], -60, 60))


def split2(x, e=0):
    """x 2^e = h1 + h2: two fp16 pieces (round to nearest even, denormals kept), returned as fp32 arrays."""
    xs = np.ldexp(np.asarray(x, np.float32), e).astype(np.float32)
    h1 = xs.astype(np.float16)
    r = (xs - h1.astype(np.float32)).astype(np.float32)           # exact: |xs - h1| <= 2^-11 |xs|
    h2 = r.astype(np.float16)
    return h1.astype(np.float32), h2.astype(np.float32)


def matmul_h2(a, b, products=3, ea=None, eb=None):
    """a [M, K] @ b [K, N] as K-HEADS computes it: block scales, two fp16 pieces per operand, the piece products accumulated
    in fp32 (small terms first), the scales taken out.  products = 2 drops a2 b1 (what a narrower build would do)."""
    ea = block_exp(a) if ea is None else ea
    eb = block_exp(b) if eb is None else eb
    A, B = split2(a, ea), split2(b, eb)
    terms = [(1, 0), (0, 1), (0, 0)][3 - products:]
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for i, j in terms:
        acc = (acc + (A[i] @ B[j]).astype(np.float32)).astype(np.float32)
    return np.ldexp(acc.astype(np.float64), -(ea + eb))
