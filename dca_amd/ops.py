"""HipOps: tensor-level view of the C ABI (one method per entry point of include/dcahip.h).

The training engine (engine.py) is written against this small interface.  The product always
instantiates HipOps -- construction fails loudly without the HIP library or without a GPU.
(Tests exercise the engine's host logic on CPU by injecting an oracle-backed object with the
same methods; that object lives under oracle/ and is never imported from this package.)
"""
import ctypes

import torch

from . import hip


class HipOps:
    name = 'hip'
    device_type = 'cuda'

    def __init__(self):
        hip.require_gpu()
        self.L = hip.lib()
        self.max_partials = self.L.dcahip_zinb_max_partials()

    # ------------------------------------------------------------------ loss
    def zinb_nll(self, a_mean, a_disp, a_pi, lda, theta_w, Y, ldy, sf, perm, cursor, B, G, ridge,
                 inv_n, flags, d_mean, d_disp, d_pi, ldd, partials):
        n = ctypes.c_int(0)
        p = hip.ptr
        hip.check(self.L.dcahip_zinb_nll(p(a_mean), p(a_disp), p(a_pi), lda, p(theta_w), p(Y), ldy,
                                         p(sf), p(perm), p(cursor), B, G, ridge, inv_n, flags,
                                         p(d_mean), p(d_disp), p(d_pi), ldd, p(partials),
                                         ctypes.byref(n), hip.stream()), 'zinb_nll')
        return n.value

    def zinb_nll_planes(self, a_mean, a_disp, a_pi, lda, theta_w, Y, ldy, sf, perm, cursor, B, G, ridge, inv_n, flags,
                        planes, col_mean, col_disp, col_pi, d_theta, ldd_theta, partials):
        """zinb_nll with the gradient planes written as pre-split bf16 pieces into planes [3, rows, ld] at the given
        columns (the operand of gemm_p3); the per-gene dispersion gradient (const. dispersion) stays fp32 in d_theta."""
        n = ctypes.c_int(0)
        p = hip.ptr
        hip.check(self.L.dcahip_zinb_nll_planes(p(a_mean), p(a_disp), p(a_pi), lda, p(theta_w), p(Y), ldy, p(sf), p(perm),
                                                p(cursor), B, G, ridge, inv_n, flags, p(planes), planes.stride(1),
                                                planes.stride(0), col_mean, col_disp, col_pi, p(d_theta), ldd_theta,
                                                p(partials), ctypes.byref(n), hip.stream()), 'zinb_nll_planes')
        return n.value

    def zinb_nll_planes_h2(self, a_mean, a_disp, a_pi, lda, theta_w, Y, ldy, sf, perm, cursor, B, G, ridge, inv_n, flags, d_exp,
                           planes, col_mean, col_disp, col_pi, d_theta, ldd_theta, partials):
        """zinb_nll_planes with the head planes as TWO fp16 pieces of the unscaled gradient g 2^d_exp (the operand of gemm_h2)."""
        n = ctypes.c_int(0)
        p = hip.ptr
        hip.check(self.L.dcahip_zinb_nll_planes_h2(p(a_mean), p(a_disp), p(a_pi), lda, p(theta_w), p(Y), ldy, p(sf), p(perm),
                                                   p(cursor), B, G, ridge, inv_n, flags, int(d_exp), p(planes), planes.stride(1),
                                                   planes.stride(0), col_mean, col_disp, col_pi, p(d_theta), ldd_theta,
                                                   p(partials), ctypes.byref(n), hip.stream()), 'zinb_nll_planes_h2')
        return n.value

    def loss_finalize(self, partials, n, scale, loss_out):
        hip.check(self.L.dcahip_loss_finalize(hip.ptr(partials), n, scale, hip.ptr(loss_out),
                                              hip.stream()), 'loss_finalize')

    def step_end(self, loss, weight, hist, rows_per_slot, acc, cursor, advance):
        p = hip.ptr
        hip.check(self.L.dcahip_step_end(p(loss), weight, p(hist), rows_per_slot, p(acc), p(cursor),
                                         advance, hip.stream()), 'step_end')

    def heads_infer(self, a_mean, a_disp, a_pi, lda, sf, B, G, mean_sf, theta, pi, ldo, flags=0):
        p = hip.ptr
        hip.check(self.L.dcahip_zinb_heads_infer(p(a_mean), p(a_disp), p(a_pi), lda, p(sf), B, G,
                                                 p(mean_sf), p(theta), p(pi), ldo, flags, hip.stream()),
                  'heads_infer')

    # ------------------------------------------------------------------ fused heads
    def heads_fused_workspace_bytes(self, B, hL, G, plane, flags):
        return self.L.dcahip_heads_fused_workspace_bytes(B, hL, G, plane, flags)

    def x3_product_32x32(self, A, B, C, K):
        hip.check(self.L.dcahip_x3_product_32x32(hip.ptr(A), hip.ptr(B), hip.ptr(C), K, hip.stream()), "x3_product_32x32")

    def has(self, entry):
        """Whether the loaded library exports `entry` (the experiment entry points exist only in -D builds)."""
        return hasattr(self.L, entry)

    def heads_tile_order_len(self, G):
        return int(self.L.dcahip_heads_tile_order_len(G))

    def heads_fused(self, H, ldh, Wh, ldw, bh, plane, theta_w, Y, ldy, sf, perm, cursor, B, hL, G,
                    ridge, inv_n, flags, gW, ldg, g_theta, dH, lddh, partials, ws, tile_order=None, loss_out=None,
                    compact=None, d_exp=0):
        """loss_out (a device float): the call also finishes the batch loss there (no loss_finalize launch needed).
        compact (dca_amd.compact.CompactCounts): the counts are read from the byte store instead of Y.
        d_exp (<= 0): scale exponent of the gradient planes for datasets with very large counts (include/dcahip.h)."""
        n = ctypes.c_int(0)
        p = hip.ptr
        c = compact
        hip.check(self.L.dcahip_heads_fused_compact(p(H), ldh, p(Wh), ldw, p(bh), plane, p(theta_w), p(Y), ldy,
                                                    p(c.Yc) if c is not None else None, c.ldc if c is not None else 0,
                                                    p(c.ovf_ptr) if c is not None else None,
                                                    p(c.ovf_col) if c is not None else None,
                                                    p(c.ovf_val) if c is not None else None,
                                                    p(sf), p(perm), p(cursor), B, hL, G, ridge, inv_n, flags,
                                                    p(gW), ldg, p(g_theta), p(dH), lddh, p(partials),
                                                    ctypes.byref(n), p(ws), ws.numel() * ws.element_size(),
                                                    p(tile_order), p(loss_out), int(d_exp), hip.stream()), 'heads_fused')
        return n.value

    # ------------------------------------------------------------------ compact counts, sparse first layer
    def counts_compact_ld(self, G):
        return int(self.L.dcahip_counts_compact_ld(G))

    def counts_compact(self, Y, ldy, n, G, Yc, ldc, status):
        hip.check(self.L.dcahip_counts_compact(hip.ptr(Y), ldy, n, G, hip.ptr(Yc), ldc, hip.ptr(status), hip.stream()),
                  'counts_compact')

    def enc0_lut_entries(self):
        """Entries per cell of the table enc0_lut writes."""
        return int(self.L.dcahip_enc0_lut_entries())

    def enc0_lut(self, fac, do_log, n, lutp):
        hip.check(self.L.dcahip_enc0_lut(hip.ptr(fac), int(do_log), n, hip.ptr(lutp), hip.stream()), 'enc0_lut')

    def enc0_sparse_supported(self, H1):
        return bool(self.L.dcahip_enc0_sparse_supported(int(H1)))

    def enc0_dw_sparse_workspace_bytes(self, B, G, H1):
        return int(self.L.dcahip_enc0_dw_sparse_workspace_bytes(B, G, H1))

    def enc0_fwd_sparse_workspace_bytes(self, H1):
        """EXPERIMENT build (-DDCA_EXP_ENC0_SPARSE_FWD) only."""
        return int(self.L.dcahip_enc0_fwd_sparse_workspace_bytes(H1))

    def enc0_dw_sparse(self, c, perm, cursor, row_base, B, G, H1, dZ, ldz, gW, ldg, ws, form=0):
        """gW [G + 1, ldg] = [X^T dZ ; colsum dZ] with X described by the compact counts c (its normalisation fields).
        form: which of the two 64-unit kernels (0: by the shape, 1: the first, 2: the ring kernel; include/dcahip.h)."""
        p = hip.ptr
        c.ensure_lut(self)                   # the per-cell table of the common counts: made on the first call for this store
        hip.check(self.L.dcahip_enc0_dw_sparse(p(c.Yc), c.ldc, p(c.ovf_ptr), p(c.ovf_col), p(c.ovf_val), p(c.fac),
                                               int(c.do_log), p(c.lutp), p(c.mean), p(c.std), p(perm), p(cursor), int(row_base),
                                               B, G, H1, p(dZ), ldz, p(gW), ldg, p(ws), ws.numel() * ws.element_size(),
                                               int(form), hip.stream()), 'enc0_dw_sparse')

    @property
    def enc0_dw_small_max_rows(self):
        return int(self.L.dcahip_enc0_dw_small_max_rows())

    def enc0_dw_small(self, c, perm, cursor, row_base, B, G, H1, dZ, ldz, gW, ldg):
        """EXPERIMENT build (-DDCA_EXP_DW_SMALL) only: gW [G + 1, ldg] = X^T dZ (+ column sums of dZ in row G) for a batch of
        at most enc0_dw_small_max_rows rows, X read from the byte store (non-zero counts only)."""
        p = hip.ptr
        hip.check(self.L.dcahip_enc0_dw_small(p(c.Yc), c.ldc, p(c.ovf_ptr), p(c.ovf_col), p(c.ovf_val), p(c.fac),
                                              int(c.do_log), p(c.mean), p(c.std), p(perm), p(cursor), int(row_base),
                                              B, G, H1, p(dZ), ldz, p(gW), ldg, hip.stream()), 'enc0_dw_small')

    def enc0_fwd_sparse(self, c, perm, cursor, row_base, B, G, H1, W, ldw, bias, Z, ldz, ws):
        """EXPERIMENT build (-DDCA_EXP_ENC0_SPARSE_FWD) only: Z [B, ldz] = X W + bias over the non-zero counts; ws:
        zero-initialised once, private to this call site."""
        p = hip.ptr
        hip.check(self.L.dcahip_enc0_fwd_sparse(p(c.Yc), c.ldc, p(c.ovf_ptr), p(c.ovf_col), p(c.ovf_val), p(c.fac),
                                                int(c.do_log), p(c.mean), p(c.std), p(perm), p(cursor), int(row_base),
                                                B, G, H1, p(W), ldw, p(bias), p(Z), ldz, p(ws),
                                                ws.numel() * ws.element_size(), hip.stream()), 'enc0_fwd_sparse')

    def enc0_fwd_lut_workspace_bytes(self, B, G, H1):
        """0: this first-layer width is not taken by the matrix-pipe forward."""
        return int(self.L.dcahip_enc0_fwd_lut_workspace_bytes(B, G, H1))

    def enc0_fwd_lut(self, c, perm, cursor, row_base, B, G, H1, W, ldw, bias, Z, ldz, ws):
        """Z [B, ldz] = X W + bias on the matrix pipe, X looked up from the byte store; ws zero-initialised once."""
        p = hip.ptr
        c.ensure_lut(self)
        hip.check(self.L.dcahip_enc0_fwd_lut(p(c.Yc), c.ldc, p(c.ovf_ptr), p(c.ovf_col), p(c.ovf_val), p(c.fac),
                                             int(c.do_log), p(c.lutp), p(c.mean), p(c.std), p(perm), p(cursor), int(row_base),
                                             B, G, H1, p(W), ldw, p(bias), p(Z), ldz, p(ws),
                                             ws.numel() * ws.element_size(), hip.stream()), 'enc0_fwd_lut')

    # ------------------------------------------------------------------ gemm
    def sgemm_workspace_bytes(self, ta, tb, M, N, K, colsum_row=False, split_k=0):
        return self.L.dcahip_sgemm_workspace_bytes(int(ta), int(tb), M, N, K, int(colsum_row), split_k)

    def sgemm(self, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias=None, perm=None, cursor=None,
              colsum_row=False, split_k=0, ws=None):
        p = hip.ptr
        wsb = ws.numel() * ws.element_size() if ws is not None else 0
        hip.check(self.L.dcahip_sgemm(int(ta), int(tb), M, N, K, p(A), lda, p(B), ldb, p(C), ldc,
                                      p(bias), p(perm), p(cursor), int(colsum_row), split_k, p(ws),
                                      wsb, hip.stream()), 'sgemm')

    # ------------------------------------------------------------------ products from pre-split operands
    def planes_alloc(self, rows, cols, device):
        """Three bf16 planes [3, rows, r8(cols)] (zeros)."""
        return torch.zeros(3, rows, (cols + 7) // 8 * 8, dtype=torch.bfloat16, device=device)

    def split_planes(self, src, ld, R, C, planes, perm=None, cursor=None):
        """planes [3, >= R, ldp] <- the three bf16 pieces of src [R, C] (rows gathered through perm / cursor)."""
        hip.check(self.L.dcahip_split_planes(hip.ptr(src), ld, hip.ptr(perm), hip.ptr(cursor), R, C, hip.ptr(planes),
                                             planes.shape[2], planes.stride(0), hip.stream()), 'split_planes')

    def gemm_p3_workspace_bytes(self, M, N, K, colsum_row=False, split_k=0):
        return self.L.dcahip_gemm_p3_workspace_bytes(M, N, K, int(colsum_row), split_k)

    def gemm_p3(self, ta, tb, M, N, K, A, B, C, ldc, bias=None, perm=None, cursor=None, colsum_row=False, split_k=0,
                ws=None):
        """C = op(A) op(B) from planes A, B ([3, rows, ld] bf16 tensors or views of them)."""
        p = hip.ptr
        wsb = ws.numel() * ws.element_size() if ws is not None else 0
        hip.check(self.L.dcahip_gemm_p3(int(ta), int(tb), M, N, K, p(A), A.stride(1), A.stride(0), p(B), B.stride(1),
                                        B.stride(0), p(C), ldc, p(bias), p(perm), p(cursor), int(colsum_row), split_k,
                                        p(ws), wsb, hip.stream()), 'gemm_p3')

    # ---- fp16 x 2 planes (include/dcahip.h: dcahip_gemm_h2).  exp: an int32 device tensor [2] = (block exponent, scratch)
    def absmax_exp(self, src, ld, R, C, exp):
        """exp[0] <- the exponent that brings max |src [R, C]| into [2^13, 2^14) (stays on the device: capturable)."""
        hip.check(self.L.dcahip_absmax_exp(hip.ptr(src), ld, R, C, hip.ptr(exp), hip.ptr(exp[1:]), hip.stream()), 'absmax_exp')

    def split_planes_h2(self, src, ld, R, C, planes, exp=None, perm=None, cursor=None):
        """planes [>= 2, >= R, ldp] <- the two fp16 pieces of src [R, C] 2^exp[0] (rows gathered through perm / cursor)."""
        hip.check(self.L.dcahip_split_planes_h2(hip.ptr(src), ld, hip.ptr(perm), hip.ptr(cursor), R, C, hip.ptr(planes),
                                                planes.shape[2], planes.stride(0), hip.ptr(exp), hip.stream()), 'split_planes_h2')

    def gemm_h2_supported(self, M, N, K):
        return bool(self.L.dcahip_gemm_h2_supported(M, N, K))

    def gemm_h2_workspace_bytes(self, M, N, K, colsum_row=False, split_k=0):
        return self.L.dcahip_gemm_h2_workspace_bytes(M, N, K, int(colsum_row), split_k)

    def gemm_h2(self, ta, tb, M, N, K, A, B, C, ldc, exp_a=None, exp_b=None, exp_a_add=0, exp_b_add=0, alpha=1.0, bias=None,
                colsum_row=False, split_k=0, ws=None):
        """C = alpha 2^-(ea + eb) op(A) op(B) from fp16 x 2 planes A, B ([>= 2, rows, ld] tensors or views of them)."""
        p = hip.ptr
        wsb = ws.numel() * ws.element_size() if ws is not None else 0
        hip.check(self.L.dcahip_gemm_h2(int(ta), int(tb), M, N, K, p(A), A.stride(1), A.stride(0), p(exp_a), int(exp_a_add),
                                        p(B), B.stride(1), B.stride(0), p(exp_b), int(exp_b_add), float(alpha), p(C), ldc, p(bias),
                                        int(colsum_row), split_k, p(ws), wsb, hip.stream()), 'gemm_h2')

    def transpose(self, src, ld_src, R, C, dst, ld_dst, perm=None, cursor=None):
        """dst [C, R] = src[rows]^T; rows = perm[cursor : cursor + R] when perm is given."""
        hip.check(self.L.dcahip_transpose_rows(hip.ptr(src), ld_src, hip.ptr(perm), hip.ptr(cursor), R, C,
                                               hip.ptr(dst), ld_dst, hip.stream()), 'transpose')

    # ------------------------------------------------------------------ batch norm
    def col_moments_chunks(self, B):
        return self.L.dcahip_col_moments_chunks(B)

    def col_moments(self, Z, ldz, B, H, part):
        hip.check(self.L.dcahip_col_moments(hip.ptr(Z), ldz, B, H, hip.ptr(part), hip.stream()),
                  'col_moments')

    def moments_combine(self, entries, counts, E, H, out):
        hip.check(self.L.dcahip_moments_combine(hip.ptr(entries), hip.ptr(counts), E, H, hip.ptr(out),
                                                hip.stream()), 'moments_combine')

    def bn_relu_apply(self, Z, ldz, B, H, entries, counts, E, beta, mm, mv, momentum, eps, relu,
                      Hout, ldh, xhat, ldx, inv_std):
        p = hip.ptr
        hip.check(self.L.dcahip_bn_relu_apply(p(Z), ldz, B, H, p(entries), p(counts), E, p(beta),
                                              p(mm), p(mv), momentum, eps, int(relu), p(Hout), ldh,
                                              p(xhat), ldx, p(inv_std), hip.stream()), 'bn_relu_apply')

    def bn_bwd_sums(self, dH, ldd, Hact, ldh, xhat, ldx, B, H, part, act=1):
        p = hip.ptr
        hip.check(self.L.dcahip_bn_bwd_sums(p(dH), ldd, p(Hact), ldh, p(xhat), ldx, B, H, p(part),
                                            act, hip.stream()), 'bn_bwd_sums')

    def bn_bwd_apply(self, dH, ldd, Hact, ldh, xhat, ldx, inv_std, sums, E, n_total, B, H, dZ, ldz,
                     dbeta, act=1):
        p = hip.ptr
        hip.check(self.L.dcahip_bn_bwd_apply(p(dH), ldd, p(Hact), ldh, p(xhat), ldx, p(inv_std),
                                             p(sums), E, float(n_total), B, H, p(dZ), ldz, p(dbeta),
                                             act, hip.stream()), 'bn_bwd_apply')

    @property
    def bn_fused_max_rows(self):
        return int(self.L.dcahip_bn_fused_max_rows())

    def bn_relu_train_small(self, Z, ldz, B, H, beta, mm, mv, momentum, eps, act, Hout, ldh, xhat, ldx, inv_std):
        p = hip.ptr
        hip.check(self.L.dcahip_bn_relu_train_small(p(Z), ldz, B, H, p(beta), p(mm), p(mv), momentum, eps, int(act),
                                                    p(Hout), ldh, p(xhat), ldx, p(inv_std), hip.stream()),
                  'bn_relu_train_small')

    def bn_bwd_small(self, dH, ldd, Hact, ldh, xhat, ldx, inv_std, n_total, B, H, dZ, ldz, dbeta, act=1):
        p = hip.ptr
        hip.check(self.L.dcahip_bn_bwd_small(p(dH), ldd, p(Hact), ldh, p(xhat), ldx, p(inv_std), float(n_total), B, H,
                                             p(dZ), ldz, p(dbeta), act, hip.stream()), 'bn_bwd_small')

    @property
    def dense_small_max_k(self):
        return int(self.L.dcahip_dense_small_max_k())

    def dense_bn_small(self, Hp, ldp, W, ldw, bias, B, K, H, batchnorm, beta, mm, mv, momentum, eps, act, Z, ldz,
                       xhat, ldx, Hout, ldh, inv_std):
        p = hip.ptr
        hip.check(self.L.dcahip_dense_bn_small(p(Hp), ldp, p(W), ldw, p(bias), B, K, H, int(batchnorm), p(beta), p(mm), p(mv),
                                               momentum, eps, int(act), p(Z), ldz, p(xhat), ldx, p(Hout), ldh, p(inv_std),
                                               hip.stream()), 'dense_bn_small')

    def hidden_small_chain(self, layers, Hin, ldin, B, batchnorm, momentum, eps, act):
        """layers: dicts with the fields of dcahip_small_layer (tensors or None); one launch for the whole list."""
        arr = (hip.SmallLayer * len(layers))()
        for q, d in zip(arr, layers):
            for k in ('W', 'bias', 'beta', 'moving_mean', 'moving_var', 'Z', 'xhat', 'Hout', 'inv_std'):
                setattr(q, k, hip.ptr(d.get(k)))
            for k in ('ldw', 'K', 'H', 'ldz', 'ldx', 'ldh'):
                setattr(q, k, int(d.get(k, 0)))
        hip.check(self.L.dcahip_hidden_small_chain(arr, len(layers), hip.ptr(Hin), ldin, B, int(batchnorm), momentum, eps,
                                                   int(act), hip.stream()), 'hidden_small_chain')

    # ------------------------------------------------------------------ K-STACK (throughput batches, one launch per direction)
    @property
    def hidden_stack_max_rows(self):
        return int(self.L.dcahip_hidden_stack_max_rows())

    def hidden_stack_workspace_bytes(self, n_layers, B):
        return int(self.L.dcahip_hidden_stack_workspace_bytes(n_layers, B))

    def hidden_stack_fwd(self, layers, B, momentum, eps, act, ws, rows_per_wg=64, steps=None):
        """layers: dicts with the fields of dcahip_small_layer; entry 0 without a kernel (its Z comes from the first GEMM).
        steps = (first, last) of the pass's n + 1 steps in ONE launch (default: all of them, cooperative)."""
        arr = (hip.SmallLayer * len(layers))()
        for q, d in zip(arr, layers):
            for k in ('W', 'bias', 'beta', 'moving_mean', 'moving_var', 'Z', 'xhat', 'Hout', 'inv_std'):
                setattr(q, k, hip.ptr(d.get(k)))
            for k in ('ldw', 'K', 'H', 'ldz', 'ldx', 'ldh'):
                setattr(q, k, int(d.get(k, 0)))
        first, last = steps if steps is not None else (0, len(layers))
        hip.check(self.L.dcahip_hidden_stack_fwd(arr, len(layers), B, momentum, eps, int(act), int(rows_per_wg), first, last,
                                                 hip.ptr(ws), ws.numel() * ws.element_size(), hip.stream()), 'hidden_stack_fwd')

    def hidden_stack_bwd(self, layers, B, n_total, act, dZ0, ldz0, ws, rows_per_wg=64, steps=None):
        """layers: dicts with the fields of dcahip_stack_bwd_layer; steps = (first, last) of the pass's n + 2 steps.
        A whole pass over a batch of at most 64 rows is one workgroup's work and needs no workspace (ws=None)."""
        arr = (hip.StackBwdLayer * len(layers))()
        for q, d in zip(arr, layers):
            for k in ('W', 'Hact', 'xhat', 'inv_std', 'Hprev', 'gW', 'dbeta', 'dH'):
                setattr(q, k, hip.ptr(d.get(k)))
            for k in ('ldw', 'K', 'H', 'ldh', 'ldx', 'ldp', 'ldg', 'lddh'):
                setattr(q, k, int(d.get(k, 0)))
        first, last = steps if steps is not None else (0, len(layers) + 1)
        hip.check(self.L.dcahip_hidden_stack_bwd(arr, len(layers), B, float(n_total), int(act), hip.ptr(dZ0), ldz0,
                                                 int(rows_per_wg), first, last, hip.ptr(ws),
                                                 0 if ws is None else ws.numel() * ws.element_size(),
                                                 hip.stream()), 'hidden_stack_bwd')

    # ---- K-STACK between the exchanges of a data-parallel step (SyncBN): one step per call
    def _small_layers(self, layers):
        arr = (hip.SmallLayer * len(layers))()
        for q, d in zip(arr, layers):
            for k in ('W', 'bias', 'beta', 'moving_mean', 'moving_var', 'Z', 'xhat', 'Hout', 'inv_std'):
                setattr(q, k, hip.ptr(d.get(k)))
            for k in ('ldw', 'K', 'H', 'ldz', 'ldx', 'ldh'):
                setattr(q, k, int(d.get(k, 0)))
        return arr

    def _bwd_layers(self, layers):
        arr = (hip.StackBwdLayer * len(layers))()
        for q, d in zip(arr, layers):
            for k in ('W', 'Hact', 'xhat', 'inv_std', 'Hprev', 'gW', 'dbeta', 'dH'):
                setattr(q, k, hip.ptr(d.get(k)))
            for k in ('ldw', 'K', 'H', 'ldh', 'ldx', 'ldp', 'ldg', 'lddh'):
                setattr(q, k, int(d.get(k, 0)))
        return arr

    def hidden_stack_fwd_sync(self, layers, B, momentum, eps, act, step, ext_entries, ext_counts, ext_E, stat_out, ws):
        """Step `step` of the forward pass with the input layer's statistics from every rank (ext_entries [E, 2, H],
        ext_counts [E]); stat_out [2, H'] <- this rank's (mean, M2) of the layer made."""
        arr = self._small_layers(layers)
        hip.check(self.L.dcahip_hidden_stack_fwd_sync(arr, len(layers), B, momentum, eps, int(act), int(step),
                                                      hip.ptr(ext_entries), hip.ptr(ext_counts), int(ext_E), hip.ptr(stat_out),
                                                      hip.ptr(ws), ws.numel() * ws.element_size(), hip.stream()),
                  'hidden_stack_fwd_sync')

    def hidden_stack_bwd_sync(self, layers, B, n_total, act, dZ0, ldz0, step, ext_sums, sums_out, ws):
        arr = self._bwd_layers(layers)
        hip.check(self.L.dcahip_hidden_stack_bwd_sync(arr, len(layers), B, float(n_total), int(act), hip.ptr(dZ0), ldz0,
                                                      int(step), hip.ptr(ext_sums), hip.ptr(sums_out), hip.ptr(ws),
                                                      ws.numel() * ws.element_size(), hip.stream()), 'hidden_stack_bwd_sync')

    def dense_bn_bwd_small(self, dH, ldd, Hact, ldh, xhat, ldx, inv_std, Hp, ldp, W, ldw, B, K, H, batchnorm, n_total, act,
                           gW, ldg, dbeta, dHp, lddp):
        p = hip.ptr
        hip.check(self.L.dcahip_dense_bn_bwd_small(p(dH), ldd, p(Hact), ldh, p(xhat), ldx, p(inv_std), p(Hp), ldp, p(W), ldw,
                                                   B, K, H, int(batchnorm), float(n_total), int(act), p(gW), ldg, p(dbeta),
                                                   p(dHp), lddp, hip.stream()), 'dense_bn_bwd_small')

    def relu_bwd(self, dH, ldd, Hact, ldh, B, H, dZ, ldz, act=1):
        p = hip.ptr
        hip.check(self.L.dcahip_relu_bwd(p(dH), ldd, p(Hact), ldh, B, H, p(dZ), ldz, act, hip.stream()),
                  'relu_bwd')

    def relu_fwd(self, Z, ldz, B, H, Hout, ldh, act=1):
        hip.check(self.L.dcahip_relu_fwd(hip.ptr(Z), ldz, B, H, hip.ptr(Hout), ldh, act, hip.stream()),
                  'relu_fwd')

    def colsum_chain(self, x, ldx, B, N, theta_w, out):
        p = hip.ptr
        hip.check(self.L.dcahip_colsum_chain(p(x), ldx, B, N, p(theta_w), p(out), hip.stream()),
                  'colsum_chain')

    # ------------------------------------------------------------------ other optimizers, regularisers
    def optimizer_step(self, kind, w, g, slot1, slot2, n, lr, it, clip):
        p = hip.ptr
        hip.check(self.L.dcahip_optimizer_step(hip.OPT_KINDS[kind], p(w), p(g), p(slot1), p(slot2), n, p(lr),
                                               p(it), clip, hip.stream()), 'optimizer_step')

    def prelu_workspace_doubles(self, h):
        return int(self.L.dcahip_prelu_workspace_doubles(h))

    def prelu_fwd(self, x, ldx, alpha, B, h, out, ldo):
        p = hip.ptr
        hip.check(self.L.dcahip_prelu_fwd(p(x), ldx, p(alpha), B, h, p(out), ldo, hip.stream()), 'prelu_fwd')

    def prelu_bwd(self, d, ldd, x, ldx, alpha, B, h, galpha, ws):
        p = hip.ptr
        hip.check(self.L.dcahip_prelu_bwd(p(d), ldd, p(x), ldx, p(alpha), B, h, p(galpha), p(ws), hip.stream()),
                  'prelu_bwd')

    def elempi_workspace_doubles(self, G):
        return int(self.L.dcahip_elempi_workspace_doubles(G))

    def elempi_fwd(self, a_mean, lda, k, c, B, G, a_pi, ldp):
        p = hip.ptr
        hip.check(self.L.dcahip_elempi_fwd(p(a_mean), lda, p(k), p(c), B, G, p(a_pi), ldp, hip.stream()), 'elempi_fwd')

    def elempi_bwd(self, m, lda, d_mean, d_pi, ldd, k, B, G, gk, gc, ws):
        p = hip.ptr
        hip.check(self.L.dcahip_elempi_bwd(p(m), lda, p(d_mean), p(d_pi), ldd, p(k), B, G, p(gk), p(gc), p(ws),
                                           hip.stream()), 'elempi_bwd')

    def bcast_cols(self, s, lds, B, G, out, ldo):
        hip.check(self.L.dcahip_bcast_cols(hip.ptr(s), lds, B, G, hip.ptr(out), ldo, hip.stream()), 'bcast_cols')

    def row_sums_strided(self, x, ldx, B, G, out, ldo):
        hip.check(self.L.dcahip_row_sums_strided(hip.ptr(x), ldx, B, G, hip.ptr(out), ldo, hip.stream()),
                  'row_sums_strided')

    def nadam_step(self, w, g, m, v, n, lr, it, m_schedule, clip):
        p = hip.ptr
        hip.check(self.L.dcahip_nadam_step(p(w), p(g), p(m), p(v), n, p(lr), p(it), p(m_schedule), clip, hip.stream()),
                  'nadam_step')

    def dropout_apply(self, x, ldx, perm, cursor, B, h, rate, seed, step, layer, row0, out, ldo):
        p = hip.ptr
        hip.check(self.L.dcahip_dropout_apply(p(x), ldx, p(perm), p(cursor), B, h, float(rate), int(seed),
                                              p(step), int(layer), int(row0), p(out), ldo, hip.stream()),
                  'dropout_apply')

    def counter_add(self, counter, v):
        hip.check(self.L.dcahip_counter_add(hip.ptr(counter), v, hip.stream()), 'counter_add')

    def reg_desc(self, segs):
        """segs: list of (start, end, l1, l2) over the flat parameter buffer."""
        d = hip.RegDesc()
        d.nseg = len(segs)
        for k, (a, b, l1, l2) in enumerate(segs):
            d.start[k], d.end[k], d.l1[k], d.l2[k] = int(a), int(b), float(l1), float(l2)
        return d

    def l1l2_workspace_doubles(self):
        return self.L.dcahip_l1l2_workspace_doubles()

    def l1l2_apply(self, desc, w, g, loss_inout, ws):
        p = hip.ptr
        hip.check(self.L.dcahip_l1l2_apply(ctypes.byref(desc), p(w), p(g), p(loss_inout), p(ws),
                                           hip.stream()), 'l1l2_apply')

    # ------------------------------------------------------------------ preprocessing
    def prep_chunks(self, n):
        return self.L.dcahip_prep_chunks(n)

    def prep_row_sums(self, Y, ldy, n, G, out):
        hip.check(self.L.dcahip_prep_row_sums(hip.ptr(Y), ldy, n, G, hip.ptr(out), hip.stream()),
                  'prep_row_sums')

    def prep_col_pass(self, Y, ldy, n, G, fac, do_log, X, ldx, col_part):
        p = hip.ptr
        hip.check(self.L.dcahip_prep_col_pass(p(Y), ldy, n, G, p(fac), int(do_log), p(X), ldx,
                                              p(col_part), hip.stream()), 'prep_col_pass')

    def prep_col_finish(self, col_part, R, G, n_total, sums, mean, stdv):
        p = hip.ptr
        hip.check(self.L.dcahip_prep_col_finish(p(col_part), R, G, float(n_total), p(sums), p(mean),
                                                p(stdv), hip.stream()), 'prep_col_finish')

    def prep_scale(self, X, ldx, n, G, mean, stdv):
        p = hip.ptr
        hip.check(self.L.dcahip_prep_scale(p(X), ldx, n, G, p(mean), p(stdv), hip.stream()), 'prep_scale')

    # ------------------------------------------------------------------ optimizer
    def rmsprop_clip_end(self, w, g, ms, n, lr, rho, eps, clip, loss, weight, hist, rows_per_slot, acc, cursor, advance):
        p = hip.ptr
        hip.check(self.L.dcahip_rmsprop_clip_end(p(w), p(g), p(ms), n, p(lr), rho, eps, clip, p(loss), weight, p(hist),
                                                 rows_per_slot, p(acc), p(cursor), advance, hip.stream()),
                  'rmsprop_clip_end')

    def rmsprop_clip(self, w, g, ms, n, lr, rho, eps, clip):
        p = hip.ptr
        hip.check(self.L.dcahip_rmsprop_clip(p(w), p(g), p(ms), n, p(lr), rho, eps, clip,
                                             hip.stream()), 'rmsprop_clip')
