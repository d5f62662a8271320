"""Oracle (test infrastructure only): a CPU object with the SAME method interface as
dca_amd.ops.HipOps, every method computed with numpy (fp64 inside, fp32 buffers) following
the documented contract of include/dcahip.h.

Purpose: lets tests/ run the engine's HOST logic (buffer layout, step orchestration, fit loop,
data-parallel sharding + collectives over gloo) in a container without a GPU, and compare it
with the straight-line oracle in net_np.py.  It is injected explicitly by tests
(Engine(ops=CpuRefOps())); nothing in dca_amd/ imports it and the product never falls back to it.
"""
import numpy as np
import torch

from . import zinb_np as Z


def _mat(t, rows, cols, ld):
    """numpy view [rows, cols] with row stride ld starting at t's first element."""
    if t is None:
        return None
    assert not t.is_cuda
    return torch.as_strided(t, (rows, cols), (ld, 1)).numpy()


def _vec(t, n):
    return None if t is None else torch.as_strided(t, (n,), (1,)).numpy()


def _chunks(B):
    r = (B + 63) // 64
    return max(1, min(256, r))


class CpuRefOps:
    name = 'cpu-oracle'
    device_type = 'cpu'
    max_partials = 2048

    # ------------------------------------------------------------------ loss
    def zinb_nll(self, a_mean, a_disp, a_pi, lda, theta_w, Y, ldy, sf, perm, cursor, B, G, ridge,
                 inv_n, flags, d_mean, d_disp, d_pi, ldd, partials):
        has_pi, cdisp = bool(flags & 1), bool(flags & 2)
        c = int(cursor.item()) if cursor is not None else 0
        rows = (perm[c:c + B].numpy().astype(np.int64) if perm is not None else np.arange(c, c + B))
        n_store = int(rows.max()) + 1
        y = _mat(Y, n_store, G, ldy)[rows].astype(np.float64)
        sfv = _vec(sf, n_store)[rows].astype(np.float64)
        am = _mat(a_mean, B, G, lda).astype(np.float64)
        ad = None if (cdisp or a_disp is None) else _mat(a_disp, B, G, lda).astype(np.float64)
        tw = _vec(theta_w, G).astype(np.float64) if cdisp else None
        n_total = 1.0 / inv_n
        if flags & 12:                       # Poisson / squared error: mean head only
            fn = Z.poisson_loss_and_grads if flags & 4 else Z.mse_loss_and_grads
            ls, _, dm = fn(am, y, sfv, n_total)
            if d_mean is not None:
                _mat(d_mean, B, G, ldd)[:] = dm
            partials[0] = float(ls)
            return 1
        if has_pi:
            ap = _mat(a_pi, B, G, lda).astype(np.float64)
            if not cdisp:
                ls, _, dm, dd, dp = Z.zinb_loss_and_grads(am, ad, ap, y, sfv, ridge, n_total, None)
            else:       # per-element d nll / d theta * inv_n (chain applied by colsum_chain)
                mu, _, pi = Z.heads_forward(am, None, ap, sfv)
                th = np.broadcast_to(Z.const_disp(tw).reshape(1, -1), am.shape)
                ls = Z.zinb_nll(y, mu, th, pi, ridge).sum()
                dmu, dth, dpi = Z.zinb_grads(y, mu, th, pi, ridge)
                gm, _ = Z._act_grads(am, None)
                dm = dmu * sfv.reshape(-1, 1) * gm * inv_n
                dd = dth * inv_n
                dp = dpi * pi * (1 - pi) * inv_n
        else:
            if cdisp:
                mu, _, _ = Z.heads_forward(am, None, None, sfv)
                th = np.broadcast_to(Z.const_disp(tw).reshape(1, -1), am.shape)
                ls = Z.nb_nll(y, mu, th).sum()
                dmu, dth = Z.nb_grads(y, mu, th)
                gm, _ = Z._act_grads(am, None)
                dm = dmu * sfv.reshape(-1, 1) * gm * inv_n
                dd = dth * inv_n
            else:
                ls, _, dm, dd = Z.nb_loss_and_grads(am, ad, y, sfv, n_total, None)
            dp = None
        if d_mean is not None:
            _mat(d_mean, B, G, ldd)[:] = dm
            _mat(d_disp, B, G, ldd)[:] = dd
            if has_pi:
                _mat(d_pi, B, G, ldd)[:] = dp
        partials[0] = float(ls)
        return 1

    def loss_finalize(self, partials, n, scale, loss_out):
        v = float(partials[:n].sum().item()) * scale
        loss_out[0] = float(np.float32(np.inf if np.isnan(v) else v))

    def step_end(self, loss, weight, hist, rows_per_slot, acc, cursor, advance):
        c = int(cursor.item()) if cursor is not None else 0
        if loss is not None:
            lv = float(loss[0].item())
            if hist is not None:
                hist[c // rows_per_slot if rows_per_slot > 0 else 0] = lv
            if acc is not None:
                acc[0] += lv * weight
        if cursor is not None:
            cursor[0] = c + advance

    def heads_infer(self, a_mean, a_disp, a_pi, lda, sf, B, G, mean_sf, theta, pi, ldo, flags=0):
        sfv = _vec(sf, B).astype(np.float64)
        if mean_sf is not None:
            am = _mat(a_mean, B, G, lda).astype(np.float64)
            _mat(mean_sf, B, G, ldo)[:] = (am if flags & 8 else Z.mean_act(am)) * sfv[:, None]
        if theta is not None:
            _mat(theta, B, G, ldo)[:] = Z.disp_act(_mat(a_disp, B, G, lda).astype(np.float64))
        if pi is not None:
            _mat(pi, B, G, ldo)[:] = Z.sigmoid(_mat(a_pi, B, G, lda).astype(np.float64))

    # ------------------------------------------------------------------ fused heads
    def heads_fused_workspace_bytes(self, B, hL, G, plane, flags):
        return 16 if (1 <= hL <= 64 and not flags & 12) else 0

    def heads_tile_order_len(self, G):
        return ((G + 31) // 32 + 1) // 2 * 2

    def heads_fused(self, H, ldh, Wh, ldw, bh, plane, theta_w, Y, ldy, sf, perm, cursor, B, hL, G,
                    ridge, inv_n, flags, gW, ldg, g_theta, dH, lddh, partials, ws, tile_order=None, loss_out=None):
        """Contract of dcahip_heads_fused_loss = the composition of the separate entry points (tile_order only
        changes which workgroup computes what; loss_out = dcahip_loss_finalize on the partials)."""
        has_pi, cdisp = bool(flags & 1), bool(flags & 2)
        nh = 1 + (0 if cdisp else 1) + (1 if has_pi else 0)
        NH = nh * plane
        A = torch.zeros(B, NH, dtype=torch.float32)
        D = torch.zeros(B, NH + plane, dtype=torch.float32)
        self.sgemm(0, 0, B, NH, hL, H, ldh, Wh, ldw, A, NH, bias=bh)
        k_pi = nh - 1
        a_disp = None if cdisp else A[:, plane:]
        a_pi = A[:, k_pi * plane:] if has_pi else None
        d_disp = D[:, NH:] if cdisp else D[:, plane:]
        d_pi = D[:, k_pi * plane:] if has_pi else None
        n = self.zinb_nll(A, a_disp, a_pi, NH, theta_w, Y, ldy, sf, perm, cursor, B, G, ridge, inv_n,
                          flags, D, d_disp, d_pi, NH + plane, partials)
        self.sgemm(1, 0, hL, NH, B, H, ldh, D, NH + plane, gW, ldg, colsum_row=True)
        if cdisp:
            self.colsum_chain(d_disp, NH + plane, B, G, theta_w, g_theta)
        self.sgemm(0, 1, B, hL, NH, D, NH + plane, Wh, ldw, dH, lddh)
        if loss_out is not None:
            self.loss_finalize(partials, n, inv_n, loss_out)
        return n

    # ------------------------------------------------------------------ gemm
    def sgemm_workspace_bytes(self, ta, tb, M, N, K, colsum_row=False, split_k=0):
        return 0

    def sgemm(self, ta, tb, M, N, K, A, lda, B, ldb, C, ldc, bias=None, perm=None, cursor=None,
              colsum_row=False, split_k=0, ws=None):
        ra, ca = (K, M) if ta else (M, K)
        rb, cb = (N, K) if tb else (K, N)
        if perm is not None:
            c = int(cursor.item()) if cursor is not None else 0
            rows = perm[c:c + ra].numpy().astype(np.int64)
            a = _mat(A, int(rows.max()) + 1, ca, lda)[rows].astype(np.float64)
        else:
            a = _mat(A, ra, ca, lda).astype(np.float64)
        b = _mat(B, rb, cb, ldb).astype(np.float64)
        opa = a.T if ta else a
        opb = b.T if tb else b
        out = opa @ opb
        if bias is not None:
            out = out + _vec(bias, N).astype(np.float64)
        _mat(C, M, N, ldc)[:] = out
        if colsum_row:
            assert not tb, 'column sums are taken where B is stored [K, N]'
            torch.as_strided(C, (N,), (1,), C.storage_offset() + M * ldc).numpy()[:] = b.sum(axis=0)

    def transpose(self, src, ld_src, R, C, dst, ld_dst, perm=None, cursor=None):
        """dcahip_transpose_rows: dst [C, R] = src[rows]^T, rows = perm[cursor : cursor + R] when perm is given."""
        if perm is not None:
            c = int(cursor.item()) if cursor is not None else 0
            rows = perm[c:c + R].numpy().astype(np.int64)
            a = _mat(src, int(rows.max()) + 1, C, ld_src)[rows]
        else:
            a = _mat(src, R, C, ld_src)
        _mat(dst, C, R, ld_dst)[:] = a.T

    # ------------------------------------------------------------------ batch norm
    def col_moments_chunks(self, B):
        return _chunks(B)

    def col_moments(self, Zt, ldz, B, H, part):
        R = _chunks(B)
        cr = -(-B // R)
        z = _mat(Zt, B, H, ldz).astype(np.float64)
        p = _vec(part, R * 2 * H).reshape(R, 2, H)
        for r in range(R):
            blk = z[r * cr:min(B, (r + 1) * cr)]
            if len(blk):
                mu = blk.mean(0)
                p[r, 0] = mu
                p[r, 1] = ((blk - mu) ** 2).sum(0)
            else:
                p[r] = 0

    @staticmethod
    def _merge(ent, cnt):
        n = 0.0
        mean = np.zeros(ent.shape[2]); m2 = np.zeros(ent.shape[2])
        for e in range(ent.shape[0]):
            ne = float(cnt[e])
            if ne <= 0:
                continue
            tot = n + ne
            delta = ent[e, 0].astype(np.float64) - mean
            mean = mean + delta * ne / tot
            m2 = m2 + ent[e, 1].astype(np.float64) + delta * delta * n * ne / tot
            n = tot
        return n, mean, m2

    def moments_combine(self, entries, counts, E, H, out):
        n, mean, m2 = self._merge(_vec(entries, E * 2 * H).reshape(E, 2, H), _vec(counts, E))
        o = _vec(out, 2 * H)
        o[:H] = mean; o[H:] = m2

    def bn_relu_apply(self, Zt, ldz, B, H, entries, counts, E, beta, mm, mv, momentum, eps, relu,
                      Hout, ldh, xhat, ldx, inv_std):
        z = _mat(Zt, B, H, ldz).astype(np.float64)
        mmv, mvv = _vec(mm, H), _vec(mv, H)
        if entries is not None:
            if counts is None:
                cr = -(-B // E)
                cnt = [max(0, min(B, (r + 1) * cr) - r * cr) for r in range(E)]
            else:
                cnt = _vec(counts, E)
            n, mean, m2 = self._merge(_vec(entries, E * 2 * H).reshape(E, 2, H), cnt)
            mean = mean.astype(np.float32); var = (m2 / n).astype(np.float32)
            mmv[:] = mmv - (mmv - mean) * np.float32(1 - momentum)
            mvv[:] = mvv - (mvv - var) * np.float32(1 - momentum)
        else:
            mean, var = mmv.copy(), mvv.copy()
        inv = (1 / np.sqrt(var.astype(np.float64) + eps))
        xh = (z - mean) * inv
        if xhat is not None:
            _mat(xhat, B, H, ldx)[:] = xh
        y = xh + (_vec(beta, H) if beta is not None else 0.0)
        from . import net_np as N
        y = N.act_fwd(int(relu), y)
        _mat(Hout, B, H, ldh)[:] = y
        if inv_std is not None:
            _vec(inv_std, H)[:] = inv

    def bn_bwd_sums(self, dH, ldd, Hact, ldh, xhat, ldx, B, H, part, act=1):
        from . import net_np as N
        R = _chunks(B)
        cr = -(-B // R)
        dy = _mat(dH, B, H, ldd).astype(np.float64) * N.act_grad_from_out(act, _mat(Hact, B, H, ldh).astype(np.float64))
        xh = _mat(xhat, B, H, ldx).astype(np.float64)
        p = _vec(part, R * 2 * H).reshape(R, 2, H)
        for r in range(R):
            s = slice(r * cr, min(B, (r + 1) * cr))
            p[r, 0] = dy[s].sum(0)
            p[r, 1] = (dy[s] * xh[s]).sum(0)

    def bn_bwd_apply(self, dH, ldd, Hact, ldh, xhat, ldx, inv_std, sums, E, n_total, B, H, dZ, ldz,
                     dbeta, act=1):
        from . import net_np as N
        s = _vec(sums, E * 2 * H).reshape(E, 2, H).astype(np.float64).sum(0)
        dy = _mat(dH, B, H, ldd).astype(np.float64) * N.act_grad_from_out(act, _mat(Hact, B, H, ldh).astype(np.float64))
        xh = _mat(xhat, B, H, ldx).astype(np.float64)
        inv = _vec(inv_std, H).astype(np.float64)
        _mat(dZ, B, H, ldz)[:] = inv * (dy - s[0] / n_total - xh * s[1] / n_total)
        if dbeta is not None:
            _vec(dbeta, H)[:] = s[0]

    def relu_bwd(self, dH, ldd, Hact, ldh, B, H, dZ, ldz, act=1):
        from . import net_np as N
        _mat(dZ, B, H, ldz)[:] = _mat(dH, B, H, ldd).astype(np.float64) * N.act_grad_from_out(act, _mat(Hact, B, H, ldh).astype(np.float64))

    def relu_fwd(self, Zt, ldz, B, H, Hout, ldh, act=1):
        from . import net_np as N
        _mat(Hout, B, H, ldh)[:] = N.act_fwd(act, _mat(Zt, B, H, ldz).astype(np.float64))

    def colsum_chain(self, x, ldx, B, N, theta_w, out):
        s = _mat(x, B, N, ldx).astype(np.float64).sum(0)
        if theta_w is not None:
            e = np.exp(_vec(theta_w, N).astype(np.float64))
            s = s * np.where((e >= 1e-3) & (e <= 1e4), e, 0.0)
        _vec(out, N)[:] = s

    # ------------------------------------------------------------------ other optimizers, regularisers
    def optimizer_step(self, kind, w, g, slot1, slot2, n, lr, it, clip):
        from . import net_np as N
        wv = _vec(w, n)
        gv = _vec(g, n).astype(np.float64)
        a = _vec(slot1, n) if slot1 is not None else None
        b = _vec(slot2, n) if slot2 is not None else None
        t = (int(it[0].item()) if it is not None else 0) + 1
        nw, na, nb = N.optimizer_update(kind, wv.astype(np.float64), gv,
                                        None if a is None else a.astype(np.float64),
                                        None if b is None else b.astype(np.float64),
                                        float(lr[0].item()), t, clip)
        wv[:] = nw
        if a is not None:
            a[:] = na
        if b is not None:
            b[:] = nb

    def prelu_workspace_doubles(self, h):
        return 4

    def prelu_fwd(self, x, ldx, alpha, B, h, out, ldo):
        xv = _mat(x, B, h, ldx)
        _mat(out, B, h, ldo)[:] = np.where(xv > 0, xv, xv * _vec(alpha, h))

    def prelu_bwd(self, d, ldd, x, ldx, alpha, B, h, galpha, ws):
        xv = _mat(x, B, h, ldx).astype(np.float64)
        dv = _mat(d, B, h, ldd)
        d64 = dv.astype(np.float64)
        _vec(galpha, h)[:] = (d64 * np.minimum(xv, 0)).sum(axis=0)
        dv[:] = np.where(xv > 0, d64, d64 * _vec(alpha, h).astype(np.float64))

    def elempi_workspace_doubles(self, G):
        return 4

    def elempi_fwd(self, a_mean, lda, k, c, B, G, a_pi, ldp):
        a = _mat(a_mean, B, G, lda)
        a[:] = -a
        _mat(a_pi, B, G, ldp)[:] = a * _vec(k, G) + _vec(c, G)

    def elempi_bwd(self, m, lda, d_mean, d_pi, ldd, k, B, G, gk, gc, ws):
        mm = _mat(m, B, G, lda).astype(np.float64)
        dp = _mat(d_pi, B, G, ldd).astype(np.float64)
        dm = _mat(d_mean, B, G, ldd)
        _vec(gk, G)[:] = (dp * mm).sum(axis=0)
        _vec(gc, G)[:] = dp.sum(axis=0)
        dm[:] = -(dm.astype(np.float64) + dp * _vec(k, G).astype(np.float64))

    def bcast_cols(self, s, lds, B, G, out, ldo):
        _mat(out, B, G, ldo)[:] = _mat(s, B, 1, lds)

    def row_sums_strided(self, x, ldx, B, G, out, ldo):
        _mat(out, B, 1, ldo)[:, 0] = _mat(x, B, G, ldx).astype(np.float64).sum(axis=1)

    def nadam_step(self, w, g, m, v, n, lr, it, m_schedule, clip):
        from . import net_np as N
        t = int(it[0].item()) + 1
        wv, mv, vv = _vec(w, n), _vec(m, n), _vec(v, n)
        nw, na, nb = N.optimizer_update('nadam', wv.astype(np.float64), _vec(g, n).astype(np.float64),
                                        mv.astype(np.float64), vv.astype(np.float64), float(lr[0].item()), t, clip)
        wv[:] = nw
        mv[:] = na
        vv[:] = nb
        m_schedule[0] = float(m_schedule[0].item()) * N.nadam_mu(t)

    def dropout_apply(self, x, ldx, perm, cursor, B, h, rate, seed, step, layer, row0, out, ldo):
        from . import net_np as N
        if B == 0:
            return
        if perm is not None:
            c = int(cursor.item()) if cursor is not None else 0
            rows = perm[c:c + B].numpy().astype(np.int64)
            src = _mat(x, int(rows.max()) + 1, h, ldx)[rows]
        else:
            src = _mat(x, B, h, ldx)
        keep = N.dropout_keep(seed, int(step[0].item()) if step is not None else 0, layer, row0, B, h, rate)
        _mat(out, B, h, ldo)[:] = np.where(keep, src * N.dropout_scale(rate), np.float32(0))

    def counter_add(self, counter, v):
        counter[0] += v

    def reg_desc(self, segs):
        return [(int(a), int(b), float(l1), float(l2)) for a, b, l1, l2 in segs]

    def l1l2_workspace_doubles(self):
        return 16

    def l1l2_apply(self, desc, w, g, loss_inout, ws):
        pen = 0.0
        for a, b, l1, l2 in desc:
            if b <= a or (l1 == 0 and l2 == 0):
                continue
            x = torch.as_strided(w, (b - a,), (1,), w.storage_offset() + a).numpy().astype(np.float64)
            if g is not None:
                gv = torch.as_strided(g, (b - a,), (1,), g.storage_offset() + a).numpy()
                gv[:] = gv + l1 * np.sign(x) + 2 * l2 * x
            pen += l1 * np.abs(x).sum() + l2 * np.square(x).sum()
        if loss_inout is not None and pen != 0.0:
            loss_inout[0] = float(np.float32(float(loss_inout[0].item()) + pen))

    # ------------------------------------------------------------------ preprocessing
    def prep_chunks(self, n):
        return max(1, min(512, (n + 127) // 128))

    def prep_row_sums(self, Y, ldy, n, G, out):
        _vec(out, n)[:] = _mat(Y, n, G, ldy).astype(np.float64).sum(axis=1)

    def prep_col_pass(self, Y, ldy, n, G, fac, do_log, X, ldx, col_part):
        x = _mat(Y, n, G, ldy).astype(np.float32)
        if fac is not None:
            x = (x / _vec(fac, n).astype(np.float32)[:, None]).astype(np.float32)
        if do_log:
            x = np.log1p(x).astype(np.float32)
        if X is not None:
            _mat(X, n, G, ldx)[:] = x
        R = self.prep_chunks(n)
        Gp = (G + 3) // 4 * 4
        cr = -(-n // R)
        part = torch.as_strided(col_part, (R, 2, Gp), (2 * Gp, Gp, 1)).numpy()
        for r in range(R):
            blk = x[r * cr:min(n, (r + 1) * cr)]
            part[r, 0, :G] = blk.sum(axis=0, dtype=np.float64)
            part[r, 1, :G] = np.multiply(blk, blk).sum(axis=0, dtype=np.float64)

    def prep_col_finish(self, col_part, R, G, n_total, sums, mean, stdv):
        Gp = (G + 3) // 4 * 4
        part = torch.as_strided(col_part, (R, 2, Gp), (2 * Gp, Gp, 1)).numpy()
        s1 = part[:, 0, :G].sum(axis=0); s2 = part[:, 1, :G].sum(axis=0)
        if sums is not None:
            _vec(sums, G)[:] = s1
        if mean is not None:
            m = s1 / n_total
            var = (s2 / n_total - m * m) * (n_total / (n_total - 1.0)) if n_total > 1 else np.zeros_like(m)
            sd = np.sqrt(np.maximum(var, 0)); sd[sd == 0] = 1
            _vec(mean, G)[:] = m; _vec(stdv, G)[:] = sd

    def prep_scale(self, X, ldx, n, G, mean, stdv):
        x = _mat(X, n, G, ldx)
        x[:] = (x - _vec(mean, G).astype(np.float32)) / _vec(stdv, G).astype(np.float32)

    # ------------------------------------------------------------------ optimizer
    def rmsprop_clip(self, w, g, ms, n, lr, rho, eps, clip):
        wv, gv, mv = _vec(w, n), _vec(g, n).astype(np.float64), _vec(ms, n)
        if clip > 0:
            gv = np.clip(gv, -clip, clip)
        m = rho * mv.astype(np.float64) + (1 - rho) * gv * gv
        mv[:] = m
        wv[:] = wv - float(lr[0].item()) * gv / (np.sqrt(m) + eps)
