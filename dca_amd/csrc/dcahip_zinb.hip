// K-ZINB: fused NB / ZINB negative log-likelihood + gradient over the cells x genes tile,
// inference heads, deterministic loss reduction.  gfx950 (MI355X), wave64.
//
// Memory-bound by design: per element the training kernel reads 3 pre-activations + y
// (16 B) and writes 3 gradients (12 B) = 28 B; everything else (exp, softplus, sigmoid,
// log1p, pow, lgamma/digamma differences) stays in registers.  Each lane owns one 16-byte
// quad of consecutive genes so every global access is a coalesced dwordx4; the minibatch
// gather (row -> perm[cursor + row]) is a wave-uniform scalar load.
//
// Reference arithmetic restated here: dca/network.py:38-39 (MeanAct, DispAct),
// dca/layers.py:21,85, dca/loss.py:72-114 (NB.loss), dca/loss.py:122-156 (ZINB.loss).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdlib.h>
#include "dcahip.h"
#include "zinb_math.hpp"
#include "h2_math.hpp"

namespace {

constexpr int kMaxPartials = 2048;    // 256 CUs x 8 blocks

struct NllArgs {
    const float *a_mean, *a_disp, *a_pi, *theta_w, *y, *sf;
    const int* perm;
    const long long* cursor;
    long lda, ldy, ldd;
    float *d_mean, *d_disp, *d_pi;
    double* partials;
    int B, G;
    float ridge, inv_n;
    // gradient planes as pre-split bf16 pieces (dcahip_zinb_nll_planes): piece q of head plane h, element (row, g) at
    // pl[h] + q * pstride + row * ldp + g
    unsigned short* pl[3];
    long ldp, pstride;
    // fp16 x 2 planes (dcahip_zinb_nll_planes_h2): the head planes hold the UNSCALED gradient g 2^kD = pscale g in TWO pieces
    // (h2_math.hpp); a per-gene dispersion's fp32 plane keeps inv_n
    float pscale;
};

// the three bf16 pieces of fp32 values (x = p0 + p1 + p2 to 2^-24 |x|: the split of dcahip_sgemm / dcahip_split_planes)
using bf16x2v = __attribute__((ext_vector_type(2))) __bf16;
using f32x2v = __attribute__((ext_vector_type(2))) float;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, bf16x2v));
}
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(p0 << 16), r1 = x1 - __uint_as_float(p0 & 0xffff0000u);
    p1 = pk_bf16(r0, r1);
    const float q0 = r0 - __uint_as_float(p1 << 16), q1 = r1 - __uint_as_float(p1 & 0xffff0000u);
    p2 = pk_bf16(q0, q1);
}
__device__ __forceinline__ void store_planes4(unsigned short* base, long pstride, const float (&v)[4]) {
    unsigned a0, a1, a2, b0, b1, b2;
    split_pair(v[0], v[1], a0, a1, a2);
    split_pair(v[2], v[3], b0, b1, b2);
    *reinterpret_cast<u32x2*>(base) = u32x2{a0, b0};
    *reinterpret_cast<u32x2*>(base + pstride) = u32x2{a1, b1};
    *reinterpret_cast<u32x2*>(base + 2 * pstride) = u32x2{a2, b2};
}
// two fp16 pieces (the values arrive scaled: h2_math.hpp)
__device__ __forceinline__ void store_planes4_h2(unsigned short* base, long pstride, const float (&v)[4]) {
    unsigned a0, a1, b0, b1;
    h2_split_pair(v[0], v[1], a0, a1);
    h2_split_pair(v[2], v[3], b0, b1);
    *reinterpret_cast<u32x2*>(base) = u32x2{a0, b0};
    *reinterpret_cast<u32x2*>(base + pstride) = u32x2{a1, b1};
}
__device__ __forceinline__ void store_planes1(unsigned short* base, long pstride, float v) {
    unsigned a0, a1, a2;
    split_pair(v, 0.f, a0, a1, a2);
    base[0] = (unsigned short)a0; base[pstride] = (unsigned short)a1; base[2 * pstride] = (unsigned short)a2;
}

__device__ __forceinline__ double block_reduce_sum(double v) {
    __shared__ double red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    double r = 0.0;
    if (threadIdx.x == 0) r = (red[0] + red[1]) + (red[2] + red[3]);
    return r;
}

template <int V> struct Vec;
template <> struct Vec<4> { using T = float4; };
template <> struct Vec<1> { using T = float; };

template <int V>
__device__ __forceinline__ void ldv(const float* p, float (&o)[V]) {
    if (V == 4) { const float4 t = *reinterpret_cast<const float4*>(p); o[0] = t.x; o[1 % V] = t.y; o[2 % V] = t.z; o[3 % V] = t.w; }
    else o[0] = *p;
}
template <int V>
__device__ __forceinline__ void stv(float* p, const float (&o)[V]) {
    if (V == 4) *reinterpret_cast<float4*>(p) = make_float4(o[0], o[1 % V], o[2 % V], o[3 % V]);
    else *p = o[0];
}

// LOSS: 0 = NB / ZINB, 1 = Poisson, 2 = squared error (mean head only)
template <bool HAS_PI, bool CONST_DISP, bool GRAD, int V, int LOSS = 0>
__global__ __launch_bounds__(256) void zinb_nll_kernel(NllArgs a) {
    // grid.x walks the gene segments (256 lanes x V genes), grid.y strides over the batch rows:
    // no integer division on the path, the per-row gather index and size factor are scalar loads
    const int nvec = (a.G + V - 1) / V;              // lanes' work units per row
    const int nseg = (nvec + 255) >> 8;
    const long long cur = a.cursor ? *a.cursor : 0;
    double dacc = 0.0;
    for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const int q = seg * 256 + threadIdx.x;
        if (q >= nvec) continue;
        const int g = q * V;
        float vd[V];
        if (CONST_DISP) ldv<V>(a.theta_w + g, vd);
        for (int row = blockIdx.y; row < a.B; row += gridDim.y) {
            const long srow = a.perm ? (long)a.perm[cur + row] : (long)(cur + row);
            const float sf = a.sf[srow];
            const long ao = (long)row * a.lda + g;
            float vm[V], vp[V], vy[V];
            ldv<V>(a.a_mean + ao, vm);
            if (!CONST_DISP && LOSS == 0) ldv<V>(a.a_disp + ao, vd);
            if (HAS_PI) ldv<V>(a.a_pi + ao, vp);
            ldv<V>(a.y + srow * a.ldy + g, vy);
            float om[V], od[V], op[V];
            float lacc = 0.f;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const bool valid = (g + j) < a.G;
                if (LOSS != 0) {
                    float d_am;
                    const float nll = LOSS == 1 ? poisson_elem(vm[j], sf, vy[j], d_am)
                                                : mse_elem(vm[j], sf, vy[j], d_am);
                    lacc += valid ? nll : 0.f;
                    if (GRAD) om[j] = valid ? d_am * a.inv_n : 0.f;
                    continue;
                }
                float dmu = 0.f, dth = 0.f, dpi = 0.f;
                const Heads h = head_acts<HAS_PI, CONST_DISP>(vm[j], vd[j], HAS_PI ? vp[j] : 0.f, sf);
                const float nll = nll_elem<HAS_PI, GRAD>(h, vy[j], a.ridge, dmu, dth, dpi);
                lacc += valid ? nll : 0.f;
                if (GRAD) {
                    om[j] = valid ? dmu * h.gm * a.inv_n : 0.f;
                    od[j] = valid ? dth * h.gd * a.inv_n : 0.f;
                    op[j] = valid ? dpi * h.pi * h.omp * a.inv_n : 0.f;
                }
            }
            dacc += (double)lacc;
            if (GRAD) {
                const long dof = (long)row * a.ldd + g;
                stv<V>(a.d_mean + dof, om);
                if (LOSS == 0) stv<V>(a.d_disp + dof, od);
                if (HAS_PI) stv<V>(a.d_pi + dof, op);
            }
        }
    }
    const double r = block_reduce_sum(dacc);
    if (threadIdx.x == 0) a.partials[blockIdx.y * gridDim.x + blockIdx.x] = r;
}

// ---- NB / ZINB, 16-byte aligned operands: the training / validation kernel of the separate-head path (decoders
// wider than 64 hidden units, validation).  Same decomposition as above -- one quad of genes per lane, rows strided
// over grid.y -- but the two branches of the likelihood are no longer evaluated under divergence:
//   * dense pass: the y = 0 formulas for every element (zinb_zero_elem / nb_zero_elem: ~100 VALU, the form K-HEADS
//     uses); the ~7 % non-zero elements are queued per wave in LDS (pre-activations, count, size factor, destination);
//   * sparse pass: whenever a wave holds 64 entries it evaluates the NB branch (lgamma / digamma differences) with
//     all lanes busy and overwrites the three gradient values of those elements (4-byte stores behind the dense
//     16-byte stores of the same wave: `s_waitcnt vmcnt(0)` in between orders them).
// Before: both branches for every wave-row (~370 VALU per element) at 3.0-3.2 TB/s = 0.38-0.40 of the HBM peak.
constexpr int kNzCap = 64 + 256;      // entries per wave: < 64 left over + up to 4 x 64 pushed per row
struct NzEntry { float am, ad, ap, y, sf; unsigned dof, dof2; };     // dof2: the fp32 dispersion plane beside bf16 planes

// PL: the gradient planes leave as pre-split bf16 pieces (a.pl; the per-gene dispersion plane of CONST_DISP stays fp32)
template <bool HAS_PI, bool CONST_DISP, bool GRAD, bool PL = false>
__global__ __launch_bounds__(256, 4) void zinb_nll_compact_kernel(NllArgs a) {     // 4 waves per SIMD: 128 registers
    constexpr int V = 4;
    __shared__ NzEntry queue[4][kNzCap];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    NzEntry* Q = queue[wave];
    const int nvec = (a.G + V - 1) / V;
    const int nseg = (nvec + 255) >> 8;
    const long long cur = a.cursor ? *a.cursor : 0;
    double dacc = 0.0;
    int qn = 0;
    float lsp = 0.f;
    auto flush = [&](bool all) {
        while (qn >= 64 || (all && qn > 0)) {
            const int c = qn < 64 ? qn : 64;
            const bool act = lane < c;
            const NzEntry e = Q[qn - c + (act ? lane : 0)];
            float o1, o2, o3 = 0.f, nll;
            if (HAS_PI) {
                nll = zinb_nz_elem<CONST_DISP>(e.am, e.ad, e.ap, e.sf, e.y, a.ridge, o1, o2, o3);
            } else {
                float dmu = 0.f, dth = 0.f, dpi = 0.f;
                const Heads hd = head_acts<false, CONST_DISP>(e.am, e.ad, 0.f, e.sf);
                nll = nll_elem<false, true, true>(hd, e.y, a.ridge, dmu, dth, dpi);
                o1 = dmu * hd.gm; o2 = dth * hd.gd;
            }
            lsp += act ? nll : 0.f;
            if (GRAD && act) {
                __builtin_amdgcn_s_waitcnt(0x0f70);                       // vmcnt(0): the dense stores of these addresses are done
                if (PL) {
                    store_planes1(a.pl[0] + e.dof, a.pstride, o1 * a.inv_n);
                    if (CONST_DISP) a.d_disp[e.dof2] = o2 * a.inv_n; else store_planes1(a.pl[1] + e.dof, a.pstride, o2 * a.inv_n);
                    if (HAS_PI) store_planes1(a.pl[2] + e.dof, a.pstride, o3 * a.inv_n);
                } else {
                    a.d_mean[e.dof] = o1 * a.inv_n;
                    a.d_disp[e.dof] = o2 * a.inv_n;
                    if (HAS_PI) a.d_pi[e.dof] = o3 * a.inv_n;
                }
            }
            qn -= c;
        }
    };
    for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const int q = seg * 256 + threadIdx.x;
        const bool qv = q < nvec;
        const int g = (qv ? q : 0) * V;
        float vd[V] = {0.f, 0.f, 0.f, 0.f};
        if (CONST_DISP && qv) ldv<V>(a.theta_w + g, vd);
        // storage row of the NEXT iteration requested one iteration ahead: perm -> (size factor, counts) is a chain of
        // two memory round trips per row otherwise
        long srow_n = a.perm ? (long)a.perm[cur + (blockIdx.y < a.B ? blockIdx.y : 0)] : (long)(cur + blockIdx.y);
        for (int row = blockIdx.y; row < a.B; row += gridDim.y) {
            const long srow = srow_n;
            const int rown = row + gridDim.y < a.B ? row + gridDim.y : row;
            srow_n = a.perm ? (long)a.perm[cur + rown] : (long)(cur + rown);
            const float sf = a.sf[srow];
            const long ao = (long)row * a.lda + g;
            const long dof2 = (long)row * a.ldd + g;
            const long dof = PL ? (long)row * a.ldp + g : dof2;
            float vm[V] = {0.f, 0.f, 0.f, 0.f}, vp[V] = {0.f, 0.f, 0.f, 0.f}, vy[V] = {0.f, 0.f, 0.f, 0.f};
            if (qv) {
                ldv<V>(a.a_mean + ao, vm);
                if (!CONST_DISP) ldv<V>(a.a_disp + ao, vd);
                if (HAS_PI) ldv<V>(a.a_pi + ao, vp);
                ldv<V>(a.y + srow * a.ldy + g, vy);
            }
            float om[V], od[V], op[V];
            float lacc = 0.f;
            bool nz[V];
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const bool valid = qv && (g + j) < a.G;
                nz[j] = valid && (HAS_PI ? !(vy[j] < kZeroThresh) : (vy[j] != 0.f));
                float gmv, gdv, gpv = 0.f, nll;
                if (HAS_PI) nll = zinb_zero_elem<CONST_DISP>(vm[j], vd[j], vp[j], sf, a.ridge, gmv, gdv, gpv);
                else nll = nb_zero_elem<CONST_DISP>(vm[j], vd[j], sf, gmv, gdv);
                lacc += (valid && !nz[j]) ? nll : 0.f;
                const float sc = valid ? a.inv_n : 0.f;
                om[j] = gmv * sc; od[j] = gdv * sc; op[j] = gpv * sc;
            }
            if (GRAD && qv) {
                if (PL) {
                    store_planes4(a.pl[0] + dof, a.pstride, om);
                    if (CONST_DISP) stv<V>(a.d_disp + dof2, od); else store_planes4(a.pl[1] + dof, a.pstride, od);
                    if (HAS_PI) store_planes4(a.pl[2] + dof, a.pstride, op);
                } else {
                    stv<V>(a.d_mean + dof, om);
                    stv<V>(a.d_disp + dof, od);
                    if (HAS_PI) stv<V>(a.d_pi + dof, op);
                }
            }
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const unsigned long long m = __ballot(nz[j]);
                if (m) {
                    const int slot = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (nz[j]) Q[slot] = NzEntry{vm[j], vd[j], vp[j], vy[j], sf, (unsigned)(dof + j), (unsigned)(dof2 + j)};
                    qn += __popcll(m);
                }
            }
            dacc += (double)lacc;
            flush(false);
        }
    }
    flush(true);
    dacc += (double)lsp;
    const double r = block_reduce_sum(dacc);
    if (threadIdx.x == 0) a.partials[blockIdx.y * gridDim.x + blockIdx.x] = r;
}

// ---- the training form of the kernel above (gradients wanted): nothing is stored twice.  The wave takes TWO batch rows of
// its 256 genes per iteration, evaluates the y = 0 formulas of both into registers, queues the non-zero elements of both
// rows in LDS (row 0 first), evaluates the queue 64 entries at a time with the NB branch -- each result written back over
// its queue entry -- and every lane then fetches the results of its own non-zero elements (its slot follows from the
// wave-uniform ballot masks) before the one dense store of the two rows.  Against the patching form: no 2- / 4-byte
// scattered stores behind the dense ones (measured at configs[4]'s decoder, 2048 x 25 000, bf16 pieces out: 0.445 ms, of
// which 0.145 ms were the patches -- partial-line writes -- profiles/r03_zinb_rows_notes.txt).
#ifdef DCA_ZINB_ROWS
constexpr int kRowsPerIter = DCA_ZINB_ROWS;
#else
constexpr int kRowsPerIter = 2;
#endif
template <bool HAS_PI, bool CONST_DISP, int PL>          // PL: 0 fp32 gradient planes, 1 three bf16 pieces, 2 two fp16 pieces of g 2^kD
__global__ __launch_bounds__(256, kRowsPerIter <= 2 ? 4 : 3) void zinb_nll_rows_kernel(NllArgs a) {
    const float psc = PL == 2 ? a.pscale : a.inv_n;        // scale of the head planes; a per-gene dispersion's plane: inv_n
    constexpr int V = 4, R = kRowsPerIter;
    __shared__ float4 queue[4][R * 256];                   // 8 KB per wave: every element of both rows may be non-zero
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float4* Q = queue[wave];
    const int nvec = (a.G + V - 1) / V;
    const int nseg = (nvec + 255) >> 8;
    const long long cur = a.cursor ? *a.cursor : 0;
    double dacc = 0.0;
    float lsp = 0.f;
    for (int seg = blockIdx.x; seg < nseg; seg += gridDim.x) {
        const int q = seg * 256 + threadIdx.x;
        const bool qv = q < nvec;
        const int g = (qv ? q : 0) * V;
        float vd[R][V];
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < V; ++j) vd[r][j] = 0.f;
        if (CONST_DISP && qv) {
            ldv<V>(a.theta_w + g, vd[0]);
#pragma unroll
            for (int r = 1; r < R; ++r)
#pragma unroll
                for (int j = 0; j < V; ++j) vd[r][j] = vd[0][j];
        }
        // storage rows of the NEXT iteration requested one iteration ahead (perm -> size factor / counts is a chain)
        long srow_n[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int rw = blockIdx.y + r * gridDim.y;
            const int rc = rw < a.B ? rw : 0;
            srow_n[r] = a.perm ? (long)a.perm[cur + rc] : (long)(cur + rc);
        }
        for (int row0 = blockIdx.y; row0 < a.B; row0 += R * gridDim.y) {
            long srow[R];
            float sf[R];
            bool rv[R];
            float vm[R][V], vp[R][V], vy[R][V];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int row = row0 + r * gridDim.y;
                rv[r] = row < a.B;
                srow[r] = srow_n[r];
                const int rn = row + R * gridDim.y;
                const int rc = rn < a.B ? rn : (rv[r] ? row : 0);
                srow_n[r] = a.perm ? (long)a.perm[cur + rc] : (long)(cur + rc);
                sf[r] = a.sf[srow[r]];
#pragma unroll
                for (int j = 0; j < V; ++j) { vm[r][j] = 0.f; vp[r][j] = 0.f; vy[r][j] = 0.f; }
                if (qv && rv[r]) {
                    const long ao = (long)row * a.lda + g;
                    ldv<V>(a.a_mean + ao, vm[r]);
                    if (!CONST_DISP) ldv<V>(a.a_disp + ao, vd[r]);
                    if (HAS_PI) ldv<V>(a.a_pi + ao, vp[r]);
                    ldv<V>(a.y + srow[r] * a.ldy + g, vy[r]);
                }
            }
            float om[R][V], od[R][V], op[R][V];
            bool nz[R][V];
            unsigned long long mask[R][V];
            int base[R][V];
            int qn = 0, nend[R];
            float lacc = 0.f;
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    const bool valid = qv && rv[r] && (g + j) < a.G;
                    nz[r][j] = valid && (HAS_PI ? !(vy[r][j] < kZeroThresh) : (vy[r][j] != 0.f));
                    float gmv, gdv, gpv = 0.f, nll;
                    if (HAS_PI) nll = zinb_zero_elem<CONST_DISP>(vm[r][j], vd[r][j], vp[r][j], sf[r], a.ridge, gmv, gdv, gpv);
                    else nll = nb_zero_elem<CONST_DISP>(vm[r][j], vd[r][j], sf[r], gmv, gdv);
                    lacc += (valid && !nz[r][j]) ? nll : 0.f;
                    const float sc = valid ? psc : 0.f, scd = valid ? (CONST_DISP ? a.inv_n : psc) : 0.f;
                    om[r][j] = gmv * sc; od[r][j] = gdv * scd; op[r][j] = gpv * sc;
                    const unsigned long long m = __ballot(nz[r][j]);
                    mask[r][j] = m; base[r][j] = qn;
                    if (nz[r][j]) {
                        const int slot = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        Q[slot] = make_float4(vm[r][j], vd[r][j], vp[r][j], vy[r][j]);
                    }
                    qn += __popcll(m);
                }
                nend[r] = qn;
            }
            dacc += (double)lacc;
            __builtin_amdgcn_wave_barrier();
            for (int b = 0; b < qn; b += 64) {                 // the NB branch, 64 queued elements at a time
                const int idx = b + lane;
                const bool act = idx < qn;
                const float4 e = Q[act ? idx : b];
                float esf = sf[R - 1];
#pragma unroll
                for (int r = R - 2; r >= 0; --r) esf = idx < nend[r] ? sf[r] : esf;
                float o1, o2, o3 = 0.f, nll;
                if (HAS_PI) {
                    nll = zinb_nz_elem<CONST_DISP>(e.x, e.y, e.z, esf, e.w, a.ridge, o1, o2, o3);
                } else {
                    float dmu = 0.f, dth = 0.f, dpi = 0.f;
                    const Heads hd = head_acts<false, CONST_DISP>(e.x, e.y, 0.f, esf);
                    nll = nll_elem<false, true, true>(hd, e.w, a.ridge, dmu, dth, dpi);
                    o1 = dmu * hd.gm; o2 = dth * hd.gd;
                }
                lsp += act ? nll : 0.f;
                if (act) Q[idx] = make_float4(o1 * psc, o2 * (CONST_DISP ? a.inv_n : psc), o3 * psc, 0.f);
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int r = 0; r < R; ++r) {
#pragma unroll
                for (int j = 0; j < V; ++j) {
                    if (nz[r][j]) {
                        const unsigned long long m = mask[r][j];
                        const float4 o = Q[base[r][j] + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))];
                        om[r][j] = o.x; od[r][j] = o.y; op[r][j] = o.z;
                    }
                }
                if (qv && rv[r]) {
                    const int row = row0 + r * gridDim.y;
                    const long dof2 = (long)row * a.ldd + g;
                    if (PL == 2) {
                        const long dof = (long)row * a.ldp + g;
                        store_planes4_h2(a.pl[0] + dof, a.pstride, om[r]);
                        if (CONST_DISP) stv<V>(a.d_disp + dof2, od[r]); else store_planes4_h2(a.pl[1] + dof, a.pstride, od[r]);
                        if (HAS_PI) store_planes4_h2(a.pl[2] + dof, a.pstride, op[r]);
                    } else if (PL == 1) {
                        const long dof = (long)row * a.ldp + g;
                        store_planes4(a.pl[0] + dof, a.pstride, om[r]);
                        if (CONST_DISP) stv<V>(a.d_disp + dof2, od[r]); else store_planes4(a.pl[1] + dof, a.pstride, od[r]);
                        if (HAS_PI) store_planes4(a.pl[2] + dof, a.pstride, op[r]);
                    } else {
                        stv<V>(a.d_mean + dof2, om[r]);
                        stv<V>(a.d_disp + dof2, od[r]);
                        if (HAS_PI) stv<V>(a.d_pi + dof2, op[r]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    dacc += (double)lsp;
    const double r = block_reduce_sum(dacc);
    if (threadIdx.x == 0) a.partials[blockIdx.y * gridDim.x + blockIdx.x] = r;
}

__global__ __launch_bounds__(256) void loss_finalize_kernel(const double* partials, int n,
                                                            double scale, float* out) {
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += partials[i];
    const double r = block_reduce_sum(v);
    if (threadIdx.x == 0) {
        double l = r * scale;
        float lf = (float)l;
        if (isnan(lf)) lf = INFINITY;                 // loss.py:148 _nan2inf
        *out = lf;
    }
}

__global__ void step_end_kernel(const float* loss, double weight, float* hist, int rows_per_slot,
                                double* acc, long long* cursor, int advance) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const long long c = cursor ? *cursor : 0;
        if (loss) {
            const float l = *loss;
            if (hist) hist[rows_per_slot > 0 ? c / rows_per_slot : 0] = l;
            if (acc) *acc += (double)l * weight;
        }
        if (cursor) *cursor = c + advance;
    }
}

struct InferArgs {
    const float *a_mean, *a_disp, *a_pi, *sf;
    float *mean_sf, *theta, *pi;
    long lda, ldo;
    int B, G;
    int linear_mean;
};

template <int V>
__global__ __launch_bounds__(256) void heads_infer_kernel(InferArgs a) {
    const int nvec = (a.G + V - 1) / V;
    const int nseg = (nvec + 255) >> 8;
    const long total = (long)a.B * nseg;
    for (long item = blockIdx.x; item < total; item += gridDim.x) {
        const int row = (int)(item / nseg);
        const int q = (int)(item - (long)row * nseg) * 256 + threadIdx.x;
        if (q >= nvec) continue;
        const int g = q * V;
        const float sf = a.sf[row];
        const long ao = (long)row * a.lda + g, oo = (long)row * a.ldo + g;
        float v[V], o[V];
        if (a.mean_sf) {
            ldv<V>(a.a_mean + ao, v);
#pragma unroll
            for (int j = 0; j < V; ++j) o[j] = a.linear_mean ? v[j] * sf : fminf(fmaxf(expf(v[j]), 1e-5f), 1e6f) * sf;
            stv<V>(a.mean_sf + oo, o);
        }
        if (a.theta) {
            ldv<V>(a.a_disp + ao, v);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float sp = fmaxf(v[j], 0.f) + log1pf(expf(-fabsf(v[j])));
                o[j] = fminf(fmaxf(sp, 1e-4f), 1e4f);
            }
            stv<V>(a.theta + oo, o);
        }
        if (a.pi) {
            ldv<V>(a.a_pi + ao, v);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                const float ex = expf(-fabsf(v[j]));
                const float s = 1.f / (1.f + ex);
                o[j] = v[j] >= 0.f ? s : ex * s;
            }
            stv<V>(a.pi + oo, o);
        }
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

template <bool HAS_PI, bool CONST_DISP, bool GRAD>
int launch_nll(const NllArgs& a, bool vec, dim3 grid, hipStream_t s) {
    // the compacted kernel addresses gradient elements with 32 bits
#ifdef DCA_ZINB_PLAIN          // A/B builds only: K-ZINB without the non-zero compaction
    constexpr bool plain = true;
#else
    constexpr bool plain = false;
#endif
#ifdef DCA_ZINB_PATCH          // A/B builds only: the non-zero elements patched behind the dense stores (round-2 form)
    constexpr bool patching = true;
#else
    constexpr bool patching = false;
#endif
    const bool fits = !GRAD || (long)a.B * a.ldd < (1L << 32);
    if (vec && fits && !plain && GRAD && !patching) hipLaunchKernelGGL((zinb_nll_rows_kernel<HAS_PI, CONST_DISP, 0>), grid, dim3(256), 0, s, a);
    else if (vec && fits && !plain) hipLaunchKernelGGL((zinb_nll_compact_kernel<HAS_PI, CONST_DISP, GRAD>), grid, dim3(256), 0, s, a);
    else if (vec) hipLaunchKernelGGL((zinb_nll_kernel<HAS_PI, CONST_DISP, GRAD, 4>), grid, dim3(256), 0, s, a);
    else     hipLaunchKernelGGL((zinb_nll_kernel<HAS_PI, CONST_DISP, GRAD, 1>), grid, dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

template <int LOSS, bool GRAD>
int launch_nll_simple(const NllArgs& a, bool vec, dim3 grid, hipStream_t s) {
    if (vec) hipLaunchKernelGGL((zinb_nll_kernel<false, false, GRAD, 4, LOSS>), grid, dim3(256), 0, s, a);
    else     hipLaunchKernelGGL((zinb_nll_kernel<false, false, GRAD, 1, LOSS>), grid, dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int dcahip_version(void) { return DCAHIP_VERSION; }
// size of the caller's loss-partial buffer: this file's kernels write at most kMaxPartials (their grid cap), the small-batch
// K-HEADS kernel one per (gene tile, row tile) workgroup, up to 8192 (dcahip_heads.hip kMaxSmallGrid)
extern "C" int dcahip_zinb_max_partials(void) { return 8192; }

extern "C" int dcahip_zinb_nll(const float* a_mean, const float* a_disp, const float* a_pi, long lda,
                               const float* theta_w, const float* y, long ldy, const float* sf,
                               const int* perm, const long long* cursor, int B, int G, float ridge,
                               float inv_n, int flags, float* d_mean, float* d_disp, float* d_pi,
                               long ldd, double* loss_partials, int* n_partials_out, void* stream) {
    const bool has_pi = flags & DCAHIP_NLL_HAS_PI, cdisp = flags & DCAHIP_NLL_CONST_DISP;
    const bool grad = d_mean != nullptr;
    const int loss = (flags & DCAHIP_NLL_POISSON) ? 1 : ((flags & DCAHIP_NLL_MSE) ? 2 : 0);
    if (B <= 0 || G <= 0 || !a_mean || !y || !sf || !loss_partials) return DCAHIP_EINVAL;
    if (loss != 0 && (has_pi || cdisp)) return DCAHIP_EINVAL;
    if (has_pi && !a_pi) return DCAHIP_EINVAL;
    if (loss == 0 && (cdisp ? (theta_w == nullptr) : (a_disp == nullptr))) return DCAHIP_EINVAL;
    if (grad && loss == 0 && (!d_disp || (has_pi && !d_pi))) return DCAHIP_EINVAL;
    bool vec = (lda % 4 == 0) && (ldy % 4 == 0) && al16(a_mean) && al16(y) &&
               (loss != 0 || (cdisp ? al16(theta_w) : al16(a_disp))) && (!has_pi || al16(a_pi)) &&
               lda >= ((G + 3) & ~3) && ldy >= ((G + 3) & ~3);
    if (grad) vec = vec && (ldd % 4 == 0) && ldd >= ((G + 3) & ~3) && al16(d_mean) && (loss != 0 || al16(d_disp)) && (!has_pi || al16(d_pi));
    NllArgs a{a_mean, a_disp, a_pi, theta_w, y, sf, perm, cursor, lda, ldy, ldd,
              d_mean, d_disp, d_pi, loss_partials, B, G, ridge, inv_n, {nullptr, nullptr, nullptr}, 0, 0};
    const int V = vec ? 4 : 1;
    const int nvec = (G + V - 1) / V;
    const int nseg = (nvec + 255) / 256;
    const int gx = nseg < kMaxPartials ? nseg : kMaxPartials;
    int gy = kMaxPartials / gx;
    if (gy > B) gy = B;
    if (gy < 1) gy = 1;
    const dim3 grid(gx, gy);
    if (n_partials_out) *n_partials_out = gx * gy;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (loss == 1) return grad ? launch_nll_simple<1, true>(a, vec, grid, s) : launch_nll_simple<1, false>(a, vec, grid, s);
    if (loss == 2) return grad ? launch_nll_simple<2, true>(a, vec, grid, s) : launch_nll_simple<2, false>(a, vec, grid, s);
#define DCA_DISPATCH(P, C)                                                  \
    return grad ? launch_nll<P, C, true>(a, vec, grid, s) : launch_nll<P, C, false>(a, vec, grid, s)
    if (has_pi && cdisp) { DCA_DISPATCH(true, true); }
    if (has_pi) { DCA_DISPATCH(true, false); }
    if (cdisp) { DCA_DISPATCH(false, true); }
    DCA_DISPATCH(false, false);
#undef DCA_DISPATCH
}

static int zinb_nll_planes_impl(int h2, float pscale, const float* a_mean, const float* a_disp, const float* a_pi, long lda,
                                      const float* theta_w, const float* y, long ldy, const float* sf,
                                      const int* perm, const long long* cursor, int B, int G, float ridge,
                                      float inv_n, int flags, void* d_planes, long ldp, long plane_stride,
                                      long col_mean, long col_disp, long col_pi, float* d_theta, long ldd_theta,
                                      double* loss_partials, int* n_partials_out, void* stream) {
    const bool has_pi = flags & DCAHIP_NLL_HAS_PI, cdisp = flags & DCAHIP_NLL_CONST_DISP;
    if (flags & (DCAHIP_NLL_POISSON | DCAHIP_NLL_MSE)) return DCAHIP_EINVAL;
    if (B <= 0 || G <= 0 || !a_mean || !y || !sf || !loss_partials || !d_planes) return DCAHIP_EINVAL;
    if ((has_pi && !a_pi) || (cdisp ? (!theta_w || !d_theta) : !a_disp)) return DCAHIP_EINVAL;
    const int G4 = (G + 3) & ~3;
    // 16-byte vectors in, 8-byte piece vectors out; 32-bit element offsets inside a plane
    if (lda % 4 || ldy % 4 || lda < G4 || ldy < G4 || !al16(a_mean) || !al16(y) || (has_pi && !al16(a_pi)) ||
        (cdisp ? !al16(theta_w) : !al16(a_disp)))
        return DCAHIP_EINVAL;
    if (ldp % 4 || plane_stride % 4 || !al16(d_planes) || col_mean % 4 || (!cdisp && col_disp % 4) || (has_pi && col_pi % 4) ||
        (long)B * ldp >= (1L << 32) || plane_stride < (long)B * ldp)
        return DCAHIP_EINVAL;
    if (col_mean + G4 > ldp || (!cdisp && col_disp + G4 > ldp) || (has_pi && col_pi + G4 > ldp)) return DCAHIP_EINVAL;
    if (cdisp && (ldd_theta % 4 || ldd_theta < G4 || !al16(d_theta) || (long)B * ldd_theta >= (1L << 32))) return DCAHIP_EINVAL;
    unsigned short* P = static_cast<unsigned short*>(d_planes);
    NllArgs a{a_mean, a_disp, a_pi, theta_w, y, sf, perm, cursor, lda, ldy, cdisp ? ldd_theta : 0,
              nullptr, cdisp ? d_theta : nullptr, nullptr, loss_partials, B, G, ridge, inv_n,
              {P + col_mean, cdisp ? nullptr : P + col_disp, has_pi ? P + col_pi : nullptr}, ldp, plane_stride, pscale};
    const int nvec = (G + 3) / 4;
    const int nseg = (nvec + 255) / 256;
    const int gx = nseg < kMaxPartials ? nseg : kMaxPartials;
    int gy = kMaxPartials / gx;
    if (gy > B) gy = B;
    if (gy < 1) gy = 1;
    const dim3 grid(gx, gy);
    if (n_partials_out) *n_partials_out = gx * gy;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (h2) {
        if (has_pi && cdisp) hipLaunchKernelGGL((zinb_nll_rows_kernel<true, true, 2>), grid, dim3(256), 0, s, a);
        else if (has_pi) hipLaunchKernelGGL((zinb_nll_rows_kernel<true, false, 2>), grid, dim3(256), 0, s, a);
        else if (cdisp) hipLaunchKernelGGL((zinb_nll_rows_kernel<false, true, 2>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((zinb_nll_rows_kernel<false, false, 2>), grid, dim3(256), 0, s, a);
        return (int)hipGetLastError();
    }
#ifdef DCA_ZINB_PATCH
#define DCA_ZK(P, C) zinb_nll_compact_kernel<P, C, true, true>
#else
#define DCA_ZK(P, C) zinb_nll_rows_kernel<P, C, 1>
#endif
    if (has_pi && cdisp) hipLaunchKernelGGL((DCA_ZK(true, true)), grid, dim3(256), 0, s, a);
    else if (has_pi) hipLaunchKernelGGL((DCA_ZK(true, false)), grid, dim3(256), 0, s, a);
    else if (cdisp) hipLaunchKernelGGL((DCA_ZK(false, true)), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((DCA_ZK(false, false)), grid, dim3(256), 0, s, a);
#undef DCA_ZK
    return (int)hipGetLastError();
}

extern "C" int dcahip_zinb_nll_planes(const float* a_mean, const float* a_disp, const float* a_pi, long lda,
                                      const float* theta_w, const float* y, long ldy, const float* sf,
                                      const int* perm, const long long* cursor, int B, int G, float ridge,
                                      float inv_n, int flags, void* d_planes, long ldp, long plane_stride,
                                      long col_mean, long col_disp, long col_pi, float* d_theta, long ldd_theta,
                                      double* loss_partials, int* n_partials_out, void* stream) {
    return zinb_nll_planes_impl(0, 0.f, a_mean, a_disp, a_pi, lda, theta_w, y, ldy, sf, perm, cursor, B, G, ridge, inv_n, flags,
                                d_planes, ldp, plane_stride, col_mean, col_disp, col_pi, d_theta, ldd_theta, loss_partials,
                                n_partials_out, stream);
}

extern "C" int dcahip_zinb_nll_planes_h2(const float* a_mean, const float* a_disp, const float* a_pi, long lda,
                                         const float* theta_w, const float* y, long ldy, const float* sf,
                                         const int* perm, const long long* cursor, int B, int G, float ridge,
                                         float inv_n, int flags, int d_exp, void* d_planes, long ldp, long plane_stride,
                                         long col_mean, long col_disp, long col_pi, float* d_theta, long ldd_theta,
                                         double* loss_partials, int* n_partials_out, void* stream) {
    if (d_exp < -40 || d_exp > 40 || !(ridge >= 0.f)) return DCAHIP_EINVAL;
    return zinb_nll_planes_impl(1, ldexpf(1.f, d_exp), a_mean, a_disp, a_pi, lda, theta_w, y, ldy, sf, perm, cursor, B, G, ridge, inv_n,
                                flags, d_planes, ldp, plane_stride, col_mean, col_disp, col_pi, d_theta, ldd_theta, loss_partials,
                                n_partials_out, stream);
}

extern "C" int dcahip_loss_finalize(const double* partials, int n_partials, double scale,
                                    float* loss_out, void* stream) {
    if (!partials || n_partials <= 0 || !loss_out) return DCAHIP_EINVAL;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream),
                       partials, n_partials, scale, loss_out);
    return (int)hipGetLastError();
}

extern "C" int dcahip_step_end(const float* loss, double weight, float* hist, int rows_per_slot,
                               double* acc, long long* cursor, int advance, void* stream) {
    hipLaunchKernelGGL(step_end_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream),
                       loss, weight, hist, rows_per_slot, acc, cursor, advance);
    return (int)hipGetLastError();
}

extern "C" int dcahip_zinb_heads_infer(const float* a_mean, const float* a_disp, const float* a_pi,
                                       long lda, const float* sf, int B, int G, float* mean_sf,
                                       float* theta, float* pi, long ldo, int flags, void* stream) {
    if (B <= 0 || G <= 0 || !sf) return DCAHIP_EINVAL;
    if ((mean_sf && !a_mean) || (theta && !a_disp) || (pi && !a_pi)) return DCAHIP_EINVAL;
    const int Gp = (G + 3) & ~3;
    bool vec = (lda % 4 == 0) && (ldo % 4 == 0) && lda >= Gp && ldo >= Gp;
    const void* ps[6] = {a_mean, a_disp, a_pi, mean_sf, theta, pi};
    for (const void* p : ps) vec = vec && (p == nullptr || al16(p));
    InferArgs a{a_mean, a_disp, a_pi, sf, mean_sf, theta, pi, lda, ldo, B, G, (flags & DCAHIP_NLL_MSE) ? 1 : 0};
    const int V = vec ? 4 : 1;
    const long total = (long)B * (((G + V - 1) / V + 255) / 256);
    const int grid = (int)(total < 4096 ? total : 4096);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vec) hipLaunchKernelGGL(heads_infer_kernel<4>, dim3(grid), dim3(256), 0, s, a);
    else     hipLaunchKernelGGL(heads_infer_kernel<1>, dim3(grid), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}
