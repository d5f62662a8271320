// K-SPARSE: compact count storage and the first Dense layer on the non-zero counts only.  gfx950, wave64.
//
// The network input of the reference is X = scale(log1p(normalize_per_cell(counts))) (dca/io.py:88-111): a DENSE
// fp32 matrix made from counts that are ~93 % zeros, consumed by the first Dense layer (dca/network.py:124-126) and
// -- through TensorFlow's autodiff -- by its weight gradient.  With
//       x[c, g] = (L[c, g] - mean[g]) / std[g],     L[c, g] = log1p(y[c, g] / fac[c])   (L = 0 where y = 0)
// both products only need the non-zero counts:
//       Z0 = X W0 + b0        =  sum_{y > 0} L[c, g] (W0[g, :] / std[g])  +  (b0 - sum_g (mean[g] / std[g]) W0[g, :])
//       dW0 = X^T dZ0         =  (sum_{c: y > 0} L[c, g] dZ0[c, :]  -  mean[g] colsum(dZ0)) / std[g]
// (each normalisation step optional: no size factors -> fac = 1, no log, no scaling -> mean = 0, std = 1).
//
// Compact counts: one byte per count (0 .. 254 as is, 255 = escape -> the value sits in a per-row overflow list
// sorted by column), rows padded to a multiple of 16 bytes: 1.37 GB instead of 5.49 GB at 68 579 x 20 000, one
// 16-byte load per lane fetches 16 genes.  K-HEADS reads its counts from the same store.
//
// Kernels
//   counts_compact      fp32 counts -> bytes (+ the number of values that are not counts / that need the escape)
//   enc0_dw             weight gradient: a WAVE owns 16 genes x a range of batch rows.  Per 256 rows: the counts of the
//                       strip are read once (16 B per lane and row), per gene the non-zero rows are compacted into an LDS
//                       queue (ballot + mbcnt), converted 64 at a time (escape, / fac, log1p) and accumulated by groups of
//                       H1/4 lanes -- one queue entry per group and iteration, four columns per lane, dZ0 rows from L2.
//                       Fixed order everywhere: deterministic.  One partial per row split; enc0_dw_finish adds them and
//                       applies mean / std and writes the bias gradient row.
//   enc0_c0             b_eff = b0 - sum_g (mean / std) W0[g, :] (fp64 partials, finished by the last-arriving workgroup)
//   enc0_fwd            forward: a wave owns one batch row; 1 KB of counts per load, byte positions compacted into the
//                       queue, entries converted 64 at a time (x = L / std[g]) and accumulated against gathered W0 rows.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "dcahip.h"

namespace {

struct Compact {
    const unsigned char* yc; long ldc;
    const int* optr; const int* ocol; const float* oval;
};

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// the count behind an escape byte (rare: counts >= 255)
__device__ __forceinline__ float escaped_count(const Compact& c, long srow, int col) {
    float v = 255.f;
    if (c.optr) {
        for (int i = c.optr[srow], e = c.optr[srow + 1]; i < e; ++i)
            if (c.ocol[i] == col) { v = c.oval[i]; break; }
    }
    return v;
}

__device__ __forceinline__ int mbcnt64(unsigned long long m) {
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
}

// ------------------------------------------------------------------------------------------------- compact
__global__ __launch_bounds__(256) void counts_compact_kernel(const float* Y, long ldy, int n, int G, unsigned char* Yc,
                                                             long ldc, int* status) {
    const long nq = ldc >> 4;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)n * nq) return;
    const long r = idx / nq;
    const int g0 = (int)(idx - r * nq) * 16;
    const float* src = Y + r * ldy + g0;
    unsigned w[4] = {0u, 0u, 0u, 0u};
    int bad = 0, esc = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (g0 + 4 * q + 3 < ldy && ((ldy & 3) == 0) && ((reinterpret_cast<uintptr_t>(Y) & 15) == 0)) {
            const float4 t = *reinterpret_cast<const float4*>(src + 4 * q);
            v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) if (g0 + 4 * q + j < ldy) v[j] = src[4 * q + j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float x = (g0 + 4 * q + j < G) ? v[j] : 0.f;
            if (!(x >= 0.f) || x != floorf(x) || x > 16777216.f) { ++bad; x = 0.f; }
            unsigned code = x >= 255.f ? 255u : (unsigned)x;
            esc += code == 255u;
            w[q] |= code << (8 * j);
        }
    }
    *reinterpret_cast<uint4*>(Yc + r * ldc + g0) = make_uint4(w[0], w[1], w[2], w[3]);
    if (bad) atomicAdd(status, bad);
    if (esc) atomicAdd(status + 1, esc);
}

// ------------------------------------------------------------------------------------------------- weight gradient
// lut [n, 8]: lut[r][k] = f(k / fac[r]) for the counts k = 0 .. 7 that make up ~99 % of the non-zero entries (f = log1p
// or the identity): one table row per cell, made once per dataset; larger counts (and escapes) take the formula.
constexpr int kLut = 8;

__global__ __launch_bounds__(256) void enc0_lut_kernel(const float* fac, int do_log, int n, float* lut) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)n * kLut) return;
    const int r = (int)(idx / kLut), k = (int)(idx - (long)r * kLut);
    float x = (float)k;
    if (fac) x = __fdiv_rn(x, fac[r]);
    if (do_log) x = log1pf(x);
    lut[idx] = x;
}

struct DwArgs {
    Compact c;
    const float* fac; int do_log;
    const float* lut;               // [n, 8]
    const int* perm; const long long* cursor; long row_base;
    int B, G;
    const float* dZ; long ldz;
    float* P; long Gs;              // [NS][Gs][H1] partial sums over the non-zero counts
    float* Sp;                      // [NS][H1] partial column sums of dZ
    int RS;                         // batch rows per split (multiple of the row block)
};

// rows of dZ a workgroup keeps in LDS at a time: 64 KB of them
constexpr int dw_row_block(int H1) { return H1 <= 64 ? 256 : (H1 == 128 ? 128 : 64); }
constexpr int kDwWaves = 16;        // waves per workgroup: 256 genes

template <int H1>
__global__ __launch_bounds__(64 * kDwWaves) void enc0_dw_kernel(DwArgs a) {
    constexpr int LPE = H1 / 4;             // lanes per queue entry (4 columns each)
    constexpr int EPI = 64 / LPE;           // entries per iteration
    constexpr int RB = dw_row_block(H1);
    constexpr int NCH = RB / 64;
    constexpr int NW = kDwWaves;
    constexpr int LS = kLut + 1;            // odd row stride of the table: rows x codes spread over the banks
    static_assert(LPE >= 1 && LPE <= 64 && (64 % LPE) == 0, "first-layer width");
    __shared__ __attribute__((aligned(16))) float dzs[RB * H1];
    __shared__ float luts[RB * LS];
    __shared__ float rfac[RB];
    __shared__ int srows[RB];
    __shared__ uint2 qe[NW][RB + 2 * EPI];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int split = blockIdx.y;
    const int rb = split * a.RS;
    const int re = min(a.B, rb + a.RS);
    const long long cur = (a.cursor ? *a.cursor : 0) + a.row_base;
    const int g0 = (blockIdx.x * NW + wave) * 16;
    const bool wave_on = g0 < a.G;

    uint2* const QE = qe[wave];
    const int grp = lane / LPE, lidx = lane - grp * LPE;
    const char* const dzl = reinterpret_cast<const char*>(dzs) + lidx * 16;

    float4 acc[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) acc[b] = make_float4(0.f, 0.f, 0.f, 0.f);
    float colsum = 0.f;                     // workgroup 0 of the split: thread j < H1 sums column j of dZ

#pragma unroll 1
    for (int rg0 = rb; rg0 < re; rg0 += RB) {
        const int nrow = min(RB, re - rg0);
        __syncthreads();                    // everyone is done with the previous block
        // ---- the block's rows of dZ (zero beyond the batch), table rows, divisors, storage rows
        for (int i = tid; i < RB * (H1 / 4); i += 64 * NW) {
            const int r = i / (H1 / 4), c4 = i - r * (H1 / 4);
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (r < nrow) v = *reinterpret_cast<const float4*>(a.dZ + (long)(rg0 + r) * a.ldz + 4 * c4);
            *reinterpret_cast<float4*>(dzs + r * H1 + 4 * c4) = v;
        }
        for (int i = tid; i < RB; i += 64 * NW) {
            const int rc = i < nrow ? i : nrow - 1;
            const long sr = a.perm ? (long)a.perm[cur + rg0 + rc] : cur + rg0 + rc;
            srows[i] = (int)sr;
            rfac[i] = a.fac ? a.fac[sr] : 1.f;
#pragma unroll
            for (int k = 0; k < kLut; ++k) luts[i * LS + k] = a.lut[sr * kLut + k];
        }
        // ---- this wave's counts: row i * 64 + lane of the block, 16 genes
        uint4 codes[NCH];
#pragma unroll
        for (int i = 0; i < NCH; ++i) {
            const int cl = i * 64 + lane;
            const int clc = cl < nrow ? cl : nrow - 1;
            const long sr = a.perm ? (long)a.perm[cur + rg0 + clc] : cur + rg0 + clc;
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (wave_on && cl < nrow) v = *reinterpret_cast<const uint4*>(a.c.yc + sr * a.c.ldc + g0);
            codes[i] = v;
        }
        __syncthreads();
        if (blockIdx.x == 0 && tid < H1)
            for (int r = 0; r < nrow; ++r) colsum += dzs[r * H1 + tid];
        if (!wave_on) continue;
#pragma unroll
        for (int b = 0; b < 16; ++b) {
            const int d = b >> 2, sh = 8 * (b & 3);
            int qn = 0;
#pragma unroll
            for (int i = 0; i < NCH; ++i) {
                const unsigned dw = d == 0 ? codes[i].x : d == 1 ? codes[i].y : d == 2 ? codes[i].z : codes[i].w;
                const unsigned code = (dw >> sh) & 255u;
                const int cl = i * 64 + lane;
                const bool nz = code != 0u;
                float x = luts[cl * LS + (code < kLut ? code : 0u)];
                if (code >= kLut) {                        // rare: a large count (or an escape): the formula itself
                    float val = (float)code;
                    if (code == 255u) val = escaped_count(a.c, srows[cl], g0 + b);
                    x = a.fac ? __fdiv_rn(val, rfac[cl]) : val;
                    if (a.do_log) x = log1pf(x);
                }
                const unsigned long long m = __ballot(nz);
                if (nz) QE[qn + mbcnt64(m)] = make_uint2(__float_as_uint(x), (unsigned)(cl * H1 * 4));
                qn += __popcll(m);
            }
            if (qn == 0) continue;
            // harmless entries up to the next multiple of two iterations
            const int nit = ((qn + 2 * EPI - 1) / (2 * EPI)) * 2;
            if (lane < nit * EPI - qn) QE[qn + lane] = make_uint2(0u, 0u);
            wave_sync();
            float4 ac = acc[b];
            for (int it = 0; it < nit; it += 2) {
                const uint2 e0 = QE[it * EPI + grp], e1 = QE[(it + 1) * EPI + grp];
                const float4 d0 = *reinterpret_cast<const float4*>(dzl + e0.y);
                const float4 d1 = *reinterpret_cast<const float4*>(dzl + e1.y);
                const float x0 = __uint_as_float(e0.x), x1 = __uint_as_float(e1.x);
                ac.x = fmaf(x0, d0.x, ac.x); ac.y = fmaf(x0, d0.y, ac.y); ac.z = fmaf(x0, d0.z, ac.z); ac.w = fmaf(x0, d0.w, ac.w);
                ac.x = fmaf(x1, d1.x, ac.x); ac.y = fmaf(x1, d1.y, ac.y); ac.z = fmaf(x1, d1.z, ac.z); ac.w = fmaf(x1, d1.w, ac.w);
            }
            acc[b] = ac;
            wave_sync();                                   // the queue is free for the next gene
        }
    }
    if (blockIdx.x == 0 && tid < H1) a.Sp[(long)split * H1 + tid] = colsum;
    if (!wave_on) return;
    // ---- the EPI groups hold partial sums over different entries: add them (fixed order), one row of P per gene
#pragma unroll
    for (int b = 0; b < 16; ++b) {
        float4 v = acc[b];
#pragma unroll
        for (int off = LPE; off < 64; off <<= 1) {
            v.x += __shfl_xor(v.x, off, 64); v.y += __shfl_xor(v.y, off, 64);
            v.z += __shfl_xor(v.z, off, 64); v.w += __shfl_xor(v.w, off, 64);
        }
        if (lane < LPE)
            *reinterpret_cast<float4*>(a.P + ((long)split * a.Gs + g0 + b) * H1 + 4 * lane) = v;
    }
}

struct DwFinishArgs {
    const float* P; long Gs; const float* Sp; int NS;
    const float* mean; const float* stdv;
    int G, H1;
    float* gW; long ldg;            // [G + 1, ldg]: row G = bias gradient
};

__global__ __launch_bounds__(256) void enc0_dw_finish_kernel(DwFinishArgs a) {
    const int hq = a.H1 >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)(a.G + 1) * hq) return;
    const int g = (int)(idx / hq), j = (int)(idx - (long)g * hq) * 4;
    float4 s = *reinterpret_cast<const float4*>(a.Sp + j);
    for (int k = 1; k < a.NS; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(a.Sp + (long)k * a.H1 + j);
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    float4 v;
    if (g == a.G) {
        v = s;
    } else {
        v = *reinterpret_cast<const float4*>(a.P + (long)g * a.H1 + j);
        for (int k = 1; k < a.NS; ++k) {
            const float4 t = *reinterpret_cast<const float4*>(a.P + ((long)k * a.Gs + g) * a.H1 + j);
            v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
        }
        if (a.mean) {
            const float m = a.mean[g];
            v.x -= m * s.x; v.y -= m * s.y; v.z -= m * s.z; v.w -= m * s.w;
        }
        if (a.stdv) {
            const float sd = a.stdv[g];
            v.x = __fdiv_rn(v.x, sd); v.y = __fdiv_rn(v.y, sd); v.z = __fdiv_rn(v.z, sd); v.w = __fdiv_rn(v.w, sd);
        }
    }
    float* dst = a.gW + (long)g * a.ldg + j;
    if ((a.ldg & 3) == 0 && (reinterpret_cast<uintptr_t>(a.gW) & 15) == 0) {
        *reinterpret_cast<float4*>(dst) = v;
    } else {
        dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
}

// ------------------------------------------------------------------------------------------------- forward
constexpr int kC0Blocks = 128;

struct C0Args {
    const float* W; long ldw; const float* bias; const float* mean; const float* stdv;
    int G, H1;
    double* part;                   // [kC0Blocks][H1]
    unsigned* ticket;
    float* beff;                    // [H1]
};

// b_eff[j] = bias[j] - sum_g (mean[g] / std[g]) W[g, j]; the last-arriving workgroup adds the partials in order
__global__ __launch_bounds__(256) void enc0_c0_kernel(C0Args a) {
    __shared__ double red[256];
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const int col = tid % a.H1, ph = tid / a.H1, nph = 256 / a.H1;
    const int per = (a.G + kC0Blocks - 1) / kC0Blocks;
    const int gb0 = blockIdx.x * per, gb1 = min(a.G, gb0 + per);
    double s = 0.0;
    for (int g = gb0 + ph; g < gb1; g += nph) {
        const float t = a.stdv ? __fdiv_rn(a.mean[g], a.stdv[g]) : a.mean[g];
        s += (double)t * (double)a.W[(long)g * a.ldw + col];
    }
    red[tid] = s;
    __syncthreads();
    if (tid < a.H1) {
        double v = 0.0;
        for (int k = 0; k < nph; ++k) v += red[k * a.H1 + tid];
        a.part[(long)blockIdx.x * a.H1 + tid] = v;
    }
    __syncthreads();
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (!is_last) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (tid < a.H1) {
        double v = 0.0;
        for (unsigned k = 0; k < gridDim.x; ++k)
            v += __hip_atomic_load(a.part + (long)k * a.H1 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a.beff[tid] = (a.bias ? a.bias[tid] : 0.f) - (float)v;
    }
    if (tid == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct FwArgs {
    Compact c;
    const float* fac; int do_log;
    const float* stdv;
    const int* perm; const long long* cursor; long row_base;
    int B, G;
    const float* W; long ldw;
    const float* beff;
    float* Z; long ldz;
};

constexpr int kFwCap = 320;

template <int H1>
__global__ __launch_bounds__(256) void enc0_fwd_kernel(FwArgs a) {
    constexpr int LPE = H1 / 4, EPI = 64 / LPE;
    __shared__ unsigned q[4][kFwCap];
    __shared__ uint2 qe[4][64];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = blockIdx.x * 4 + wave;
    if (c >= a.B) return;
    const long long cur = (a.cursor ? *a.cursor : 0) + a.row_base;
    const long sr = a.perm ? (long)a.perm[cur + c] : cur + c;
    const float facr = a.fac ? a.fac[sr] : 1.f;
    unsigned* const Q = q[wave];
    uint2* const QE = qe[wave];
    const int grp = lane / LPE, lidx = lane - grp * LPE;
    const char* const wb = reinterpret_cast<const char*>(a.W) + lidx * 16;
    const unsigned ldwb = (unsigned)(a.ldw * 4);
    const unsigned char* const yrow = a.c.yc + sr * a.c.ldc;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int qn = 0;

    auto flush = [&](bool all) {
        wave_sync();
        while (qn >= 64 || (all && qn > 0)) {
            const int n = min(64, qn);
            const unsigned e = lane < n ? Q[qn - n + lane] : 0u;
            const unsigned code = e & 255u;
            const int gene = (int)(e >> 8);
            float val = (float)code;
            if (code == 255u) val = escaped_count(a.c, sr, gene);
            float x = a.fac ? __fdiv_rn(val, facr) : val;
            if (a.do_log) x = log1pf(x);
            if (a.stdv) x = __fdiv_rn(x, a.stdv[gene]);
            QE[lane] = make_uint2(__float_as_uint(x), (unsigned)gene * ldwb);
            wave_sync();
            // every row of W the batch needs is requested before the first product (lanes beyond n hold x = 0, row 0):
            // the gathers come from L2 and their latency, not their number, is what a wave waits for
            constexpr int NIT = 64 / EPI, HALF = NIT > 8 ? 8 : NIT;
            const int nit = (n + EPI - 1) / EPI;
#pragma unroll 1
            for (int h0 = 0; h0 < nit; h0 += HALF) {
                uint2 en[HALF];
                float4 w[HALF];
#pragma unroll
                for (int it = 0; it < HALF; ++it) en[it] = QE[(h0 + it) * EPI + grp];
#pragma unroll
                for (int it = 0; it < HALF; ++it) w[it] = *reinterpret_cast<const float4*>(wb + en[it].y);
#pragma unroll
                for (int it = 0; it < HALF; ++it) {
                    const float xv = __uint_as_float(en[it].x);
                    acc.x = fmaf(xv, w[it].x, acc.x); acc.y = fmaf(xv, w[it].y, acc.y);
                    acc.z = fmaf(xv, w[it].z, acc.z); acc.w = fmaf(xv, w[it].w, acc.w);
                }
            }
            qn -= n;
            wave_sync();
        }
    };

    const int nch = (int)((a.c.ldc + 1023) >> 10);
    uint4 nxt = make_uint4(0u, 0u, 0u, 0u);
    if (lane * 16 < a.c.ldc) nxt = *reinterpret_cast<const uint4*>(yrow + lane * 16);
#pragma unroll 1
    for (int ch = 0; ch < nch; ++ch) {
        const uint4 codes = nxt;
        const int gbase = ch * 1024 + lane * 16;
        nxt = make_uint4(0u, 0u, 0u, 0u);
        if (ch + 1 < nch && gbase + 1024 < a.c.ldc) nxt = *reinterpret_cast<const uint4*>(yrow + gbase + 1024);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const unsigned dw = d == 0 ? codes.x : d == 1 ? codes.y : d == 2 ? codes.z : codes.w;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned code = (dw >> (8 * k)) & 255u;
                const bool nz = code != 0u;
                const unsigned long long m = __ballot(nz);
                if (nz) Q[qn + mbcnt64(m)] = code | ((unsigned)(gbase + 4 * d + k) << 8);
                qn += __popcll(m);
            }
            if (qn > kFwCap - 256) flush(false);
        }
    }
    flush(true);
#pragma unroll
    for (int off = LPE; off < 64; off <<= 1) {
        acc.x += __shfl_xor(acc.x, off, 64); acc.y += __shfl_xor(acc.y, off, 64);
        acc.z += __shfl_xor(acc.z, off, 64); acc.w += __shfl_xor(acc.w, off, 64);
    }
    if (lane < LPE) {
        const float4 b = a.beff ? *reinterpret_cast<const float4*>(a.beff + 4 * lane) : make_float4(0.f, 0.f, 0.f, 0.f);
        float* dst = a.Z + (long)c * a.ldz + 4 * lane;
        *reinterpret_cast<float4*>(dst) = make_float4(acc.x + b.x, acc.y + b.y, acc.z + b.z, acc.w + b.w);
    }
}

inline bool width_ok(int H1) { return H1 == 16 || H1 == 32 || H1 == 64 || H1 == 128 || H1 == 256; }

// row splits: as many as keep the grid at (or just below) one workgroup per CU -- every workgroup resident, one round
inline int dw_splits(int B, int G, int H1) {
    const int RB = dw_row_block(H1);
    const int groups = (G + 16 * kDwWaves - 1) / (16 * kDwWaves);
    int ns = 256 / groups;
    const int maxs = (B + RB - 1) / RB;
    if (ns > maxs) ns = maxs;
    if (ns > 16) ns = 16;
    if (ns < 1) ns = 1;
    return ns;
}
inline int dw_rows_per_split(int B, int ns, int H1) {
    const int RB = dw_row_block(H1);
    return (((B + ns - 1) / ns) + RB - 1) / RB * RB;
}

}  // namespace

extern "C" long dcahip_counts_compact_ld(int G) { return ((long)G + 15) / 16 * 16; }

extern "C" int dcahip_counts_compact(const float* Y, long ldy, int n, int G, unsigned char* Yc, long ldc, int* status,
                                     void* stream) {
    if (n < 0 || G <= 0 || ldc < G || (ldc & 15) || ldy < G || !Y || !Yc || !status) return DCAHIP_EINVAL;
    if (n == 0) return 0;
    const long total = (long)n * (ldc >> 4);
    hipLaunchKernelGGL(counts_compact_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       Y, ldy, n, G, Yc, ldc, status);
    return (int)hipGetLastError();
}

extern "C" int dcahip_enc0_sparse_supported(int H1) { return width_ok(H1) ? 1 : 0; }

extern "C" long dcahip_enc0_dw_sparse_workspace_bytes(int B, int G, int H1) {
    if (!width_ok(H1) || B <= 0 || G <= 0) return 0;
    const int ns = dw_splits(B, G, H1);
    const long Gs = ((long)G + 255) / 256 * 256;
    return ((long)ns * Gs * H1 + (long)ns * H1) * 4;
}

extern "C" int dcahip_enc0_lut(const float* fac, int do_log, int n, float* lut, void* stream) {
    if (n <= 0 || !lut) return DCAHIP_EINVAL;
    hipLaunchKernelGGL(enc0_lut_kernel, dim3((unsigned)(((long)n * kLut + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       fac, do_log, n, lut);
    return (int)hipGetLastError();
}

extern "C" int dcahip_enc0_dw_sparse(const unsigned char* Yc, long ldc, const int* ovf_ptr, const int* ovf_col,
                                     const float* ovf_val, const float* fac, int do_log, const float* lut, const float* mean,
                                     const float* stdv, const int* perm, const long long* cursor, long row_base,
                                     int B, int G, int H1, const float* dZ, long ldz, float* gW, long ldg,
                                     void* workspace, long workspace_bytes, void* stream) {
    if (!width_ok(H1) || B <= 0 || G <= 0 || !Yc || !lut || (ldc & 15) || ldc < G || !dZ || (ldz & 3) || ldz < H1 || !gW ||
        ldg < H1 || (reinterpret_cast<uintptr_t>(dZ) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15) || !workspace)
        return DCAHIP_EINVAL;
    if (workspace_bytes < dcahip_enc0_dw_sparse_workspace_bytes(B, G, H1)) return DCAHIP_EINVAL;
    const int ns = dw_splits(B, G, H1);
    const long Gs = ((long)G + 255) / 256 * 256;
    DwArgs a;
    a.c = Compact{Yc, ldc, ovf_ptr, ovf_col, ovf_val};
    a.fac = fac; a.do_log = do_log; a.lut = lut; a.perm = perm; a.cursor = cursor; a.row_base = row_base;
    a.B = B; a.G = G; a.dZ = dZ; a.ldz = ldz;
    a.P = static_cast<float*>(workspace); a.Gs = Gs; a.Sp = a.P + (long)ns * Gs * H1;
    a.RS = dw_rows_per_split(B, ns, H1);
    const dim3 grid((unsigned)((G + 16 * kDwWaves - 1) / (16 * kDwWaves)), (unsigned)ns);
    const dim3 block(64 * kDwWaves);
    hipStream_t s = (hipStream_t)stream;
    switch (H1) {
        case 16: hipLaunchKernelGGL(enc0_dw_kernel<16>, grid, block, 0, s, a); break;
        case 32: hipLaunchKernelGGL(enc0_dw_kernel<32>, grid, block, 0, s, a); break;
        case 64: hipLaunchKernelGGL(enc0_dw_kernel<64>, grid, block, 0, s, a); break;
        case 128: hipLaunchKernelGGL(enc0_dw_kernel<128>, grid, block, 0, s, a); break;
        default: hipLaunchKernelGGL(enc0_dw_kernel<256>, grid, block, 0, s, a); break;
    }
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    DwFinishArgs f{a.P, Gs, a.Sp, ns, mean, stdv, G, H1, gW, ldg};
    const long total = (long)(G + 1) * (H1 / 4);
    hipLaunchKernelGGL(enc0_dw_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, f);
    return (int)hipGetLastError();
}

extern "C" long dcahip_enc0_fwd_sparse_workspace_bytes(int H1) {
    return width_ok(H1) ? (long)kC0Blocks * H1 * 8 + 256 + (long)H1 * 4 : 0;
}

extern "C" int dcahip_enc0_fwd_sparse(const unsigned char* Yc, long ldc, const int* ovf_ptr, const int* ovf_col,
                                      const float* ovf_val, const float* fac, int do_log, const float* mean,
                                      const float* stdv, const int* perm, const long long* cursor, long row_base,
                                      int B, int G, int H1, const float* W, long ldw, const float* bias,
                                      float* Z, long ldz, void* workspace, long workspace_bytes, void* stream) {
    if (!width_ok(H1) || B <= 0 || G <= 0 || !Yc || (ldc & 15) || ldc < G || !W || (ldw & 3) || ldw < H1 || !Z ||
        (ldz & 3) || ldz < H1 || (reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(Z) & 15) ||
        (reinterpret_cast<uintptr_t>(workspace) & 15) || !workspace)
        return DCAHIP_EINVAL;
    if (workspace_bytes < dcahip_enc0_fwd_sparse_workspace_bytes(H1)) return DCAHIP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    // workspace: [kC0Blocks][H1] doubles | ticket (zero before the first call, left at zero by every call) | b_eff [H1]
    double* part = static_cast<double*>(workspace);
    unsigned* ticket = reinterpret_cast<unsigned*>(part + (long)kC0Blocks * H1);
    float* beff = reinterpret_cast<float*>(reinterpret_cast<char*>(ticket) + 256);
    C0Args c{W, ldw, bias, mean, stdv, G, H1, part, ticket, beff};
    if (mean) {
        hipLaunchKernelGGL(enc0_c0_kernel, dim3(kC0Blocks), dim3(256), 0, s, c);
        int rc = (int)hipGetLastError();
        if (rc) return rc;
    }
    FwArgs f;
    f.c = Compact{Yc, ldc, ovf_ptr, ovf_col, ovf_val};
    f.fac = fac; f.do_log = do_log; f.stdv = stdv; f.perm = perm; f.cursor = cursor; f.row_base = row_base;
    f.B = B; f.G = G; f.W = W; f.ldw = ldw; f.beff = mean ? beff : bias; f.Z = Z; f.ldz = ldz;
    const dim3 grid((unsigned)((B + 3) / 4));
    switch (H1) {
        case 16: hipLaunchKernelGGL(enc0_fwd_kernel<16>, grid, dim3(256), 0, s, f); break;
        case 32: hipLaunchKernelGGL(enc0_fwd_kernel<32>, grid, dim3(256), 0, s, f); break;
        case 64: hipLaunchKernelGGL(enc0_fwd_kernel<64>, grid, dim3(256), 0, s, f); break;
        case 128: hipLaunchKernelGGL(enc0_fwd_kernel<128>, grid, dim3(256), 0, s, f); break;
        default: hipLaunchKernelGGL(enc0_fwd_kernel<256>, grid, dim3(256), 0, s, f); break;
    }
    return (int)hipGetLastError();
}
