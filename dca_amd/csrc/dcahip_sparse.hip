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
//   enc0_lut            per cell: f(k / fac) for the counts k = 0 .. 63 as three bf16 pieces -- both products below LOOK
//                       their operand up instead of dividing, taking logarithms and splitting
//   enc0_split_dz, enc0_dw, enc0_dw_finish
//                       weight gradient on the matrix pipe: a wave owns 32 genes x all H1 columns, K = the batch rows; the
//                       counts of a 64-row block go through an LDS tile (a lane needs 8 rows of ONE gene), the rows'
//                       tables and the pre-split dZ are staged with them, double buffered.  One partial per row split;
//                       the finish adds them, applies mean / std and writes the bias-gradient row.  Deterministic.
//   enc0_wsplit, enc0_fwd_lut, enc0_fwd_reduce
//                       forward on the matrix pipe (batches from 1 024 rows, 32 / 64 units): a wave owns 32 batch rows, one
//                       per lane -- each lane reads ITS row's bytes straight from memory, looks them up in its row's table
//                       (LDS, resident for the whole kernel); W0 / std pre-split into MFMA-ordered tiles shared by the
//                       workgroup; K = the genes in 16 chunks whose partials are added in order.  Deterministic.
//   enc0_c0             b_eff = b0 - sum_g (mean / std) W0[g, :] (fp64 partials, finished by the last-arriving workgroup)
//   enc0_fwd            forward over the non-zero counts only (vector pipe): a wave owns one batch row; 1 KB of counts per
//                       load, byte positions compacted into a queue, entries converted 64 at a time (x = L / std[g]) and
//                       accumulated against gathered W0 rows.  Slower than the dense GEMM at every batch size; kept for
//                       the widths the matrix-pipe forward does not take, off by default.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <type_traits>
#include <utility>
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
// dW0 = X^T dZ0 on the matrix pipe, X built on the fly from the byte store:
//     sum_c x[c, g] dZ[c, :] = (sum_c L[c, g] dZ[c, :] - mean[g] colsum(dZ)) / std[g],   L = f(y / fac[c])
// A wave owns a 32-gene tile and all H1 columns; K = the batch rows, 16 per step.  The A operand (32 genes x 16 rows
// of L as three bf16 pieces) is LOOKED UP, not computed: 8 byte loads of counts per lane and step, each the index
// into the cell's table of pre-split values lutp[cell][count] (the first 64 of its 128 entries; larger ones take the formula -- a
// wave-uniform branch that is rare on count data; an escape byte looks its count up in the row's overflow list there).
// The B operand (dZ as three bf16 pieces, laid out for the MFMA by enc0_split_dz once per call) is shared by the
// workgroup's 8 waves through LDS, 64 rows at a time.
// Six bf16 products per fp32 product, fp32 accumulation (the arithmetic of K-HEADS / K-GEMM); fixed order.
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2v = __attribute__((ext_vector_type(2))) __bf16;
using f32x2v = __attribute__((ext_vector_type(2))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)
// the six products of one K = 16 step, small terms first (index 0 = leading piece)
#define MFMA_X3(A, Bf, ACC) { ACC = MFMA16(A[2], Bf[0], ACC); ACC = MFMA16(A[1], Bf[1], ACC); ACC = MFMA16(A[0], Bf[2], ACC); \
                              ACC = MFMA16(A[1], Bf[0], ACC); ACC = MFMA16(A[0], Bf[1], ACC); ACC = MFMA16(A[0], Bf[0], ACC); }

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {           // v_cvt_pk_bf16_f32: a -> low half
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, bf16x2v));
}
// x = p0 + p1 + p2 (bf16 each, round-to-nearest residuals), two values per call
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(p0 << 16), r1 = x1 - __uint_as_float(p0 & 0xffff0000u);
    p1 = pk_bf16(r0, r1);
    const float q0 = r0 - __uint_as_float(p1 << 16), q1 = r1 - __uint_as_float(p1 & 0xffff0000u);
    p2 = pk_bf16(q0, q1);
}
// one value -> table entry {p0 | p1 << 16, p2}
// log1p for counts BEYOND the per-cell table (x >= kLut / size factor), in the table kernels' formula paths: on the
// transcendental unit (v_log_f32, 1 ulp) with Kahan's exact-ratio correction of the rounded 1 + x -- ~2e-7 relative.  The
// library's log1pf (the table's own formula) costs such a path ~1100 more cycles per visit, and a whole workgroup waits at its
// barrier for the wave that is in it (profiles/r05q_*).
__device__ __forceinline__ float log1p_beyond_table(float x) {
    const float u = 1.f + x, d1 = u - 1.f;
    const float lg = __builtin_amdgcn_logf(u) * 0.69314718055994531f;
    return d1 == 0.f ? x : lg * (x * __builtin_amdgcn_rcpf(d1));
}
__device__ __forceinline__ uint2 split_entry(float x) {
    unsigned a, b, c;
    split_pair(x, 0.f, a, b, c);
    return make_uint2((a & 0xffffu) | (b << 16), c & 0xffffu);
}
__device__ __forceinline__ int rowmap(int e, int hi) { return (e & 3) + 8 * (e >> 2) + 4 * hi; }   // row of accumulator element e

// LDS reads as instructions.  The destination is written when the data arrives, not where the statement stands: EVERY
// destination must appear in a tie ("+v") of an s_waitcnt statement before it dies -- a destination the compiler considers dead
// is handed to another value and overwritten by the late data (seen: the last step's storage-row read landing in `any`).
template <int OFF> __device__ __forceinline__ void lds_read_u8(unsigned& r, unsigned ad) { asm volatile("ds_read_u8 %0, %1 offset:%2" : "=v"(r) : "v"(ad), "n"(OFF)); }
template <int OFF> __device__ __forceinline__ void lds_read_b64(u32x2& r, unsigned ad) { asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r) : "v"(ad), "n"(OFF)); }
template <int OFF> __device__ __forceinline__ void lds_read_b128(u32x4& r, unsigned ad) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(ad), "n"(OFF)); }
template <int... I, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, I...>, F&& f) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, f); }


constexpr int kLut = 128;           // table entries per cell in memory (counts 0 .. 127); the kernels hold the first 32 / 64 / 128 in LDS

__global__ __launch_bounds__(256) void enc0_lut_kernel(const float* fac, int do_log, int n, uint2* lutp) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)n * kLut) return;
    const int r = (int)(idx / kLut), k = (int)(idx - (long)r * kLut);
    float x = (float)k;
    if (fac) x = __fdiv_rn(x, fac[r]);
    if (do_log) x = log1pf(x);
    lutp[idx] = split_entry(x);
}

constexpr int kKS = 16;             // batch rows per K step
constexpr int kDwWaves = 8;         // waves (32-gene tiles) per workgroup: 256 genes
constexpr int kDwRB = 64;           // batch rows a workgroup holds in LDS at a time (4 K steps)
constexpr int kDwMaxRS = 2048;      // batch rows per row split at most (their storage rows sit in LDS)
constexpr int kCodeLd = 272;        // row stride of the count tile in LDS (256 + 16: rows 8 apart sit 32 banks apart)
constexpr int dz_step_elems(int H1) { return 3 * (H1 / 32) * 2 * 32 * 8; }   // bf16 elements of one K step of split dZ

// dZ [B, ldz] -> DZP [ceil(B / 16)][piece][column tile][k half][32 columns][8 rows] bf16 (zero rows beyond B), the
// column sums of each step's 16 rows -> Spp [steps][H1], and the storage row of every batch row -> srowb [B]
template <int H1>
__global__ __launch_bounds__(256) void enc0_split_dz_kernel(const float* dZ, long ldz, int B, const int* perm,
                                                            const long long* cursor, long row_base,
                                                            unsigned short* DZP, float* Spp, int* srowb) {
    constexpr int NTL = H1 / 32;
    __shared__ float part[8][H1];
    const int ks = blockIdx.x, tid = threadIdx.x;
    if (tid < kKS && ks * kKS + tid < B) {
        const long long cur = (cursor ? *cursor : 0) + row_base;
        const int r = ks * kKS + tid;
        srowb[r] = perm ? perm[cur + r] : (int)(cur + r);
    }
    unsigned* out = reinterpret_cast<unsigned*>(DZP + (long)ks * dz_step_elems(H1));
    for (int idx = tid; idx < 8 * H1; idx += 256) {
        const int col = idx % H1, rp = idx / H1;            // row pair rp: rows 8 hi + 2 jp, + 1
        const int hi = rp >> 2, jp = rp & 3;
        const int r0 = ks * kKS + 8 * hi + 2 * jp;
        const float x0 = r0 < B ? dZ[(long)r0 * ldz + col] : 0.f;
        const float x1 = r0 + 1 < B ? dZ[(long)(r0 + 1) * ldz + col] : 0.f;
        unsigned p[3];
        split_pair(x0, x1, p[0], p[1], p[2]);
        const int t = col >> 5, c = col & 31;
#pragma unroll
        for (int q = 0; q < 3; ++q) out[((((q * NTL + t) * 2 + hi) * 32 + c) * 8 + 2 * jp) >> 1] = p[q];
        part[rp][col] = x0 + x1;
    }
    __syncthreads();
    for (int col = tid; col < H1; col += 256) {
        float v = 0.f;
#pragma unroll
        for (int rp = 0; rp < 8; ++rp) v += part[rp][col];
        Spp[(long)ks * H1 + col] = v;
    }
}

struct DwArgs {
    Compact c;
    const float* fac; int do_log;
    const uint2* lutp;              // [n, kLut]
    int B, G;
    const unsigned short* DZP; const float* Spp; const int* srowb;
    float* P; long Gs;              // [NS][Gs][H1] partial sums over the batch rows of a split
    float* Sp;                      // [NS][H1] column sums of dZ per split
    int RS;                         // batch rows per split (multiple of the row block)
};

constexpr int kDwLut = 64;          // table entries per cell held in LDS (with 32, one block in two waits for a wave in the formula path)

// Workgroups are dealt to the eight XCDs round-robin by their linear index, and each XCD has its own L2.  The weight-gradient
// kernels want the workgroups of ONE row split (same table rows, same dZ pieces, all gene groups) behind the same L2: this
// gives workgroup L the (group, split) cell that keeps each XCD on a contiguous range of the split-major order.
__device__ __forceinline__ void xcd_cell(int& bx, int& by) {
#ifdef DCA_EXP_NO_XCD_MAP
    bx = blockIdx.x; by = blockIdx.y;
#else
    const int total = gridDim.x * gridDim.y, L = blockIdx.x + blockIdx.y * gridDim.x;
    const int q = total >> 3, r = total & 7, x = L & 7;
    const int cell = x * q + (x < r ? x : r) + (L >> 3);
    bx = cell % (int)gridDim.x; by = cell / (int)gridDim.x;
#endif
}

// Column sums of one row split's dZ = sum of its K steps' sums, IN ORDER (the bias gradient's summation order), by the H1
// threads of one workgroup per split.  All loads are issued before the first addition (64 at a time): as a loop of dependent
// load -> add round trips this tail took 30 000 cycles and made those workgroups the last to finish.
template <int H1>
__device__ __forceinline__ void split_column_sums(const float* Spp, float* Sp, int split, int ks0, int ks1, int tid) {
    float sum = 0.f;
    for (int base = ks0; base < ks1; base += 64) {
        float v[64];
#pragma unroll
        for (int i = 0; i < 64; ++i) v[i] = base + i < ks1 ? Spp[(long)(base + i) * H1 + tid] : 0.f;
#pragma unroll
        for (int i = 0; i < 64; ++i) sum += v[i];
    }
    Sp[(long)split * H1 + tid] = sum;
}

template <int H1>
__global__ __launch_bounds__(64 * kDwWaves) void enc0_dw_kernel(DwArgs a) {
    constexpr int NTL = H1 / 32;
    constexpr int RB = kDwRB;
    constexpr int NKS = RB / kKS;
    constexpr int KSE = dz_step_elems(H1);
    constexpr int NT = 64 * kDwWaves;
    constexpr int DZ_UNITS = NKS * KSE * 2 / 16;         // 16-byte units of the block's split dZ
    constexpr int DZ_PER = (DZ_UNITS + NT - 1) / NT;
    constexpr int LSEG = kDwLut / 2;                     // 16-byte units of one table row
    constexpr int LUT_PER = RB * LSEG / NT;              // ... of the block's table rows per thread
    constexpr int NBUF = H1 <= 64 ? 2 : 1;               // LDS tiles of a block: double buffered where they fit
    constexpr int GK = H1 <= 64 ? NKS : 1;               // K steps whose lookups are issued together (registers)
    static_assert(RB * 16 == 2 * NT && RB * LSEG == LUT_PER * NT && NT % LSEG == 0, "tile units per thread");
    __shared__ __attribute__((aligned(16))) unsigned short dzp[NBUF][NKS * KSE];
    __shared__ __attribute__((aligned(16))) uint2 lutl[NBUF][RB * kDwLut];
    __shared__ __attribute__((aligned(16))) unsigned char codes[NBUF][RB * kCodeLd];
    __shared__ float rfac[NBUF][RB];
    __shared__ int srows[NBUF][RB];
    __shared__ int srow_all[kDwMaxRS];                   // storage rows of the split's batch rows

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int bx, by;
    xcd_cell(bx, by);
    const int split = by;
    const int rb = split * a.RS;
    const int re = min(a.B, rb + a.RS);
    const int gbase = bx * (32 * kDwWaves);
    const int g0 = gbase + wave * 32;
    const bool wave_on = g0 < a.G;
    const int gene = g0 + l31;
    const int nks_total = (a.B + kKS - 1) / kKS;

    f32x16 acc[NTL];
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // ---- the loader: every thread moves fixed pieces of a block -- two 16-byte units of the count tile (rows tid / 16
    // and 32 + tid / 16), LUT_PER of the table rows, DZ_PER of the split dZ, and (tid < RB) one row's divisor.
    // Block i is computed from LDS buffer i % NBUF while the pieces of block i + 1 (requested during block i - 1, in
    // registers) are written to the other buffer behind the products and block i + 2 is requested: ONE barrier per block
    // (with a single buffer: a second one between the products and the writes).
    const int crow = tid >> 4, cseg = tid & 15;
    constexpr int LROWS = NT / LSEG;                     // table rows one pass of the workgroup covers
    const int lrow = tid / LSEG, lseg = tid % LSEG;
    const bool cseg_ok = gbase + cseg * 16 < a.c.ldc;
    // the storage rows of the split come from LDS: a storage row fetched from memory and carried in a register to the
    // next block's requests made the compiler wait for EVERY request in flight before a block's products (the copy of a
    // loaded register waits for the whole in-order load queue)
    for (int i = tid; i < re - rb; i += NT) srow_all[i] = a.srowb[rb + i];
    __syncthreads();
    auto srow_of = [&](int rg0, int row) __attribute__((always_inline)) {               // storage row of the block's row (clamped inside the split)
        const int r = rg0 + row;
        return srow_all[(r < re ? r : re - 1) - rb];
    };
    u32x4 cq[2], r_l[LUT_PER], r_dz[DZ_PER];
    float r_fac = 1.f;
    int r_srow = 0;
    auto request = [&](int rg0) __attribute__((always_inline)) {
        const u32x4 z = {0u, 0u, 0u, 0u};
        const int sc0 = srow_of(rg0, crow), sc1 = srow_of(rg0, 32 + crow);
        int sr_l[LUT_PER];
#pragma unroll
        for (int i = 0; i < LUT_PER; ++i) sr_l[i] = srow_of(rg0, lrow + i * LROWS);
        const int sr_t = srow_of(rg0, tid < RB ? tid : 0);
        cq[0] = (cseg_ok && rg0 + crow < re) ? *reinterpret_cast<const u32x4*>(a.c.yc + (long)sc0 * a.c.ldc + gbase + cseg * 16) : z;
        cq[1] = (cseg_ok && rg0 + 32 + crow < re) ? *reinterpret_cast<const u32x4*>(a.c.yc + (long)sc1 * a.c.ldc + gbase + cseg * 16) : z;
#pragma unroll
        for (int i = 0; i < LUT_PER; ++i) r_l[i] = *reinterpret_cast<const u32x4*>(a.lutp + (long)sr_l[i] * kLut + lseg * 2);
        const int ks0 = rg0 / kKS;
        const u32x4* src = reinterpret_cast<const u32x4*>(a.DZP + (long)ks0 * KSE);
#pragma unroll
        for (int i = 0; i < DZ_PER; ++i) {
            const int u = tid + i * NT;
            const int k = u / (KSE * 2 / 16);
            r_dz[i] = (u < DZ_UNITS && ks0 + k < nks_total) ? src[u] : z;
        }
        r_fac = a.fac ? a.fac[sr_t] : 1.f;
        r_srow = sr_t;
    };
    auto deposit = [&](auto bufc) __attribute__((always_inline)) {
        constexpr int b = decltype(bufc)::value;
        *reinterpret_cast<u32x4*>(codes[b] + crow * kCodeLd + cseg * 16) = cq[0];
        *reinterpret_cast<u32x4*>(codes[b] + (32 + crow) * kCodeLd + cseg * 16) = cq[1];
#pragma unroll
        for (int i = 0; i < LUT_PER; ++i) *reinterpret_cast<u32x4*>(lutl[b] + (lrow + i * LROWS) * kDwLut + lseg * 2) = r_l[i];
#pragma unroll
        for (int i = 0; i < DZ_PER; ++i) {
            const int u = tid + i * NT;
            if (u < DZ_UNITS) reinterpret_cast<u32x4*>(dzp[b])[u] = r_dz[i];
        }
        if (tid < RB) { rfac[b][tid] = r_fac; srows[b][tid] = r_srow; }
    };

    using Buf0 = std::integral_constant<int, 0>;
    using Buf1 = std::integral_constant<int, NBUF - 1>;
    if (rb < re) {
        request(rb);
        deposit(Buf0{});
        if (rb + RB < re) request(rb + RB);
    }
    auto block = [&](auto bufc, auto nextc, int rg0) __attribute__((always_inline)) {
        constexpr int b = decltype(bufc)::value;
        __syncthreads();                    // block rg0 is in buffer b; everyone is done with the block before it
        if (wave_on) {
            // all lookups of GK K steps first (two LDS round trips for the group instead of two per step), then their
            // products.  Rows beyond the split's end hold zero counts (table entry 0 = 0): their K steps add nothing.
#pragma unroll
          for (int kg = 0; kg < NKS; kg += GK) {
            unsigned code[GK * 8], lo[GK * 8], hx[GK * 8];
#pragma unroll
            for (int ks = 0; ks < GK; ++ks) {
                const unsigned char* cp = codes[b] + ((kg + ks) * kKS + 8 * hi) * kCodeLd + wave * 32 + l31;
#pragma unroll
                for (int j = 0; j < 8; ++j) code[ks * 8 + j] = cp[j * kCodeLd];
            }
            unsigned any = 0u;
#pragma unroll
            for (int i = 0; i < GK * 8; ++i) {
                const unsigned idx = code[i] < (unsigned)(kDwLut - 1) ? code[i] : (unsigned)(kDwLut - 1);
                const uint2 e = lutl[b][((kg + (i >> 3)) * kKS + 8 * hi + (i & 7)) * kDwLut + idx];
                lo[i] = e.x; hx[i] = e.y;
                any |= code[i];
            }
            if (__ballot(any >= (unsigned)kDwLut)) {         // rare: counts beyond the table take the formula itself, one at a time
                unsigned bad = 0u;
#pragma unroll
                for (int i = 0; i < GK * 8; ++i) bad |= (code[i] >= (unsigned)kDwLut ? 1u : 0u) << i;
#pragma unroll 1
                while (__ballot(bad != 0u)) {
                    const bool on = bad != 0u;
                    const int i = on ? __builtin_ctz(bad) : 0;
                    bad &= bad - 1u;
                    unsigned ci = code[0];
#pragma unroll
                    for (int k = 1; k < GK * 8; ++k) ci = i == k ? code[k] : ci;
                    const int rl = (kg + (i >> 3)) * kKS + 8 * hi + (i & 7);
                    float val = (float)ci;
                    if (__ballot(on && ci == 255u)) {        // an escape: the count itself from the row's overflow list
                        if (on && ci == 255u) val = escaped_count(a.c, srows[b][rl], gene);
                    }
                    float x = a.fac ? __fdiv_rn(val, rfac[b][rl]) : val;
                    if (a.do_log) x = log1p_beyond_table(x);
                    const uint2 e = split_entry(x);
#pragma unroll
                    for (int k = 0; k < GK * 8; ++k)
                        if (on && i == k) { lo[k] = e.x; hx[k] = e.y; }
                }
            }
#pragma unroll
            for (int ks = 0; ks < GK; ++ks) {
                u32x4 A[3];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) {
                    A[0][jj] = __builtin_amdgcn_perm(lo[8 * ks + 2 * jj + 1], lo[8 * ks + 2 * jj], 0x05040100u);
                    A[1][jj] = __builtin_amdgcn_perm(lo[8 * ks + 2 * jj + 1], lo[8 * ks + 2 * jj], 0x07060302u);
                    A[2][jj] = __builtin_amdgcn_perm(hx[8 * ks + 2 * jj + 1], hx[8 * ks + 2 * jj], 0x05040100u);
                }
                const unsigned short* bstep = dzp[b] + (kg + ks) * KSE + (hi * 32 + l31) * 8;
#pragma unroll
                for (int t = 0; t < NTL; ++t) {
                    u32x4 Bf[3];
#pragma unroll
                    for (int q = 0; q < 3; ++q)
                        Bf[q] = *reinterpret_cast<const u32x4*>(bstep + ((q * NTL + t) * 2) * 32 * 8);
                    MFMA_X3(A, Bf, acc[t])
                }
            }
          }
        }
        if (NBUF == 1) __syncthreads();     // a single buffer: everyone is done reading it
        if (rg0 + RB < re) deposit(nextc);  // the next block (in registers since the previous step) behind the products
        if (rg0 + 2 * RB < re) request(rg0 + 2 * RB);
    };
#pragma unroll 1
    for (int rg0 = rb; rg0 < re; rg0 += 2 * RB) {
        block(Buf0{}, Buf1{}, rg0);
        if (rg0 + RB < re) block(Buf1{}, Buf0{}, rg0 + RB);
    }
    if (bx == 0 && tid < H1) split_column_sums<H1>(a.Spp, a.Sp, split, rb / kKS, (re + kKS - 1) / kKS, tid);
    if (!wave_on) return;
#pragma unroll
    for (int t = 0; t < NTL; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            a.P[((long)split * a.Gs + g0 + rowmap(e, hi)) * H1 + 32 * t + l31] = acc[t][e];
}

// ---- second form of the weight gradient (64 first-layer units): operands staged global -> LDS directly into a deep ring,
// a software pipeline over the K steps, each wave's LDS / vector work between its own matrix instructions.  What changed against enc0_dw_kernel above and why
// (its counters: MfmaUtil 25 %, 46 % of a wave's cycles at s_waitcnt, 343 vector + 103 LDS instructions per 48 matrix
// instructions; each block ran barrier -> counts -> table entries -> products with every LDS round trip exposed, both waves of
// a SIMD in the same phase -- profiles/r05k_*, r05l_*):
//   * a wave owns TWO 32-gene tiles (a workgroup 512 genes): the dZ fragments of a K step are read from LDS once for both,
//     the table rows and dZ of a batch row are fetched by half as many workgroups;
//   * a stage is ONE K step (16 batch rows: their counts, 128-entry table rows and split dZ, 30.25 KB); five stages in a ring
//     filled by global_load_lds_dwordx4 (no staging registers, no LDS store pass) FOUR steps ahead, retired with a counted
//     s_waitcnt vmcnt; one barrier per step.  (Four stages -- two workgroups of four waves per CU -- were measured: the two
//     steps of lead do not cover the memory round trip, 46 % of the wave cycles at the vmcnt wait.)
//   * software pipeline over the steps: in step k a wave requests step k + 4, reads the counts of step k + 1, the table
//     entries of step k (its counts arrived during step k - 1) and the dZ fragments of step k, and issues the 24 matrix
//     instructions of step k - 1: every LDS round trip stands behind matrix work;
//   * the step is written as 2 x 12 matrix instructions with the LDS reads, the request and the repacking v_perms between
//     them, every accumulator tied to an empty instruction statement behind its MFMA so that the order survives the compiler:
//     the two waves of a SIMD do not hide each other's vector / LDS work (tried: one half of the waves reading first, the
//     other multiplying first; priorities; two workgroups per CU -- the times add), a wave's own matrix instructions do;
//   * all LDS reads of the loop are written as instructions (lds_read_*): the compiler does not know which LDS bytes a
//     global_load_lds writes and puts s_waitcnt vmcnt(0) in front of every LDS read it generates itself (measured with plain
//     reads: the ring stands still for a memory round trip per step).
// Same operands, same six products, same summation order over the rows of a split as the first form (the split count differs:
// dw2_splits): results agree to the association of the split sums.
constexpr int kD2Waves = 8;
constexpr int kD2MT = 2;                                   // 32-gene tiles per wave
constexpr int kD2Genes = kD2Waves * kD2MT * 32;            // 512 genes per workgroup
constexpr int kD2RB = kKS;                                 // rows per split are a multiple of this
constexpr int kD2MaxRS = 1024;                             // batch rows per split at most (their storage rows sit in LDS)
constexpr int kD2Stages = 5;
constexpr int kD2Lead = kD2Stages - 1;                     // a request goes into the stage the step before the current one used
constexpr int kD2LutB = kKS * kLut * 8;                    // 16384: 16 table rows of 1 KB (128 entries: with 64, the workgroups that hold
                                                           // the most expressed genes visited the formula path 10-15 times and set the kernel's time)
constexpr int kD2DzB = dz_step_elems(64) * 2;              // 6144: one K step of split dZ
constexpr int kD2CodeChunk = 1024 + 32;                    // two count rows of 512 bytes land as one chunk; rows 8 apart sit 32 banks apart
constexpr int kD2CodeB = 8 * kD2CodeChunk;                 // 8448
constexpr int kD2StageB = kD2LutB + kD2DzB + kD2CodeB;     // 30976

#ifdef DCA_DW_TIMING
__device__ long long* g_dw_timing = nullptr;         // [workgroup][8]: clock at kernel entry, after the prologue, after the loop, at the end (wave 0) + realtime at entry / end
#define DWSTAMP(i) if (g_dw_timing && tid == 0) g_dw_timing[(blockIdx.x + blockIdx.y * gridDim.x) * 8 + (i)] = (i) >= 4 ? (long long)__builtin_amdgcn_s_memrealtime() : (long long)__builtin_readcyclecounter();
#else
#define DWSTAMP(i)
#endif
__global__ __launch_bounds__(64 * kD2Waves) __attribute__((amdgpu_waves_per_eu(2, 2))) void enc0_dw2_kernel(DwArgs a) {
    constexpr int H1 = 64, NTL = 2;
    constexpr int KSE = dz_step_elems(H1);
    constexpr int NT = 64 * kD2Waves;
    static_assert(kLut == 128 && kKS == 16 && kD2DzB == 6 * 1024 && kD2Waves == 8, "chunks of the loader");
    __shared__ __attribute__((aligned(16))) unsigned char ring[kD2Stages * kD2StageB];
    __shared__ int srow_all[kD2MaxRS];
    __shared__ float fac_all[kD2MaxRS];                    // size factors of the split's rows (the formula path)

    const int tid = threadIdx.x, lane = tid & 63;
    DWSTAMP(0) DWSTAMP(4)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int bx, by;
    xcd_cell(bx, by);
    const int split = by;
    const int rb = split * a.RS;
    const int re = min(a.B, rb + a.RS);
    const int gbase = bx * kD2Genes;
    const int g0 = gbase + wave * (32 * kD2MT);

    f32x16 acc[kD2MT][NTL];
#pragma unroll
    for (int m = 0; m < kD2MT; ++m)
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][t][e] = 0.f;

    // storage rows of the split's rows; entries up to the end of the last K step repeat the last row (its dZ rows are zero
    // beyond B: enc0_split_dz_kernel)
    const int nrows = re > rb ? re - rb : 0;           // (a split behind the end of the batch is empty: it writes zero partials)
    const int nsteps = (nrows + kKS - 1) / kKS;
    for (int i = tid; i < nsteps * kKS; i += NT) {
        const int sr = a.srowb[rb + (i < nrows ? i : nrows - 1)];
        srow_all[i] = sr;
        fac_all[i] = a.fac ? a.fac[sr] : 1.f;
    }
    // count bytes of genes behind the end of a stored row are never loaded: those places of the ring stay zero
    for (int i = tid; i < kD2Stages * kD2CodeB / 16; i += NT) {
        const int st = i / (kD2CodeB / 16), u = i % (kD2CodeB / 16);
        *reinterpret_cast<u32x4*>(ring + st * kD2StageB + kD2LutB + kD2DzB + u * 16) = u32x4{0u, 0u, 0u, 0u};
    }
    __syncthreads();

    // ---- the loader: a wave instruction moves 1 KB (lane-linear in LDS).  Per step 16 chunks of table rows and 8 of counts
    // (rows 2 wave and 2 wave + 1 of the step, one of each per wave) and 6 of split dZ (waves 0..5).
    const unsigned ring_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)ring;
    const unsigned srow_base = (unsigned)(size_t)(__attribute__((address_space(3))) int*)srow_all;
    const int ks_first = rb / kKS;
    const bool cseg_ok = gbase + l31 * 16 < a.c.ldc;
    const bool three = wave < 6;                           // requests of this wave per step: 4 (with a dZ chunk) or 3
    const unsigned sr_ad = srow_base + (unsigned)(2 * wave + hi) * 4u;
    auto issue_srow = [&](int k, int& s0) __attribute__((always_inline)) {   // storage row of batch row 2 wave + hi of step k
        asm volatile("ds_read_b32 %0, %1" : "=v"(s0) : "v"(sr_ad + (unsigned)k * (kKS * 4u)));
    };
    auto request = [&](int k, int stage, int s0) __attribute__((always_inline)) {
        unsigned char* st = ring + stage * kD2StageB;
        const unsigned char* lut = reinterpret_cast<const unsigned char*>(a.lutp);
        const unsigned char* dz = reinterpret_cast<const unsigned char*>(a.DZP + (long)(ks_first + k) * KSE);
        // table rows 2 wave and 2 wave + 1 of the step (1 KB each = one chunk): their storage rows sit in lanes 0 and 32
        const long sA = __builtin_amdgcn_readlane(s0, 0), sB = __builtin_amdgcn_readlane(s0, 32);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(lut + sA * (kLut * 8) + lane * 16),
                                         (__attribute__((address_space(3))) void*)(st + (2 * wave) * 1024), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(lut + sB * (kLut * 8) + lane * 16),
                                         (__attribute__((address_space(3))) void*)(st + (2 * wave + 1) * 1024), 16, 0, 0);
        if (cseg_ok)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.c.yc + (long)s0 * a.c.ldc + gbase + l31 * 16),
                                             (__attribute__((address_space(3))) void*)(st + kD2LutB + kD2DzB + wave * kD2CodeChunk), 16, 0, 0);
        if (three)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(dz + wave * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(st + kD2LutB + wave * 1024), 16, 0, 0);
    };
    // own requests of all steps but the `newer` newest have landed
    auto retire = [&](int newer) __attribute__((always_inline)) {
        if (newer <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (three) {
            if (newer == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (newer == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        } else {
            if (newer == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (newer == 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
        }
    };

    // ---- the stages of a K step: counts -> table entries -> six products per (gene tile, dZ tile)
    const unsigned code_off = ring_base + kD2LutB + kD2DzB + (4 * hi) * kD2CodeChunk + wave * (32 * kD2MT) + l31;   // row 8 hi + j: chunk 4 hi + j / 2, row j % 2 of it
    const unsigned lut_off = ring_base + 8 * hi * (kLut * 8);
    const unsigned dz_off = ring_base + kD2LutB + (hi * 32 + l31) * 16;
    auto issue_counts = [&](int stage, unsigned (&code)[kD2MT][8]) __attribute__((always_inline)) {
        const unsigned ad = code_off + (unsigned)stage * kD2StageB;
        static_for<kD2MT * 8>([&](auto ic) __attribute__((always_inline)) {
            constexpr int m = decltype(ic)::value >> 3, j = decltype(ic)::value & 7;
            lds_read_u8<(j >> 1) * kD2CodeChunk + (j & 1) * 512 + 32 * m>(code[m][j], ad);      // gene g0 + 32 m + l31
        });
    };
    // (a count beyond the table reads past its row -- still inside the ring: the value is replaced below)
    auto issue_entries = [&](int stage, const unsigned (&code)[kD2MT][8], u32x2 (&ent)[kD2MT][8]) __attribute__((always_inline)) {
        const unsigned ad = lut_off + (unsigned)stage * kD2StageB;
        static_for<kD2MT * 8>([&](auto ic) __attribute__((always_inline)) {
            constexpr int m = decltype(ic)::value >> 3, j = decltype(ic)::value & 7;
            lds_read_b64<j * kLut * 8>(ent[m][j], ad + code[m][j] * 8u);
        });
    };
    auto issue_dz = [&](int stage, u32x4 (&Bf)[NTL][3]) __attribute__((always_inline)) {
        const unsigned ad = dz_off + (unsigned)stage * kD2StageB;
        static_for<NTL * 3>([&](auto ic) __attribute__((always_inline)) {
            constexpr int t = decltype(ic)::value / 3, q = decltype(ic)::value % 3;
            lds_read_b128<(q * NTL + t) * 1024>(Bf[t][q], ad);
        });
    };
    // rare: counts beyond the table take the formula itself -- only the (tile, row) places some lane of the wave needs, one at
    // a time; count, size factor and storage row come from LDS (a workgroup that holds a highly expressed gene comes here in
    // most steps, and all its waves wait at the barrier meanwhile: each visit cost ~3300 cycles with select chains over the
    // register arrays and global loads -- those workgroups ran 25 % longer than the others, profiles/r05n_*)
    const unsigned fac_base = (unsigned)(size_t)(__attribute__((address_space(3))) float*)fac_all;
#ifdef DCA_DW_TIMING
    int n_visits = 0, n_places = 0;
#endif
    auto beyond_table = [&](int k, int stage, const unsigned (&code)[kD2MT][8], u32x2 (&ent)[kD2MT][8], unsigned any_dbg) __attribute__((always_inline)) {
        unsigned places = 0u;
        static_for<kD2MT * 8>([&](auto ic) __attribute__((always_inline)) {
            constexpr int q = decltype(ic)::value;
            if (__ballot(code[q >> 3][q & 7] >= (unsigned)kLut)) places |= 1u << q;
        });
#ifdef DCA_DW_TIMING
        n_visits += 1; n_places += __builtin_popcount(places);
        if (places == 0u && g_dw_timing) {                 // (must not happen) leave the evidence: step, lane's sixteen counts packed
            unsigned long long pk0 = 0ull, pk1 = 0ull;
            for (int q = 0; q < 8; ++q) { pk0 |= (unsigned long long)(code[0][q] & 0xffu) << (8 * q); pk1 |= (unsigned long long)(code[1][q] & 0xffu) << (8 * q); }
            unsigned mx = 0u;
            for (int q = 0; q < 16; ++q) mx |= code[q >> 3][q & 7];
            if (any_dbg >= (unsigned)kLut) {
                long long* d = g_dw_timing + 4096 * 8;
                d[0] = k; d[1] = (long long)pk0; d[2] = (long long)pk1; d[3] = lane + 1000 * wave; d[4] = mx; d[5] = any_dbg;
            }
        }
#endif
#pragma unroll 1
        while (places) {
            const int q = __builtin_ctz(places);
            places &= places - 1u;
            const int m = q >> 3, j = q & 7;
            const unsigned ro = (unsigned)(k * kKS + 8 * hi + j) * 4u;
            unsigned c; float fc; int srow;
            asm volatile("ds_read_u8 %0, %1" : "=v"(c) : "v"(code_off + (unsigned)(stage * kD2StageB + (j >> 1) * kD2CodeChunk + (j & 1) * 512 + 32 * m)));
            asm volatile("ds_read_b32 %0, %1" : "=v"(fc) : "v"(fac_base + ro));
            asm volatile("ds_read_b32 %0, %1" : "=v"(srow) : "v"(srow_base + ro));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(c), "+v"(fc), "+v"(srow));
            const bool on = c >= (unsigned)kLut;
            float val = (float)c;
            if (__ballot(on && c == 255u)) { if (on && c == 255u) val = escaped_count(a.c, srow, g0 + 32 * m + l31); }
            float x = a.fac ? __fdiv_rn(val, fc) : val;
            if (a.do_log) x = log1p_beyond_table(x);
            const uint2 e = split_entry(x);
            static_for<kD2MT * 8>([&](auto ic) __attribute__((always_inline)) {
                constexpr int Q = decltype(ic)::value;
                if (q == Q) {                                   // (uniform: one of the sixteen runs)
                    ent[Q >> 3][Q & 7][0] = on ? e.x : ent[Q >> 3][Q & 7][0];
                    ent[Q >> 3][Q & 7][1] = on ? e.y : ent[Q >> 3][Q & 7][1];
                }
            });
        }
    };
#define DCA_TIE8(x, m) "+v"(x[m][0]), "+v"(x[m][1]), "+v"(x[m][2]), "+v"(x[m][3]), "+v"(x[m][4]), "+v"(x[m][5]), "+v"(x[m][6]), "+v"(x[m][7])
#define DCA_TIE_ACC "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[1][0]), "+v"(acc[1][1])

    if (nsteps > 0) {
        u32x4 A0[kD2MT][3], A1[kD2MT][3], B0[NTL][3], B1[NTL][3];
        unsigned c0[kD2MT][8], c1[kD2MT][8];
        u32x2 ent[kD2MT][8];
        int s0 = 0;
#pragma unroll
        for (int m = 0; m < kD2MT; ++m)
#pragma unroll
            for (int q = 0; q < 3; ++q) A0[m][q] = u32x4{0u, 0u, 0u, 0u};        // (step "-1": zero products)
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int q = 0; q < 3; ++q) B0[t][q] = u32x4{0u, 0u, 0u, 0u};
#pragma unroll
        for (int k = 0; k < kD2Lead; ++k)
            if (k < nsteps) {
                issue_srow(k, s0);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(s0));
                request(k, k, s0);
            }
        issue_srow(min(kD2Lead, nsteps - 1), s0);
        retire(min(kD2Lead, nsteps) - 2);               // steps 0 and 1 have landed (mine)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_counts(0, c0);
        int sk = 0;                                       // stage of step k
        DWSTAMP(1)
        // One matrix instruction of the six-product scheme: product PR (0..5, small terms first) of gene tile M with dZ tile T.
        // The accumulator is tied to an (empty) instruction statement behind it: the compiler keeps the matrix instruction
        // between the LDS / vector work written before and after it.
#ifdef DCA_EXP_DW2_NOMFMA
#define DCA_MF(Ap, Bp, M, T, PR) { acc[M][T][PR] += __uint_as_float(Ap[M][PR % 3][0] ^ Bp[T][PR % 3][1]); asm volatile("" : "+v"(acc[M][T])); }
#else
#define DCA_MF(Ap, Bp, M, T, PR) { constexpr int PA_[6] = {2, 1, 0, 1, 0, 0}, PB_[6] = {0, 1, 2, 0, 1, 0}; \
            acc[M][T] = MFMA16(Ap[M][PA_[PR]], Bp[T][PB_[PR]], acc[M][T]); asm volatile("" : "+v"(acc[M][T])); }
#endif
        // step k: cK = its counts (requested in the step before), Ap / Bp = table entries / dZ fragments of step k - 1; leaves
        // the counts of step k + 1 in cN, the operands of step k in An / Bk.  A wave's vector and LDS instructions ride behind
        // its OWN matrix instructions (about five per matrix instruction are free, tools/microbench/mfma_valu_interleave.hip);
        // the other wave of the SIMD does not hide them (measured: the two waves' times add).
        auto step = [&](int k, unsigned (&cK)[kD2MT][8], unsigned (&cN)[kD2MT][8], u32x4 (&Ap)[kD2MT][3], u32x4 (&An)[kD2MT][3],
                        u32x4 (&Bp)[NTL][3], u32x4 (&Bk)[NTL][3]) __attribute__((always_inline)) {
            const int sn = sk + 1 == kD2Stages ? 0 : sk + 1, sp = sk == 0 ? kD2Stages - 1 : sk - 1;
            // (counts of this step, dZ fragments of the last one, storage row of the next request: all LDS reads so far)
            asm volatile("s_waitcnt lgkmcnt(0)" : DCA_TIE8(cK, 0), DCA_TIE8(cK, 1), "+v"(s0));
            asm volatile("" : "+v"(Bp[0][0]), "+v"(Bp[0][1]), "+v"(Bp[0][2]), "+v"(Bp[1][0]), "+v"(Bp[1][1]), "+v"(Bp[1][2]));
            unsigned any = 0u;
            // ---- gene tile 0 of step k - 1, behind it: the request of the step kD2Lead ahead, the table entries and dZ fragments of step k
            const unsigned ead = lut_off + (unsigned)sk * kD2StageB, dad = dz_off + (unsigned)sk * kD2StageB;
            static_for<12>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value, T = i & 1, PR = i >> 1;
                DCA_MF(Ap, Bp, 0, T, PR)
                if constexpr (i == 0) {
                    if (k + kD2Lead < nsteps) request(k + kD2Lead, sp, s0);   // into the stage of step k - 1: everyone read it before the barrier
                }
                if constexpr (i < 8) {                    // entries of (tile, row) places 2 i and 2 i + 1
                    constexpr int q0 = 2 * i, q1 = 2 * i + 1;
                    lds_read_b64<(q0 & 7) * kLut * 8>(ent[q0 >> 3][q0 & 7], ead + cK[q0 >> 3][q0 & 7] * 8u);
                    lds_read_b64<(q1 & 7) * kLut * 8>(ent[q1 >> 3][q1 & 7], ead + cK[q1 >> 3][q1 & 7] * 8u);
                } else if constexpr (i < 11) {            // dZ fragments 2 (i - 8) and 2 (i - 8) + 1
                    constexpr int f0 = 2 * (i - 8), f1 = f0 + 1;
                    lds_read_b128<((f0 % 3) * NTL + f0 / 3) * 1024>(Bk[f0 / 3][f0 % 3], dad);
                    lds_read_b128<((f1 % 3) * NTL + f1 / 3) * 1024>(Bk[f1 / 3][f1 % 3], dad);
                } else {
                    issue_srow(min(k + 1 + kD2Lead, nsteps - 1), s0);
#pragma unroll
                    for (int m = 0; m < kD2MT; ++m)
#pragma unroll
                        for (int j = 0; j < 8; ++j) any |= cK[m][j];
                }
            });
            // the sixteen entries are in registers (LDS reads return in order: the six dZ fragments and the storage row
            // behind them may still be in flight -- they are waited for at the top of the next step)
            asm volatile("s_waitcnt lgkmcnt(7)" : DCA_TIE8(ent, 0), DCA_TIE8(ent, 1));
            if (__ballot(any >= (unsigned)kLut)) beyond_table(k, sk, cK, ent, any);
            // ---- gene tile 1 of step k - 1, behind it: the operands of step k from its entries, the counts of step k + 1
            const unsigned cad = code_off + (unsigned)sn * kD2StageB;
            static_for<12>([&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value, T = i & 1, PR = i >> 1;
                DCA_MF(Ap, Bp, 1, T, PR)
                if constexpr (i < 8) {                    // counts of places 2 i, 2 i + 1 of step k + 1 (behind the last step: a stale stage, not used)
                    constexpr int q0 = 2 * i, q1 = 2 * i + 1;
                    lds_read_u8<((q0 & 7) >> 1) * kD2CodeChunk + (q0 & 1) * 512 + 32 * (q0 >> 3)>(cN[q0 >> 3][q0 & 7], cad);
                    lds_read_u8<((q1 & 7) >> 1) * kD2CodeChunk + (q1 & 1) * 512 + 32 * (q1 >> 3)>(cN[q1 >> 3][q1 & 7], cad);
                }
                {                                         // two of the 24 operand registers of step k
                    constexpr int M = i / 6, Q = (i % 6) / 2, J0 = (i & 1) * 2;
#pragma unroll
                    for (int jj = J0; jj < J0 + 2; ++jj) {
                        const unsigned e1 = ent[M][2 * jj + 1][Q == 2 ? 1 : 0], e0 = ent[M][2 * jj][Q == 2 ? 1 : 0];
                        An[M][Q][jj] = __builtin_amdgcn_perm(e1, e0, Q == 1 ? 0x07060302u : 0x05040100u);
                    }
                    if constexpr ((i & 1) == 1) asm volatile("" : "+v"(An[M][Q]));
                }
            });
            retire(min(k + kD2Lead, nsteps - 1) - (k + 2));   // step k + 2 has landed (mine)
            // the LDS reads of stage sk (six dZ fragments, the storage row) are in registers BEFORE another wave may refill that
            // stage behind the barrier: LDS reads return in order, the sixteen count reads of stage sn issued last may stay in flight
            asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
#ifndef DCA_EXP_DW2_NOBARRIER
            __builtin_amdgcn_s_barrier();
#endif
            asm volatile("" ::: "memory");
            sk = sn;
        };
        int k = 0;
#pragma unroll 1
        for (; k + 1 < nsteps; k += 2) {
            step(k, c0, c1, A0, A1, B0, B1);
            step(k + 1, c1, c0, A1, A0, B1, B0);
        }
        if (k < nsteps) {
            step(k, c0, c1, A0, A1, B0, B1);
            asm volatile("s_waitcnt lgkmcnt(0)" : DCA_TIE8(c1, 0), DCA_TIE8(c1, 1), "+v"(s0));
            asm volatile("" : "+v"(B1[0][0]), "+v"(B1[0][1]), "+v"(B1[0][2]), "+v"(B1[1][0]), "+v"(B1[1][1]), "+v"(B1[1][2]));
#pragma unroll
            for (int m = 0; m < kD2MT; ++m)
#pragma unroll
                for (int t = 0; t < NTL; ++t) { MFMA_X3(A1[m], B1[t], acc[m][t]) }
        } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : DCA_TIE8(c0, 0), DCA_TIE8(c0, 1), "+v"(s0));
            asm volatile("" : "+v"(B0[0][0]), "+v"(B0[0][1]), "+v"(B0[0][2]), "+v"(B0[1][0]), "+v"(B0[1][1]), "+v"(B0[1][2]));
#pragma unroll
            for (int m = 0; m < kD2MT; ++m)
#pragma unroll
                for (int t = 0; t < NTL; ++t) { MFMA_X3(A0[m], B0[t], acc[m][t]) }
        }
#undef DCA_MF
        DWSTAMP(2)
#ifdef DCA_DW_TIMING
        if (g_dw_timing && lane == 0 && n_visits) {
            atomicAdd((unsigned long long*)&g_dw_timing[(blockIdx.x + blockIdx.y * gridDim.x) * 8 + 6], (unsigned long long)n_visits);
            atomicAdd((unsigned long long*)&g_dw_timing[(blockIdx.x + blockIdx.y * gridDim.x) * 8 + 7], (unsigned long long)n_places);
        }
#endif
    }
#undef DCA_TIE8
#undef DCA_TIE_ACC
    if (bx == 0 && tid < H1) split_column_sums<H1>(a.Spp, a.Sp, split, rb / kKS, (re + kKS - 1) / kKS, tid);
#pragma unroll
    for (int m = 0; m < kD2MT; ++m)
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                a.P[((long)split * a.Gs + g0 + 32 * m + rowmap(e, hi)) * H1 + 32 * t + l31] = acc[m][t][e];
    DWSTAMP(3) DWSTAMP(5)
}

// splits of the batch for the second form: one workgroup per CU (256 slots), at least one 32-row block each, at most
// kD2MaxRS rows each
inline int dw2_splits(int B, int G) {
    const int groups = (G + kD2Genes - 1) / kD2Genes;
    int ns = 256 / groups;
    const int maxs = (B + kD2RB - 1) / kD2RB;
    if (ns > maxs) ns = maxs;
    if (ns > 16) ns = 16;
    if (ns < 1) ns = 1;
    const int need = (B + kD2MaxRS - 1) / kD2MaxRS;
    if (ns < need) ns = need;
    return ns;
}
inline int dw2_rows_per_split(int B, int ns) { return (((B + ns - 1) / ns) + kD2RB - 1) / kD2RB * kD2RB; }

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

// (The round-2 forward over the NON-ZERO counts only -- gathers of W0 rows on the vector pipe -- lost to the dense product at
// every batch size and is an experiment build: -DDCA_EXP_ENC0_SPARSE_FWD, tools/bench_enc0.py.)
#ifdef DCA_EXP_ENC0_SPARSE_FWD
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
    {                                                    // the partials in a fixed order: phases of blocks, then the phases
        double v = 0.0;
        for (unsigned k = ph; k < gridDim.x; k += nph)
            v += __hip_atomic_load(a.part + (long)k * a.H1 + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red[tid] = v;
    }
    __syncthreads();
    if (tid < a.H1) {
        double v = 0.0;
        for (int k = 0; k < nph; ++k) v += red[k * a.H1 + tid];
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

#endif  // DCA_EXP_ENC0_SPARSE_FWD

// ------------------------------------------------------------------------------------------------- forward on the matrix pipe
// Z0 = L (W0 / std) + b_eff as a dense product whose A operand is LOOKED UP from the byte store (the mirror image of
// enc0_dw): a wave owns 32 batch rows (one per lane and K half) and all H1 columns; K = the genes, 128 per super step:
// the two lanes of a row load its 128 consecutive count bytes (one cache line, requested a super step ahead and touched
// two ahead: the rows are gathered, every line is a DRAM access of its own and the loop is otherwise latency bound),
// consumed as two macro steps of 64 genes (4 K steps of 8 bytes per lane each).  Every byte indexes the cell's table of
// pre-split values (counts 0 .. 31 from LDS, one padded table row per batch row so that equal counts of different
// rows fall into different banks; larger counts take the formula -- a wave-uniform branch that is rare on count data).
// The B operand (W0 / std as three bf16 pieces in MFMA order, written once per call by enc0_wsplit) is shared by the
// workgroup's 8 waves through LDS, one 64-gene tile at a time, double buffered.  The genes are cut into chunks (one
// partial per chunk, added in order by enc0_fwd_reduce): fixed order everywhere, deterministic.
constexpr int kFlRows = 256;        // batch rows per workgroup (8 waves x 32)
constexpr int kFlMS = 64;           // genes per macro step (one tile of split W0 in LDS); two macro steps = one super step
constexpr int kFlLut = 32;          // table entries per cell held in LDS
constexpr int kFlLutLd = 33;        // their row stride (entries)
constexpr int fl_ms_elems(int H1) { return 3 * (H1 / 32) * 4 * 2 * 32 * 8; }   // bf16 elements of one macro step of split W0

// W0 [G, ldw] (/ std) -> WP [macro step][piece][column tile][K step][k half][32 columns][8 genes] bf16; K step ks of
// macro step ms holds genes 128 (ms / 2) + 64 half + 32 (ms % 2) + 8 ks + j (the bytes a lane holds); zero beyond G.  With a mean: the
// macro step's share of the bias correction, C0P [macro step][H1] = sum_g (mean[g] / std[g]) W0[g, :] in fp64.
template <int H1>
__global__ __launch_bounds__(256) void enc0_wsplit_kernel(const float* W, long ldw, const float* mean, const float* stdv,
                                                          int G, unsigned short* WP, double* C0P) {
    constexpr int NTL = H1 / 32;
    __shared__ double red[8][32];
    const int ms = blockIdx.x, t = blockIdx.y, u = threadIdx.x;      // one (macro step, column tile) per workgroup
    const int col = u & 31, hi = (u >> 5) & 1, ks = u >> 6;
    const int gb = (ms >> 1) * (2 * kFlMS) + 64 * hi + 32 * (ms & 1) + 8 * ks;
    float w[8], sd[8], mu[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int g = gb + j;
        w[j] = g < G ? W[(long)g * ldw + 32 * t + col] : 0.f;
        sd[j] = (stdv && g < G) ? stdv[g] : 1.f;
        mu[j] = (mean && g < G) ? mean[g] : 0.f;
    }
    double c0 = 0.0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (mean) c0 += (double)(stdv ? __fdiv_rn(mu[j], sd[j]) : mu[j]) * (double)w[j];
        if (stdv) w[j] = __fdiv_rn(w[j], sd[j]);
    }
    u32x4 p[3];
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
        unsigned a, b, c;
        split_pair(w[2 * jj], w[2 * jj + 1], a, b, c);
        p[0][jj] = a; p[1][jj] = b; p[2][jj] = c;
    }
    u32x4* out = reinterpret_cast<u32x4*>(WP + (long)ms * fl_ms_elems(H1));
#pragma unroll
    for (int q = 0; q < 3; ++q) out[((q * NTL + t) * 4 + ks) * 64 + hi * 32 + col] = p[q];
    if (!mean) return;
    red[ks * 2 + hi][col] = c0;
    __syncthreads();
    if (u < 32) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) v += red[k][u];
        C0P[(long)ms * H1 + 32 * t + u] = v;
    }
}

struct FlArgs {
    Compact c;
    const float* fac; int do_log;
    const uint2* lutp;              // [n, kLut]
    const int* perm; const long long* cursor; long row_base;
    int B, G;
    const unsigned short* WP;       // [n_ms][fl_ms_elems]
    const double* C0P;              // [n_ms][H1] shares of the bias correction, or NULL
    float* P; long Bp;              // [chunks][Bp][H1] partial products
    int n_ms, ms_per;               // macro steps in all / per gene chunk
};

// RT = 32-row tiles per wave: 1 -> eight waves of 32 rows (two per SIMD), 2 -> four waves of 64 rows (one per SIMD, the W
// fragments of a K step read from LDS once for both tiles).
template <int H1, int RT>
__global__ __launch_bounds__(kFlRows / (32 * RT) * 64) void enc0_fwd_lut_kernel(FlArgs a) {
    constexpr int NT = kFlRows / (32 * RT) * 64;
    constexpr int NTL = H1 / 32, MSE = fl_ms_elems(H1), UNITS = MSE / 8, UPT = (UNITS + NT - 1) / NT;
    __shared__ __attribute__((aligned(16))) uint2 lutl[kFlRows * kFlLutLd];
    __shared__ __attribute__((aligned(16))) unsigned short wl[2][MSE];
    __shared__ int srows[kFlRows];
    __shared__ float csum[H1];
    __shared__ double cpart[NT / H1][H1];
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg0 = blockIdx.x * kFlRows;
    const int ms0 = blockIdx.y * a.ms_per, ms1 = min(a.n_ms, ms0 + a.ms_per);
    for (int i = tid; i < kFlRows; i += NT) {
        const long long cur = (a.cursor ? *a.cursor : 0) + a.row_base;
        const int r = min(rg0 + i, a.B - 1);
        srows[i] = a.perm ? a.perm[cur + r] : (int)(cur + r);
    }
    {                                       // this chunk's share of -sum_g (mean / std) W0[g, :], macro steps spread over the threads
        const int col = tid % H1, part = tid / H1;
        double v = 0.0;
        if (a.C0P)
            for (int ms = ms0 + part; ms < ms1; ms += NT / H1) v += a.C0P[(long)ms * H1 + col];
        cpart[part][col] = v;
    }
    __syncthreads();
    if (tid < H1) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < NT / H1; ++k) v += cpart[k][tid];
        csum[tid] = (float)(-v);            // (read after the loop's barriers)
    }
    // the table rows of the workgroup's cells: requested here, stored behind the first counts / tile requests below (one
    // round trip to memory for the whole prologue instead of three)
    constexpr int NU = kFlRows * (kFlLut / 2) / NT;
    u32x4 lv[NU];
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int u = tid + NT * k;
        lv[k] = *reinterpret_cast<const u32x4*>(a.lutp + (long)srows[u / (kFlLut / 2)] * kLut + (u % (kFlLut / 2)) * 2);
    }
    long sr[RT]; float facr[RT]; const unsigned char* yrow[RT]; unsigned lp_ad[RT];
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        const int myrow = wave * (32 * RT) + 32 * r + l31;
        sr[r] = srows[myrow];
        facr[r] = a.fac ? a.fac[sr[r]] : 1.f;
        yrow[r] = a.c.yc + sr[r] * a.c.ldc + 64 * hi;
        lp_ad[r] = (unsigned)(size_t)(__attribute__((address_space(3))) uint2*)lutl + (unsigned)myrow * (kFlLutLd * 8);
    }
    f32x16 acc[RT][NTL];
#pragma unroll
    for (int r = 0; r < RT; ++r)
#pragma unroll
        for (int t = 0; t < NTL; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[r][t][e] = 0.f;
    u32x4 cq[RT][4], cn[RT][4], wr[UPT];
    // every load of the loop is unconditional (addresses clamped, results selected): the loads of a step then form one
    // straight queue -- tile first, counts after -- and the wait before the tile's LDS store leaves the counts in flight
    const int ss_last = a.n_ms / 2 - 1;
    auto codes_ok = [&](int ss, int i) __attribute__((always_inline)) {
        return (long)min(ss, ss_last) * (2 * kFlMS) + 16 * i + 64 * hi + 16 <= a.c.ldc;
    };
    auto load_codes = [&](int ss) __attribute__((always_inline)) {        // raw: take_codes masks them
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int i = 0; i < 4; ++i)
                cn[r][i] = *reinterpret_cast<const u32x4*>(yrow[r] + (codes_ok(ss, i) ? (long)min(ss, ss_last) * (2 * kFlMS) + 16 * i : -64L * hi));
    };
    auto take_codes = [&](int ss) __attribute__((always_inline)) {        // the first use of the loaded bytes
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = codes_ok(ss, i);
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int k = 0; k < 4; ++k) cq[r][i][k] = ok ? cn[r][i][k] : 0u;
        }
    };
    auto load_w = [&](int ms) __attribute__((always_inline)) {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.WP + (long)min(ms, a.n_ms - 1) * MSE);
#pragma unroll
        for (int i = 0; i < UPT; ++i)
            if ((i + 1) * NT <= UNITS || tid + i * NT < UNITS) wr[i] = src[tid + i * NT];
    };
    auto store_w = [&](int b) __attribute__((always_inline)) {
        u32x4* dst = reinterpret_cast<u32x4*>(wl[b]);
#pragma unroll
        for (int i = 0; i < UPT; ++i)
            if ((i + 1) * NT <= UNITS || tid + i * NT < UNITS) dst[tid + i * NT] = wr[i];
    };
    using Half0 = std::integral_constant<int, 0>;
    using Half1 = std::integral_constant<int, 1>;
    if (ms0 < ms1) { load_codes(ms0 >> 1); load_w(ms0); }
#pragma unroll
    for (int k = 0; k < NU; ++k) {
        const int u = tid + NT * k, row = u / (kFlLut / 2), seg = u % (kFlLut / 2);
        lutl[row * kFlLutLd + seg * 2] = make_uint2(lv[k][0], lv[k][1]);
        lutl[row * kFlLutLd + seg * 2 + 1] = make_uint2(lv[k][2], lv[k][3]);
    }
    if (ms0 < ms1) { store_w(0); take_codes(ms0 >> 1); }
    // ---- software pipeline over the K steps (16 genes; four per macro step): while the 6 NTL RT matrix instructions of K step
    // k run, the wave looks up the 8 RT values of K step k + 1, reads its W fragments and repacks the entries -- its OWN LDS /
    // vector instructions between its own matrix instructions (the other wave of the SIMD does not hide them: round 5, DESIGN
    // 4.3).  The LDS reads are instruction statements in the order written; each matrix instruction's accumulator is tied to
    // an empty statement so that the compiler keeps it between them.  One barrier per macro step, between its K steps 2 and
    // 3: K step 3 already prepares K step 0 of the next macro step from the other W tile.
    const unsigned wl_ad = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)wl[0] + (unsigned)(hi * 32 + l31) * 16u;
    u32x4 A0[RT][3], A1[RT][3], Bf0[NTL][3], Bf1[NTL][3];
    u32x2 ent[RT][8];
    // the formula for the values of one K step of row tile r beyond the table (rare; `gene0` = gene of the step's first value)
    auto beyond_table = [&](auto rc, unsigned d0, unsigned d1, int gene0) __attribute__((always_inline)) {
        constexpr int r = decltype(rc)::value;
        unsigned bad = 0u;
#pragma unroll
        for (int j = 0; j < 8; ++j) bad |= ((((j >> 2) ? d1 : d0) >> (8 * (j & 3) + 5)) & 7u) != 0u ? 1u << j : 0u;
#pragma unroll 1
        while (__ballot(bad != 0u)) {
            const bool on = bad != 0u;
            const int j = on ? __builtin_ctz(bad) : 0;
            bad &= bad - 1u;
            const unsigned dw = (j >> 2) ? d1 : d0;
            const unsigned code = (dw >> (8 * (j & 3))) & 255u;
            float val = (float)code;
            if (__ballot(on && code == 255u)) {     // an escape: the count itself from the row's overflow list
                if (on && code == 255u) val = escaped_count(a.c, sr[r], gene0 + j);
            }
            float x = a.fac ? __fdiv_rn(val, facr[r]) : val;
            if (a.do_log) x = log1p_beyond_table(x);
            const uint2 e = split_entry(x);
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (on && j == k) { ent[r][k][0] = e.x; ent[r][k][1] = e.y; }
        }
    };
    // the work for K step `kn` (of the macro step whose counts are d[][], W tile `bn`), cut into twelve pieces that ride behind
    // the matrix instructions of the K step before it: piece v of prepare<KN>(...)
    auto prepare = [&](auto kn_c, auto v_c, const unsigned (&d)[RT][8], int bn, int gene0, u32x4 (&An)[RT][3], u32x4 (&Bn)[NTL][3]) __attribute__((always_inline)) {
        constexpr int KN = decltype(kn_c)::value, V = decltype(v_c)::value;
        const unsigned wad = wl_ad + (unsigned)bn * (MSE * 2);
        if constexpr (V < 4) {                       // lookups 2 V, 2 V + 1 of every row tile
#pragma unroll
            for (int r = 0; r < RT; ++r)
#pragma unroll
                for (int j = 2 * V; j < 2 * V + 2; ++j) {
                    const unsigned idx = (d[r][2 * KN + (j >> 2)] >> (8 * (j & 3))) & (unsigned)(kFlLut - 1);
                    lds_read_b64<0>(ent[r][j], lp_ad[r] + idx * 8u);
                }
        } else if constexpr (V < 7) {                // W fragments 2 (V - 4), 2 (V - 4) + 1 of 3 NTL
            constexpr int f0 = 2 * (V - 4), f1 = f0 + 1;
            if constexpr (f0 < 3 * NTL) lds_read_b128<(((f0 % 3) * NTL + f0 / 3) * 4 + KN) * 1024>(Bn[f0 / 3][f0 % 3], wad);
            if constexpr (f1 < 3 * NTL) lds_read_b128<(((f1 % 3) * NTL + f1 / 3) * 4 + KN) * 1024>(Bn[f1 / 3][f1 % 3], wad);
        } else if constexpr (V == 7) {               // the entries are in registers (the W fragments behind them may be in flight)
#define DCA_ENT8(r) "+v"(ent[r][0]), "+v"(ent[r][1]), "+v"(ent[r][2]), "+v"(ent[r][3]), "+v"(ent[r][4]), "+v"(ent[r][5]), "+v"(ent[r][6]), "+v"(ent[r][7])
            if constexpr (NTL == 2) asm volatile("s_waitcnt lgkmcnt(6)" : DCA_ENT8(0));
            else asm volatile("s_waitcnt lgkmcnt(3)" : DCA_ENT8(0));
            if constexpr (RT == 2) asm volatile("" : DCA_ENT8(RT - 1));
#undef DCA_ENT8
#ifndef DCA_EXP_FWD_NOFORMULA
            static_for<RT>([&](auto rc) __attribute__((always_inline)) {
                constexpr int r = decltype(rc)::value;
                if (__ballot(((d[r][2 * KN] | d[r][2 * KN + 1]) & 0xe0e0e0e0u) != 0u)) beyond_table(rc, d[r][2 * KN], d[r][2 * KN + 1], gene0 + 8 * KN);
            });
#endif
        } else {                                     // V = 8 .. 11: three of the twelve operand registers of every row tile
#pragma unroll
            for (int r = 0; r < RT; ++r) {
#pragma unroll
                for (int x = 3 * (V - 8); x < 3 * (V - 8) + 3; ++x) {
                    const int q = x >> 2, jj = x & 3;
                    An[r][q][jj] = __builtin_amdgcn_perm(ent[r][2 * jj + 1][q == 2 ? 1 : 0], ent[r][2 * jj][q == 2 ? 1 : 0], q == 1 ? 0x07060302u : 0x05040100u);
                }
                if constexpr (V == 9) asm volatile("" : "+v"(An[r][0]));
                if constexpr (V == 10) asm volatile("" : "+v"(An[r][1]));
                if constexpr (V == 11) asm volatile("" : "+v"(An[r][2]));
            }
        }
    };
    // K step K of the current macro step: its 6 NTL RT matrix instructions (operands Ac, Bc), the next K step's preparation between
    auto kstep = [&](auto kn_c, const unsigned (&dn)[RT][8], int bn, int gene0n, u32x4 (&Ac)[RT][3], u32x4 (&Bc)[NTL][3], u32x4 (&An)[RT][3], u32x4 (&Bn)[NTL][3]) __attribute__((always_inline)) {
        // (the W fragments of this K step: the only LDS reads still in flight)
        if constexpr (NTL == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bc[0][0]), "+v"(Bc[0][1]), "+v"(Bc[0][2]), "+v"(Bc[1][0]), "+v"(Bc[1][1]), "+v"(Bc[1][2]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bc[0][0]), "+v"(Bc[0][1]), "+v"(Bc[0][2]));
        constexpr int NS = 6 * NTL * RT;
        static_for<NS>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value, T = i % NTL, R = (i / NTL) % RT, PR = i / (NTL * RT);
            constexpr int PA_[6] = {2, 1, 0, 1, 0, 0}, PB_[6] = {0, 1, 2, 0, 1, 0};
#ifdef DCA_EXP_FWD_NOMFMA
            acc[R][T][PR] += __uint_as_float(Ac[R][PA_[PR]][0] ^ Bc[T][PB_[PR]][1]);
#else
            acc[R][T] = MFMA16(Ac[R][PA_[PR]], Bc[T][PB_[PR]], acc[R][T]);
#endif
            // (one wave per SIMD: the accumulators live in the AGPR half of its 512 registers -- a "+v" tie would copy them
            // to vector registers and back around every matrix instruction)
            if constexpr (RT == 2) asm volatile("" : "+a"(acc[R][T])); else asm volatile("" : "+v"(acc[R][T]));
            constexpr int v0 = i * 12 / NS, v1 = (i + 1) * 12 / NS;          // pieces [v0, v1) behind this matrix instruction
            static_for<v1 - v0>([&](auto jc) __attribute__((always_inline)) {
                prepare(kn_c, std::integral_constant<int, v0 + decltype(jc)::value>{}, dn, bn, gene0n, An, Bn);
            });
        });
    };
    auto codes_of = [&](int h, unsigned (&d)[RT][8]) __attribute__((always_inline)) {
#pragma unroll
        for (int r = 0; r < RT; ++r)
#pragma unroll
            for (int k = 0; k < 4; ++k) { d[r][k] = cq[r][2 * h][k]; d[r][4 + k] = cq[r][2 * h + 1][k]; }
    };
    using K0 = std::integral_constant<int, 0>; using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>; using K3 = std::integral_constant<int, 3>;
    // one macro step (half h of its super step; W tile h of the LDS pair: ms0 is even).  On entry the operands of its K step 0
    // are in A0 / Bf0 (fragments possibly in flight) and tile h is complete in LDS.
    auto step = [&](auto half, int ms) __attribute__((always_inline)) {
        constexpr int h = decltype(half)::value, b = h;
        load_w(ms + 1);                     // (past the chunk's end: a tile nobody reads)
        if (h == 0) load_codes((ms >> 1) + 1);
        unsigned d[RT][8], dn[RT][8];
        codes_of(h, d);
        const int gene0 = (ms >> 1) * (2 * kFlMS) + 64 * hi + 32 * h;
        kstep(K1{}, d, b, gene0, A0, Bf0, A1, Bf1);
        kstep(K2{}, d, b, gene0, A1, Bf1, A0, Bf0);
        kstep(K3{}, d, b, gene0, A0, Bf0, A1, Bf1);
        store_w(b ^ 1);
        if (h == 1) take_codes((ms >> 1) + 1);              // the next super step's counts (requested a super step ago)
        codes_of(h ^ 1, dn);
#ifndef DCA_EXP_FWD_NOBARRIER
        __syncthreads();                    // tile b ^ 1 is in LDS; everyone's reads of tile b are in registers
#endif
        kstep(K0{}, dn, b ^ 1, (((ms + 1) >> 1) * (2 * kFlMS)) + 64 * hi + 32 * (h ^ 1), A1, Bf1, A0, Bf0);
    };
    if (ms0 < ms1) {                        // the operands of the first K step (no matrix instructions to put them behind)
        __syncthreads();                    // tile 0 and the tables are in LDS
        unsigned d[RT][8];
        codes_of(0, d);
        const int gene0 = (ms0 >> 1) * (2 * kFlMS) + 64 * hi;
        static_for<12>([&](auto vc) __attribute__((always_inline)) { prepare(K0{}, vc, d, 0, gene0, A0, Bf0); });
    }
#pragma unroll 1
    for (int ms = ms0; ms < ms1; ms += 2) {
        step(Half0{}, ms);
        step(Half1{}, ms + 1);
    }
    if (ms0 < ms1) {                        // (the last K step prepared operands nobody uses: their reads must not outlive the loop)
        if constexpr (NTL == 2) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bf0[0][0]), "+v"(Bf0[0][1]), "+v"(Bf0[0][2]), "+v"(Bf0[1][0]), "+v"(Bf0[1][1]), "+v"(Bf0[1][2]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Bf0[0][0]), "+v"(Bf0[0][1]), "+v"(Bf0[0][2]));
    }
    if (ms0 >= ms1) __syncthreads();        // (an empty chunk never passed a barrier after csum was written)
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        float* const dst = a.P + ((long)blockIdx.y * a.Bp + rg0 + wave * (32 * RT) + 32 * r) * H1;
#pragma unroll
        for (int t = 0; t < NTL; ++t) {
            const float c0 = csum[32 * t + l31];
#pragma unroll
            for (int e = 0; e < 16; ++e) dst[(long)rowmap(e, hi) * H1 + 32 * t + l31] = acc[r][t][e] + c0;
        }
    }
}

// Z[r, :] = b_eff + the chunk partials in order
__global__ __launch_bounds__(256) void enc0_fwd_reduce_kernel(const float* P, long Bp, int nsk, const float* beff, int B,
                                                              int H1, float* Z, long ldz) {
    const int hq = H1 >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * hq) return;
    const int r = (int)(idx / hq), j = (int)(idx - (long)r * hq) * 4;
    float4 v = beff ? *reinterpret_cast<const float4*>(beff + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k0 = 0; k0 < nsk; k0 += 8) {                 // batches of 8 independent loads (one round trip each), added in chunk order
        float4 t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = *reinterpret_cast<const float4*>(P + ((long)(k0 + u < nsk ? k0 + u : 0) * Bp + r) * H1 + j);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + u < nsk) { v.x += t[u].x; v.y += t[u].y; v.z += t[u].z; v.w += t[u].w; }
    }
    *reinterpret_cast<float4*>(Z + (long)r * ldz + j) = v;
}

inline bool fl_width_ok(int H1) { return H1 == 32 || H1 == 64; }
inline int fl_row_groups(int B) { return (B + kFlRows - 1) / kFlRows; }
inline int fl_n_ms(int G) { return 2 * ((G + 2 * kFlMS - 1) / (2 * kFlMS)); }        // whole super steps
// gene chunks (whole super steps): about one workgroup per CU in all
inline int fl_ms_per(int B, int G) {
    int nsk = 256 / fl_row_groups(B);
    if (nsk > 32) nsk = 32;
    if (nsk < 1) nsk = 1;
    const int n_ss = fl_n_ms(G) / 2;
    return 2 * ((n_ss + nsk - 1) / nsk);
}

inline bool width_ok(int H1) { return H1 == 16 || H1 == 32 || H1 == 64 || H1 == 128 || H1 == 256; }

inline bool dw_width_ok(int H1) { return H1 == 32 || H1 == 64 || H1 == 128; }

// row splits: as many workgroups as are resident at once (or just below): one round
// 64 units: the ring form (enc0_dw2_kernel) from kD2MinRows batch rows up (below, its longer prologue and the 512-gene
// workgroups cost more than the pipeline saves: tools/ab_enc0_dw.py); the `form` argument of dcahip_enc0_dw_sparse (1 / 2)
// forces the first / the ring form at every size (A/B runs, tests), 0 = by the shape
constexpr int kD2MinRows = 1024;
inline bool dw_second_form(int H1, int B, int form) { return H1 == 64 && (form == 2 || (form == 0 && B >= kD2MinRows)); }

inline int dw_splits(int B, int G, int H1, int form) {
    if (dw_second_form(H1, B, form)) return dw2_splits(B, G);
    const int groups = (G + 32 * kDwWaves - 1) / (32 * kDwWaves);
    int ns = 256 / groups;                               // one 8-wave workgroup per CU (registers)
    const int maxs = (B + kDwRB - 1) / kDwRB;
    if (ns > maxs) ns = maxs;
    if (ns > 16) ns = 16;
    if (ns < 1) ns = 1;
    const int need = (B + kDwMaxRS - 1) / kDwMaxRS;       // a split's storage rows sit in LDS
    if (ns < need) ns = need;
    return ns;
}
inline int dw_rows_per_split(int B, int ns, int H1, int form) {
    if (dw_second_form(H1, B, form)) return dw2_rows_per_split(B, ns);
    return (((B + ns - 1) / ns) + kDwRB - 1) / kDwRB * kDwRB;
}
inline long r16(long x) { return (x + 15) / 16 * 16; }


// (The byte-store weight gradient for batches of at most 64 rows measured 11.2 us against the GEMM's 7.4 us at batch 32
// (profiles/r05g_*): an experiment build, -DDCA_EXP_DW_SMALL.)
#ifdef DCA_EXP_DW_SMALL
// ------------------------------------------------------------------------------------------------- small batches
// The first layer's weight gradient at the reference's default batch (32 rows, dca/train.py:37; up to 64 here) straight
// from the byte store: dW0[g, :] = (sum over the batch rows with a NON-ZERO count of f(y / fac) dZ[r, :] - mean[g] colsum(dZ)) / std[g]
// -- about two terms per gene at 93 % zeros -- instead of a rank-32 update through the GEMM (14 us + a split-K reduce at
// G = 20 000).  A group of 16 lanes owns one gene (lane t: hidden units 4 t .. 4 t + 3: one 16-byte store), a workgroup
// kGenesSmall genes; dZ, its column sums, the storage rows and the per-cell divisors sit in LDS.  fp32 FMAs in row order:
// deterministic.  Row G of gW = colsum(dZ) (the bias gradient, as dcahip_sgemm's colsum_row).  The kernel is bound by
// memory round trips and instruction fetch, not by work: the workgroup requests its 16 x B count bytes in one round trip into
// LDS and walks them in ONE compact loop body (measured at G = 20 000, batch 32: four genes per group with eight requests in
// flight 44 us; all 32 requests at once but unrolled bodies with libm's log1pf 35 us, with the fast logarithm 22 us; the GEMM
// + split-K reduce it replaces 19 us).
constexpr int kSmallRows = 64;
constexpr int kGenesSmall = 16;          // per workgroup: one gene per group of 16 lanes

struct DwSmallArgs {
    Compact c;
    const float* fac; int do_log;
    const float* mean; const float* stdv;
    const int* perm; const long long* cursor; long row_base;
    int B, G, H1;
    const float* dZ; long ldz;
    float* gW; long ldg;
};

__global__ __launch_bounds__(256) void enc0_dw_small_kernel(DwSmallArgs a) {
    __shared__ __attribute__((aligned(16))) float dz[kSmallRows * 64];
    __shared__ float cs[64];
    __shared__ int srow[kSmallRows];
    __shared__ float rfac[kSmallRows];
    __shared__ unsigned char codes[kSmallRows][kGenesSmall];
    const int tid = threadIdx.x;
    const long cur = a.cursor ? (long)*a.cursor : 0;
    const int H4 = a.H1 >> 2;
    const int g0 = blockIdx.x * kGenesSmall;
    // every request of the workgroup in ONE memory round trip behind the row indices: thread (row, gene) takes the count
    // byte of its pair straight away (its row index from memory, not from LDS), rows 0 .. 15, then 16 .. 31, ...
    {
        const int gl = tid & 15;
        const int gene = g0 + gl < a.G ? g0 + gl : a.G - 1;
        for (int r = tid >> 4; r < a.B; r += 16) {
            const long rr = cur + a.row_base + r;
            const int sr = a.perm ? a.perm[rr] : (int)rr;
            codes[r][gl] = a.c.yc[(unsigned long long)(unsigned)sr * (unsigned long long)a.c.ldc + gene];
            if (gl == 0) { srow[r] = sr; rfac[r] = a.fac ? 1.f / a.fac[sr] : 1.f; }
        }
    }
    for (int i = tid; i < a.B * 16; i += 256) {
        const int r = i >> 4, q = i & 15;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < H4) v = *reinterpret_cast<const float4*>(a.dZ + (long)r * a.ldz + 4 * q);
        *reinterpret_cast<float4*>(dz + r * 64 + 4 * q) = v;
    }
    __syncthreads();
    if (tid < 64) {
        float v = 0.f;
        for (int r = 0; r < a.B; ++r) v += dz[r * 64 + tid];
        cs[tid] = v;
    }
    __syncthreads();
    const int t = tid & 15, grp = tid >> 4;
    if (blockIdx.x == 0 && tid < H4)           // the bias gradient
        *reinterpret_cast<float4*>(a.gW + (long)a.G * a.ldg + 4 * tid) = *reinterpret_cast<const float4*>(cs + 4 * tid);
    const int gene = g0 + grp;
    if (t >= H4 || gene >= a.G) return;
    const float4 c4 = *reinterpret_cast<const float4*>(cs + 4 * t);
    const float m = a.mean ? a.mean[gene] : 0.f;
    const float sd = a.stdv ? a.stdv[gene] : 1.f;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 1
    for (int r = 0; r < a.B; ++r) {             // (one compact body: the unrolled form spent its time fetching instructions)
        const unsigned code = codes[r][grp];
        if (code == 0u) continue;
        float val = (float)code;
        if (code == 255u) val = escaped_count(a.c, srow[r], gene);
        float L = val * rfac[r];
        if (a.do_log) {
            // log1p on the transcendental unit (v_log_f32, 1 ulp) with Kahan's exact-ratio correction of the rounded 1 + L:
            // ~2e-7 relative for the arguments counts produce (L >= 1 / fac)
            const float u = 1.f + L, d1 = u - 1.f;
            const float lg = __builtin_amdgcn_logf(u) * 0.69314718055994531f;
            L = d1 == 0.f ? L : lg * (L * __builtin_amdgcn_rcpf(d1));
        }
        const float4 d = *reinterpret_cast<const float4*>(dz + r * 64 + 4 * t);
        acc.x = fmaf(L, d.x, acc.x); acc.y = fmaf(L, d.y, acc.y); acc.z = fmaf(L, d.z, acc.z); acc.w = fmaf(L, d.w, acc.w);
    }
    const float is = 1.f / sd;
    float4 o;
    o.x = (acc.x - m * c4.x) * is; o.y = (acc.y - m * c4.y) * is; o.z = (acc.z - m * c4.z) * is; o.w = (acc.w - m * c4.w) * is;
    *reinterpret_cast<float4*>(a.gW + (long)gene * a.ldg + 4 * t) = o;
}

#endif  // DCA_EXP_DW_SMALL

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

#ifdef DCA_EXP_DW_SMALL
extern "C" int dcahip_enc0_dw_small_max_rows(void) { return kSmallRows; }

extern "C" int dcahip_enc0_dw_small(const unsigned char* Yc, long ldc, const int* ovf_ptr, const int* ovf_col,
                                    const float* ovf_val, const float* fac, int do_log, const float* mean, const float* stdv,
                                    const int* perm, const long long* cursor, long row_base, int B, int G, int H1,
                                    const float* dZ, long ldz, float* gW, long ldg, void* stream) {
    if (!Yc || !dZ || !gW || B <= 0 || B > kSmallRows || G <= 0 || H1 <= 0 || H1 > 64 || (H1 & 3) || ldc < G || ldz < H1 || ldg < H1 ||
        (ldz & 3) || (ldg & 3) || ((reinterpret_cast<uintptr_t>(dZ) | reinterpret_cast<uintptr_t>(gW)) & 15))
        return DCAHIP_EINVAL;
    DwSmallArgs a{Compact{Yc, ldc, ovf_ptr, ovf_col, ovf_val}, fac, do_log, mean, stdv, perm, cursor, row_base, B, G, H1, dZ, ldz, gW, ldg};
    hipLaunchKernelGGL(enc0_dw_small_kernel, dim3((unsigned)((G + kGenesSmall - 1) / kGenesSmall)), dim3(256), 0,
                       (hipStream_t)stream, a);
    return (int)hipGetLastError();
}

#endif  // DCA_EXP_DW_SMALL

extern "C" int dcahip_enc0_sparse_supported(int H1) { return dw_width_ok(H1) ? 1 : 0; }

extern "C" int dcahip_enc0_lut(const float* fac, int do_log, int n, void* lutp, void* stream) {
    if (n <= 0 || !lutp) return DCAHIP_EINVAL;
    hipLaunchKernelGGL(enc0_lut_kernel, dim3((unsigned)(((long)n * kLut + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       fac, do_log, n, static_cast<uint2*>(lutp));
    return (int)hipGetLastError();
}

// workspace: P [NS][Gs][H1] | Sp [NS][H1] | Spp [steps][H1] | DZP [steps][...] bf16
extern "C" long dcahip_enc0_dw_sparse_workspace_bytes(int B, int G, int H1) {
    if (!dw_width_ok(H1) || B <= 0 || G <= 0) return 0;
    // (sufficient for either form of the 64-unit kernel, whichever the launch's `form` argument names)
    int ns = dw_splits(B, G, H1, 1);
    if (H1 == 64) { const int n2 = dw2_splits(B, G); if (n2 > ns) ns = n2; }
    const long Gs = ((long)G + 511) / 512 * 512;
    const long steps = (B + kKS - 1) / kKS;
    return r16(((long)ns * Gs * H1 + (long)ns * H1) * 4) + r16(steps * H1 * 4) + r16(steps * (long)dz_step_elems(H1) * 2) + r16((long)B * 4);
}


#ifdef DCA_DW_TIMING
extern "C" void dcahip_enc0_dw_set_timing(long long* buf) { hipMemcpyToSymbol(HIP_SYMBOL(g_dw_timing), &buf, sizeof(buf)); }
#endif
extern "C" int dcahip_enc0_lut_entries(void) { return kLut; }

extern "C" int dcahip_enc0_dw_sparse(const unsigned char* Yc, long ldc, const int* ovf_ptr, const int* ovf_col,
                                     const float* ovf_val, const float* fac, int do_log, const void* lutp, const float* mean,
                                     const float* stdv, const int* perm, const long long* cursor, long row_base,
                                     int B, int G, int H1, const float* dZ, long ldz, float* gW, long ldg,
                                     void* workspace, long workspace_bytes, int form, void* stream) {
    if (!dw_width_ok(H1) || B <= 0 || G <= 0 || !Yc || !lutp || (ldc & 15) || ldc < G || !dZ || ldz < H1 || !gW ||
        ldg < H1 || (reinterpret_cast<uintptr_t>(workspace) & 15) || !workspace || form < 0 || form > 2)
        return DCAHIP_EINVAL;
    if (workspace_bytes < dcahip_enc0_dw_sparse_workspace_bytes(B, G, H1)) return DCAHIP_EINVAL;
    const int ns = dw_splits(B, G, H1, form);
    if (dw_second_form(H1, B, form) && dw_rows_per_split(B, ns, H1, form) > kD2MaxRS) return DCAHIP_EINVAL;
    const long Gs = ((long)G + 511) / 512 * 512;
    const long steps = (B + kKS - 1) / kKS;
    char* wsb = static_cast<char*>(workspace);
    DwArgs a;
    a.c = Compact{Yc, ldc, ovf_ptr, ovf_col, ovf_val};
    a.fac = fac; a.do_log = do_log; a.lutp = static_cast<const uint2*>(lutp);
    a.B = B; a.G = G;
    a.P = reinterpret_cast<float*>(wsb); a.Gs = Gs; a.Sp = a.P + (long)ns * Gs * H1;
    float* Spp = reinterpret_cast<float*>(wsb + r16(((long)ns * Gs * H1 + (long)ns * H1) * 4));
    unsigned short* DZP = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(Spp) + r16(steps * H1 * 4));
    int* srowb = reinterpret_cast<int*>(reinterpret_cast<char*>(DZP) + r16(steps * (long)dz_step_elems(H1) * 2));
    a.DZP = DZP; a.Spp = Spp; a.srowb = srowb;
    a.RS = dw_rows_per_split(B, ns, H1, form);
    const dim3 grid((unsigned)((G + 32 * kDwWaves - 1) / (32 * kDwWaves)), (unsigned)ns);
    const dim3 block(64 * kDwWaves);
    hipStream_t s = (hipStream_t)stream;
    switch (H1) {
        case 32:
            hipLaunchKernelGGL(enc0_split_dz_kernel<32>, dim3((unsigned)steps), dim3(256), 0, s, dZ, ldz, B, perm, cursor, row_base, DZP, Spp, srowb);
            hipLaunchKernelGGL(enc0_dw_kernel<32>, grid, block, 0, s, a); break;
        case 64:
            hipLaunchKernelGGL(enc0_split_dz_kernel<64>, dim3((unsigned)steps), dim3(256), 0, s, dZ, ldz, B, perm, cursor, row_base, DZP, Spp, srowb);
            if (dw_second_form(H1, B, form))
                hipLaunchKernelGGL(enc0_dw2_kernel, dim3((unsigned)((G + kD2Genes - 1) / kD2Genes), (unsigned)ns), dim3(64 * kD2Waves), 0, s, a);
            else
                hipLaunchKernelGGL(enc0_dw_kernel<64>, grid, block, 0, s, a);
            break;
        default:
            hipLaunchKernelGGL(enc0_split_dz_kernel<128>, dim3((unsigned)steps), dim3(256), 0, s, dZ, ldz, B, perm, cursor, row_base, DZP, Spp, srowb);
            hipLaunchKernelGGL(enc0_dw_kernel<128>, grid, block, 0, s, a); break;
    }
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    DwFinishArgs f{a.P, Gs, a.Sp, ns, mean, stdv, G, H1, gW, ldg};
    const long total = (long)(G + 1) * (H1 / 4);
    hipLaunchKernelGGL(enc0_dw_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, f);
    return (int)hipGetLastError();
}

#ifdef DCA_EXP_ENC0_SPARSE_FWD
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

#endif  // DCA_EXP_ENC0_SPARSE_FWD

// 32-row tiles per wave of the matrix-pipe forward: 1 = eight waves (the product).  2 = four waves of 64 rows: built,
// bit-identical, measured SLOWER (0.080 vs 0.067 ms) -- an experiment build (-DDCA_EXP_FWD_FORM2), not a switch.
#ifdef DCA_EXP_FWD_FORM2
constexpr int kFlRT = 2;
#else
constexpr int kFlRT = 1;
#endif

// workspace: WP | C0P | P
extern "C" long dcahip_enc0_fwd_lut_workspace_bytes(int B, int G, int H1) {
    if (!fl_width_ok(H1) || B <= 0 || G <= 0) return 0;
    const int ms_per = fl_ms_per(B, G), n_ms = fl_n_ms(G);
    const long nsk = (n_ms + ms_per - 1) / ms_per, Bp = (long)fl_row_groups(B) * kFlRows;
    return r16((long)n_ms * fl_ms_elems(H1) * 2) + r16((long)n_ms * H1 * 8) + r16(nsk * Bp * H1 * 4);
}

extern "C" int dcahip_enc0_fwd_lut(const unsigned char* Yc, long ldc, const int* ovf_ptr, const int* ovf_col,
                                   const float* ovf_val, const float* fac, int do_log, const void* lutp, const float* mean,
                                   const float* stdv, const int* perm, const long long* cursor, long row_base,
                                   int B, int G, int H1, const float* W, long ldw, const float* bias,
                                   float* Z, long ldz, void* workspace, long workspace_bytes, void* stream) {
    if (!fl_width_ok(H1) || B <= 0 || G <= 0 || !Yc || !lutp || (ldc & 15) || ldc < G || !W || ldw < H1 || !Z || (ldz & 3) ||
        ldz < H1 || (reinterpret_cast<uintptr_t>(Z) & 15) || (reinterpret_cast<uintptr_t>(workspace) & 15) || !workspace)
        return DCAHIP_EINVAL;
    if (workspace_bytes < dcahip_enc0_fwd_lut_workspace_bytes(B, G, H1)) return DCAHIP_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    char* wsb = static_cast<char*>(workspace);
    const int ms_per = fl_ms_per(B, G), n_ms = fl_n_ms(G);
    const int nsk = (n_ms + ms_per - 1) / ms_per;
    const long Bp = (long)fl_row_groups(B) * kFlRows;
    unsigned short* WP = reinterpret_cast<unsigned short*>(wsb);
    double* C0P = reinterpret_cast<double*>(wsb + r16((long)n_ms * fl_ms_elems(H1) * 2));
    float* P = reinterpret_cast<float*>(reinterpret_cast<char*>(C0P) + r16((long)n_ms * H1 * 8));
    FlArgs f;
    f.c = Compact{Yc, ldc, ovf_ptr, ovf_col, ovf_val};
    f.fac = fac; f.do_log = do_log; f.lutp = static_cast<const uint2*>(lutp);
    f.perm = perm; f.cursor = cursor; f.row_base = row_base; f.B = B; f.G = G;
    f.WP = WP; f.C0P = mean ? C0P : nullptr; f.P = P; f.Bp = Bp; f.n_ms = n_ms; f.ms_per = ms_per;
    const dim3 grid((unsigned)fl_row_groups(B), (unsigned)nsk);
    if (H1 == 32) {
        hipLaunchKernelGGL(enc0_wsplit_kernel<32>, dim3((unsigned)n_ms, 1), dim3(256), 0, s, W, ldw, mean, stdv, G, WP, C0P);
        hipLaunchKernelGGL((enc0_fwd_lut_kernel<32, kFlRT>), grid, dim3(512 / kFlRT), 0, s, f);
    } else {
        hipLaunchKernelGGL(enc0_wsplit_kernel<64>, dim3((unsigned)n_ms, 2), dim3(256), 0, s, W, ldw, mean, stdv, G, WP, C0P);
        hipLaunchKernelGGL((enc0_fwd_lut_kernel<64, kFlRT>), grid, dim3(512 / kFlRT), 0, s, f);
    }
    int rc = (int)hipGetLastError();
    if (rc) return rc;
    const long total = (long)B * (H1 / 4);
    hipLaunchKernelGGL(enc0_fwd_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, P, Bp, nsk,
                       bias, B, H1, Z, ldz);
    return (int)hipGetLastError();
}
