// K-GEMM: fp32 GEMM on the gfx950 matrix pipe (v_mfma_f32_32x32x2_f32: exact fp32, bitwise a
// k-ordered fmaf chain), LDS-tiled, 4 waves per workgroup, register-staged prefetch of the
// next K-chunk, deterministic split-K through a workspace + ordered reduce.
//
// One template covers the three operand layouts the Dense layers of the autoencoder need
// (dca/network.py:124-126, 369-380 and their autodiff):
//   NN  C = A  B      forward             A [M,K] k-contiguous, B [K,N] n-contiguous
//   TN  C = A^T B     weight gradient     A [K,M] m-contiguous, B [K,N] n-contiguous
//   NT  C = A  B^T    input gradient      A [M,K] k-contiguous, B [N,K] k-contiguous
// The minibatch gather (storage row = perm[*cursor + r]) is folded into the A loads, so the
// shuffled batch of the resident [n_cells, n_genes] matrix is never copied.
//
// Three kernel families live here (all fp32 results):
//   gemm_kernel       exact fp32 MFMA (the yardstick; TN and narrow NN shapes of the 64-unit networks)
//   gemm_x3_kernel    operands split into three bf16 pieces inside the K loop, six products on the bf16 pipe
//   gemm_p3_kernel / gemm_p3w_kernel   the same six products from operands that arrive PRE-SPLIT ("planes"):
//                     128 x 128 register-staged tiles, and 256 x 256 tiles fed global -> LDS by DMA through a
//                     three-stage ring (the wide networks' step)
//
// LDS image: As[k][m], Bs[k][n] (k-major).  A wave's MFMA fragment read is then 32
// consecutive floats per half-wave: conflict-free ds_read_b32.  k-contiguous sources are
// transposed on the way in (4 x ds_write_b32, odd row stride -> conflict-free), m/n-
// contiguous sources are stored with one aligned ds_write_b128 per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "dcahip.h"
#include "h2_math.hpp"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    const int* perm;
    const long long* cursor;
    float* ws;
    long lda, ldb, ldc;
    int M, N, K;
    int split, kslab;
    int colsum;
    int mtiles, ntiles;
};

struct RowMap {
    const int* perm;
    long long cur;
    __device__ __forceinline__ long operator()(int r) const {
        return perm ? (long)perm[cur + r] : (long)(cur + r);
    }
};
struct Identity {
    __device__ __forceinline__ long operator()(int r) const { return r; }
};

// ---- tile loaders: global -> registers, registers -> LDS --------------------------------
// KContig: source row index = m/n (ROWS per tile), contiguous along k.
template <int ROWS, int BK, int V>
struct KContig {
    static constexpr int TK = BK / V;            // threads along k
    static constexpr int RPP = 256 / TK;         // rows per pass
    static constexpr int PASSES = ROWS / RPP;
    float r[PASSES][V];
    long off[PASSES];                            // storage offset of this thread's rows: looked up ONCE per tile (the
                                                 // gather index is a load; inside load() it was a dependent round trip per chunk)
    template <class Map>
    __device__ __forceinline__ void prepare(long ld, int row0, int nrows, int, int, const Map& map) {
        const int mr = threadIdx.x / TK;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int m = row0 + mr + i * RPP;
            off[i] = map(m < nrows ? m : nrows - 1) * ld;
        }
    }
    template <class Map>
    __device__ __forceinline__ void load(const float* base, long ld, int row0, int nrows, int k0,
                                         int kend, const Map&) {
        const int t = threadIdx.x;
        const int k = k0 + (t % TK) * V;
        const int mr = t / TK;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int m = row0 + mr + i * RPP;
#pragma unroll
            for (int j = 0; j < V; ++j) r[i][j] = 0.f;
            if (m < nrows && k < kend) {
                const float* p = base + off[i] + k;
                if (V == 4 && k + 4 <= kend) {
                    const float4 v = *reinterpret_cast<const float4*>(p);
                    r[i][0] = v.x; r[i][1 % V] = v.y; r[i][2 % V] = v.z; r[i][3 % V] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < V; ++j) if (k + j < kend) r[i][j] = p[j];
                }
            }
        }
    }
    // LDS image S[k][row], row stride LDS_ (odd)
    template <int LDS_>
    __device__ __forceinline__ void store(float* S) const {
        const int t = threadIdx.x;
        const int k = (t % TK) * V;
        const int mr = t / TK;
#pragma unroll
        for (int i = 0; i < PASSES; ++i)
#pragma unroll
            for (int j = 0; j < V; ++j) S[(k + j) * LDS_ + mr + i * RPP] = r[i][j];
    }
};

// MNContig: source row index = k, contiguous along m/n (COLS per tile).
template <int COLS, int BK, int V>
struct MNContig {
    static constexpr int TC = COLS / V;          // threads along m/n
    static constexpr int KPP = 256 / TC;         // k rows per pass
    static constexpr int PASSES = (BK + KPP - 1) / KPP;
    float r[PASSES][V];
    // (the gather index of chunk c + 1 requested together with the data of chunk c was measured on the first layer's
    // weight gradient: 16 more registers, 9 -> 6 workgroups per CU, 0.158 -> 0.164 ms; not kept)
    template <class Map>
    __device__ __forceinline__ void prepare(long, int, int, int, int, const Map&) {}
    template <class Map>
    __device__ __forceinline__ void load(const float* base, long ld, int col0, int ncols, int k0,
                                         int kend, const Map& map) {
        const int t = threadIdx.x;
        const int c = col0 + (t % TC) * V;
        const int kr = t / TC;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int k = k0 + kr + i * KPP;
#pragma unroll
            for (int j = 0; j < V; ++j) r[i][j] = 0.f;
            if (kr + i * KPP < BK && k < kend && c < ncols) {
                const float* p = base + map(k) * ld + c;
                if (V == 4 && c + 4 <= ncols) {
                    const float4 v = *reinterpret_cast<const float4*>(p);
                    r[i][0] = v.x; r[i][1 % V] = v.y; r[i][2 % V] = v.z; r[i][3 % V] = v.w;
                } else {
#pragma unroll
                    for (int j = 0; j < V; ++j) if (c + j < ncols) r[i][j] = p[j];
                }
            }
        }
    }
    // LDS image S[k][col], row stride LDS_ (multiple of 4)
    template <int LDS_>
    __device__ __forceinline__ void store(float* S) const {
        const int t = threadIdx.x;
        const int c = (t % TC) * V;
        const int kr = t / TC;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int k = kr + i * KPP;
            if (k < BK) {
                if (V == 4) *reinterpret_cast<float4*>(&S[k * LDS_ + c]) = make_float4(r[i][0], r[i][1 % V], r[i][2 % V], r[i][3 % V]);
                else S[k * LDS_ + c] = r[i][0];
            }
        }
    }
};

template <bool KC, int DIM, int BK, int V> struct LoaderSel;
template <int DIM, int BK, int V> struct LoaderSel<true, DIM, BK, V> { using T = KContig<DIM, BK, V>; };
template <int DIM, int BK, int V> struct LoaderSel<false, DIM, BK, V> { using T = MNContig<DIM, BK, V>; };

template <int BM, int BN, int BK, int WGM, int WGN, bool TA, bool TB, int V>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr bool A_KC = !TA;                  // A k-contiguous unless transposed
    constexpr bool B_KC = TB;                   // B k-contiguous only when stored [N,K]
    constexpr int LDA_S = BM + (A_KC ? 1 : 0);
    constexpr int LDB_S = BN + (B_KC ? 1 : 0);
    __shared__ __attribute__((aligned(16))) float As[BK * LDA_S];
    __shared__ __attribute__((aligned(16))) float Bs[BK * LDB_S];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int id = blockIdx.x;
    const int s = id % p.split; id /= p.split;
    const int nt = id % p.ntiles;
    const int mt = id / p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kbeg = s * p.kslab;
    const int kend = min(p.K, kbeg + p.kslab);
    const RowMap amap{p.perm, p.cursor ? *p.cursor : 0};
    const Identity ident;

    typename LoaderSel<A_KC, BM, BK, V>::T la;
    typename LoaderSel<B_KC, BN, BK, V>::T lb;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool do_colsum = !TB && p.colsum && mt == 0;
    constexpr int NPH = 256 / BN > 0 ? 256 / BN : 1;   // column-sum phases (BN <= 256)
    const int cn = t % BN, cph = t / BN;
    float csum = 0.f;

    const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
    const int nchunks = (kend - kbeg + BK - 1) / BK;

    // (requesting chunk c + 2 while chunk c is multiplied -- two loader register sets -- was measured: the registers cost
    // more occupancy than the deeper prefetch returns, 0.164 -> 0.297 ms on the first layer's weight gradient)
    la.prepare(p.lda, m0, p.M, kbeg, kend, amap);
    lb.prepare(p.ldb, n0, p.N, kbeg, kend, ident);
    la.load(p.A, p.lda, m0, p.M, kbeg, kend, amap);
    lb.load(p.B, p.ldb, n0, p.N, kbeg, kend, ident);
    for (int c = 0; c < nchunks; ++c) {
        la.template store<LDA_S>(As);
        lb.template store<LDB_S>(Bs);
        __syncthreads();
        if (c + 1 < nchunks) {
            const int k0 = kbeg + (c + 1) * BK;
            la.load(p.A, p.lda, m0, p.M, k0, kend, amap);
            lb.load(p.B, p.ldb, n0, p.N, k0, kend, ident);
        }
#pragma unroll
        for (int kk = 0; kk < BK / 2; ++kk) {
            const int k = 2 * kk + (lane >> 5);
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[k * LDA_S + wm0 + i * 32 + (lane & 31)];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[k * LDB_S + wn0 + j * 32 + (lane & 31)];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (do_colsum && cph < NPH) {
#pragma unroll 4
            for (int k = cph; k < BK; k += NPH) csum += Bs[k * LDB_S + cn];
        }
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------
    const int Mo = p.M + (p.colsum ? 1 : 0);
    float* out = p.split > 1 ? p.ws + (long)s * Mo * p.N : p.C;
    const long ldo = p.split > 1 ? (long)p.N : p.ldc;
    const bool add_bias = p.split == 1 && p.bias != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + (lane & 31);
        const float bv = (add_bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M && n < p.N) out[(long)m * ldo + n] = acc[i][j][r] + bv;
            }
        }
    }
    if (do_colsum) {
        float* red = As;                          // safe: last loop iteration ended with a barrier
        if (cph < NPH) red[cph * BN + cn] = csum;
        __syncthreads();
        if (cph == 0) {
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < NPH; ++q) v += red[q * BN + cn];
            const int n = n0 + cn;
            if (n < p.N) out[(long)p.M * ldo + n] = v;
        }
    }
}


// =====================================================================================================
// fp32 GEMM on the bf16 matrix pipe: every operand element is split into three bf16 pieces (x = x1 + x2 + x3,
// round-to-nearest residuals) on its way into LDS and the six products a1b1, a1b2, a2b1, a1b3, a2b2, a3b1 are
// accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The dropped terms are below 2^-24 of sum|ab|: the result has the
// accuracy of an fp32 dot product (measured on the MI355X against the exact-fp32 MFMA above: 9.6e-8 vs 1.6e-7 of
// sum|ab| at K = 64, 1.1e-7 vs 1.2e-7 at K = 4096; tests/test_heads_fused_gpu.py::test_x3_products_are_fp32_accurate)
// at 6/16 of the matrix-pipe cycles -- the fp32 MFMA runs at the vector rate on gfx950.
//
// LDS images (per piece): a k-contiguous source is stored [row][32 k] (64-byte rows, the four 16-byte units of a
// row rotated by row >> 2: the fragment read -- one ds_read_b128 per lane, lanes = rows -- is conflict-free); an
// m/n-contiguous source is stored as it comes, [k][cols], and read with the transposing ds_read_b64_tr_b16 (row
// stride = 64 bytes modulo 128, so the four k rows of a lane group fall into different bank quarters).
// =====================================================================================================
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2v = __attribute__((ext_vector_type(2))) __bf16;
using f32x2v = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using s16x4 = __attribute__((ext_vector_type(4))) short;

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
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)

// image of one operand tile: DIM rows/cols x 32 k, three pieces
template <bool KC, int DIM>
struct X3Image {
    static constexpr int STRIDE = KC ? 64 : (DIM * 2 + ((DIM * 2) % 128 == 64 ? 0 : 64 - (DIM * 2) % 128 + ((DIM * 2) % 128 > 64 ? 128 : 0)));
    static constexpr int PIECE = KC ? DIM * 64 : 32 * STRIDE;       // bytes
    static constexpr int BYTES = 3 * PIECE;
    static_assert(KC || STRIDE % 128 == 64, "transposed-read images: row stride = 64 bytes modulo 128 (four k rows -> four bank quarters)");
    static_assert(KC || STRIDE % 8 == 0, "8-byte aligned rows");

    // ---- stores (registers of the tile loaders above -> pieces)
    template <int ROWS, int V>
    static __device__ __forceinline__ void store(unsigned char* S, const KContig<ROWS, 32, V>& l) {
        static_assert(KC, "layout");
        using L = KContig<ROWS, 32, V>;
        const int t = threadIdx.x;
        const int k = (t % L::TK) * V, mr = t / L::TK;
#pragma unroll
        for (int i = 0; i < L::PASSES; ++i) {
            const int m = mr + i * L::RPP;
            unsigned char* row = S + m * 64;
            if (V == 4) {
                unsigned a0, a1, a2, b0, b1, b2;
                split_pair(l.r[i][0], l.r[i][1 % V], a0, a1, a2);
                split_pair(l.r[i][2 % V], l.r[i][3 % V], b0, b1, b2);
                const int off = ((((k >> 3) + (m >> 2)) & 3) << 4) + (((k >> 2) & 1) << 3);
                *reinterpret_cast<u32x2*>(row + off) = u32x2{a0, b0};
                *reinterpret_cast<u32x2*>(row + PIECE + off) = u32x2{a1, b1};
                *reinterpret_cast<u32x2*>(row + 2 * PIECE + off) = u32x2{a2, b2};
            } else {
                unsigned a0, a1, a2;
                split_pair(l.r[i][0], 0.f, a0, a1, a2);
                const int off = ((((k >> 3) + (m >> 2)) & 3) << 4) + ((k & 7) << 1);
                *reinterpret_cast<unsigned short*>(row + off) = (unsigned short)a0;
                *reinterpret_cast<unsigned short*>(row + PIECE + off) = (unsigned short)a1;
                *reinterpret_cast<unsigned short*>(row + 2 * PIECE + off) = (unsigned short)a2;
            }
        }
    }
    template <int COLS, int V>
    static __device__ __forceinline__ void store(unsigned char* S, const MNContig<COLS, 32, V>& l, float (&csum)[V], bool colsum) {
        static_assert(!KC, "layout");
        using L = MNContig<COLS, 32, V>;
        const int t = threadIdx.x;
        const int c = (t % L::TC) * V, kr = t / L::TC;
#pragma unroll
        for (int i = 0; i < L::PASSES; ++i) {
            const int k = kr + i * L::KPP;
            if (k < 32) {
                unsigned char* row = S + k * STRIDE + c * 2;
                if (colsum) {
#pragma unroll
                    for (int j = 0; j < V; ++j) csum[j] += l.r[i][j];
                }
                if (V == 4) {
                    unsigned a0, a1, a2, b0, b1, b2;
                    split_pair(l.r[i][0], l.r[i][1 % V], a0, a1, a2);
                    split_pair(l.r[i][2 % V], l.r[i][3 % V], b0, b1, b2);
                    *reinterpret_cast<u32x2*>(row) = u32x2{a0, b0};
                    *reinterpret_cast<u32x2*>(row + PIECE) = u32x2{a1, b1};
                    *reinterpret_cast<u32x2*>(row + 2 * PIECE) = u32x2{a2, b2};
                } else {
                    unsigned a0, a1, a2;
                    split_pair(l.r[i][0], 0.f, a0, a1, a2);
                    *reinterpret_cast<unsigned short*>(row) = (unsigned short)a0;
                    *reinterpret_cast<unsigned short*>(row + PIECE) = (unsigned short)a1;
                    *reinterpret_cast<unsigned short*>(row + 2 * PIECE) = (unsigned short)a2;
                }
            }
        }
    }
    // ---- MFMA operand fragment: row/col `idx0 + (lane & 31)`, k = 16 ks + 8 (lane >> 5) .. + 7, piece q
    static __device__ __forceinline__ u32x4 frag(const unsigned char* S, int idx0, int ks, int q, int lane) {
        const int l31 = lane & 31, hi = lane >> 5;
        if (KC) {
            const int m = idx0 + l31;
            return *reinterpret_cast<const u32x4*>(S + q * PIECE + m * 64 + (((2 * ks + hi + (m >> 2)) & 3) << 4));
        } else {
            const int t16 = lane & 15;
            const unsigned char* b = S + q * PIECE + (16 * ks + 8 * hi + (t16 >> 2)) * STRIDE
                                     + (idx0 + 16 * ((lane >> 4) & 1) + 4 * (t16 & 3)) * 2;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b));
            const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b + 4 * STRIDE));
            const u32x2 a = __builtin_bit_cast(u32x2, lo), c = __builtin_bit_cast(u32x2, hi4);
            return u32x4{a[0], a[1], c[0], c[1]};
        }
    }
};

template <int BM, int BN, int WGM, int WGN, bool TA, bool TB, int V>
__global__ __launch_bounds__(256) void gemm_x3_kernel(GemmArgs p) {
    static_assert(WGM * WGN == 4, "4 waves per workgroup");
    constexpr int BK = 32;
    constexpr int WM = BM / WGM, WN = BN / WGN, TM = WM / 32, TN = WN / 32;
    constexpr bool A_KC = !TA, B_KC = TB;
    using IA = X3Image<A_KC, BM>;
    using IB = X3Image<B_KC, BN>;
    __shared__ __attribute__((aligned(16))) unsigned char As[IA::BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[IB::BYTES];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int id = blockIdx.x;
    const int s = id % p.split; id /= p.split;
    const int nt = id % p.ntiles;
    const int mt = id / p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kbeg = s * p.kslab;
    const int kend = min(p.K, kbeg + p.kslab);
    const RowMap amap{p.perm, p.cursor ? *p.cursor : 0};
    const Identity ident;

    typename LoaderSel<A_KC, BM, BK, V>::T la;
    typename LoaderSel<B_KC, BN, BK, V>::T lb;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool do_colsum = !TB && p.colsum && mt == 0;       // B is n-contiguous there (TN, NN): column sums at store time
    float csum[V];
#pragma unroll
    for (int j = 0; j < V; ++j) csum[j] = 0.f;
    float dummy[V];

    const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
    const int nchunks = (kend - kbeg + BK - 1) / BK;

    la.prepare(p.lda, m0, p.M, kbeg, kend, amap);
    lb.prepare(p.ldb, n0, p.N, kbeg, kend, ident);
    la.load(p.A, p.lda, m0, p.M, kbeg, kend, amap);
    lb.load(p.B, p.ldb, n0, p.N, kbeg, kend, ident);
    for (int c = 0; c < nchunks; ++c) {
        if constexpr (A_KC) IA::store(As, la); else IA::store(As, la, dummy, false);
        if constexpr (B_KC) IB::store(Bs, lb); else IB::store(Bs, lb, csum, do_colsum);
        __syncthreads();
        if (c + 1 < nchunks) {
            const int k0 = kbeg + (c + 1) * BK;
            la.load(p.A, p.lda, m0, p.M, k0, kend, amap);
            lb.load(p.B, p.ldb, n0, p.N, k0, kend, ident);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 a[TM][3], b[TN][3];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) a[i][q] = IA::frag(As, wm0 + i * 32, ks, q, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) b[j][q] = IB::frag(Bs, wn0 + j * 32, ks, q, lane);
            // six products per accumulator, small terms first, round-robin over the accumulators
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = MFMA16(a[i][PA[pr]], b[j][PB[pr]], acc[i][j]);
            }
        }
        __syncthreads();
    }

    // ---- epilogue ------------------------------------------------------------------------
    const int Mo = p.M + (p.colsum ? 1 : 0);
    float* out = p.split > 1 ? p.ws + (long)s * Mo * p.N : p.C;
    const long ldo = p.split > 1 ? (long)p.N : p.ldc;
    const bool add_bias = p.split == 1 && p.bias != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + (lane & 31);
        const float bv = (add_bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M && n < p.N) out[(long)m * ldo + n] = acc[i][j][r] + bv;
            }
        }
    }
    if constexpr (!TB) {
        if (do_colsum) {
            // column sums of B (the bias gradient): per-thread partial sums of the rows this thread stored, combined over
            // the KPP row groups in fixed order
            using L = MNContig<BN, BK, V>;
            float* red = reinterpret_cast<float*>(As);             // safe: the last loop iteration ended with a barrier
            const int c = (t % L::TC) * V, kr = t / L::TC;
            static_assert(L::KPP * BN * 4 <= IA::BYTES, "column-sum scratch");
#pragma unroll
            for (int j = 0; j < V; ++j) red[kr * BN + c + j] = csum[j];
            __syncthreads();
            if (t < BN) {
                float v = 0.f;
                for (int q = 0; q < L::KPP; ++q) v += red[q * BN + t];
                const int n = n0 + t;
                if (n < p.N) out[(long)p.M * ldo + n] = v;
            }
        }
    }
}

// C[m,n] = bias[n] + sum_s ws[s][m][n]  (ordered: deterministic).  GL threads split the S slabs of
// one output (many slabs of a small C: the batch-32 regime), combined in fixed order through LDS.
template <int GL>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* ws, int S, int Mo, int N,
                                                            const float* bias, int Mbias,
                                                            float* C, long ldc) {
    constexpr int OUT = 256 / GL;
    __shared__ float red[256];
    const long total = (long)Mo * N;
    const int o = threadIdx.x % OUT, gl = threadIdx.x / OUT;
    for (long base = (long)blockIdx.x * OUT; base < total; base += (long)gridDim.x * OUT) {
        const long i = base + o;
        float v = 0.f;
        if (i < total)
            for (int s = gl; s < S; s += GL) v += ws[(long)s * total + i];
        if (GL > 1) {
            __syncthreads();
            red[threadIdx.x] = v;
            __syncthreads();
            if (gl == 0) {
#pragma unroll
                for (int k = 1; k < GL; ++k) v += red[k * OUT + o];
            }
        }
        if (gl == 0 && i < total) {
            const int m = (int)(i / N), n = (int)(i - (long)m * N);
            if (bias && m < Mbias) v += bias[n];
            C[(long)m * ldc + n] = v;
        }
    }
}

// =====================================================================================================
// K-GEMM from PRE-SPLIT operands ("planes"): the same six-product arithmetic, the three bf16 pieces of every element
// already in memory as three planes [3][rows][ld] of bf16 (dcahip_split_planes, or a producer kernel that writes them
// directly).  An operand that is used by several products per step (the gradient planes of the heads: weight AND input
// gradient; the head weights: forward AND input gradient; the normalised counts: every epoch) is split once instead of
// in every K loop, and the loop has no vector arithmetic left: 16-byte loads -> LDS -> fragments -> MFMA.  Either
// operand may be k-contiguous (fragment = one ds_read_b128) or m/n-contiguous (the transposing ds_read_b64_tr_b16), so
// ONE stored layout of a matrix serves the products that contract over its rows and over its columns: no transposed
// copies.  128 x 128 tiles, 4 waves, 64 x 64 per wave, K chunks of 32 staged through registers.
// =====================================================================================================
struct Gemm3Args {
    const unsigned short* A;
    const unsigned short* B;
    long lda, ldb, pa, pb;              // leading dimensions and plane strides, in elements
    float* C;
    const float* bias;
    const int* perm;
    const long long* cursor;
    float* ws;
    long ldc;
    int M, N, K;
    int split, kslab;
    int colsum;
    int mtiles, ntiles;
    // 256 x 256 kernel, split == 1: the tiles of the last, partial round of workgroups (tile index >= tail_first) are cut
    // into tail_split K slices each, so that the round is as short as its share of the work (a partial round of whole
    // tiles costs a full tile's time); their partial tiles [tile - tail_first][slice][256][256] are summed afterwards
    int tail_first, tail_split, tail_kslab;
    float* tail_ws;
};

// one operand tile (DIM rows/cols x 32 k, three pieces) as 16-byte units: 12 DIM units over 256 threads
template <bool KC, int DIM>
struct PlaneTile {
    static constexpr int PER = 12 * DIM / 256;           // units per thread (6 at DIM = 128): piece = i / (PER / 3)
    static constexpr int RPT = PER / 3;                  // rows (KC) or k rows (MN) per thread and piece
    static_assert(PER % 3 == 0 && RPT >= 1, "tile");
    u32x4 r[PER];
    long off[RPT];                                       // KC: storage offset of this thread's rows (looked up once)
    long koff[RPT], knext[RPT];                          // MN: storage offset of this chunk's / the next chunk's k rows

    // KC: unit (row, kq): row = t / 4 + 64 j, kq = t % 4.   MN: unit (k, cq): k = t / (DIM / 8) + (256 * 8 / DIM) j
    template <class Map>
    __device__ __forceinline__ void prepare(long ld, int idx0, int nidx, int k0, int kend, const Map& map) {
        const int t = threadIdx.x;
        if (KC) {
#pragma unroll
            for (int j = 0; j < RPT; ++j) {
                const int row = idx0 + (t >> 2) + 64 * j;
                off[j] = map(row < nidx ? row : nidx - 1) * ld;
            }
        } else {
            lookup(ld, k0, kend, map, knext);
        }
    }
    template <class Map>
    __device__ __forceinline__ void lookup(long ld, int k0, int kend, const Map& map, long (&dst)[RPT]) {
        constexpr int KSTEP = 256 * 8 / DIM;
        const int t = threadIdx.x;
#pragma unroll
        for (int j = 0; j < RPT; ++j) {
            const int k = k0 + t / (DIM / 8) + KSTEP * j;
            dst[j] = k < kend ? map(k) * ld : -1;
        }
    }
    // the chunk at k0 into registers; MN: also looks up the storage rows of the chunk after it
    template <class Map>
    __device__ __forceinline__ void load(const unsigned short* base, long ld, long pstride, int idx0, int nidx, int k0,
                                         int kend, const Map& map) {
        const int t = threadIdx.x;
        if (KC) {
            const int k = k0 + (t & 3) * 8;
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int j = 0; j < RPT; ++j) {
                    r[q * RPT + j] = u32x4{0u, 0u, 0u, 0u};
                    if (k < kend) r[q * RPT + j] = *reinterpret_cast<const u32x4*>(base + q * pstride + off[j] + k);
                }
        } else {
#pragma unroll
            for (int j = 0; j < RPT; ++j) koff[j] = knext[j];
            const int c = idx0 + (t % (DIM / 8)) * 8;
#pragma unroll
            for (int q = 0; q < 3; ++q)
#pragma unroll
                for (int j = 0; j < RPT; ++j) {
                    r[q * RPT + j] = u32x4{0u, 0u, 0u, 0u};
                    if (koff[j] >= 0 && c < nidx) r[q * RPT + j] = *reinterpret_cast<const u32x4*>(base + q * pstride + koff[j] + c);
                }
            lookup(ld, k0 + 32, kend, map, knext);
        }
    }
};

template <bool KC, int DIM>
__device__ __forceinline__ void plane_store(unsigned char* S, const PlaneTile<KC, DIM>& l) {
    using I = X3Image<KC, DIM>;
    using T = PlaneTile<KC, DIM>;
    const int t = threadIdx.x;
#pragma unroll
    for (int q = 0; q < 3; ++q)
#pragma unroll
        for (int j = 0; j < T::RPT; ++j) {
            if (KC) {
                const int row = (t >> 2) + 64 * j, kq = t & 3;
                *reinterpret_cast<u32x4*>(S + q * I::PIECE + row * 64 + (((kq + (row >> 2)) & 3) << 4)) = l.r[q * T::RPT + j];
            } else {
                const int k = t / (DIM / 8) + (256 * 8 / DIM) * j, cq = t % (DIM / 8);
                *reinterpret_cast<u32x4*>(S + q * I::PIECE + k * I::STRIDE + cq * 16) = l.r[q * T::RPT + j];
            }
        }
}

__device__ __forceinline__ float bf16_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf16_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(256) void gemm_p3_kernel(Gemm3Args p) {
    constexpr int BM = 128, BN = 128, BK = 32, WGN = 2, WM = 64, WN = 64, TM = 2, TN = 2;
    using IA = X3Image<A_KC, BM>;
    using IB = X3Image<B_KC, BN>;
    __shared__ __attribute__((aligned(16))) unsigned char As[IA::BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[IB::BYTES];

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int id = blockIdx.x;
    const int s = id % p.split; id /= p.split;
    const int nt = id % p.ntiles;
    const int mt = id / p.ntiles;
    const int m0 = mt * BM, n0 = nt * BN;
    const int kbeg = s * p.kslab;
    const int kend = min(p.K, kbeg + p.kslab);
    const RowMap amap{p.perm, p.cursor ? *p.cursor : 0};
    const Identity ident;

    PlaneTile<A_KC, BM> la;
    PlaneTile<B_KC, BN> lb;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const bool do_colsum = !B_KC && p.colsum && mt == 0;     // column sums of B (the Dense bias gradient)
    float csum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) csum[j] = 0.f;

    const int wm0 = (wave / WGN) * WM, wn0 = (wave % WGN) * WN;
    const int nchunks = (kend - kbeg + BK - 1) / BK;

    la.prepare(p.lda, m0, p.M, kbeg, kend, amap);
    lb.prepare(p.ldb, n0, p.N, kbeg, kend, ident);
    la.load(p.A, p.lda, p.pa, m0, p.M, kbeg, kend, amap);
    lb.load(p.B, p.ldb, p.pb, n0, p.N, kbeg, kend, ident);
    for (int c = 0; c < nchunks; ++c) {
        plane_store<A_KC, BM>(As, la);
        plane_store<B_KC, BN>(Bs, lb);
        if constexpr (!B_KC) {
            if (do_colsum) {
                using T = PlaneTile<false, BN>;
#pragma unroll
                for (int u = 0; u < T::PER; ++u)
#pragma unroll
                    for (int w = 0; w < 4; ++w) { csum[2 * w] += bf16_lo(lb.r[u][w]); csum[2 * w + 1] += bf16_hi(lb.r[u][w]); }
            }
        }
        __syncthreads();
        if (c + 1 < nchunks) {
            const int k0 = kbeg + (c + 1) * BK;
            la.load(p.A, p.lda, p.pa, m0, p.M, k0, kend, amap);
            lb.load(p.B, p.ldb, p.pb, n0, p.N, k0, kend, ident);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            u32x4 a[TM][3], b[TN][3];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int q = 0; q < 3; ++q) a[i][q] = IA::frag(As, wm0 + i * 32, ks, q, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 3; ++q) b[j][q] = IB::frag(Bs, wn0 + j * 32, ks, q, lane);
#pragma unroll
            for (int pr = 0; pr < 6; ++pr) {
                constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = MFMA16(a[i][PA[pr]], b[j][PB[pr]], acc[i][j]);
            }
        }
        __syncthreads();
    }

    const int Mo = p.M + (p.colsum ? 1 : 0);
    float* out = p.split > 1 ? p.ws + (long)s * Mo * p.N : p.C;
    const long ldo = p.split > 1 ? (long)p.N : p.ldc;
    const bool add_bias = p.split == 1 && p.bias != nullptr;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + wn0 + j * 32 + (lane & 31);
        const float bv = (add_bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M && n < p.N) out[(long)m * ldo + n] = acc[i][j][r] + bv;
            }
        }
    }
    if constexpr (!B_KC) {
        if (do_colsum) {
            // per-thread sums of the k rows this thread stored (columns 8 (t % 16) .. + 7), combined over the 16 row groups
            float* red = reinterpret_cast<float*>(As);             // safe: the loop ended with a barrier
            static_assert(16 * BN * 4 <= IA::BYTES, "column-sum scratch");
            const int c8 = (t & 15) * 8, kr = t >> 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) red[kr * BN + c8 + j] = csum[j];
            __syncthreads();
            if (t < BN) {
                float v = 0.f;
                for (int q = 0; q < 16; ++q) v += red[q * BN + t];
                const int n = n0 + t;
                if (n < p.N) out[(long)p.M * ldo + n] = v;
            }
        }
    }
}

// ---- the same product for LARGE outputs: 256 x 256 tiles, 8 waves (2 x 4, 128 x 64 per wave), K steps of 16 staged
// DIRECTLY global -> LDS (global_load_lds_dwordx4: no staging registers, no LDS store pass) into a ring of three stages,
// requested two steps ahead and retired with a counted s_waitcnt vmcnt (never 0 in the steady state), ONE barrier per
// step.  Per step and wave: 18 fragment reads feed 48 MFMAs (the 128 x 128 kernel above: 24 feed 48, with a store pass
// and two barriers per 32 k) -- the LDS pipe is what bounds that one (75 % busy at its MFMA rate).
// LDS images, one plane of one operand of one stage:
//   k-contiguous    [256 rows][16 k] bf16 = 32-byte rows; a fragment (8 k of one row) is half a row.  Thread t brings row
//                   t / 2, half t % 2 -- with the halves of rows 8..15 (mod 16) swapped: the 16 lanes of one phase of a
//                   ds_read_b128 (16 consecutive rows, one k half) then cover the 64 banks once (unswapped, rows r and
//                   r + 8 collide: SQ_LDS_BANK_CONFLICT 40 % of the LDS cycles).
//   m/n-contiguous  16 k rows of 256 columns; wave w brings k rows w and w + 8 as ONE 1 KB chunk (a wave's
//                   global_load_lds lands lane-linear), chunks 1088 bytes apart: the four consecutive k rows a lane group
//                   of the transposing read touches sit 64 bytes modulo 128 apart (X3Image's rule).
// Order of events in step k (stage k % 3):  wait vmcnt(6) [own requests of step k have landed; step k + 1's 6 stay in
// flight] -> barrier [everybody's have; everybody has read step k - 1] -> request step k + 2 into the stage step k - 1
// used -> fragments of step k -> MFMAs.
constexpr int kWBK = 16;
template <bool KC>
struct WideImage {
    static constexpr int CHUNK = KC ? 1024 : 1088;         // bytes between the LDS destinations of consecutive waves
    static constexpr int PLANE = 8 * CHUNK;
    static constexpr int BYTES = 3 * PLANE;
    // Fragment reads are ISSUED here (inline asm: the compiler neither reorders them nor guards them with a vmcnt(0)
    // against the LDS-DMA requests in flight) and RETIRED by the caller with a counted s_waitcnt lgkmcnt.
    // lane_off: this lane's byte offset inside a plane for row / column tile 0 of the wave
    static __device__ __forceinline__ unsigned lane_off(int idx0, int lane) {
        const int l31 = lane & 31, hi = lane >> 5;
        if (KC) return (unsigned)((idx0 + l31) * 32 + ((hi ^ ((l31 >> 3) & 1)) * 16));   // (halves of rows 8..15 mod 16 swapped)
        const int t16 = lane & 15, r = t16 >> 2;           // k rows 8 hi + r and 8 hi + r + 4
        return (unsigned)(r * CHUNK + hi * 512 + (idx0 + 16 * ((lane >> 4) & 1) + 4 * (t16 & 3)) * 2);
    }
    static constexpr int READS = KC ? 1 : 2;               // LDS instructions per fragment
    // fragment of piece Q, tile TILE (32 rows / columns further per tile) at LDS address `addr` (stage + operand + lane_off)
    template <int Q, int TILE>
    static __device__ __forceinline__ void issue(u32x4& f, unsigned addr) {
        if constexpr (KC) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(Q * PLANE + TILE * 32 * 32));
        } else {
            u32x2 lo, hi;
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(Q * PLANE + TILE * 64));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(Q * PLANE + TILE * 64 + 4 * CHUNK));
            f = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
    }
    // this thread's source element offset inside a plane at k = 0 (add k for KC, k * ld for MN)
    static __device__ __forceinline__ long src_off(int t, long ld, int idx0, int nidx) {
        if (KC) {
            // LDS slot t = (row t / 2, physical half t % 2); rows 8..15 (mod 16) keep their halves swapped, so that the
            // 16 lanes of a ds_read_b128 phase (rows r .. r + 15, one k half) cover the 64 banks once
            const int r = t >> 1, row = idx0 + r;
            return (long)(row < nidx ? row : nidx - 1) * ld + (((t & 1) ^ ((r >> 3) & 1)) * 8);
        } else {
            const int l = t & 63, kk = (t >> 6) + 8 * (l >> 5);
            int c = idx0 + (l & 31) * 8;
            if (c >= nidx) c = 0;                            // (columns outside the matrix: any readable address)
            return (long)kk * ld + c;
        }
    }
};

template <bool A_KC, bool B_KC, bool CS>
__global__ __launch_bounds__(512) void gemm_p3w_kernel(Gemm3Args p) {
    using IA = WideImage<A_KC>;
    using IB = WideImage<B_KC>;
    constexpr int STAGE = IA::BYTES + IB::BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int id = blockIdx.x;
    int s, kbeg, kend;
    const bool tail = p.tail_split > 1 && id >= p.tail_first;          // (tail_split > 1 only with split == 1)
    if (tail) {
        const int r = id - p.tail_first;
        id = p.tail_first + r / p.tail_split;
        s = r % p.tail_split;
        kbeg = s * p.tail_kslab;
        kend = min(p.K, kbeg + p.tail_kslab);
    } else {
        s = id % p.split; id /= p.split;
        kbeg = s * p.kslab;
        kend = min(p.K, kbeg + p.kslab);
    }
    const int tile = id;
    const int nt = id % p.ntiles;
    const int mt = id / p.ntiles;
    const int m0 = mt * 256, n0 = nt * 256;
    const int nsteps = (kend - kbeg) / kWBK;

    // the source of this thread's unit in the step to request next (advanced by one step per request)
    const unsigned short* ga = p.A + IA::src_off(t, p.lda, m0, p.M) + (A_KC ? (long)kbeg : (long)kbeg * p.lda);
    const unsigned short* gb = p.B + IB::src_off(t, p.ldb, n0, p.N) + (B_KC ? (long)kbeg : (long)kbeg * p.ldb);
    const long sa = A_KC ? (long)kWBK : (long)kWBK * p.lda, sb = B_KC ? (long)kWBK : (long)kWBK * p.ldb;
    const int wa = wave * IA::CHUNK, wb = IA::BYTES + wave * IB::CHUNK;

    auto request = [&](int stage) __attribute__((always_inline)) {
        unsigned char* st = lds + stage * STAGE;
#pragma unroll
        for (int q = 0; q < 3; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + q * p.pa),
                                             (__attribute__((address_space(3))) void*)(st + q * IA::PLANE + wa), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 3; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + q * p.pb),
                                             (__attribute__((address_space(3))) void*)(st + q * IB::PLANE + wb), 16, 0, 0);
        ga += sa; gb += sb;
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm0 = (wave >> 2) * 128, wn0 = (wave & 3) * 64;
    const unsigned aoff = IA::lane_off(wm0, lane), boff = IB::lane_off(wn0, lane);
    const bool do_colsum = CS && !B_KC && p.colsum && mt == 0 && (wave >> 2) == 0;
    float csum[2] = {0.f, 0.f};

    if (nsteps > 0) request(0);
    if (nsteps > 1) request(1);
    int cur = 0, nxt = 2;                                  // stage of step k, stage step k + 2 goes to
#pragma unroll 1
    for (int k = 0; k < nsteps; ++k) {
        if (k + 1 < nsteps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned sa_ = (unsigned)(cur * STAGE) + aoff, sb_ = (unsigned)(cur * STAGE + IA::BYTES) + boff;
        const int rq = nxt;
        cur = cur == 2 ? 0 : cur + 1; nxt = nxt == 2 ? 0 : nxt + 1;
        // fragments: B (2 column tiles x 3 pieces), then A row tile by row tile, each requested one tile ahead of its
        // products; waits are counted in LDS instructions still allowed in flight (they return in order)
        u32x4 b[2][3], a[2][3];
        IB::template issue<0, 0>(b[0][0], sb_); IB::template issue<1, 0>(b[0][1], sb_); IB::template issue<2, 0>(b[0][2], sb_);
        IB::template issue<0, 1>(b[1][0], sb_); IB::template issue<1, 1>(b[1][1], sb_); IB::template issue<2, 1>(b[1][2], sb_);
        IA::template issue<0, 0>(a[0][0], sa_); IA::template issue<1, 0>(a[0][1], sa_); IA::template issue<2, 0>(a[0][2], sa_);
#define DCA_TIE6(x) "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[0][2]), "+v"(x[1][0]), "+v"(x[1][1]), "+v"(x[1][2])
#define DCA_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : DCA_TIE6(a), DCA_TIE6(b))
#define DCA_PRODUCTS(I, AI) do { _Pragma("unroll") for (int pr = 0; pr < 6; ++pr) { \
            constexpr int PA[6] = {2, 1, 0, 1, 0, 0}, PB[6] = {0, 1, 2, 0, 1, 0}; \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[I][j] = MFMA16(a[AI][PA[pr]], b[j][PB[pr]], acc[I][j]); } } while (0)
        IA::template issue<0, 1>(a[1][0], sa_); IA::template issue<1, 1>(a[1][1], sa_); IA::template issue<2, 1>(a[1][2], sa_);
        if constexpr (IA::READS == 1) DCA_WAIT_LGKM(3); else DCA_WAIT_LGKM(6);
        if constexpr (CS && !B_KC) {
            if (do_colsum) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 3; ++q)
#pragma unroll
                        for (int w = 0; w < 4; ++w) csum[j] += bf16_lo(b[j][q][w]) + bf16_hi(b[j][q][w]);
            }
        }
        DCA_PRODUCTS(0, 0);
        // the next-but-one step's requests go out BEHIND the first products: their issue (an M0 write and an address per
        // piece) fills the gaps of the matrix pipe instead of standing between the barrier and the first product
        if (k + 2 < nsteps) request(rq);
        IA::template issue<0, 2>(a[0][0], sa_); IA::template issue<1, 2>(a[0][1], sa_); IA::template issue<2, 2>(a[0][2], sa_);
        if constexpr (IA::READS == 1) DCA_WAIT_LGKM(3); else DCA_WAIT_LGKM(6);
        DCA_PRODUCTS(1, 1);
        IA::template issue<0, 3>(a[1][0], sa_); IA::template issue<1, 3>(a[1][1], sa_); IA::template issue<2, 3>(a[1][2], sa_);
        if constexpr (IA::READS == 1) DCA_WAIT_LGKM(3); else DCA_WAIT_LGKM(6);
        DCA_PRODUCTS(2, 0);
        DCA_WAIT_LGKM(0);
        DCA_PRODUCTS(3, 1);
#undef DCA_PRODUCTS
#undef DCA_WAIT_LGKM
#undef DCA_TIE6
    }

    const int Mo = p.M + (p.colsum ? 1 : 0);
    if (tail) {
        // a K slice of a tail tile: the partial tile, tile-local, into the tail workspace
        float* pt = p.tail_ws + ((long)(tile - p.tail_first) * p.tail_split + s) * (256L * 256);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    pt[(wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 256 + wn0 + j * 32 + (lane & 31)] = acc[i][j][r];
        return;                                            // (no column sums here: the host keeps m tile 0 out of the tail)
    }
    float* out = p.split > 1 ? p.ws + (long)s * Mo * p.N : p.C;
    const long ldo = p.split > 1 ? (long)p.N : p.ldc;
    const bool add_bias = p.split == 1 && p.bias != nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn0 + j * 32 + (lane & 31);
        const float bv = (add_bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M && n < p.N) out[(long)m * ldo + n] = acc[i][j][r] + bv;
            }
        }
        if (CS && do_colsum) {
            // the lane holds the k rows 8 hi .. + 7 of column n: the other half's share through a lane exchange
            const float tot = csum[j] + __shfl_xor(csum[j], 32, 64);
            if ((lane >> 5) == 0 && n < p.N) out[(long)p.M * ldo + n] = tot;
        }
    }
}

// =====================================================================================================
// The plane GEMM on TWO fp16 pieces per operand and three products per fp32 product (round 6; arithmetic: h2_math.hpp,
// stated in full in dcahip_heads.hip): dcahip_split_planes_h2 scales a matrix by the power of two of its largest magnitude
// (dcahip_absmax_exp: a device word, so nothing leaves the stream) and writes two planes; the 256 x 256 kernel below is
// gemm_p3w_kernel with two planes per operand in its ring (stages of 32 KB instead of 48), HALF the matrix instructions and a
// third fewer fragment reads per step; the block scales leave in the epilogue together with the caller's factor `alpha`.
// =====================================================================================================
#define MFMAH2(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h2_f16x8, a), __builtin_bit_cast(h2_f16x8, b), c, 0, 0, 0)

struct GemmH2Args {
    const unsigned short* A;
    const unsigned short* B;
    long lda, ldb, pa, pb;              // leading dimensions and plane strides, in elements
    float* C;
    const float* bias;
    float* ws;
    long ldc;
    int M, N, K;
    int split, kslab;
    int colsum;
    int mtiles, ntiles;
    int tail_first, tail_split, tail_kslab;
    float* tail_ws;
    const int* exp_a; const int* exp_b;  // device words: the operands' block exponents (NULL: 0) ...
    int exp_a_add, exp_b_add;            // ... plus these
    float alpha;
};

template <bool KC, int NCH = 8>                             // NCH: 1 KB chunks per plane (8: 256 rows / columns; 4: 128, k-contiguous only)
struct WideImageH {
    static constexpr int CHUNK = KC ? 1024 : 1088;         // bytes between the LDS destinations of consecutive waves
    static constexpr int PLANE = NCH * CHUNK;
    static constexpr int BYTES = 2 * PLANE;
    // Fragment reads are ISSUED here (inline asm: the compiler neither reorders them nor guards them with a vmcnt(0)
    // against the LDS-DMA requests in flight) and RETIRED by the caller with a counted s_waitcnt lgkmcnt.
    // lane_off: this lane's byte offset inside a plane for row / column tile 0 of the wave
    static __device__ __forceinline__ unsigned lane_off(int idx0, int lane) {
        const int l31 = lane & 31, hi = lane >> 5;
        if (KC) return (unsigned)((idx0 + l31) * 32 + ((hi ^ ((l31 >> 3) & 1)) * 16));   // (halves of rows 8..15 mod 16 swapped)
        const int t16 = lane & 15, r = t16 >> 2;           // k rows 8 hi + r and 8 hi + r + 4
        return (unsigned)(r * CHUNK + hi * 512 + (idx0 + 16 * ((lane >> 4) & 1) + 4 * (t16 & 3)) * 2);
    }
    static constexpr int READS = KC ? 1 : 2;               // LDS instructions per fragment
    // fragment of piece Q, tile TILE (32 rows / columns further per tile) at LDS address `addr` (stage + operand + lane_off)
    template <int Q, int TILE>
    static __device__ __forceinline__ void issue(u32x4& f, unsigned addr) {
        if constexpr (KC) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "n"(Q * PLANE + TILE * 32 * 32));
        } else {
            u32x2 lo, hi;
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(Q * PLANE + TILE * 64));
            asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(Q * PLANE + TILE * 64 + 4 * CHUNK));
            f = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
    }
    // this thread's source element offset inside a plane at k = 0 (add k for KC, k * ld for MN)
    static __device__ __forceinline__ long src_off(int t, long ld, int idx0, int nidx) {
        if (KC) {
            // LDS slot t = (row t / 2, physical half t % 2); rows 8..15 (mod 16) keep their halves swapped, so that the
            // 16 lanes of a ds_read_b128 phase (rows r .. r + 15, one k half) cover the 64 banks once
            const int r = t >> 1, row = idx0 + r;
            return (long)(row < nidx ? row : nidx - 1) * ld + (((t & 1) ^ ((r >> 3) & 1)) * 8);
        } else {
            const int l = t & 63, kk = (t >> 6) + 8 * (l >> 5);
            int c = idx0 + (l & 31) * 8;
            if (c >= nidx) c = 0;                            // (columns outside the matrix: any readable address)
            return (long)kk * ld + c;
        }
    }
};

template <bool A_KC, bool B_KC, bool CS>
__global__ __launch_bounds__(512) void gemm_h2w_kernel(GemmH2Args p) {
    using IA = WideImageH<A_KC>;
    using IB = WideImageH<B_KC>;
    constexpr int STAGE = IA::BYTES + IB::BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int id = blockIdx.x;
    int s, kbeg, kend;
    const bool tail = p.tail_split > 1 && id >= p.tail_first;          // (tail_split > 1 only with split == 1)
    if (tail) {
        const int r = id - p.tail_first;
        id = p.tail_first + r / p.tail_split;
        s = r % p.tail_split;
        kbeg = s * p.tail_kslab;
        kend = min(p.K, kbeg + p.tail_kslab);
    } else {
        s = id % p.split; id /= p.split;
        kbeg = s * p.kslab;
        kend = min(p.K, kbeg + p.kslab);
    }
    const int tile = id;
    const int nt = id % p.ntiles;
    const int mt = id / p.ntiles;
    const int m0 = mt * 256, n0 = nt * 256;
    const int nsteps = (kend - kbeg) / kWBK;

    // the source of this thread's unit in the step to request next (advanced by one step per request)
    const unsigned short* ga = p.A + IA::src_off(t, p.lda, m0, p.M) + (A_KC ? (long)kbeg : (long)kbeg * p.lda);
    const unsigned short* gb = p.B + IB::src_off(t, p.ldb, n0, p.N) + (B_KC ? (long)kbeg : (long)kbeg * p.ldb);
    const long sa = A_KC ? (long)kWBK : (long)kWBK * p.lda, sb = B_KC ? (long)kWBK : (long)kWBK * p.ldb;
    const int wa = wave * IA::CHUNK, wb = IA::BYTES + wave * IB::CHUNK;

    auto request = [&](int stage) __attribute__((always_inline)) {
        unsigned char* st = lds + stage * STAGE;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + q * p.pa),
                                             (__attribute__((address_space(3))) void*)(st + q * IA::PLANE + wa), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb + q * p.pb),
                                             (__attribute__((address_space(3))) void*)(st + q * IB::PLANE + wb), 16, 0, 0);
        ga += sa; gb += sb;
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wm0 = (wave >> 2) * 128, wn0 = (wave & 3) * 64;
    const unsigned aoff = IA::lane_off(wm0, lane), boff = IB::lane_off(wn0, lane);
    const bool do_colsum = CS && !B_KC && p.colsum && mt == 0 && (wave >> 2) == 0;
    float csum[2] = {0.f, 0.f};

    if (nsteps > 0) request(0);
    if (nsteps > 1) request(1);
    int cur = 0, nxt = 2;                                  // stage of step k, stage step k + 2 goes to
#pragma unroll 1
    for (int k = 0; k < nsteps; ++k) {
        if (k + 1 < nsteps) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned sa_ = (unsigned)(cur * STAGE) + aoff, sb_ = (unsigned)(cur * STAGE + IA::BYTES) + boff;
        const int rq = nxt;
        cur = cur == 2 ? 0 : cur + 1; nxt = nxt == 2 ? 0 : nxt + 1;
        // fragments: B (2 column tiles x 3 pieces), then A row tile by row tile, each requested one tile ahead of its
        // products; waits are counted in LDS instructions still allowed in flight (they return in order)
        u32x4 b[2][2], a[2][2];
        IB::template issue<0, 0>(b[0][0], sb_); IB::template issue<1, 0>(b[0][1], sb_);
        IB::template issue<0, 1>(b[1][0], sb_); IB::template issue<1, 1>(b[1][1], sb_);
        IA::template issue<0, 0>(a[0][0], sa_); IA::template issue<1, 0>(a[0][1], sa_);
#define DCA_TIE4(x) "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1])
#define DCA_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : DCA_TIE4(a), DCA_TIE4(b))
#define DCA_PRODUCTS(I, AI) do { _Pragma("unroll") for (int pr = 0; pr < 3; ++pr) { \
            constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0}; \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[I][j] = MFMAH2(a[AI][PA[pr]], b[j][PB[pr]], acc[I][j]); } } while (0)
        IA::template issue<0, 1>(a[1][0], sa_); IA::template issue<1, 1>(a[1][1], sa_);
        if constexpr (IA::READS == 1) DCA_WAIT_LGKM(2); else DCA_WAIT_LGKM(4);
        if constexpr (CS && !B_KC) {
            if (do_colsum) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const unsigned pw = b[j][q][w];
                            csum[j] = __builtin_amdgcn_fdot2(__builtin_bit_cast(h2_f16x2, pw), h2_f16x2{(_Float16)1.f, (_Float16)1.f}, csum[j], false);
                        }
            }
        }
        DCA_PRODUCTS(0, 0);
        // the next-but-one step's requests go out BEHIND the first products: their issue (an M0 write and an address per
        // piece) fills the gaps of the matrix pipe instead of standing between the barrier and the first product
        if (k + 2 < nsteps) request(rq);
        IA::template issue<0, 2>(a[0][0], sa_); IA::template issue<1, 2>(a[0][1], sa_);
        if constexpr (IA::READS == 1) DCA_WAIT_LGKM(2); else DCA_WAIT_LGKM(4);
        DCA_PRODUCTS(1, 1);
        IA::template issue<0, 3>(a[1][0], sa_); IA::template issue<1, 3>(a[1][1], sa_);
        if constexpr (IA::READS == 1) DCA_WAIT_LGKM(2); else DCA_WAIT_LGKM(4);
        DCA_PRODUCTS(2, 0);
        DCA_WAIT_LGKM(0);
        DCA_PRODUCTS(3, 1);
#undef DCA_PRODUCTS
#undef DCA_WAIT_LGKM
#undef DCA_TIE4
    }

    // the operands' block scales out again (exact: powers of two) and the caller's factor in
    const int ea = (p.exp_a ? *p.exp_a : 0) + p.exp_a_add, eb = (p.exp_b ? *p.exp_b : 0) + p.exp_b_add;
    const float un = p.alpha * h2_pow2i(-(ea + eb)), un_b = p.alpha * h2_pow2i(-eb);
    const int Mo = p.M + (p.colsum ? 1 : 0);
    if (tail) {
        // a K slice of a tail tile: the partial tile, tile-local, into the tail workspace
        float* pt = p.tail_ws + ((long)(tile - p.tail_first) * p.tail_split + s) * (256L * 256);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    pt[(wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 256 + wn0 + j * 32 + (lane & 31)] = acc[i][j][r] * un;
        return;                                            // (no column sums here: the host keeps m tile 0 out of the tail)
    }
    float* out = p.split > 1 ? p.ws + (long)s * Mo * p.N : p.C;
    const long ldo = p.split > 1 ? (long)p.N : p.ldc;
    const bool add_bias = p.split == 1 && p.bias != nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn0 + j * 32 + (lane & 31);
        const float bv = (add_bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M && n < p.N) out[(long)m * ldo + n] = fmaf(acc[i][j][r], un, bv);
            }
        }
        if (CS && do_colsum) {
            // the lane holds the k rows 8 hi .. + 7 of column n: the other half's share through a lane exchange
            const float tot = csum[j] + __shfl_xor(csum[j], 32, 64);
            if ((lane >> 5) == 0 && n < p.N) out[(long)p.M * ldo + n] = tot * un_b;
        }
    }
}


// The same products from 128 x 256 tiles by FOUR waves (A k-contiguous only: forward products and input gradients), two
// workgroups per CU (76 KB of LDS each): a workgroup's epilogue -- 128 KB of fp32 through dword stores, during which its waves issue
// no matrix instruction -- runs beside the other workgroup's K loop instead of idling the CU (one 8-wave workgroup per CU: 40 us of
// a 98 us tile at K = 512).  The wave's work is that of gemm_h2w_kernel (128 x 64 per wave, the same fragment reads and product
// order); the B image (256 columns) is requested by 256 threads in two halves.
template <bool B_KC>
__global__ __launch_bounds__(256, 2) void gemm_h2m_kernel(GemmH2Args p) {      // (two waves per SIMD: two workgroups per CU)
    using IA = WideImageH<true, 4>;
    using IB = WideImageH<B_KC, 8>;
    constexpr int STAGE = IA::BYTES + IB::BYTES;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    int id = blockIdx.x;
    const int s = id % p.split; id /= p.split;
    const int kbeg = s * p.kslab, kend = min(p.K, kbeg + p.kslab);
    const int nt = id % p.ntiles, mt = id / p.ntiles;
    const int m0 = mt * 128, n0 = nt * 256;
    const int nsteps = (kend - kbeg) / kWBK;

    const unsigned short* ga = p.A + IA::src_off(t, p.lda, m0, p.M) + (long)kbeg;
    const unsigned short* gb0 = p.B + IB::src_off(t, p.ldb, n0, p.N) + (B_KC ? (long)kbeg : (long)kbeg * p.ldb);
    const unsigned short* gb1 = p.B + IB::src_off(t + 256, p.ldb, n0, p.N) + (B_KC ? (long)kbeg : (long)kbeg * p.ldb);
    const long sa = (long)kWBK, sb = B_KC ? (long)kWBK : (long)kWBK * p.ldb;
    const int wa = wave * IA::CHUNK, wb0 = IA::BYTES + wave * IB::CHUNK, wb1 = IA::BYTES + (wave + 4) * IB::CHUNK;

    auto request = [&](int stage) __attribute__((always_inline)) {
        unsigned char* st = lds + stage * STAGE;
#pragma unroll
        for (int q = 0; q < 2; ++q)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ga + q * p.pa),
                                             (__attribute__((address_space(3))) void*)(st + q * IA::PLANE + wa), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb0 + q * p.pb),
                                             (__attribute__((address_space(3))) void*)(st + q * IB::PLANE + wb0), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gb1 + q * p.pb),
                                             (__attribute__((address_space(3))) void*)(st + q * IB::PLANE + wb1), 16, 0, 0);
        }
        ga += sa; gb0 += sb; gb1 += sb;
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int wn0 = wave * 64;
    const unsigned aoff = IA::lane_off(0, lane), boff = IB::lane_off(wn0, lane);

    if (nsteps > 0) request(0);
    if (nsteps > 1) request(1);
    int cur = 0, nxt = 2;
#pragma unroll 1
    for (int k = 0; k < nsteps; ++k) {
        if (k + 1 < nsteps) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned sa_ = (unsigned)(cur * STAGE) + aoff, sb_ = (unsigned)(cur * STAGE + IA::BYTES) + boff;
        const int rq = nxt;
        cur = cur == 2 ? 0 : cur + 1; nxt = nxt == 2 ? 0 : nxt + 1;
        u32x4 b[2][2], a[2][2];
        IB::template issue<0, 0>(b[0][0], sb_); IB::template issue<1, 0>(b[0][1], sb_);
        IB::template issue<0, 1>(b[1][0], sb_); IB::template issue<1, 1>(b[1][1], sb_);
        IA::template issue<0, 0>(a[0][0], sa_); IA::template issue<1, 0>(a[0][1], sa_);
#define DCA_TIE4(x) "+v"(x[0][0]), "+v"(x[0][1]), "+v"(x[1][0]), "+v"(x[1][1])
#define DCA_WAIT_LGKM(n) asm volatile("s_waitcnt lgkmcnt(" #n ")" : DCA_TIE4(a), DCA_TIE4(b))
#define DCA_PRODUCTS(I, AI) do { _Pragma("unroll") for (int pr = 0; pr < 3; ++pr) { \
            constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0}; \
            _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[I][j] = MFMAH2(a[AI][PA[pr]], b[j][PB[pr]], acc[I][j]); } } while (0)
        IA::template issue<0, 1>(a[1][0], sa_); IA::template issue<1, 1>(a[1][1], sa_);
        DCA_WAIT_LGKM(2);
        DCA_PRODUCTS(0, 0);
        if (k + 2 < nsteps) request(rq);
        IA::template issue<0, 2>(a[0][0], sa_); IA::template issue<1, 2>(a[0][1], sa_);
        DCA_WAIT_LGKM(2);
        DCA_PRODUCTS(1, 1);
        IA::template issue<0, 3>(a[1][0], sa_); IA::template issue<1, 3>(a[1][1], sa_);
        DCA_WAIT_LGKM(2);
        DCA_PRODUCTS(2, 0);
        DCA_WAIT_LGKM(0);
        DCA_PRODUCTS(3, 1);
#undef DCA_PRODUCTS
#undef DCA_WAIT_LGKM
#undef DCA_TIE4
    }

    const int ea = (p.exp_a ? *p.exp_a : 0) + p.exp_a_add, eb = (p.exp_b ? *p.exp_b : 0) + p.exp_b_add;
    const float un = p.alpha * h2_pow2i(-(ea + eb));
    float* out = p.split > 1 ? p.ws + (long)s * p.M * p.N : p.C;
    const long ldo = p.split > 1 ? (long)p.N : p.ldc;
    const bool add_bias = p.split == 1 && p.bias != nullptr;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0 + wn0 + j * 32 + (lane & 31);
        const float bv = (add_bias && n < p.N) ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (m < p.M && n < p.N) out[(long)m * ldo + n] = fmaf(acc[i][j][r], un, bv);
            }
        }
    }
}

// |x| maxima of a matrix as the bits of a non-negative float (order-preserving as unsigned): atomicMax is deterministic.
// One atomic per WORKGROUP, and only when it can raise the word (the plain read is a filter: the word only grows, so a stale
// read sends an atomic that changes nothing, never skips one that would) -- 16 k same-address atomics cost 130 us here.
__global__ __launch_bounds__(256) void absmax_kernel(const float* src, long ld, long R, int C, unsigned* amax) {
    const long units = (C + 3) / 4;
    const long total = R * units;
    const bool vec = (ld % 4 == 0) && ((reinterpret_cast<unsigned long long>(src) & 15) == 0);
    float m0 = 0.f, m1 = 0.f;
    const long stride = (long)gridDim.x * 256;
    long u = (long)blockIdx.x * 256 + threadIdx.x;
    if (vec && ld == C) {                       // a dense matrix: one flat run of 16-byte units, four loads in flight
        const float4* s4 = reinterpret_cast<const float4*>(src);
        for (; u + 3 * stride < total; u += 4 * stride) {
            const float4 a = s4[u], b = s4[u + stride], c = s4[u + 2 * stride], d = s4[u + 3 * stride];
            m0 = fmaxf(m0, fmaxf(fmaxf(fabsf(a.x), fabsf(a.y)), fmaxf(fabsf(a.z), fabsf(a.w))));
            m1 = fmaxf(m1, fmaxf(fmaxf(fabsf(b.x), fabsf(b.y)), fmaxf(fabsf(b.z), fabsf(b.w))));
            m0 = fmaxf(m0, fmaxf(fmaxf(fabsf(c.x), fabsf(c.y)), fmaxf(fabsf(c.z), fabsf(c.w))));
            m1 = fmaxf(m1, fmaxf(fmaxf(fabsf(d.x), fabsf(d.y)), fmaxf(fabsf(d.z), fabsf(d.w))));
        }
    }
    for (; u < total; u += stride) {
        const long r = u / units;
        const int c = (int)(u - r * units) * 4;
        const float* sp = src + r * ld + c;
        if (vec && c + 4 <= C) {
            const float4 v = *reinterpret_cast<const float4*>(sp);
            m0 = fmaxf(m0, fmaxf(fabsf(v.x), fabsf(v.y)));
            m1 = fmaxf(m1, fmaxf(fabsf(v.z), fabsf(v.w)));
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) m0 = fmaxf(m0, c + j < C ? fabsf(sp[j]) : 0.f);
        }
    }
    float m = fmaxf(m0, m1);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
        if (m > 0.f && __float_as_uint(m) > __atomic_load_n(amax, __ATOMIC_RELAXED)) atomicMax(amax, __float_as_uint(m));
    }
}
__global__ void absmax_exp_kernel(const unsigned* amax, int* exp_out) { *exp_out = h2_block_exp(__uint_as_float(*amax)); }

// fp32 [R, C] (leading dimension ld) x 2^*exp -> two fp16 planes [2][R][ldp]; columns C .. ldp - 1 are written as zeros
__global__ __launch_bounds__(256) void split_planes_h2_kernel(const float* src, long ld, const int* perm, const long long* cursor,
                                                              long R, int C, unsigned short* dst, long ldp, long pstride, const int* exp) {
    const long units = ldp / 8;
    const long total = R * units;
    const long long cur = cursor ? *cursor : 0;
    const float sc = h2_pow2i(exp ? *exp : 0);
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
        const long r = u / units;
        const int c = (int)(u - r * units) * 8;
        const long sr = perm ? (long)perm[cur + r] : (cursor ? (long)(cur + r) : r);
        const float* sp = src + sr * ld + c;
        float v[8];
        if (c + 8 <= C && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            const float4 a = *reinterpret_cast<const float4*>(sp), b = *reinterpret_cast<const float4*>(sp + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = c + j < C ? sp[j] : 0.f;
        }
        u32x4 q0, q1;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned a0, a1;
            h2_split_pair(v[2 * w] * sc, v[2 * w + 1] * sc, a0, a1);
            q0[w] = a0; q1[w] = a1;
        }
        unsigned short* dp = dst + r * ldp + c;
        *reinterpret_cast<u32x4*>(dp) = q0;
        *reinterpret_cast<u32x4*>(dp + pstride) = q1;
    }
}

// C tile = bias + sum over the K slices of a tail tile (ordered: deterministic); one workgroup per 16 rows of a tile
__global__ __launch_bounds__(256) void gemm_p3w_tail_sum_kernel(const float* tail_ws, int tail_first, int tail_split, int ntiles,
                                                               int M, int N, const float* bias, float* C, long ldc) {
    const int tt = blockIdx.x >> 4, rg = blockIdx.x & 15;            // tail tile, row group of 16
    const int tile = tail_first + tt;
    const int m0 = (tile / ntiles) * 256, n0 = (tile % ntiles) * 256;
    const int col = threadIdx.x;
    const int n = n0 + col;
    const float bv = (bias && n < N) ? bias[n] : 0.f;
    const float* src = tail_ws + (long)tt * tail_split * (256L * 256) + (long)(rg * 16) * 256 + col;
#pragma unroll 4
    for (int r = 0; r < 16; ++r) {
        float v = 0.f;
        for (int q = 0; q < tail_split; ++q) v += src[(long)q * (256L * 256) + r * 256];
        const int m = m0 + rg * 16 + r;
        if (m < M && n < N) C[(long)m * ldc + n] = v + bv;
    }
}

// fp32 [R, C] (leading dimension ld) -> three bf16 planes [3][R][ldp]: x = p0 + p1 + p2 to 2^-24 |x|; columns C .. ldp - 1
// are written as zeros (ldp % 8 == 0: 16-byte rows).  One thread = 8 consecutive elements of a row.
__global__ __launch_bounds__(256) void split_planes_kernel(const float* src, long ld, const int* perm, const long long* cursor,
                                                           long R, int C, unsigned short* dst, long ldp, long pstride) {
    const long units = ldp / 8;
    const long total = R * units;
    const long long cur = cursor ? *cursor : 0;
    for (long u = (long)blockIdx.x * 256 + threadIdx.x; u < total; u += (long)gridDim.x * 256) {
        const long r = u / units;
        const int c = (int)(u - r * units) * 8;
        const long sr = perm ? (long)perm[cur + r] : (cursor ? (long)(cur + r) : r);
        const float* sp = src + sr * ld + c;
        float v[8];
        if (c + 8 <= C && (ld & 3) == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
            const float4 a = *reinterpret_cast<const float4*>(sp), b = *reinterpret_cast<const float4*>(sp + 4);
            v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = c + j < C ? sp[j] : 0.f;
        }
        u32x4 q0, q1, q2;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned a0, a1, a2;
            split_pair(v[2 * w], v[2 * w + 1], a0, a1, a2);
            q0[w] = a0; q1[w] = a1; q2[w] = a2;
        }
        unsigned short* dp = dst + r * ldp + c;
        *reinterpret_cast<u32x4*>(dp) = q0;
        *reinterpret_cast<u32x4*>(dp + pstride) = q1;
        *reinterpret_cast<u32x4*>(dp + 2 * pstride) = q2;
    }
}

struct Plan {
    int cfg;       // 0: 128x64, 1: 64x128, 2: 128x128, 3: 64x64
    int BM, BN;
    int mtiles, ntiles, split, kslab;
};

constexpr int kBK = 32;

Plan make_plan(int M, int N, int K, int split_k) {
    Plan p;
    // skinny outputs with many rows (the first layer at throughput batches: 4096 x 64 x 20000 and 20000 x 64 x 4096):
    // 64 x 64 tiles, 16 KB of LDS -> up to 9 workgroups per CU interleave their load / barrier / MFMA phases
    // (tools/bench_gemm_stages.py: 0.140 ms against 0.159 for the weight gradient, 0.144 against 0.148 forward;
    // 256 x 64 tiles: 0.21-0.29 ms)
    if (N <= 64 && M >= 2048) { p.cfg = 3; p.BM = 64; p.BN = 64; }
    else if (N <= 64) { p.cfg = 0; p.BM = 128; p.BN = 64; }
    else if (M <= 64) { p.cfg = 1; p.BM = 64; p.BN = 128; }
    else { p.cfg = 2; p.BM = 128; p.BN = 128; }
    // mid-size outputs (the hidden stack of the wide networks at throughput batches: 2048 x 256 x 512, 512 x 256 x 2048 ...):
    // at most 64 tiles of 128 x 128 -- a quarter of the chip -- which only a deep split-K (+ its reduce launch) could spread;
    // 64 x 64 tiles fill it with little or no split (tools/bench_hidden_gemm.py: the twelve products of configs[4]'s stack
    // 283 -> 160 us; profiles/r03_hidden_gemm_notes.txt)
#ifndef DCA_GEMM_NO_MID64
    const bool mid = p.cfg == 2 && (long)M * N <= 2048L * 512;
#else
    const bool mid = false;
#endif
    if (mid) { p.cfg = 3; p.BM = 64; p.BN = 64; }
    p.mtiles = (M + p.BM - 1) / p.BM;
    p.ntiles = (N + p.BN - 1) / p.BN;
    const int nchunks = (K + kBK - 1) / kBK;
    int S = split_k;
    if (S <= 0) {
        const long tiles = (long)p.mtiles * p.ntiles;
        // ~4 workgroups per CU (256 CUs), never more: measured on the step's skinny shapes at
        // B = 4096 (tools/bench_gemm.py): 4096x64x20000 best at 32 tiles x 32 splits = 1024
        // workgroups (0.147 ms vs 0.239 at 384), 20000x64x4096 at 157 x 6 = 942 (0.160 vs 0.214)
        S = 1;
        const long target = p.cfg == 3 ? 1536 : 1024;
        if (mid) {                                       // one workgroup per CU, K slices of at least 128
            S = (int)(256 / tiles);
            if (S > nchunks / 4) S = nchunks / 4;
            if (S < 1) S = 1;
        } else if (tiles < target) {
            S = (int)(target / tiles);
            if (S > nchunks / 2) S = nchunks / 2;
            if (S > 128) S = 128;
            if (S < 1) S = 1;
        }
    }
    if (S > nchunks) S = nchunks;
    if (S < 1) S = 1;
    const int cps = (nchunks + S - 1) / S;          // chunks per split
    p.kslab = cps * kBK;
    p.split = (nchunks + cps - 1) / cps;            // drop empty trailing splits
    return p;
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// A/B builds only (hipcc -DDCA_GEMM_F32MFMA / -DDCA_GEMM_X3): the exact-fp32 MFMA kernel / the split-bf16 kernel for every
// layout.  The shipped library has no run-time switches: the choice is a pure function of layout and shape.
#ifdef DCA_GEMM_F32MFMA
constexpr bool use_f32_mfma() { return true; }
#else
constexpr bool use_f32_mfma() { return false; }
#endif
#ifdef DCA_GEMM_X3
constexpr bool force_x3() { return true; }
#else
constexpr bool force_x3() { return false; }
#endif

template <int BM, int BN, int WGM, int WGN>
int launch_cfg(const GemmArgs& a, int ta, int tb, bool vec, int grid, hipStream_t s, bool exact) {
#define DCA_L(TA, TB, V) do { if (exact) hipLaunchKernelGGL((gemm_kernel<BM, BN, kBK, WGM, WGN, TA, TB, V>), dim3(grid), dim3(256), 0, s, a); \
                              else hipLaunchKernelGGL((gemm_x3_kernel<BM, BN, WGM, WGN, TA, TB, V>), dim3(grid), dim3(256), 0, s, a); } while (0)
    if (!ta && !tb) { if (vec) DCA_L(false, false, 4); else DCA_L(false, false, 1); }
    else if (ta && !tb) { if (vec) DCA_L(true, false, 4); else DCA_L(true, false, 1); }
    else if (!ta && tb) { if (vec) DCA_L(false, true, 4); else DCA_L(false, true, 1); }
    else return DCAHIP_EINVAL;
#undef DCA_L
    return (int)hipGetLastError();
}

}  // namespace

extern "C" long dcahip_sgemm_workspace_bytes(int ta, int tb, int M, int N, int K, int colsum_row,
                                             int split_k) {
    (void)ta; (void)tb;
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    const Plan p = make_plan(M, N, K, split_k);
    if (p.split <= 1) return 0;
    return (long)p.split * (M + (colsum_row ? 1 : 0)) * N * (long)sizeof(float);
}

extern "C" int dcahip_sgemm(int ta, int tb, int M, int N, int K, const float* A, long lda,
                            const float* B, long ldb, float* C, long ldc, const float* bias,
                            const int* perm, const long long* cursor, int colsum_row, int split_k,
                            void* workspace, long workspace_bytes, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C) return DCAHIP_EINVAL;
    if (ta && tb) return DCAHIP_EINVAL;
    if (colsum_row && tb) return DCAHIP_EINVAL;          // the column sums are taken where B is stored [K, N]
    const Plan p = make_plan(M, N, K, split_k);
    const long need = dcahip_sgemm_workspace_bytes(ta, tb, M, N, K, colsum_row, split_k);
    if (need > 0 && (!workspace || workspace_bytes < need)) return DCAHIP_EINVAL;
    // 16-byte vector loads need aligned bases and leading dimensions
    const bool vec = al16(A) && al16(B) && (lda % 4 == 0) && (ldb % 4 == 0);
    GemmArgs a{A, B, C, bias, perm, cursor, static_cast<float*>(workspace), lda, ldb, ldc,
               M, N, K, p.split, p.kslab, colsum_row, p.mtiles, p.ntiles};
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int grid = p.mtiles * p.ntiles * p.split;
    int rc;
    // Which arithmetic (a pure function of the layout and the shape; results of both are fp32-accurate):
    //   NT (both operands k-contiguous: direct 16-byte LDS operand reads)       split-bf16, measured 1.35x - 2.1x faster
    //   NN with >= 128 columns (B through the transposing read)                 split-bf16, 1.07x - 1.13x
    //   NN with fewer columns, TN                                               exact fp32 MFMA (the split kernel's larger LDS
    //       images cost more occupancy than the shorter matrix phase returns on these latency-bound shapes: 0.158 vs 0.157 ms
    //       and 0.195 vs 0.150 ms on the first layer at B = 4096; profiles/r02e_gemm_ab.txt)
    const bool exact = use_f32_mfma() || split_k < 0 || (ta && !force_x3()) || (!ta && !tb && N < 128 && !force_x3());
    if (p.cfg == 0) rc = launch_cfg<128, 64, 4, 1>(a, ta, tb, vec, grid, s, exact);
    else if (p.cfg == 1) rc = launch_cfg<64, 128, 1, 4>(a, ta, tb, vec, grid, s, exact);
    else if (p.cfg == 3) rc = launch_cfg<64, 64, 2, 2>(a, ta, tb, vec, grid, s, exact);
    else rc = launch_cfg<128, 128, 2, 2>(a, ta, tb, vec, grid, s, exact);
    if (rc != 0) return rc;
    if (p.split > 1) {
        const int Mo = M + (colsum_row ? 1 : 0);
        const long total = (long)Mo * N;
        long g = (total + 255) / 256;
        if (g > 4096) g = 4096;
        if (p.split >= 16 && total <= 65536) {
            long g16 = (total + 15) / 16;
            if (g16 > 4096) g16 = 4096;
            hipLaunchKernelGGL(splitk_reduce_kernel<16>, dim3((int)g16), dim3(256), 0, s, a.ws, p.split, Mo, N,
                               bias, M, C, ldc);
        } else {
            hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3((int)g), dim3(256), 0, s, a.ws, p.split, Mo, N,
                               bias, M, C, ldc);
        }
        rc = (int)hipGetLastError();
    }
    return rc;
}


// ---- pre-split operands
extern "C" int dcahip_split_planes(const float* src, long ld, const int* perm, const long long* cursor, long R, int C,
                                   void* planes, long ldp, long plane_stride, void* stream) {
    if (!src || !planes || R <= 0 || C <= 0 || ld < C || ldp < C || ldp % 8 != 0 || plane_stride < R * ldp) return DCAHIP_EINVAL;
    if (!al16(planes) || plane_stride % 8 != 0) return DCAHIP_EINVAL;
    const long total = R * (ldp / 8);
    long g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(split_planes_kernel, dim3((int)g), dim3(256), 0, static_cast<hipStream_t>(stream), src, ld, perm, cursor,
                       R, C, static_cast<unsigned short*>(planes), ldp, plane_stride);
    return (int)hipGetLastError();
}

// wide: the 256 x 256 kernel (large outputs, K a multiple of 16, no row gather: the caller says so)
static Plan make_plan3(int M, int N, int K, int split_k, bool wide) {
    Plan p;
    p.cfg = wide ? 4 : 2; p.BM = p.BN = wide ? 256 : 128;
    p.mtiles = (M + p.BM - 1) / p.BM;
    p.ntiles = (N + p.BN - 1) / p.BN;
    const int nchunks = (K + kBK - 1) / kBK;
    int S = split_k;
    if (S <= 0) {
        const long tiles = (long)p.mtiles * p.ntiles;
        const long target = wide ? 256 : 512;           // one / two workgroups per CU
        S = 1;
        if (tiles < target) {
            S = (int)(target / tiles);
            if (S > nchunks / 4) S = nchunks / 4;
            if (S > 64) S = 64;
            if (S < 1) S = 1;
        }
    }
    if (S > nchunks) S = nchunks;
    if (S < 1) S = 1;
    const int cps = (nchunks + S - 1) / S;
    p.kslab = cps * kBK;
    p.split = (nchunks + cps - 1) / cps;
    return p;
}

static bool p3_wide(int M, int N, int K, const int* perm, const long long* cursor) {
    return M >= 256 && N >= 256 && K % 16 == 0 && (long)M * N >= 512L * 512 && !perm && !cursor;
}

// tail splitting of the 256 x 256 kernel (split == 1 only): T tiles on 256 CUs run as ceil(T / 256) rounds; a last round of
// r < 256 whole tiles lasts as long as a full one.  Cut its tiles into S K slices (r S <= 256 workgroups, each 1 / S of a
// tile): the round shrinks to its share of the work, for r S partial tiles written and summed (heads' weight gradient of
// the 512-wide network: 586 tiles = 2.29 rounds ran as 3; 0.89 -> 0.7 ms)
struct TailPlan { int first, split, kslab; long ws_bytes; };
static TailPlan tail_plan(const Plan& p, int M, int K, int colsum_row) {
    TailPlan t{0, 1, 0, 0};
    if (p.cfg != 4 || p.split != 1) return t;
    const int T = p.mtiles * p.ntiles, rem = T % 256, nsteps = K / kWBK;
    if (T < 256 || rem == 0 || rem > 200) return t;
    int S = 256 / rem;
    if (S > nsteps / 4) S = nsteps / 4;
    if (S > 8) S = 8;
    if (S < 2) return t;
    const int first = T - rem;
    if (colsum_row && first / p.ntiles == 0) return t;                  // the column sums come from whole tiles of m tile 0
    const int sps = (nsteps + S - 1) / S;
    t.first = first; t.kslab = sps * kWBK; t.split = (nsteps + sps - 1) / sps;
    if (t.split < 2) { t.split = 1; return t; }
    t.ws_bytes = (long)rem * t.split * 256L * 256 * (long)sizeof(float);
    (void)M;
    return t;
}

extern "C" long dcahip_gemm_p3_workspace_bytes(int M, int N, int K, int colsum_row, int split_k) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    // (the larger of the two plans a call of this shape can take: with and without a row gather)
    const Plan pw = make_plan3(M, N, K, split_k, p3_wide(M, N, K, nullptr, nullptr));
    const Plan pn = make_plan3(M, N, K, split_k, false);
    const Plan p = pw.split > pn.split ? pw : pn;
    const long tail = tail_plan(pw, M, K, colsum_row).ws_bytes;
    if (tail > 0 && p.split <= 1) return tail;
    if (p.split <= 1) return 0;
    return (long)p.split * (M + (colsum_row ? 1 : 0)) * N * (long)sizeof(float);
}

extern "C" int dcahip_gemm_p3(int ta, int tb, int M, int N, int K, const void* A, long lda, long plane_a,
                              const void* B, long ldb, long plane_b, float* C, long ldc, const float* bias,
                              const int* perm, const long long* cursor, int colsum_row, int split_k,
                              void* workspace, long workspace_bytes, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || ldc < N) return DCAHIP_EINVAL;
    if (colsum_row && tb) return DCAHIP_EINVAL;
    if (!al16(A) || !al16(B) || lda % 8 != 0 || ldb % 8 != 0 || plane_a % 8 != 0 || plane_b % 8 != 0) return DCAHIP_EINVAL;
    // k-contiguous operands are read in units of 8 k: K must be a multiple of 8 (pad both operands with zeros)
    if ((!ta || tb) && K % 8 != 0) return DCAHIP_EINVAL;
    // rows must cover the 16-byte units the tile reads: ld >= the extent rounded up to 8
    if ((ta ? lda < (M + 7) / 8 * 8 : lda < K) || (tb ? ldb < K : ldb < (N + 7) / 8 * 8)) return DCAHIP_EINVAL;
    const Plan p = make_plan3(M, N, K, split_k, p3_wide(M, N, K, perm, cursor));
    const TailPlan tp = tail_plan(p, M, K, colsum_row);
    const long need = p.split > 1 ? (long)p.split * (M + (colsum_row ? 1 : 0)) * N * (long)sizeof(float) : tp.ws_bytes;
    if (need > 0 && (!workspace || workspace_bytes < need)) return DCAHIP_EINVAL;
    Gemm3Args a{static_cast<const unsigned short*>(A), static_cast<const unsigned short*>(B), lda, ldb, plane_a, plane_b,
                C, bias, perm, cursor, static_cast<float*>(workspace), ldc, M, N, K, p.split, p.kslab, colsum_row,
                p.mtiles, p.ntiles, tp.first, tp.split, tp.kslab, static_cast<float*>(workspace)};
    hipStream_t s = static_cast<hipStream_t>(stream);
    int grid = p.mtiles * p.ntiles * p.split;
    if (tp.split > 1) grid = tp.first + (p.mtiles * p.ntiles - tp.first) * tp.split;
    if (p.cfg == 4) {
        // (dynamic LDS above 64 KB needs the attribute once per kernel)
#define DCA_W(AKC, BKC, CSV) do { \
            constexpr int bytes = 3 * (WideImage<AKC>::BYTES + WideImage<BKC>::BYTES); \
            static bool set = false; \
            if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_p3w_kernel<AKC, BKC, CSV>), \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes); set = true; } \
            hipLaunchKernelGGL((gemm_p3w_kernel<AKC, BKC, CSV>), dim3(grid), dim3(512), bytes, s, a); } while (0)
        if (!ta && !tb) { if (colsum_row) DCA_W(true, false, true); else DCA_W(true, false, false); }
        else if (!ta && tb) DCA_W(true, true, false);
        else if (ta && !tb) { if (colsum_row) DCA_W(false, false, true); else DCA_W(false, false, false); }
        else DCA_W(false, true, false);
#undef DCA_W
    } else if (!ta && !tb) hipLaunchKernelGGL((gemm_p3_kernel<true, false>), dim3(grid), dim3(256), 0, s, a);
    else if (!ta && tb) hipLaunchKernelGGL((gemm_p3_kernel<true, true>), dim3(grid), dim3(256), 0, s, a);
    else if (ta && !tb) hipLaunchKernelGGL((gemm_p3_kernel<false, false>), dim3(grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((gemm_p3_kernel<false, true>), dim3(grid), dim3(256), 0, s, a);
    int rc = (int)hipGetLastError();
    if (rc != 0) return rc;
    if (p.split > 1) {
        const int Mo = M + (colsum_row ? 1 : 0);
        const long total = (long)Mo * N;
        long g = (total + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3((int)g), dim3(256), 0, s, a.ws, p.split, Mo, N, bias, M, C, ldc);
        rc = (int)hipGetLastError();
    } else if (tp.split > 1) {
        const int ntail = p.mtiles * p.ntiles - tp.first;
        hipLaunchKernelGGL(gemm_p3w_tail_sum_kernel, dim3(ntail * 16), dim3(256), 0, s, a.tail_ws, tp.first, tp.split, p.ntiles,
                           M, N, bias, C, ldc);
        rc = (int)hipGetLastError();
    }
    return rc;
}

// ---------------------------------------------------------------------------------------------------------------------
// fp16 x 2 planes (see gemm_h2w_kernel)
extern "C" int dcahip_absmax_exp(const float* src, long ld, long R, int C, int* exp_out, void* scratch_word, void* stream) {
    if (!src || !exp_out || !scratch_word || R <= 0 || C <= 0 || ld < C) return DCAHIP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    unsigned* amax = static_cast<unsigned*>(scratch_word);
    int rc = (int)hipMemsetAsync(amax, 0, sizeof(unsigned), s);
    if (rc != 0) return rc;
    const long total = R * ((C + 3) / 4);
    long g = (total + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(absmax_kernel, dim3((int)g), dim3(256), 0, s, src, ld, R, C, amax);
    hipLaunchKernelGGL(absmax_exp_kernel, dim3(1), dim3(1), 0, s, amax, exp_out);
    return (int)hipGetLastError();
}

extern "C" int dcahip_split_planes_h2(const float* src, long ld, const int* perm, const long long* cursor, long R, int C,
                                      void* planes, long ldp, long plane_stride, const int* exp, void* stream) {
    if (!src || !planes || R <= 0 || C <= 0 || ld < C || ldp < C || ldp % 8 != 0 || plane_stride < R * ldp) return DCAHIP_EINVAL;
    if (!al16(planes) || plane_stride % 8 != 0) return DCAHIP_EINVAL;
    const long total = R * (ldp / 8);
    long g = (total + 255) / 256;
    if (g > 16384) g = 16384;
    hipLaunchKernelGGL(split_planes_h2_kernel, dim3((int)g), dim3(256), 0, static_cast<hipStream_t>(stream), src, ld, perm, cursor,
                       R, C, static_cast<unsigned short*>(planes), ldp, plane_stride, exp);
    return (int)hipGetLastError();
}

// the shapes the 256 x 256 kernel takes (the same as the wide form of dcahip_gemm_p3)
extern "C" int dcahip_gemm_h2_supported(int M, int N, int K) { return p3_wide(M, N, K, nullptr, nullptr) ? 1 : 0; }

extern "C" long dcahip_gemm_h2_workspace_bytes(int M, int N, int K, int colsum_row, int split_k) {
    if (!p3_wide(M, N, K, nullptr, nullptr)) return 0;
    const Plan p = make_plan3(M, N, K, split_k, true);
    if (p.split > 1) return (long)p.split * (M + (colsum_row ? 1 : 0)) * N * (long)sizeof(float);
    return tail_plan(p, M, K, colsum_row).ws_bytes;
}

extern "C" int dcahip_gemm_h2(int ta, int tb, int M, int N, int K, const void* A, long lda, long plane_a, const int* exp_a, int exp_a_add,
                              const void* B, long ldb, long plane_b, const int* exp_b, int exp_b_add, float alpha,
                              float* C, long ldc, const float* bias, int colsum_row, int split_k,
                              void* workspace, long workspace_bytes, void* stream) {
    if (M <= 0 || N <= 0 || K <= 0 || !A || !B || !C || ldc < N) return DCAHIP_EINVAL;
    if (!p3_wide(M, N, K, nullptr, nullptr)) return DCAHIP_EINVAL;
    if (colsum_row && tb) return DCAHIP_EINVAL;
    if (!al16(A) || !al16(B) || lda % 8 != 0 || ldb % 8 != 0 || plane_a % 8 != 0 || plane_b % 8 != 0) return DCAHIP_EINVAL;
    if ((!ta || tb) && K % 8 != 0) return DCAHIP_EINVAL;
    if ((ta ? lda < (M + 7) / 8 * 8 : lda < K) || (tb ? ldb < K : ldb < (N + 7) / 8 * 8)) return DCAHIP_EINVAL;
    const Plan p = make_plan3(M, N, K, split_k, true);
    const TailPlan tp = tail_plan(p, M, K, colsum_row);
    const long need = p.split > 1 ? (long)p.split * (M + (colsum_row ? 1 : 0)) * N * (long)sizeof(float) : tp.ws_bytes;
    if (need > 0 && (!workspace || workspace_bytes < need)) return DCAHIP_EINVAL;
    GemmH2Args a{static_cast<const unsigned short*>(A), static_cast<const unsigned short*>(B), lda, ldb, plane_a, plane_b,
                 C, bias, static_cast<float*>(workspace), ldc, M, N, K, p.split, p.kslab, colsum_row,
                 p.mtiles, p.ntiles, tp.first, tp.split, tp.kslab, static_cast<float*>(workspace), exp_a, exp_b, exp_a_add, exp_b_add, alpha};
    hipStream_t s = static_cast<hipStream_t>(stream);
#ifndef DCA_EXP_H2_NOHALFM
    if (!ta && !tb && !colsum_row) {
        // forward products (A k-contiguous, B n-contiguous): 128 x 256 tiles by four waves, two workgroups per CU
        // (gemm_h2m_kernel).  Measured at configs[4] (tools/ab_heads_lib.py, one box): heads forward 0.87 - 0.90 -> 0.73 - 0.74 ms,
        // first layer forward (split-K) 0.368 -> 0.362; the input gradient (B k-contiguous, split-K) 0.50 -> 0.64: stays on
        // the 8-wave kernel
        GemmH2Args h = a;
        h.mtiles = (M + 127) / 128;
        h.tail_first = 0; h.tail_split = 1;
        const int gridm = h.mtiles * h.ntiles * h.split;
#define DCA_M(BKC) do { \
            constexpr int bytes = 3 * (WideImageH<true, 4>::BYTES + WideImageH<BKC, 8>::BYTES); \
            static bool set = false; \
            if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h2m_kernel<BKC>), \
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, bytes); set = true; } \
            hipLaunchKernelGGL((gemm_h2m_kernel<BKC>), dim3(gridm), dim3(256), bytes, s, h); } while (0)
        DCA_M(false);
#undef DCA_M
        int rcm = (int)hipGetLastError();
        if (rcm != 0) return rcm;
        if (h.split > 1) {
            const long total = (long)M * N;
            long g = (total + 255) / 256;
            if (g > 4096) g = 4096;
            hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3((int)g), dim3(256), 0, s, h.ws, h.split, M, N, bias, M, C, ldc);
            rcm = (int)hipGetLastError();
        }
        return rcm;
    }
#endif
    int grid = p.mtiles * p.ntiles * p.split;
    if (tp.split > 1) grid = tp.first + (p.mtiles * p.ntiles - tp.first) * tp.split;
#define DCA_W(AKC, BKC, CSV) do { \
        constexpr int bytes = 3 * (WideImageH<AKC>::BYTES + WideImageH<BKC>::BYTES); \
        static bool set = false; \
        if (!set) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_h2w_kernel<AKC, BKC, CSV>), \
                                              hipFuncAttributeMaxDynamicSharedMemorySize, bytes); set = true; } \
        hipLaunchKernelGGL((gemm_h2w_kernel<AKC, BKC, CSV>), dim3(grid), dim3(512), bytes, s, a); } while (0)
    if (!ta && !tb) { if (colsum_row) DCA_W(true, false, true); else DCA_W(true, false, false); }
    else if (!ta && tb) DCA_W(true, true, false);
    else if (ta && !tb) { if (colsum_row) DCA_W(false, false, true); else DCA_W(false, false, false); }
    else DCA_W(false, true, false);
#undef DCA_W
    int rc = (int)hipGetLastError();
    if (rc != 0) return rc;
    if (p.split > 1) {
        const int Mo = M + (colsum_row ? 1 : 0);
        const long total = (long)Mo * N;
        long g = (total + 255) / 256;
        if (g > 4096) g = 4096;
        hipLaunchKernelGGL(splitk_reduce_kernel<1>, dim3((int)g), dim3(256), 0, s, a.ws, p.split, Mo, N, bias, M, C, ldc);
        rc = (int)hipGetLastError();
    } else if (tp.split > 1) {
        const int ntail = p.mtiles * p.ntiles - tp.first;
        hipLaunchKernelGGL(gemm_p3w_tail_sum_kernel, dim3(ntail * 16), dim3(256), 0, s, a.tail_ws, tp.first, tp.split, p.ntiles,
                           M, N, bias, C, ldc);
        rc = (int)hipGetLastError();
    }
    return rc;
}
