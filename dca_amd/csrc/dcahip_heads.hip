// K-HEADS: the output heads of the autoencoder as ONE kernel -- forward GEMM of the three
// Dense heads, NB / ZINB negative log-likelihood + gradient, weight/bias gradient and input
// gradient -- so that the [cells x genes] pre-activation and gradient planes never exist in
// HBM.  gfx950 (MI355X), wave64.
//
// Reference path replaced: dca/network.py:369-385 (pi / dispersion / mean Dense heads,
// ColwiseMultLayer, SliceLayer, ZINB loss closure), dca/network.py:38-39, dca/layers.py:21,85,
// dca/loss.py:72-156 and TensorFlow's autodiff of all of it.
//
// Arithmetic of the three matrix products: fp32 results on the bf16 matrix pipe.  Every fp32 operand is
// split into three bf16 pieces (x = x1 + x2 + x3, round-to-nearest residuals) and the six products
// a1b1, a1b2, a2b1, a1b3, a2b2, a3b1 are accumulated in fp32 by v_mfma_f32_32x32x16_bf16: the dropped terms
// are below 2^-24 of sum|ab|, measured error 1.0e-7 of sum|ab| against 1.8e-7 for the fp32 MFMA
// (tools/microbench/bf16x3_mfma.hip), at 6/16 of its matrix-pipe cycles.  The fp32 MFMA runs at the VECTOR
// rate on gfx950 and shares the SIMD with the likelihood arithmetic; this form does not.
// (The first implementation, on v_mfma_f32_32x32x2_f32, is kept below for A/B runs: DCA_HEADS_F32MFMA=1.)
//
// Work decomposition (gene-stationary):
//   * a workgroup owns ONE 32-gene tile of every head: its [hL x 32 x heads] slice of the head weights sits in
//     LDS for the lifetime of the workgroup as bf16 pieces, ONE image serving both orientations (the forward
//     contracts over hidden units and reads it with the transposing ds_read_b64_tr_b16, dH contracts over
//     genes and reads it directly; a 16-byte-unit rotation per row keeps both conflict-free);
//   * its 8 waves take strided 32-row batch tiles; a wave's weight-gradient slice ([hL x 32] per head) lives
//     in MFMA accumulators for the lifetime of the wave;
//   * the decoder output H is split once per launch by a small kernel into bf16 pieces in the two operand
//     layouts the loop needs (rows x k for the forward, k x rows for dW): A operands are plain 16-byte loads;
//   * per (row tile, gene tile):
//       F   pre-activations  A = H W + b        32 rows x 32 genes x heads, K = 64
//       Z   NLL + d NLL / d A   element-wise, through a wave-private fp32 LDS staging tile (the y = 0
//           formulas densely, the non-zero elements compacted into 64-lane batches)
//       dH  = D W^T   (D read transposed from the staging tile, split on the fly) -> partial per gene tile
//       dW += H^T D   (a lane's staged D column IS its B operand)
//     No workgroup barrier inside the loop: waves drift apart, one wave's element-wise phase runs beside
//     its SIMD partner's matrix phases.
//   * HBM traffic per element: y (4 B) + the dH partial (8 B written, 8 B re-read by the
//     reduce) instead of 36 B for materialised pre-activations / gradients + 28 B K-ZINB.
//   * deterministic: fixed summation orders everywhere (no atomics).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "dcahip.h"
#include "zinb_math.hpp"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kTG = 32;        // genes per wave tile
constexpr int kTR = 32;        // batch rows per tile
constexpr int kLdS = 33;       // odd LDS row stride: both operand orientations conflict-free
constexpr int kWG = 2;         // gene tiles per workgroup
constexpr int kMaxGrid = 2048;      // grid cap of the persistent kernels
constexpr int kMaxSmallGrid = 8192; // = dcahip_zinb_max_partials(): loss partials of the four-wave kernel (one per workgroup)
constexpr int kCUs = 256;
constexpr int kZU = 4;        // staged rows per Z group (two groups per loop iteration)
constexpr int kQCap = 320;     // non-zero queue entries per wave (< 64 left over + 4 x 64 pushed)
constexpr int kLdH = 65;       // row stride of the H tile parked in the staging buffer

#ifdef DCA_HEADS_TIMING
#define TSTAMP(i) { const long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define TSTAMP(i)
#endif

__device__ __forceinline__ int rowmap(int e, int hi) { return (e & 3) + 8 * (e >> 2) + 4 * hi; }

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// =====================================================================================================
// bf16 x 3 implementation (the product path)
// =====================================================================================================
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using bf16x2v = __attribute__((ext_vector_type(2))) __bf16;
using f32x2v = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using s16x4 = __attribute__((ext_vector_type(4))) short;

constexpr int kWR2 = 8;                 // row slots (waves) per workgroup, one gene tile per workgroup
constexpr int kWr8MinNT = 5;            // the 8-wave persistent kernel from this many row tiles on (see make_heads_plan)
constexpr int kHTile = 3 * 32 * 64;     // bf16 elements of one row tile of the split decoder output (3 pieces x 32 x 64)

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0)

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {           // v_cvt_pk_bf16_f32: a -> low half
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, bf16x2v));
}
// x = p0 + p1 + p2 (bf16 each, round-to-nearest residuals), two values per call; |x - sum| <= 2^-26 |x|
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& p0, unsigned& p1, unsigned& p2) {
    p0 = pk_bf16(x0, x1);
    const float r0 = x0 - __uint_as_float(p0 << 16), r1 = x1 - __uint_as_float(p0 & 0xffff0000u);
    p1 = pk_bf16(r0, r1);
    const float q0 = r0 - __uint_as_float(p1 << 16), q1 = r1 - __uint_as_float(p1 & 0xffff0000u);
    p2 = pk_bf16(q0, q1);
}
// eight fp32 values -> three MFMA operand fragments (element j of the fragment = value j)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 (&f)[3]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned a, b, c;
        split_pair(x[2 * j], x[2 * j + 1], a, b, c);
        f[0][j] = a; f[1][j] = b; f[2][j] = c;
    }
}
// the six products of one K = 16 step, small terms first
#define MFMA_X3(A, Bf, ACC) { ACC = MFMA16(A[2], Bf[0], ACC); ACC = MFMA16(A[1], Bf[1], ACC); ACC = MFMA16(A[0], Bf[2], ACC); \
                              ACC = MFMA16(A[1], Bf[0], ACC); ACC = MFMA16(A[0], Bf[1], ACC); ACC = MFMA16(A[0], Bf[0], ACC); }

#ifdef DCA_EXP_BWD3      // experiment: three products (a1b1 + a1b2 + a2b1, error ~2e-6 of sum|ab|) in the two backward products
#define MFMA_BWD(A, Bf, ACC) { ACC = MFMA16(A[1], Bf[0], ACC); ACC = MFMA16(A[0], Bf[1], ACC); ACC = MFMA16(A[0], Bf[0], ACC); }
#else
#define MFMA_BWD MFMA_X3
#endif

struct HeadsArgs2 {
    long long* timing;
    const unsigned short* HA;         // [NT][3][32 rows][64 k] bf16 pieces of the decoder output (forward A operand)
    const unsigned short* HT;         // [NT][3][64 k][32 rows, order of the MFMA row map] (dW A operand)
    const float* H; long ldh;         // the decoder output itself: single-wave workgroups (batches below 256 rows) split it on the fly
    float* gW; long ldg;              // S == 1: weight / bias gradients go straight to their destination (no partial buffer)
    float* g_theta;
    const float* Wh; long ldw;
    const float* bh;
    const float* theta_w;
    const float* y;  long ldy;
    const unsigned char* yc; long ldc;                  // compact counts (YC kernels): one byte per count, 255 = escape
    const int* ovf_ptr; const int* ovf_col; const float* ovf_val;   // per-row overflow list behind the escapes
    const float* sf;
    const int* perm;
    const long long* cursor;
    float* ws_dw;  long dw_stride;   // [S][(hL + 2)][ldws]
    float* ws_dh;                     // [NT][npart][32][64]: one partial per workgroup of the row tile's batch split
    int npart, nitems;                // partials per row tile (= grid / S), work items (= S x gene tiles)
    const int* tile_order;
    double* partials;
    long plane, ldws;
    int B, hL, G;
    int S, NT;
    float ridge, inv_n;
};

// the count behind an escape byte of the compact store (counts >= 255: rare)
__device__ __forceinline__ float escaped_count(const HeadsArgs2& p, long srow, int col) {
    float v = 255.f;
    if (p.ovf_ptr)
        for (int i = p.ovf_ptr[srow], e = p.ovf_ptr[srow + 1]; i < e; ++i)
            if (p.ovf_col[i] == col) { v = p.ovf_val[i]; break; }
    return v;
}

// Decoder output -> bf16 pieces, once per launch (B x 64 elements: ~1 us).  One 32-row tile per block.
__global__ __launch_bounds__(256) void heads_split_h_kernel(const float* H, long ldh, int B, int hL,
                                                            unsigned short* HA, unsigned short* HT) {
    __shared__ unsigned short tr[3][64][32 + 2];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int row = tid >> 3, k0 = (tid & 7) * 8;
    const long grow = (long)t * 32 + row;
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] = (grow < B && k0 + j < hL) ? H[grow * ldh + k0 + j] : 0.f;
    u32x4 f[3];
    split8(x, f);
    // row -> position inside the tile's transposed image: the order in which a lane of the 32x32 MFMA result
    // holds its rows (row bits 2 and 3 swapped), so that 8 consecutive positions are one K = 16 operand half
    const int pos = (row & 0x13) | ((row & 4) << 1) | ((row & 8) >> 1);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        *reinterpret_cast<u32x4*>(HA + (((long)t * 3 + q) * 32 + row) * 64 + k0) = f[q];
#pragma unroll
        for (int j = 0; j < 8; ++j) tr[q][k0 + j][pos] = (unsigned short)(f[q][j >> 1] >> (16 * (j & 1)));
    }
    __syncthreads();
    for (int c = tid; c < 3 * 64 * 4; c += 256) {           // 16-byte chunks of [3][64][32]
        const int q = c >> 8, i = (c >> 2) & 63, part = c & 3;
        u32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (unsigned)tr[q][i][part * 8 + 2 * j] | ((unsigned)tr[q][i][part * 8 + 2 * j + 1] << 16);
        *reinterpret_cast<u32x4*>(HT + (((long)t * 3 + q) * 64 + i) * 32 + part * 8) = v;
    }
}

template <bool HAS_PI, bool CONST_DISP, int WR, bool YC>
__global__ __launch_bounds__(64 * WR) void heads_fused_x3_kernel(HeadsArgs2 p) {
    using YV = std::conditional_t<YC, unsigned, float>;        // a count as the kernel holds it: the byte code / the fp32 value
    constexpr int NH = 1 + (CONST_DISP ? 0 : 1) + (HAS_PI ? 1 : 0);
    constexpr int PI_H = NH - 1;
    constexpr int KT = 64;
    constexpr int ST_PLANE = kTG * kLdS;
    constexpr int NP = NH + (CONST_DISP ? 1 : 0);
    constexpr int TH_P = NH;
    constexpr int ST_WAVE = NP * ST_PLANE;
    constexpr int NTHREADS = 64 * WR;
    constexpr int W_PIECE = 64 * 64;                       // bytes of one (head, piece) weight image: 64 k x 32 genes bf16
    constexpr int W_FLOATS = NH * 3 * W_PIECE / 4;
    constexpr int NRED = NH * 2 * 16 + NH + 1;
    constexpr int BIAS_FLOATS = (NH + 1) * 32;             // biases (+ log-dispersion) of the 32 genes
    constexpr int LDS_FLOATS = W_FLOATS + WR * (ST_WAVE + kQCap) + BIAS_FLOATS;
    static_assert(WR == 1 || (WR / 2) * NRED * 64 <= LDS_FLOATS, "dW reduce scratch");
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ double lred[WR];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef DCA_HEADS_TIMING
    const long long t_entry = __builtin_readcyclecounter();
    long long t_loop0 = t_entry, t_loop1 = t_entry;
#endif
    const int l31 = lane & 31, hi = lane >> 5;
    const int r = wave;
    const long long cur = p.cursor ? *p.cursor : 0;
    double dacc = 0.0;
    // Persistent workgroups: workgroup w = (batch split s = w % S, lane wq = w / S) takes one gene tile per round
    // (every item of a wave visits the same row tiles: those of split s).
    // The input gradient of those row tiles is ACCUMULATED over the workgroup's gene tiles in a workgroup-private
    // slice of the workspace (the first item writes, the others add): grid / S partial sums per row instead of one
    // per gene tile -- 5x fewer bytes to store and to reduce at G = 20 000.  Static assignment: deterministic.
    // Gene tiles arrive sorted by cost (tile_order): round j hands them out in forward order on even rounds and in
    // reverse on odd ones, so no workgroup collects the heaviest tile of every round.
    bool first_item = true;
    const int s_wg = blockIdx.x % p.S, wq = blockIdx.x / p.S;
    const int ngb = p.nitems / p.S;
    const int full = ngb / p.npart, rem = ngb - full * p.npart;          // the last round is partial
    const int nrounds = full + ((rem > 0 && ((full & 1) ? p.npart - 1 - wq : wq) < rem) ? 1 : 0);
#pragma unroll 1
    for (int j = 0; j < nrounds; ++j) {
    const int gb = j * p.npart + ((j & 1) ? p.npart - 1 - wq : wq);
    int s = s_wg;
    asm volatile("" : "+s"(s));              // opaque per round: what depends on it is recomputed, not kept live across rounds
    const int gt = p.tile_order ? p.tile_order[gb] : gb;
    const int g0 = gt * kTG;
    const int gene = g0 + l31;
    const bool tile_ok = g0 < p.G;
    const bool gvalid = gene < p.G;

    unsigned char* const Wimg = reinterpret_cast<unsigned char*>(lds);
    float* St = lds + W_FLOATS + wave * ST_WAVE;
    unsigned* Q = reinterpret_cast<unsigned*>(lds + W_FLOATS + WR * ST_WAVE) + wave * kQCap;
    float* const Bs = lds + W_FLOATS + WR * (ST_WAVE + kQCap);          // [head][32] biases, then [32] log-dispersion

    // ---- head weights of this gene tile -> LDS as bf16 pieces, image [head][piece][k][32 genes], the four
    // 16-byte units of a row rotated by (k >> 2): conflict-free for the direct 16-byte reads of dH (lanes = rows k)
    // and for the transposing reads of F (4 consecutive rows per lane group)
    if (tile_ok) {
        // batches of UNR independent 16-byte loads (clamped addresses, zeroed afterwards): one memory round trip per
        // batch -- a load-split-store loop costs one per ITERATION, 24 of them for a single-wave workgroup
        constexpr int UNR = WR == 1 ? 8 : 3;
        constexpr int ITEMS = NH * 64 * 8;
        static_assert(ITEMS % (NTHREADS * UNR) == 0 || WR != 1, "single-wave prologue batches");
        for (int base = tid; base < ITEMS; base += NTHREADS * UNR) {
            float4 v[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int idx = base + u * NTHREADS;
                const int c4 = idx & 7, kq = (idx >> 3) & 63, h = (idx >> 9) < NH ? (idx >> 9) : NH - 1;
                const int kc = kq < p.hL ? kq : p.hL - 1;
                long gcol = g0 + c4 * 4;
                if (gcol > p.plane - 4) gcol = p.plane - 4;
                v[u] = *reinterpret_cast<const float4*>(p.Wh + (long)kc * p.ldw + (long)h * p.plane + gcol);
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int idx = base + u * NTHREADS;
                if (idx >= ITEMS) continue;
                const int c4 = idx & 7, kq = (idx >> 3) & 63, h = idx >> 9;
                const int gcol = g0 + c4 * 4;
                const bool kv = kq < p.hL && gcol <= p.plane - 4;
                float4 w = v[u];
                if (!kv || gcol + 0 >= p.G) w.x = 0.f;
                if (!kv || gcol + 1 >= p.G) w.y = 0.f;
                if (!kv || gcol + 2 >= p.G) w.z = 0.f;
                if (!kv || gcol + 3 >= p.G) w.w = 0.f;
                unsigned a0, a1, a2, b0, b1, b2;
                split_pair(w.x, w.y, a0, a1, a2);
                split_pair(w.z, w.w, b0, b1, b2);
                const int off = kq * 64 + ((((c4 >> 1) + (kq >> 2)) & 3) << 4) + ((c4 & 1) << 3);
                *reinterpret_cast<u32x2*>(Wimg + (h * 3 + 0) * W_PIECE + off) = u32x2{a0, b0};
                *reinterpret_cast<u32x2*>(Wimg + (h * 3 + 1) * W_PIECE + off) = u32x2{a1, b1};
                *reinterpret_cast<u32x2*>(Wimg + (h * 3 + 2) * W_PIECE + off) = u32x2{a2, b2};
            }
        }
        if (tid < 32) {
            const bool gv = g0 + tid < p.G;
#pragma unroll
            for (int h = 0; h < NH; ++h) Bs[h * 32 + tid] = gv ? p.bh[(long)h * p.plane + g0 + tid] : 0.f;
            Bs[NH * 32 + tid] = (CONST_DISP && gv) ? p.theta_w[g0 + tid] : 0.f;
        }
    }
    __syncthreads();

    f32x16 dW[NH][2];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int e = 0; e < 16; ++e) dW[h][ib][e] = 0.f;
    float bsum[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) bsum[h] = 0.f;
    float thsum = 0.f;

    if (tile_ok) {
        // biases / log-dispersion of the lane's gene are read from LDS where they are used: three registers and --
        // more to the point -- no spill slot whose reload would wait for every global prefetch in flight
        const int gene_c = gvalid ? gene : p.G - 1;
        const float* const ycol = p.y + gene_c;
        const unsigned char* const ycolc = p.yc + gene_c;
        const unsigned ldy_u = YC ? (unsigned)p.ldc : (unsigned)p.ldy;
        auto count_at = [&](int sr) -> YV {
#ifdef DCA_EXP_YCACHED       // experiment (wrong results): every count from the same 8 storage rows -- always cache hits: prices the count loads' latency
            sr &= 7;
#endif
            if constexpr (YC) return (unsigned)ycolc[(unsigned long long)(unsigned)sr * ldy_u];
            else return ycol[(unsigned long long)(unsigned)sr * ldy_u];
        };
        const int tstep = p.S * WR;
        int t = s * WR + r;
        const long dh_tstride = (long)p.npart * (kTR * KT);
        float* const dh_part = p.ws_dh + (long)(blockIdx.x / p.S) * (kTR * KT);            // this workgroup's partials
        const int dh_lane = (4 * hi * KT + l31) * 4;
        int srow_l = 0;
        float sf_l = 1.f;
        YV yA[kZU], yB[kZU];
        auto row_clamped = [&](int tt) { const int rl = tt * kTR + l31; return rl < p.B ? rl : p.B - 1; };
        auto load_srow = [&](int tt) { const int rlc = row_clamped(tt); return p.perm ? p.perm[cur + rlc] : (int)(cur + rlc); };
        // split decoder output through buffer resources: (per-lane offset fixed for the whole kernel) + (tile offset
        // in an SGPR) + immediate
        const __amdgpu_buffer_rsrc_t ha_rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned short*>(p.HA), 0, p.NT * (kHTile * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t ht_rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<unsigned short*>(p.HT), 0, p.NT * (kHTile * 2), 0x00020000);
        const int ha_lane = l31 * 128 + hi * 64;       // row l31, this lane half's 32 hidden units (4 K steps x 8)
        const int ht_lane = l31 * 64 + hi * 16;        // hidden unit l31 (+ 32 ib), rows of K-step half hi
        // forward A operands [piece] of one K step and dW A operands [piece][ib] of one K step: requested one step
        // ahead of their use (the first forward step of the NEXT tile during the dW products of this one), never the
        // whole tile at once -- the weight-gradient accumulators (96 registers) leave no room for that
        u32x4 ha0[3];
        auto load_ha = [&](int tt, int ks, u32x4 (&dst)[3]) {
            if constexpr (WR == 1) {
                // small batches: no split pass in front of the kernel -- row l31 of the tile, hidden units 32 hi + 8 ks ..
                const int row = tt * kTR + l31;
                const float* hp = p.H + (long)(row < p.B ? row : p.B - 1) * p.ldh;
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {                      // unconditional loads (clamped), zeroed afterwards
                    const int kk = 32 * hi + 8 * ks + j;
                    const float v = hp[kk < p.hL ? kk : p.hL - 1];
                    x[j] = (row < p.B && kk < p.hL) ? v : 0.f;
                }
                split8(x, dst);
            } else {
                const int so = tt * (kHTile * 2);
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    dst[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ha_rs, ha_lane + q * 4096 + ks * 16, so, 0));
            }
        };
        auto load_ht = [&](int tt, int ks, u32x4 (&dst)[3][2]) {
            if constexpr (WR == 1) {
#pragma unroll
                for (int ib = 0; ib < 2; ++ib) {
                    const int i = l31 + 32 * ib;              // hidden unit; rows of K-step ks in the order of the MFMA row map
                    float x[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int row = tt * kTR + rowmap(8 * ks + j, hi);
                        const float v = p.H[(long)(row < p.B ? row : p.B - 1) * p.ldh + (i < p.hL ? i : p.hL - 1)];
                        x[j] = (row < p.B && i < p.hL) ? v : 0.f;
                    }
                    u32x4 f[3];
                    split8(x, f);
                    dst[0][ib] = f[0]; dst[1][ib] = f[1]; dst[2][ib] = f[2];
                }
            } else {
                const int so = tt * (kHTile * 2);
#pragma unroll
                for (int q = 0; q < 3; ++q)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib)
                        dst[q][ib] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                            ht_rs, ht_lane + q * 4096 + ib * 2048 + ks * 32, so, 0));
            }
        };
        // LDS addresses of the weight image.  Transposing read (F): lane t of a 16-lane group supplies the 8-byte
        // chunk (row kk + t / 4, genes 16 (group & 1) + 4 (t & 3) ..) and receives rows kk .. kk + 3 of gene
        // 16 (group & 1) + t.  kk = 32 hi + 8 ks (+ 4): its rotation (kk >> 2) & 3 = (2 ks (+ 1)) & 3 is a compile-time
        // constant, the lane part of the address one of four precomputed values.
        const int t16 = lane & 15, c8 = 4 * ((lane >> 4) & 1) + (t16 & 3);
        int wtr[4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
            wtr[rr] = hi * 2048 + (t16 >> 2) * 64 + ((((c8 >> 1) + rr) & 3) << 4) + ((c8 & 1) << 3);
        // direct read (dH): lane = hidden unit l31 (+ 32 jb), 8 genes 16 gs + 8 hi: unit (2 gs + hi + rotation)
        int wdr[2];
#pragma unroll
        for (int gs = 0; gs < 2; ++gs) wdr[gs] = l31 * 64 + (((2 * gs + hi + (l31 >> 2)) & 3) << 4);
        auto w_tr = [&](int h, int q, int ks) {           // B operand of F: gene l31, k = 32 hi + 8 ks .. + 7
            const unsigned char* b0 = Wimg + (h * 3 + q) * W_PIECE + ks * 512;
            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(b0 + wtr[(2 * ks) & 3]));
            const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                (__attribute__((address_space(3))) s16x4*)(b0 + 256 + wtr[(2 * ks + 1) & 3]));
            u32x4 o;
            const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi4);
            o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1];
            return o;
        };
        auto w_dr = [&](int h, int q, int jb, int gs) {   // B operand of dH: hidden unit l31 + 32 jb, genes 16 gs + 8 hi ..
            return *reinterpret_cast<const u32x4*>(Wimg + (h * 3 + q) * W_PIECE + jb * 2048 + wdr[gs]);
        };

        if (t < p.NT) {
            srow_l = load_srow(t);
            sf_l = p.sf[srow_l];
            load_ha(t, 0, ha0);
#pragma unroll
            for (int j = 0; j < kZU; ++j) {
                const int sr = __shfl(srow_l, rowmap(j, hi), 64);
                yA[j] = count_at(sr);
            }
        }
#ifdef DCA_HEADS_TIMING
        long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        long long tlast = __builtin_readcyclecounter();
        t_loop0 = tlast;
#endif
        int tile_no = wave >> 2;
        for (; t < p.NT; t += tstep) {
#if defined(DCA_EXP_PRIO_MFMA)
            __builtin_amdgcn_s_setprio(2);           // experiment: the matrix phases outrank the partner's likelihood phase
#elif defined(DCA_EXP_PRIO_Z)
            __builtin_amdgcn_s_setprio(0);
#elif !defined(DCA_EXP_NOPRIO)
            if (WR == 8) { if ((tile_no++) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
#endif
            TSTAMP(0)
            const int row0 = t * kTR;
            const int tn = t + tstep < p.NT ? t + tstep : t;
            // ---- F: pre-activations, K = 64 as 4 steps of 16 (lane half hi covers k in [32 hi, 32 hi + 32))
            // (the accumulators start from the bias of the lane's gene: one LDS read per head instead of 16 additions)
            f32x16 acc[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const float bias_h = Bs[h * 32 + l31];
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[h][e] = bias_h;
            }
            {
                u32x4 hb[2][3];
#pragma unroll
                for (int q = 0; q < 3; ++q) hb[0][q] = ha0[q];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks < 3) load_ha(t, ks + 1, hb[(ks + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);          // the request stays AHEAD of this step's products
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        u32x4 bf[3] = {w_tr(h, 0, ks), w_tr(h, 1, ks), w_tr(h, 2, ks)};
                        MFMA_X3(hb[ks & 1], bf, acc[h])
                    }
                }
            }
            TSTAMP(1)
            TSTAMP(2)
            // ---- stage [gene][row] (row stride 1, gene stride 33)
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int e = 0; e < 16; ++e) St[h * ST_PLANE + l31 * kLdS + rowmap(e, hi)] = acc[h][e];
            const float thw = CONST_DISP ? Bs[NH * 32 + l31] : 0.f;
            const int srow_n = load_srow(tn);
            wave_sync();
            TSTAMP(3)

            // ---- Z: element-wise likelihood and gradient (dense y = 0 pass + compacted non-zero pass)
#if defined(DCA_EXP_PRIO_MFMA)
            __builtin_amdgcn_s_setprio(0);
#elif defined(DCA_EXP_PRIO_Z)
            __builtin_amdgcn_s_setprio(2);
#endif
            float lacc = 0.f;
            int qn = 0;
            auto z_dense = [&](auto fullv, int grp, const YV (&yv)[kZU]) {
                constexpr bool FULLV = decltype(fullv)::value;
                float i_am[kZU], i_ad[kZU], i_ap[kZU];
#pragma unroll
                for (int j = 0; j < kZU; ++j) {
                    const int row = rowmap(grp * kZU + j, hi);
                    const int idx = l31 * kLdS + row;
                    i_am[j] = St[idx];
                    i_ad[j] = CONST_DISP ? thw : St[ST_PLANE + idx];
                    i_ap[j] = HAS_PI ? St[PI_H * ST_PLANE + idx] : 0.f;
                }
                float o_m[kZU], o_d[kZU], o_p[kZU];
                bool o_nz[kZU];
#pragma unroll
                for (int j = 0; j < kZU; ++j) {
                    const int row = rowmap(grp * kZU + j, hi);
                    const bool valid = FULLV || ((row0 + row < p.B) && gvalid);
                    const YV yj = yv[j];
                    bool nz;
                    if constexpr (YC) nz = valid && yj != 0u;
                    else nz = valid && (HAS_PI ? !(yj < kZeroThresh) : (yj != 0.f));
                    if (HAS_PI) {
                        float gmv, gdv, gpv;
                        const float nll = zinb_zero_elem<CONST_DISP>(i_am[j], i_ad[j], i_ap[j], __shfl(sf_l, row, 64), p.ridge, gmv, gdv, gpv);
                        const float sc = valid ? p.inv_n : 0.f;
                        lacc += (valid && !nz) ? nll : 0.f;
                        o_m[j] = gmv * sc;
                        o_d[j] = gdv * sc;
                        o_p[j] = gpv * sc;
                    } else {
                        float gmv, gdv;
                        const float nll = nb_zero_elem<CONST_DISP>(i_am[j], i_ad[j], __shfl(sf_l, row, 64), gmv, gdv);
                        const float sc = valid ? p.inv_n : 0.f;
                        lacc += (valid && !nz) ? nll : 0.f;
                        o_m[j] = gmv * sc;
                        o_d[j] = gdv * sc;
                        o_p[j] = 0.f;
                    }
                    o_nz[j] = nz;
                }
                // (pinning the four elements' outputs here -- so that their arithmetic stays one interleaved block instead of
                // being sunk into the `if (!nz)` branches below -- was measured: 0.888 / 0.876 ms against 0.868 / 0.864,
                // profiles/notes/r04_heads_wave_specialised/README.md; the pass is bound by the vector pipe, not by latency)
#pragma unroll
                for (int j = 0; j < kZU; ++j) {
                    const int row = rowmap(grp * kZU + j, hi);
                    const int idx = l31 * kLdS + row;
                    const bool nz = o_nz[j];
                    const unsigned long long m = __ballot(nz);
                    const int slot = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (nz) {
                        const YV yj = yv[j];
                        unsigned y16;
                        if constexpr (YC) y16 = yj == 255u ? 0xFFFFu : yj;
                        else y16 = (yj < 65535.f && yj == floorf(yj)) ? (unsigned)yj : 0xFFFFu;
                        Q[slot] = (unsigned)idx | (y16 << 16);
                    } else {
                        St[idx] = o_m[j];
                        if (CONST_DISP) St[TH_P * ST_PLANE + idx] = o_d[j]; else St[ST_PLANE + idx] = o_d[j];
                        if (HAS_PI) St[PI_H * ST_PLANE + idx] = o_p[j];
                    }
                    qn += __popcll(m);
                }
            };
            auto z_sparse = [&](int q0, int cnt) {
                const bool act = lane < cnt;
                const unsigned e = Q[q0 + (act ? lane : 0)];
                const int idx = e & 2047;
                const int gq = (idx * 1986) >> 16;          // idx / 33 for idx < 1056
                const int row = idx - gq * kLdS;
                const float sfr = __shfl(sf_l, row, 64);
                const int sr = __shfl(srow_l, row, 64);
                const float am = St[idx];
                const float ad = CONST_DISP ? Bs[NH * 32 + gq] : St[ST_PLANE + idx];
                const float ap = HAS_PI ? St[PI_H * ST_PLANE + idx] : 0.f;
                float yq = (float)(e >> 16);
                // counts that do not fit the queue's 16 bits (rare): a wave-uniform branch around the memory access, and the
                // wait for it INSIDE the branch -- left to the compiler, the join in front of the first use of yq waits for
                // vmcnt(0), i.e. for every prefetch in flight, in every batch
                if (__ballot(act && (e >> 16) == 0xFFFFu)) {
                    if (act && (e >> 16) == 0xFFFFu) yq = YC ? escaped_count(p, sr, g0 + gq) : p.y[(long)sr * p.ldy + g0 + gq];
                    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
                }
                float o1, o2, o3 = 0.f, nll;
                if (HAS_PI) {
                    nll = zinb_nz_elem<CONST_DISP, YC>(am, ad, ap, sfr, yq, p.ridge, o1, o2, o3);
                } else {
                    float dmu = 0.f, dth = 0.f, dpi = 0.f;
                    const Heads hd = head_acts<HAS_PI, CONST_DISP>(am, ad, ap, sfr);
                    nll = nll_elem<HAS_PI, true, true, YC>(hd, yq, p.ridge, dmu, dth, dpi);
                    o1 = dmu * hd.gm; o2 = dth * hd.gd;
                }
                lacc += act ? nll : 0.f;
                if (act) {
                    St[idx] = o1 * p.inv_n;
                    const float od = o2 * p.inv_n;
                    if (CONST_DISP) St[TH_P * ST_PLANE + idx] = od; else St[ST_PLANE + idx] = od;
                    if (HAS_PI) St[PI_H * ST_PLANE + idx] = o3 * p.inv_n;
                }
            };
            auto z_flush = [&](bool last) {
                while (qn >= 64 || (last && qn > 0)) {
                    const int c = qn < 64 ? qn : 64;
                    wave_sync();
                    z_sparse(qn - c, c);
                    qn -= c;
                }
            };
            auto load_y = [&](int srow_src, int grp, YV (&yv)[kZU]) {
#pragma unroll
                for (int j = 0; j < kZU; ++j) {
                    const int sr = __shfl(srow_src, rowmap(grp * kZU + j, hi), 64);
                    yv[j] = count_at(sr);
                }
            };
            auto z_loop = [&](auto fullv) {
#pragma unroll 1
                for (int it = 0; it < 16 / (2 * kZU); ++it) {
                    const bool last = it + 1 == 16 / (2 * kZU);
                    load_y(srow_l, 2 * it + 1, yB);
                    z_dense(fullv, 2 * it, yA);
                    z_flush(false);
                    if (!last) load_y(srow_l, 2 * it + 2, yA);
                    else load_y(srow_n, 0, yA);
                    z_dense(fullv, 2 * it + 1, yB);
                    z_flush(last);
                }
            };
#ifndef DCA_EXP_X3NOZ
            if (row0 + kTR <= p.B && g0 + kTG <= p.G) z_loop(std::true_type{}); else z_loop(std::false_type{});
#endif
            dacc += (double)lacc;
            const float sf_n = p.sf[srow_n];
            wave_sync();
#if defined(DCA_EXP_PRIO_MFMA)
            __builtin_amdgcn_s_setprio(2);
#elif defined(DCA_EXP_PRIO_Z)
            __builtin_amdgcn_s_setprio(0);
#endif
            TSTAMP(4)

            // ---- dH[row, i] = sum_genes D[row, gene] W[i, gene]: A = D read transposed from the staging tile
            // (row l31, 8 genes per K-step half) and split on the fly, B = the weight image read directly
            u32x4 htb[2][3][2];
            load_ht(t, 0, htb[0]);                   // first K step of the dW operands: in flight during the dH products
            __builtin_amdgcn_sched_barrier(0);
            {
                // the accumulators start from what the workgroup's earlier gene tiles left for this row tile (no extra
                // registers; the operand reads and splits below cover part of the load latency)
                f32x16 dHa[2];
                // the 8 KB partial of (this workgroup, row tile t) through a buffer resource: scalar base, ONE per-lane
                // offset, immediates -- 32 separate 64-bit addresses would not fit the register file
                const __amdgpu_buffer_rsrc_t dh_rs = __builtin_amdgcn_make_buffer_rsrc(
                    dh_part + (long)t * dh_tstride, 0, kTR * KT * 4, 0x00020000);
#ifdef DCA_EXP_NODHLOAD      // experiment (wrong results): every item starts its dH partial from zero -- prices the load's exposed latency
                if (true) {
#else
                if (first_item) {
#endif
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int e = 0; e < 16; ++e) dHa[jb][e] = 0.f;
                } else {
#pragma unroll
                    for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                        for (int e = 0; e < 16; ++e) {
                            const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(
                                dh_rs, dh_lane + (((rowmap(e, 0) * KT + jb * 32) * 4) & 4095), ((rowmap(e, 0) * KT + jb * 32) * 4) & ~4095, 0);
                            dHa[jb][e] = __uint_as_float(u);
                        }
                }
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int gs = 0; gs < 2; ++gs) {
                        float dv[8];
#pragma unroll
                        for (int j = 0; j < 8; ++j) dv[j] = St[h * ST_PLANE + (16 * gs + 8 * hi + j) * kLdS + l31];
                        u32x4 af[3];
                        split8(dv, af);
#pragma unroll
                        for (int jb = 0; jb < 2; ++jb) {
                            u32x4 bf[3] = {w_dr(h, 0, jb, gs), w_dr(h, 1, jb, gs), w_dr(h, 2, jb, gs)};
                            MFMA_BWD(af, bf, dHa[jb])
                        }
                    }
                TSTAMP(7)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const float v = dHa[jb][e];          // (a bit_cast of the vector element itself stores element 0)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), dh_rs,
                                                              dh_lane + (((rowmap(e, 0) * KT + jb * 32) * 4) & 4095),
                                                              ((rowmap(e, 0) * KT + jb * 32) * 4) & ~4095, 0);   // 12-bit immediate + scalar offset
                    }
            }
            load_ht(t, 1, htb[1]);
            load_ha(tn, 0, ha0);                     // next tile's first forward step: in flight during the dW products
            __builtin_amdgcn_sched_barrier(0);
            TSTAMP(5)
            // ---- dW[i, gene] += sum_rows H[row, i] D[row, gene]: B = the lane's own staged D column (rows in
            // the order of the MFMA row map = the order of the transposed H image), A = H^T pieces
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    float dv[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        dv[j] = St[h * ST_PLANE + l31 * kLdS + rowmap(8 * ks + j, hi)];
                        bsum[h] += dv[j];
                    }
                    u32x4 bf[3];
                    split8(dv, bf);
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        u32x4 af[3] = {htb[ks][0][ib], htb[ks][1][ib], htb[ks][2][ib]};
                        MFMA_BWD(af, bf, dW[h][ib])
                    }
                }
            if (CONST_DISP) {
#pragma unroll
                for (int e = 0; e < 16; ++e) thsum += St[TH_P * ST_PLANE + l31 * kLdS + rowmap(e, hi)];
            }
            srow_l = srow_n;
            sf_l = sf_n;
            wave_sync();
            TSTAMP(6)
        }
#ifdef DCA_HEADS_TIMING
        t_loop1 = __builtin_readcyclecounter();
        if (p.timing && lane == 0)
            for (int i = 0; i < 8; ++i) p.timing[((long)blockIdx.x * WR + wave) * 10 + i] += tacc[i];
#endif
    }

    __syncthreads();                      // every wave is done with the LDS weights / staging of this item

    // ---- dW / bias-gradient sums of the WR row slots: ordered tree through LDS
    if (WR > 1) {
        float* red = lds;
#pragma unroll
        for (int step = 1; step < WR; step *= 2) {
            const int slot = r / (2 * step);
            float* rs = red + (long)slot * NRED * 64 + lane;
            if (r % (2 * step) == step) {
                int n = 0;
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                        for (int e = 0; e < 16; ++e) rs[(n++) * 64] = dW[h][ib][e];
#pragma unroll
                for (int h = 0; h < NH; ++h) rs[(n++) * 64] = bsum[h];
                rs[(n++) * 64] = thsum;
            }
            __syncthreads();
            if (r % (2 * step) == 0 && r + step < WR) {
                int n = 0;
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                        for (int e = 0; e < 16; ++e) dW[h][ib][e] += rs[(n++) * 64];
#pragma unroll
                for (int h = 0; h < NH; ++h) bsum[h] += rs[(n++) * 64];
                thsum += rs[(n++) * 64];
            }
            __syncthreads();
        }
    }
    if (r == 0 && tile_ok) {
        const bool direct = p.S == 1;                // one batch split: no partial buffer, no reduce launch
        float* out = direct ? p.gW : p.ws_dw + (long)s * p.dw_stride;
        const long ldo = direct ? p.ldg : p.ldws;
        const bool cw = gene < p.plane;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = ib * 32 + rowmap(e, hi);
                    if (cw && i < p.hL) out[(long)i * ldo + (long)h * p.plane + gene] = dW[h][ib][e];
                }
            const float bv = bsum[h] + __shfl_xor(bsum[h], 32, 64);
            if (cw && hi == 0) out[(long)p.hL * ldo + (long)h * p.plane + gene] = bv;
        }
        if (CONST_DISP) {
            const float tv = thsum + __shfl_xor(thsum, 32, 64);
            if (direct) {                            // ConstantDispersionLayer chain (dca/layers.py:17-21)
                if (gvalid && hi == 0) {
                    const float e = expf(p.theta_w[gene]);
                    p.g_theta[gene] = (e >= 1e-3f && e <= 1e4f) ? tv * e : 0.f;
                }
            } else if (cw && hi == 0) {
                out[(long)(p.hL + 1) * ldo + gene] = tv;
            }
        }
    }
    __syncthreads();                      // the reduce scratch (and the weight image) are free for the next item
    first_item = false;
    }   // work items

    // ---- loss: wave -> workgroup -> one partial per workgroup
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dacc += __shfl_down(dacc, off, 64);
    if (lane == 0) lred[wave] = dacc;
    __syncthreads();
    if (tid == 0) {
        double v = 0.0;
        for (int w = 0; w < WR; ++w) v += lred[w];
        p.partials[blockIdx.x] = v;
    }
#ifdef DCA_HEADS_TIMING
    if (p.timing && lane == 0) {
        long long* tp = p.timing + ((long)blockIdx.x * WR + wave) * 10;
        tp[8] = t_loop0 - t_entry;
        tp[9] = (long long)__builtin_readcyclecounter() - t_loop1;
    }
#endif
}


// gW[i, col] = sum_s ws[s][i][col], i = 0..hL (row hL = bias gradient), then the
// ConstantDispersionLayer chain (dca/layers.py:17-21) on the per-gene theta sums.
struct ReduceDwArgs {
    const float* ws; int S; long stride; int hL; long ldws, ncols; float* gW; long ldg;
    const float* theta_w; float* g_theta; int G;
};

// bid / nblk: this workgroup's index among the nblk that share the reduction (the stand-alone kernel: blockIdx / gridDim;
// the combined launch below: the workgroups behind those of the dH reduction)
__device__ __forceinline__ void reduce_dw_body(const ReduceDwArgs& q, int bid, int nblk) {
    const float* ws = q.ws; const int S = q.S; const long stride = q.stride; const int hL = q.hL;
    const long ldws = q.ldws, ncols = q.ncols; float* gW = q.gW; const long ldg = q.ldg;
    const float* theta_w = q.theta_w; float* g_theta = q.g_theta; const int G = q.G;
    if ((ncols & 3) == 0 && (ldws & 3) == 0 && (ldg & 3) == 0 && (stride & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(ws) | reinterpret_cast<uintptr_t>(gW)) & 15) == 0) {
        // 16 bytes per lane (the plane width is a multiple of 4); partial s is added in order s = 0, 1, ..
        const long nq = ncols >> 2, totalq = (long)(hL + 1) * nq;
        for (long idx = (long)bid * 256 + threadIdx.x; idx < totalq; idx += (long)nblk * 256) {
            const long i = idx / nq, c = (idx - i * nq) << 2;
            float4 v = *reinterpret_cast<const float4*>(ws + i * ldws + c);
            for (int s = 1; s < S; ++s) {
                const float4 x = *reinterpret_cast<const float4*>(ws + (long)s * stride + i * ldws + c);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
            }
            *reinterpret_cast<float4*>(gW + i * ldg + c) = v;
        }
    } else {
        const long total = (long)(hL + 1) * ncols;
        for (long idx = (long)bid * 256 + threadIdx.x; idx < total; idx += (long)nblk * 256) {
            const long i = idx / ncols, c = idx - i * ncols;
            float v = 0.f;
            for (int s = 0; s < S; ++s) v += ws[(long)s * stride + i * ldws + c];
            gW[i * ldg + c] = v;
        }
    }
    if (g_theta) {
        for (long c = (long)bid * 256 + threadIdx.x; c < G; c += (long)nblk * 256) {
            float v = 0.f;
            for (int s = 0; s < S; ++s) v += ws[(long)s * stride + (long)(hL + 1) * ldws + c];
            const float e = expf(theta_w[c]);
            g_theta[c] = (e >= 1e-3f && e <= 1e4f) ? v * e : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void heads_reduce_dw_kernel(ReduceDwArgs q) { reduce_dw_body(q, blockIdx.x, gridDim.x); }

// dH[row, i] = sum over gene tiles of ws[row tile][gt][row % 32][i]; GL threads split the gene tiles of
// one output quad, combined in fixed order through LDS.
struct ReduceDhArgs {
    const float* ws; int ntg, B, Bpad, KT, hL; float* dH; long lddh;
    const double* loss_partials; int n_partials; double loss_scale; float* loss_out;
};

template <int GL>
__device__ __forceinline__ void reduce_dh_body(const ReduceDhArgs& q, int bid) {
    const float* ws = q.ws; const int ntg = q.ntg, B = q.B, KT = q.KT, hL = q.hL; float* dH = q.dH; const long lddh = q.lddh;
    const double* loss_partials = q.loss_partials; const int n_partials = q.n_partials; const double loss_scale = q.loss_scale;
    float* loss_out = q.loss_out;
    constexpr int OUT = 256 / GL;
    __shared__ float4 red[256];
    const int o = threadIdx.x % OUT, gl = threadIdx.x / OUT;
    const int q4 = KT / 4;
    const long nq = (long)B * q4;
    const long quad = (long)bid * OUT + o;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (quad < nq) {
        const long tq = (long)kTR * q4;                      // quads of one (row tile, gene tile) partial
        const long t = quad / tq, within = quad - t * tq;
        const float4* src = reinterpret_cast<const float4*>(ws) + t * ntg * tq + within;
        // batches of 8 independent loads (one memory round trip per batch), added in tile order
        for (int gt0 = gl; gt0 < ntg; gt0 += 8 * GL) {
            float4 x[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int gt = gt0 + u * GL;
                x[u] = src[(long)(gt < ntg ? gt : gl) * tq];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (gt0 + u * GL < ntg) { v.x += x[u].x; v.y += x[u].y; v.z += x[u].z; v.w += x[u].w; }
        }
    }
    red[threadIdx.x] = v;
    __syncthreads();
    if (gl == 0 && quad < nq) {
#pragma unroll
        for (int k = 1; k < GL; ++k) {
            const float4 x = red[k * OUT + o];
            v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
        }
        const long row = quad / q4;
        const int i = (int)(quad - row * q4) * 4;
        float* d = dH + row * lddh + i;
        if (i + 0 < hL) d[0] = v.x;
        if (i + 1 < hL) d[1] = v.y;
        if (i + 2 < hL) d[2] = v.z;
        if (i + 3 < hL) d[3] = v.w;
    }
    // optionally the work of dcahip_loss_finalize on this launch (block 0): batch loss = scale * sum of the workgroup
    // partials, nan -> inf (dca/loss.py:146-148)
    if (loss_out && bid == 0) {
        __shared__ double lsum[256];
        double a = 0.0;
        for (int i = threadIdx.x; i < n_partials; i += 256) a += loss_partials[i];
        lsum[threadIdx.x] = a;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) lsum[threadIdx.x] += lsum[threadIdx.x + st];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            float lf = (float)(lsum[0] * loss_scale);
            if (isnan(lf)) lf = INFINITY;
            *loss_out = lf;
        }
    }
}

template <int GL>
__global__ __launch_bounds__(256) void heads_reduce_dh_kernel(ReduceDhArgs q) { reduce_dh_body<GL>(q, blockIdx.x); }

// Both reductions of a launch with several batch splits in ONE kernel: the first n_dh workgroups reduce the input-gradient
// partials, the others the weight-gradient partials -- they run side by side instead of one after the other.
template <int GL>
__global__ __launch_bounds__(256) void heads_reduce_both_kernel(ReduceDhArgs qh, ReduceDwArgs qw, int n_dh) {
    if ((int)blockIdx.x < n_dh) reduce_dh_body<GL>(qh, blockIdx.x);
    else reduce_dw_body(qw, blockIdx.x - n_dh, gridDim.x - n_dh);
}

// The pipelined one-wave-per-SIMD kernel (heads_p4.inc) is an EXPERIMENT build (-DDCA_EXP_HEADS_P4; the library then carries
// dcahip_heads_set_p4_min_tiles, the row-tile count from which that kernel takes a launch): measured on the MI355X it loses
// to the 8-wave kernel at every batch size (C3, 4 096 rows: 1.23 vs 0.81 ms; profiles/r05c_heads_p4_ab.txt, per-phase cycles
// in profiles/r05c_heads_p4_timing.txt, why in DESIGN.md 4.1).  The product library has neither the kernel nor the switch.
#ifdef DCA_EXP_HEADS_P4
#include "heads_p4.inc"
int g_p4_min_nt = 1 << 30;
#else
constexpr int g_p4_min_nt = 1 << 30;
constexpr int kWR4 = 4;
#endif

struct HeadsPlan {
    bool small;                          // one row tile: the four-wave kernel, one workgroup per gene tile
    bool p4;                             // the pipelined four-wave kernel (one wave per SIMD)
    int HLB, WR, S, NT, ntg, ngb, grid;
    int nitems, npart;                   // split-bf16 path: work items (S x gene tiles), dH partials per row tile
    long ldws, dw_stride, dw_bytes, dh_bytes, hs_bytes;
};

// resident workgroups of the split-bf16 kernel: its LDS (up to 149 KB with 8 waves, 51 KB with one) admits one
// 8-wave or three single-wave workgroups per CU
inline int x3_resident(int WR) { return WR == 1 ? 3 * kCUs : kCUs; }

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// a pure function of the shape: row slots per workgroup, batch splits, workspace layout
bool make_heads_plan(int B, int hL, int G, long plane, int flags, HeadsPlan* out, int p4_min_nt = -1) {
    if (p4_min_nt < 0) p4_min_nt = g_p4_min_nt;
    if (flags & (DCAHIP_NLL_POISSON | DCAHIP_NLL_MSE)) return false;     // NB / ZINB family only
    if (B <= 0 || G <= 0 || hL <= 0 || hL > 64 || plane < G || (plane & 3) || plane > ((G + 31) & ~31)) return false;
    if (B > (1 << 22)) return false;                 // H is addressed through a 32-bit buffer resource (B x 64 floats)
    const bool has_pi = flags & DCAHIP_NLL_HAS_PI, cdisp = flags & DCAHIP_NLL_CONST_DISP;
    const int NH = 1 + (cdisp ? 0 : 1) + (has_pi ? 1 : 0);
    const int wg_tiles = 1;                          // gene tiles per workgroup
    HeadsPlan p;
    p.HLB = 2;
    p.NT = (B + kTR - 1) / kTR;
    p.ntg = (G + kTG - 1) / kTG;
    p.ngb = (p.ntg + wg_tiles - 1) / wg_tiles;
    if (p.ngb > kMaxGrid) return false;
    // the 8-wave persistent kernel from 5 row tiles on (waves beyond the batch idle): measured against the four-wave kernel
    // at G = 20 000: B = 128 0.088 / 0.086 ms (kept on the four-wave kernel), 160: 0.100 / 0.104, 192: 0.103 / 0.121,
    // 224: 0.107 / 0.142 (profiles/r02z_heads_kernel_switch.txt)
    p.WR = p.NT >= kWr8MinNT ? kWR2 : 1;
    p.p4 = has_pi && p.NT >= kWr8MinNT && p.NT >= p4_min_nt;
    if (p.p4) p.WR = kWR4;
    const int smax = (p.NT + p.WR - 1) / p.WR;
    double best = 1e300;
    p.S = 1;
    // per work item: the weight prologue and the reduce (0.75 tile times); the pipelined kernel also fills and drains its
    // three-stage pipeline (two more iterations of about a third of the work each)
    const double item_cost = p.p4 ? 1.75 : 0.75;
    for (int S = 1; S <= smax && (long)S * p.ngb <= kMaxGrid; ++S) {
        const long items = (long)S * p.ngb;
        const long rounds = (items + kCUs - 1) / kCUs;
        const int tiles = (p.NT + S * p.WR - 1) / (S * p.WR);
        const double cost = (double)rounds * (tiles + item_cost);
        if (cost < best - 1e-9) { best = cost; p.S = S; }
    }
    p.nitems = p.S * p.ngb;
    p.grid = p.nitems;
    p.npart = p.ntg;
    p.small = p.WR == 1 && (long)p.ntg * p.NT <= kMaxSmallGrid;
    if (p.small) {                                   // one workgroup per (gene tile, row tile): S = NT weight-gradient partials
        p.S = p.NT;
        p.nitems = p.NT * p.ntg;
        p.grid = p.nitems;
        p.npart = p.ntg;
    } else {                                         // persistent: as many workgroups as are resident, a multiple of S
        const int res = x3_resident(p.WR) / p.S * p.S;
        if (p.grid > res) p.grid = res;
        p.npart = p.grid / p.S;
    }
    p.ldws = (long)NH * plane;
    p.dw_stride = (long)(hL + 2) * p.ldws;
    p.dw_bytes = (long)p.S * p.dw_stride * (long)sizeof(float);
    p.dh_bytes = (long)p.npart * p.NT * kTR * (p.HLB * 32) * (long)sizeof(float);
    p.hs_bytes = 2L * p.NT * kHTile * 2;             // split decoder output, both layouts
    *out = p;
    return true;
}

long long* g_timing = nullptr;

// =====================================================================================================
// K-HEADS for batches below 256 rows (B <= 32, ONE row tile, is the reference's default batch size, dca/api.py:33):
// one (gene tile, row tile) pair per workgroup, FOUR waves working on it together.  With a lone wave per tile (the persistent kernel above at NT = 1) the
// 625 tiles of G = 20 000 leave three quarters of the SIMDs idle and every phase is one wave's dependent chain:
// measured 41 us, of which 10 launch + weight prologue, 13 the likelihood pass, 18 the three products and their stores
// (tools/_dbg experiment builds).  Here
//   F   : wave h < NH computes head h (24 MFMA),
//   Z   : wave w takes the row-slot group w (4 of the 16 slots of each lane half) -- dense pass + its own non-zero queue,
//   dW  : wave h < NH accumulates and stores head h (24 MFMA), while
//   dH  : wave 3 multiplies all heads (72 MFMA) and stores the tile's partial sum,
// all through one weight image and one staging tile in LDS; operands, arithmetic and the order of every sum are
// those of the persistent kernel (same products, same six-term order; the loss partial of the tile is the sum of its
// four waves' in wave order).
// =====================================================================================================
template <bool HAS_PI, bool CONST_DISP, bool YC>
__global__ __launch_bounds__(256) void heads_fused_small_kernel(HeadsArgs2 p) {
    using YV = std::conditional_t<YC, unsigned, float>;
    constexpr int NH = 1 + (CONST_DISP ? 0 : 1) + (HAS_PI ? 1 : 0);
    constexpr int PI_H = NH - 1;
    constexpr int KT = 64;
    constexpr int ST_PLANE = kTG * kLdS;
    constexpr int NP = NH + (CONST_DISP ? 1 : 0);
    constexpr int TH_P = NH;
    constexpr int ST_TILE = NP * ST_PLANE;
    constexpr int W_PIECE = 64 * 64;
    constexpr int W_FLOATS = NH * 3 * W_PIECE / 4;
    constexpr int BIAS_FLOATS = CONST_DISP ? (NH + 1) * 32 : 0;       // conditional dispersion: biases straight from memory
    constexpr int NW = 4;
    static_assert(NW * kZU == 16, "one Z group per wave");
    // a wave queues at most its 4 x 64 elements.  With 256 entries per queue the workgroup's LDS stays below a third
    // of the CU's 160 KB: three workgroups per CU = 768 resident, the 625 gene tiles of G = 20 000 in ONE round
    // (at two per CU the last 113 tiles ran after the first 512: 30 us instead of 17)
    constexpr int QCAP = kZU * 64;
    constexpr int LDS_FLOATS = W_FLOATS + ST_TILE + NW * QCAP + BIAS_FLOATS;
    static_assert(LDS_FLOATS * 4 + NW * 8 <= 42 * 1280, "three workgroups per CU (LDS is allocated in 1280-byte granules)");
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ double lred[NW];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const long long cur = p.cursor ? *p.cursor : 0;
    // workgroup = (gene tile gb, row tile t); the row tiles of a gene tile are neighbours (they share its weights in L2)
    const int t = p.NT > 1 ? (int)(blockIdx.x % p.NT) : 0;
    const int gb = p.NT > 1 ? (int)(blockIdx.x / p.NT) : (int)blockIdx.x;
    const int row0 = t * kTR;
    const int gt = p.tile_order ? p.tile_order[gb] : gb;
    const int g0 = gt * kTG;
    float* const dh_out = p.ws_dh + ((long)t * p.npart + gb) * (kTR * KT);
    if (g0 >= p.G) {                         // padding entry of the tile order: an all-zero partial, no loss
        for (int i = tid; i < kTR * KT; i += 256) dh_out[i] = 0.f;
        if (tid == 0) p.partials[blockIdx.x] = 0.0;
        return;
    }
    const int gene = g0 + l31;
    const bool gvalid = gene < p.G;

    unsigned char* const Wimg = reinterpret_cast<unsigned char*>(lds);
    float* const St = lds + W_FLOATS;
    unsigned* const Q = reinterpret_cast<unsigned*>(lds + W_FLOATS + ST_TILE) + wave * QCAP;
    float* const Bs = lds + W_FLOATS + ST_TILE + NW * QCAP;

    // ---- requests that do not depend on the weights, in flight during the weight prologue: storage rows, size
    // factors, the counts of this wave's Z group, the decoder rows (both operand orientations) of the product waves
    const int rl = row0 + l31 < p.B ? row0 + l31 : p.B - 1;
    const int srow_l = p.perm ? p.perm[cur + rl] : (int)(cur + rl);
    const float sf_l = p.sf[srow_l];
    const float* const ycol = p.y + (gvalid ? gene : p.G - 1);
    const unsigned char* const ycolc = p.yc + (gvalid ? gene : p.G - 1);
    YV yv[kZU];
#pragma unroll
    for (int j = 0; j < kZU; ++j) {
        const int sr = __shfl(srow_l, rowmap(wave * kZU + j, hi), 64);
        if constexpr (YC) yv[j] = (unsigned)ycolc[(unsigned long long)(unsigned)sr * (unsigned)p.ldc];
        else yv[j] = ycol[(unsigned long long)(unsigned)sr * (unsigned)p.ldy];
    }
    float hx[4][8], htx[2][2][8];
    float bias_h = 0.f;
    if (wave < NH) {
        bias_h = p.bh[(long)wave * p.plane + (gvalid ? gene : p.G - 1)];
        const float* hp = p.H + (long)rl * p.ldh;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {            // row l31, hidden units 32 hi + 8 ks ..; unconditional (clamped) loads
                const int kk = 32 * hi + 8 * ks + j;
                hx[ks][j] = hp[kk < p.hL ? kk : p.hL - 1];
            }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int j = 0; j < 8; ++j) {        // hidden unit l31 + 32 ib, rows in the order of the MFMA row map
                    const int row = row0 + rowmap(8 * ks + j, hi), i = l31 + 32 * ib;
                    htx[ks][ib][j] = p.H[(long)(row < p.B ? row : p.B - 1) * p.ldh + (i < p.hL ? i : p.hL - 1)];
                }
    }

    // ---- head weights of the gene tile -> bf16 pieces in LDS (image and rotation of the persistent kernel)
    {
        constexpr int UNR = 2 * NH;                  // NH * 64 * 8 sixteen-byte items over 256 threads: one batch of loads
        float4 v[UNR];
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = tid + u * 256;
            const int c4 = idx & 7, kq = (idx >> 3) & 63, h = idx >> 9;
            const int kc = kq < p.hL ? kq : p.hL - 1;
            long gcol = g0 + c4 * 4;
            if (gcol > p.plane - 4) gcol = p.plane - 4;
            v[u] = *reinterpret_cast<const float4*>(p.Wh + (long)kc * p.ldw + (long)h * p.plane + gcol);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = tid + u * 256;
            const int c4 = idx & 7, kq = (idx >> 3) & 63, h = idx >> 9;
            const int gcol = g0 + c4 * 4;
            const bool kv = kq < p.hL && gcol <= p.plane - 4;
            float4 w = v[u];
            if (!kv || gcol + 0 >= p.G) w.x = 0.f;
            if (!kv || gcol + 1 >= p.G) w.y = 0.f;
            if (!kv || gcol + 2 >= p.G) w.z = 0.f;
            if (!kv || gcol + 3 >= p.G) w.w = 0.f;
            unsigned a0, a1, a2, b0, b1, b2;
            split_pair(w.x, w.y, a0, a1, a2);
            split_pair(w.z, w.w, b0, b1, b2);
            const int off = kq * 64 + ((((c4 >> 1) + (kq >> 2)) & 3) << 4) + ((c4 & 1) << 3);
            *reinterpret_cast<u32x2*>(Wimg + (h * 3 + 0) * W_PIECE + off) = u32x2{a0, b0};
            *reinterpret_cast<u32x2*>(Wimg + (h * 3 + 1) * W_PIECE + off) = u32x2{a1, b1};
            *reinterpret_cast<u32x2*>(Wimg + (h * 3 + 2) * W_PIECE + off) = u32x2{a2, b2};
        }
        if (CONST_DISP && tid < 32) Bs[NH * 32 + tid] = (g0 + tid < p.G) ? p.theta_w[g0 + tid] : 0.f;
    }
    __syncthreads();

    // LDS addresses of the weight image (see the persistent kernel)
    const int t16 = lane & 15, c8 = 4 * ((lane >> 4) & 1) + (t16 & 3);
    int wtr[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
        wtr[rr] = hi * 2048 + (t16 >> 2) * 64 + ((((c8 >> 1) + rr) & 3) << 4) + ((c8 & 1) << 3);
    int wdr[2];
#pragma unroll
    for (int gs = 0; gs < 2; ++gs) wdr[gs] = l31 * 64 + (((2 * gs + hi + (l31 >> 2)) & 3) << 4);
    auto w_tr = [&](int h, int q, int ks) {           // B operand of F: gene l31, k = 32 hi + 8 ks .. + 7
        const unsigned char* b0 = Wimg + (h * 3 + q) * W_PIECE + ks * 512;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(b0 + wtr[(2 * ks) & 3]));
        const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4*)(b0 + 256 + wtr[(2 * ks + 1) & 3]));
        u32x4 o;
        const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi4);
        o[0] = a[0]; o[1] = a[1]; o[2] = b[0]; o[3] = b[1];
        return o;
    };
    auto w_dr = [&](int h, int q, int jb, int gs) {   // B operand of dH: hidden unit l31 + 32 jb, genes 16 gs + 8 hi ..
        return *reinterpret_cast<const u32x4*>(Wimg + (h * 3 + q) * W_PIECE + jb * 2048 + wdr[gs]);
    };

    // ---- F: wave h computes the pre-activations of head h and stages them [gene][row]
    if (wave < NH) {
        const int h = wave;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (row0 + l31 < p.B && 32 * hi + 8 * ks + j < p.hL) ? hx[ks][j] : 0.f;
            u32x4 af[3];
            split8(x, af);
            u32x4 bf[3] = {w_tr(h, 0, ks), w_tr(h, 1, ks), w_tr(h, 2, ks)};
            MFMA_X3(af, bf, acc)
        }
        const float bias = gvalid ? bias_h : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) St[h * ST_PLANE + l31 * kLdS + rowmap(e, hi)] = acc[e] + bias;
    }
    __syncthreads();

    // ---- Z: element-wise likelihood and gradient of this wave's four row slots per lane half
    double dacc = 0.0;
    {
        const float thw = CONST_DISP ? Bs[NH * 32 + l31] : 0.f;
        float lacc = 0.f;
        int qn = 0;
        float i_am[kZU], i_ad[kZU], i_ap[kZU];
#pragma unroll
        for (int j = 0; j < kZU; ++j) {
            const int idx = l31 * kLdS + rowmap(wave * kZU + j, hi);
            i_am[j] = St[idx];
            i_ad[j] = CONST_DISP ? thw : St[ST_PLANE + idx];
            i_ap[j] = HAS_PI ? St[PI_H * ST_PLANE + idx] : 0.f;
        }
        float o_m[kZU], o_d[kZU], o_p[kZU];
        bool o_nz[kZU];
#pragma unroll
        for (int j = 0; j < kZU; ++j) {
            const int row = rowmap(wave * kZU + j, hi);
            const bool valid = (row0 + row < p.B) && gvalid;
            const YV yj = yv[j];
            bool nz;
            if constexpr (YC) nz = valid && yj != 0u;
            else nz = valid && (HAS_PI ? !(yj < kZeroThresh) : (yj != 0.f));
            const float sc = valid ? p.inv_n : 0.f;
            if (HAS_PI) {
                float gmv, gdv, gpv;
                const float nll = zinb_zero_elem<CONST_DISP>(i_am[j], i_ad[j], i_ap[j], __shfl(sf_l, row, 64), p.ridge, gmv, gdv, gpv);
                lacc += (valid && !nz) ? nll : 0.f;
                o_m[j] = gmv * sc; o_d[j] = gdv * sc; o_p[j] = gpv * sc;
            } else {
                float gmv, gdv;
                const float nll = nb_zero_elem<CONST_DISP>(i_am[j], i_ad[j], __shfl(sf_l, row, 64), gmv, gdv);
                lacc += (valid && !nz) ? nll : 0.f;
                o_m[j] = gmv * sc; o_d[j] = gdv * sc; o_p[j] = 0.f;
            }
            o_nz[j] = nz;
        }
#pragma unroll
        for (int j = 0; j < kZU; ++j) {
            const int idx = l31 * kLdS + rowmap(wave * kZU + j, hi);
            const bool nz = o_nz[j];
            const unsigned long long m = __ballot(nz);
            const int slot = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (nz) {
                const YV yj = yv[j];
                unsigned y16;
                if constexpr (YC) y16 = yj == 255u ? 0xFFFFu : yj;
                else y16 = (yj < 65535.f && yj == floorf(yj)) ? (unsigned)yj : 0xFFFFu;
                Q[slot] = (unsigned)idx | (y16 << 16);
            } else {
                St[idx] = o_m[j];
                if (CONST_DISP) St[TH_P * ST_PLANE + idx] = o_d[j]; else St[ST_PLANE + idx] = o_d[j];
                if (HAS_PI) St[PI_H * ST_PLANE + idx] = o_p[j];
            }
            qn += __popcll(m);
        }
        // compacted non-zero entries, 64 per pass
        while (qn > 0) {
            const int c = qn < 64 ? qn : 64;
            wave_sync();
            const int q0 = qn - c;
            const bool act = lane < c;
            const unsigned e = Q[q0 + (act ? lane : 0)];
            const int idx = e & 2047;
            const int gq = (idx * 1986) >> 16;          // idx / 33 for idx < 1056
            const int row = idx - gq * kLdS;
            const float sfr = __shfl(sf_l, row, 64);
            const int sr = __shfl(srow_l, row, 64);
            const float am = St[idx];
            const float ad = CONST_DISP ? Bs[NH * 32 + gq] : St[ST_PLANE + idx];
            const float ap = HAS_PI ? St[PI_H * ST_PLANE + idx] : 0.f;
            float yq = (float)(e >> 16);
            if (__ballot(act && (e >> 16) == 0xFFFFu)) {     // (see heads_fused_x3_kernel)
                if (act && (e >> 16) == 0xFFFFu) yq = YC ? escaped_count(p, sr, g0 + gq) : p.y[(long)sr * p.ldy + g0 + gq];
                __builtin_amdgcn_s_waitcnt(0x0F70);
            }
            float o1, o2, o3 = 0.f, nll;
            if (HAS_PI) {
                nll = zinb_nz_elem<CONST_DISP>(am, ad, ap, sfr, yq, p.ridge, o1, o2, o3);
            } else {
                float dmu = 0.f, dth = 0.f, dpi = 0.f;
                const Heads hd = head_acts<HAS_PI, CONST_DISP>(am, ad, ap, sfr);
                nll = nll_elem<HAS_PI, true, true>(hd, yq, p.ridge, dmu, dth, dpi);
                o1 = dmu * hd.gm; o2 = dth * hd.gd;
            }
            lacc += act ? nll : 0.f;
            if (act) {
                St[idx] = o1 * p.inv_n;
                const float od = o2 * p.inv_n;
                if (CONST_DISP) St[TH_P * ST_PLANE + idx] = od; else St[ST_PLANE + idx] = od;
                if (HAS_PI) St[PI_H * ST_PLANE + idx] = o3 * p.inv_n;
            }
            qn -= c;
        }
        dacc = (double)lacc;
    }
    __syncthreads();

    if (wave == NW - 1) {
        // ---- dH[row, i] = sum_genes D[row, gene] W[i, gene], all heads: the tile's partial sum
        f32x16 dHa[2];
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int e = 0; e < 16; ++e) dHa[jb][e] = 0.f;
#pragma unroll
        for (int h = 0; h < NH; ++h)
#pragma unroll
            for (int gs = 0; gs < 2; ++gs) {
                float dv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) dv[j] = St[h * ST_PLANE + (16 * gs + 8 * hi + j) * kLdS + l31];
                u32x4 af[3];
                split8(dv, af);
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    u32x4 bf[3] = {w_dr(h, 0, jb, gs), w_dr(h, 1, jb, gs), w_dr(h, 2, jb, gs)};
                    MFMA_BWD(af, bf, dHa[jb])
                }
            }
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = dHa[jb][e];
                dh_out[rowmap(e, hi) * KT + jb * 32 + l31] = v;
            }
        if (CONST_DISP) {                            // ConstantDispersionLayer chain (dca/layers.py:17-21)
            float thsum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) thsum += St[TH_P * ST_PLANE + l31 * kLdS + rowmap(e, hi)];
            const float tv = thsum + __shfl_xor(thsum, 32, 64);
            if (p.NT == 1) {
                if (gvalid && hi == 0) {
                    const float ex = expf(p.theta_w[gene]);
                    p.g_theta[gene] = (ex >= 1e-3f && ex <= 1e4f) ? tv * ex : 0.f;
                }
            } else if (gene < p.plane && hi == 0) {      // several row tiles: the raw sum, chained by the dW reduce
                p.ws_dw[(long)t * p.dw_stride + (long)(p.hL + 1) * p.ldws + gene] = tv;
            }
        }
    } else if (wave < NH) {
        // ---- dW[i, gene] = sum_rows H[row, i] D[row, gene] of head h, straight to the gradient buffer (one batch split)
        const int h = wave;
        f32x16 dW[2];
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int e = 0; e < 16; ++e) dW[ib][e] = 0.f;
        float bsum = 0.f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float dv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dv[j] = St[h * ST_PLANE + l31 * kLdS + rowmap(8 * ks + j, hi)];
                bsum += dv[j];
            }
            u32x4 bf[3];
            split8(dv, bf);
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    x[j] = (row0 + rowmap(8 * ks + j, hi) < p.B && l31 + 32 * ib < p.hL) ? htx[ks][ib][j] : 0.f;
                u32x4 af[3];
                split8(x, af);
                MFMA_BWD(af, bf, dW[ib])
            }
        }
        const bool cw = gene < p.plane;
        // one row tile: straight to the gradient buffer; several: row tile t's partial, summed by the dW reduce launch
        float* const out = p.NT == 1 ? p.gW : p.ws_dw + (long)t * p.dw_stride;
        const long ldo = p.NT == 1 ? p.ldg : p.ldws;
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = ib * 32 + rowmap(e, hi);
                const float v = dW[ib][e];
                if (cw && i < p.hL) out[(long)i * ldo + (long)h * p.plane + gene] = v;
            }
        const float bv = bsum + __shfl_xor(bsum, 32, 64);
        if (cw && hi == 0) out[(long)p.hL * ldo + (long)h * p.plane + gene] = bv;
    }

    // ---- loss: wave -> workgroup (wave order) -> one partial per tile
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dacc += __shfl_down(dacc, off, 64);
    if (lane == 0) lred[wave] = dacc;
    __syncthreads();
    if (tid == 0) p.partials[blockIdx.x] = ((lred[0] + lred[1]) + lred[2]) + lred[3];
}

// C [32, 32] = A [32, K] B [K, 32] with the operand split and the six bf16 products of K-HEADS, one wave
// (dcahip_x3_product_32x32: the accuracy contract of the matrix products, tested against fp64)
__global__ __launch_bounds__(64) void x3_product_kernel(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        float av[8], bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            av[j] = A[(long)l31 * K + k0 + 8 * hi + j];
            bv[j] = B[(long)(k0 + 8 * hi + j) * 32 + l31];
        }
        u32x4 af[3], bf[3];
        split8(av, af);
        split8(bv, bf);
        MFMA_X3(af, bf, acc)
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) C[rowmap(e, hi) * 32 + l31] = acc[e];
}

template <bool P, bool C, bool YC>
void launch_fused_x3(const HeadsPlan& pl, const HeadsArgs2& a, hipStream_t s) {
#ifdef DCA_EXP_HEADS_P4
    if constexpr (P) {
        if (pl.p4) { hipLaunchKernelGGL((heads_fused_p4_kernel<C, YC>), dim3(pl.grid), dim3(64 * kWR4), 0, s, a); return; }
    }
#endif
    if (pl.small) hipLaunchKernelGGL((heads_fused_small_kernel<P, C, YC>), dim3(pl.grid), dim3(256), 0, s, a);
    else if (pl.WR == kWR2) hipLaunchKernelGGL((heads_fused_x3_kernel<P, C, kWR2, YC>), dim3(pl.grid), dim3(64 * kWR2), 0, s, a);
    else hipLaunchKernelGGL((heads_fused_x3_kernel<P, C, 1, YC>), dim3(pl.grid), dim3(64), 0, s, a);
}

}  // namespace

// sufficient for every batch of at most B rows: the maximum over the plans of all of them (a smaller batch may
// split more and keep more partials); a plan depends on B only through its row-tile count
extern "C" long dcahip_heads_fused_workspace_bytes(int B, int hL, int G, long plane, int flags) {
    HeadsPlan p;
    if (!make_heads_plan(B, hL, G, plane, flags, &p)) return 0;
    long need = 0;
    for (int nt = 1; nt <= p.NT; ++nt) {
        const int b = nt * kTR < B ? nt * kTR : B;
#ifdef DCA_EXP_HEADS_P4
        for (int p4_min : {1, 1 << 30}) {           // whichever kernel the threshold (dcahip_heads_set_p4_min_tiles) picks later
#else
        for (int p4_min : {1 << 30}) {
#endif
            HeadsPlan q;
            if (!make_heads_plan(b, hL, G, plane, flags, &q, p4_min)) continue;
            const long n = q.dw_bytes + q.dh_bytes + q.hs_bytes;
            if (n > need) need = n;
        }
    }
    return need;
}

#ifdef DCA_EXP_HEADS_P4
extern "C" int dcahip_heads_set_p4_min_tiles(int nt) {
    const int old = g_p4_min_nt;
    if (nt > 0) g_p4_min_nt = nt;
    return old;
}
#endif

extern "C" int dcahip_x3_product_32x32(const float* A, const float* B, float* C, int K, void* stream) {
    if (!A || !B || !C || K <= 0 || (K & 15)) return DCAHIP_EINVAL;
    hipLaunchKernelGGL(x3_product_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), A, B, C, K);
    return (int)hipGetLastError();
}

#ifdef DCA_HEADS_TIMING
extern "C" void dcahip_heads_set_timing(long long* buf) { g_timing = buf; }
#endif

// (rounded up to the two tiles per workgroup of the fp32-MFMA variant; the product path reads the first ceil(G / 32))
extern "C" int dcahip_heads_tile_order_len(int G) { return G > 0 ? (((G + kTG - 1) / kTG + kWG - 1) / kWG) * kWG : 0; }

extern "C" int dcahip_heads_fused_ordered(const float* H, long ldh, const float* Wh, long ldw,
                                          const float* bh, long plane, const float* theta_w,
                                          const float* y, long ldy, const float* sf, const int* perm,
                                          const long long* cursor, int B, int hL, int G, float ridge,
                                          float inv_n, int flags, float* gW, long ldg, float* g_theta,
                                          float* dH, long lddh, double* loss_partials, int* n_partials_out,
                                          void* workspace, long workspace_bytes, const int* tile_order,
                                          void* stream);

extern "C" int dcahip_heads_fused_loss(const float* H, long ldh, const float* Wh, long ldw,
                                       const float* bh, long plane, const float* theta_w,
                                       const float* y, long ldy, const float* sf, const int* perm,
                                       const long long* cursor, int B, int hL, int G, float ridge,
                                       float inv_n, int flags, float* gW, long ldg, float* g_theta,
                                       float* dH, long lddh, double* loss_partials, int* n_partials_out,
                                       void* workspace, long workspace_bytes, const int* tile_order,
                                       float* loss_out, void* stream);

extern "C" int dcahip_heads_fused_compact(const float* H, long ldh, const float* Wh, long ldw,
                                          const float* bh, long plane, const float* theta_w,
                                          const float* y, long ldy,
                                          const unsigned char* yc, long ldc, const int* ovf_ptr, const int* ovf_col,
                                          const float* ovf_val, const float* sf, const int* perm,
                                          const long long* cursor, int B, int hL, int G, float ridge,
                                          float inv_n, int flags, float* gW, long ldg, float* g_theta,
                                          float* dH, long lddh, double* loss_partials, int* n_partials_out,
                                          void* workspace, long workspace_bytes, const int* tile_order,
                                          float* loss_out, void* stream);

extern "C" int dcahip_heads_fused(const float* H, long ldh, const float* Wh, long ldw,
                                  const float* bh, long plane, const float* theta_w,
                                  const float* y, long ldy, const float* sf, const int* perm,
                                  const long long* cursor, int B, int hL, int G, float ridge,
                                  float inv_n, int flags, float* gW, long ldg, float* g_theta,
                                  float* dH, long lddh, double* loss_partials, int* n_partials_out,
                                  void* workspace, long workspace_bytes, void* stream) {
    return dcahip_heads_fused_ordered(H, ldh, Wh, ldw, bh, plane, theta_w, y, ldy, sf, perm, cursor, B, hL, G, ridge,
                                      inv_n, flags, gW, ldg, g_theta, dH, lddh, loss_partials, n_partials_out,
                                      workspace, workspace_bytes, nullptr, stream);
}

extern "C" int dcahip_heads_fused_ordered(const float* H, long ldh, const float* Wh, long ldw,
                                          const float* bh, long plane, const float* theta_w,
                                          const float* y, long ldy, const float* sf, const int* perm,
                                          const long long* cursor, int B, int hL, int G, float ridge,
                                          float inv_n, int flags, float* gW, long ldg, float* g_theta,
                                          float* dH, long lddh, double* loss_partials, int* n_partials_out,
                                          void* workspace, long workspace_bytes, const int* tile_order,
                                          void* stream) {
    return dcahip_heads_fused_loss(H, ldh, Wh, ldw, bh, plane, theta_w, y, ldy, sf, perm, cursor, B, hL, G, ridge,
                                   inv_n, flags, gW, ldg, g_theta, dH, lddh, loss_partials, n_partials_out,
                                   workspace, workspace_bytes, tile_order, nullptr, stream);
}

extern "C" int dcahip_heads_fused_loss(const float* H, long ldh, const float* Wh, long ldw,
                                       const float* bh, long plane, const float* theta_w,
                                       const float* y, long ldy, const float* sf, const int* perm,
                                       const long long* cursor, int B, int hL, int G, float ridge,
                                       float inv_n, int flags, float* gW, long ldg, float* g_theta,
                                       float* dH, long lddh, double* loss_partials, int* n_partials_out,
                                       void* workspace, long workspace_bytes, const int* tile_order,
                                       float* loss_out, void* stream) {
    return dcahip_heads_fused_compact(H, ldh, Wh, ldw, bh, plane, theta_w, y, ldy, nullptr, 0, nullptr, nullptr, nullptr,
                                      sf, perm, cursor, B, hL, G, ridge, inv_n, flags, gW, ldg, g_theta, dH, lddh,
                                      loss_partials, n_partials_out, workspace, workspace_bytes, tile_order, loss_out, stream);
}

extern "C" int dcahip_heads_fused_compact(const float* H, long ldh, const float* Wh, long ldw,
                                          const float* bh, long plane, const float* theta_w,
                                          const float* y, long ldy,
                                          const unsigned char* yc, long ldc, const int* ovf_ptr, const int* ovf_col,
                                          const float* ovf_val, const float* sf, const int* perm,
                                          const long long* cursor, int B, int hL, int G, float ridge,
                                          float inv_n, int flags, float* gW, long ldg, float* g_theta,
                                          float* dH, long lddh, double* loss_partials, int* n_partials_out,
                                          void* workspace, long workspace_bytes, const int* tile_order,
                                          float* loss_out, void* stream) {
    const bool has_pi = flags & DCAHIP_NLL_HAS_PI, cdisp = flags & DCAHIP_NLL_CONST_DISP;
    HeadsPlan pl;
    if (!make_heads_plan(B, hL, G, plane, flags, &pl)) return DCAHIP_EINVAL;
    if (!H || !Wh || !bh || (!y && !yc) || !sf || !gW || !dH || !loss_partials || !workspace) return DCAHIP_EINVAL;
    if (cdisp && (!theta_w || !g_theta)) return DCAHIP_EINVAL;
    if (workspace_bytes < pl.dw_bytes + pl.dh_bytes + pl.hs_bytes) return DCAHIP_EINVAL;
    if (!al16(H) || !al16(Wh) || !al16(workspace) || (ldh & 3) || (ldw & 3) || ldh < ((hL + 3) & ~3))
        return DCAHIP_EINVAL;
    const int NH = 1 + (cdisp ? 0 : 1) + (has_pi ? 1 : 0);
    if (ldw < (long)NH * plane || ldg < (long)NH * plane) return DCAHIP_EINVAL;
    if (yc ? (ldc < G || ldc > 0xffffffffL) : (ldy < G || ldy > 0xffffffffL)) return DCAHIP_EINVAL;
    float* ws_dh = static_cast<float*>(workspace);
    float* ws_dw = ws_dh + pl.dh_bytes / sizeof(float);
    hipStream_t s = static_cast<hipStream_t>(stream);
    bool direct_dw = false;
    {
        unsigned short* HA = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(workspace) + pl.dh_bytes + pl.dw_bytes);
        unsigned short* HT = HA + (long)pl.NT * kHTile;
        if (pl.WR != 1) {                    // single-wave workgroups (batches below 256 rows) split H themselves
            hipLaunchKernelGGL(heads_split_h_kernel, dim3(pl.NT), dim3(256), 0, s, H, ldh, B, hL, HA, HT);
            int rc0 = (int)hipGetLastError();
            if (rc0 != 0) return rc0;
        }
        direct_dw = pl.S == 1;
        HeadsArgs2 a{g_timing, HA, HT, H, ldh, gW, ldg, g_theta, Wh, ldw, bh, theta_w, y, ldy, yc, ldc, ovf_ptr, ovf_col, ovf_val,
                     sf, perm, cursor, ws_dw, pl.dw_stride, ws_dh,
                     pl.npart, pl.nitems, tile_order, loss_partials, plane, pl.ldws, B, hL, G, pl.S, pl.NT, ridge, inv_n};
        if (yc) {
            if (has_pi && cdisp) launch_fused_x3<true, true, true>(pl, a, s);
            else if (has_pi) launch_fused_x3<true, false, true>(pl, a, s);
            else if (cdisp) launch_fused_x3<false, true, true>(pl, a, s);
            else launch_fused_x3<false, false, true>(pl, a, s);
        } else {
            if (has_pi && cdisp) launch_fused_x3<true, true, false>(pl, a, s);
            else if (has_pi) launch_fused_x3<true, false, false>(pl, a, s);
            else if (cdisp) launch_fused_x3<false, true, false>(pl, a, s);
            else launch_fused_x3<false, false, false>(pl, a, s);
        }
    }
    int rc = (int)hipGetLastError();
    if (rc != 0) return rc;
    if (n_partials_out) *n_partials_out = pl.grid;
    {
        const int KT = pl.HLB * 32;
        const long nq = (long)B * (KT / 4);
        const ReduceDhArgs qh{ws_dh, pl.npart, B, pl.NT * kTR, KT, hL, dH, lddh, loss_partials, pl.grid, (double)inv_n, loss_out};
        const long total = (long)(hL + 1) * pl.ldws;
        long gr = (total / 4 + 255) / 256;                  // the weight-gradient reduction moves 16 bytes per lane
        if (gr > 2048) gr = 2048;
        if (gr < 1) gr = 1;
        const ReduceDwArgs qw{ws_dw, pl.S, pl.dw_stride, hL, pl.ldws, pl.ldws, gW, ldg, cdisp ? theta_w : nullptr,
                              cdisp ? g_theta : nullptr, G};
        const int gl = (nq <= 1024 && pl.npart >= 256) ? 64 : (nq >= 64L * 512 ? 4 : 16);
        const int n_dh = (int)((nq + 256 / gl - 1) / (256 / gl));
        if (!direct_dw) {
            const dim3 grid(n_dh + (int)gr);
            if (gl == 64) hipLaunchKernelGGL(heads_reduce_both_kernel<64>, grid, dim3(256), 0, s, qh, qw, n_dh);
            else if (gl == 4) hipLaunchKernelGGL(heads_reduce_both_kernel<4>, grid, dim3(256), 0, s, qh, qw, n_dh);
            else hipLaunchKernelGGL(heads_reduce_both_kernel<16>, grid, dim3(256), 0, s, qh, qw, n_dh);
        } else {
            if (gl == 64) hipLaunchKernelGGL(heads_reduce_dh_kernel<64>, dim3(n_dh), dim3(256), 0, s, qh);
            else if (gl == 4) hipLaunchKernelGGL(heads_reduce_dh_kernel<4>, dim3(n_dh), dim3(256), 0, s, qh);
            else hipLaunchKernelGGL(heads_reduce_dh_kernel<16>, dim3(n_dh), dim3(256), 0, s, qh);
        }
        rc = (int)hipGetLastError();
    }
    return rc;
}
