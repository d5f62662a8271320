// K-HEADS: the output heads of the autoencoder as ONE kernel -- forward GEMM of the three
// Dense heads, NB / ZINB negative log-likelihood + gradient, weight/bias gradient and input
// gradient -- so that the [cells x genes] pre-activation and gradient planes never exist in
// HBM.  gfx950 (MI355X), wave64.
//
// Reference path replaced: dca/network.py:369-385 (pi / dispersion / mean Dense heads,
// ColwiseMultLayer, SliceLayer, ZINB loss closure), dca/network.py:38-39, dca/layers.py:21,85,
// dca/loss.py:72-156 and TensorFlow's autodiff of all of it.
//
// Arithmetic of the three matrix products (round 6): fp32 results on the fp16 matrix pipe.  Every fp32 operand block is
// scaled by a power of two that brings its largest magnitude into [2^13, 2^14) and split into TWO fp16 pieces,
// x 2^e = h1 + h2 (round-to-nearest residual: |x 2^e - h1 - h2| <= 2^-22 |x 2^e|, or 2^-25 absolute where h2 is an fp16
// denormal -- the matrix pipe of gfx950 preserves fp16 denormals, tools/microbench/mfma_f16_denorm.hip), and THREE products
// a1 b1 + a1 b2 + a2 b1 are accumulated in fp32 by v_mfma_f32_32x32x16_f16; the dropped a2 b2 is 2^-22 of a product.  Against
// fp64 on the operands of a training step: 3.2e-7 / 2.7e-7 / 5.6e-7 of sum|ab| for F / dH / dW (three-way bf16 splits with six
// products, rounds 2-5: 1.8e-7 / 1.9e-7 / 4.1e-7; oracle/x3_np.py restates both, tests/test_x3_arith_cpu.py) -- at HALF the
// matrix instructions and a fifth of the vector instructions per split (one v_cvt_pk_f16_f32 + two v_fma_mix_f32 + one
// v_cvt_pk_f16_f32 per pair of values).  The powers of two:
//   H  per 32-row tile, 2^eH[t] (heads_split_h_kernel)            W  per gene tile (work item), 2^eW (weight prologue)
//   D  = g 2^kD[t] with g the UNSCALED gradient d nll / d pre-activation (no 1 / n factor) and kD[t] = kD0 - (eH[t] - eHmin):
//      the weight-gradient accumulators of a wave then carry ONE scale 2^(eHmin + kD0) over all its row tiles.  g is O(1)
//      on count data (|g| <= max(theta, ~2 y) by the formulas, 0.16 in the median): kD0 = 8 + d_exp keeps |g| <= 117 inside
//      the fp16 range and every |g| >= 5e-4 at 22 bits.  The likelihood pass keeps the tile's largest |D|; a tile that meets a
//      larger value (a count in the hundreds, a dispersion at its floor) takes a slow path BEFORE anything is split: its staged
//      D and the wave's accumulators are scaled down by what the maximum needs, and back afterwards (exact: powers of two).
//      Nothing saturates, nothing is clamped.
// Every gradient leaves the kernel in g units; the 1 / n factor (inv_n) is applied ONCE per output element at the end.
//
// Work decomposition (gene-stationary):
//   * a workgroup owns ONE 32-gene tile of every head: its [hL x 32 x heads] slice of the head weights sits in
//     LDS for the lifetime of the workgroup as fp16 pieces, ONE image serving both orientations (the forward
//     contracts over hidden units and reads it with the transposing ds_read_b64_tr_b16, dH contracts over
//     genes and reads it directly; a 16-byte-unit rotation per row keeps both conflict-free);
//   * its 8 waves take strided 32-row batch tiles; a wave's weight-gradient slice ([hL x 32] per head) lives
//     in MFMA accumulators for the lifetime of the wave;
//   * the decoder output H is split once per launch by a small kernel into fp16 pieces in the two operand
//     layouts the loop needs (rows x k for the forward, k x rows for dW): A operands are plain 16-byte loads;
//   * per (row tile, gene tile):
//       F   pre-activations  A = H W + b        32 rows x 32 genes x heads, K = 64
//       Z   NLL + d NLL / d A   element-wise, through a wave-private fp32 LDS staging tile (the y = 0
//           formulas densely, the non-zero elements compacted into 64-lane batches)
//       dH  = D W^T   (D read transposed from the staging tile, split on the fly) -> partial per gene tile
//       dW += H^T D   (a lane's staged D column IS its B operand)
//     No workgroup barrier inside the loop: waves drift apart, one wave's element-wise phase runs beside
//     its SIMD partner's matrix phases.
//   * HBM traffic per element: y (1 B from the byte store) + the dH partial instead of 36 B for materialised
//     pre-activations / gradients + 28 B K-ZINB.
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
constexpr int kWG = 2;         // (tile-order length is rounded to pairs: dcahip_heads_tile_order_len)
constexpr int kMaxGrid = 2048;      // grid cap of the persistent kernel
constexpr int kMaxSmallGrid = 8192; // = dcahip_zinb_max_partials(): loss partials of the four-wave kernel (one per workgroup)
constexpr int kCUs = 256;
constexpr int kZU = 4;        // staged rows per Z group (two groups per loop iteration)
constexpr int kQCap = 320;     // non-zero queue entries per wave (< 64 left over + 4 x 64 pushed)

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
// fp16 x 2 arithmetic
// =====================================================================================================
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f16x2v = __attribute__((ext_vector_type(2))) _Float16;
using f32x2v = __attribute__((ext_vector_type(2))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
using s16x4 = __attribute__((ext_vector_type(4))) short;

constexpr int kWR2 = 8;                 // row slots (waves) per workgroup, one gene tile per workgroup
constexpr int kWr8MinNT = 5;            // the 8-wave persistent kernel from this many row tiles on (see make_heads_plan)
constexpr int kHTile = 2 * 32 * 64;     // fp16 elements of one row tile of the split decoder output (2 pieces x 32 x 64)
constexpr int kTop = 13;                // a scaled block's largest magnitude lies in [2^kTop, 2^(kTop + 1))
constexpr int kDExp0 = 8;               // D = g 2^kDExp0: |g| <= 117 (kDLim / 256) stays inside the fp16 range, |g| >= 5e-4 keeps 22 bits
constexpr float kDLim = 30000.f;        // a scaled gradient beyond this sends its tile through the slow path (rescaled as a whole)

#define MFMAH(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0)

__device__ __forceinline__ unsigned pk_f16(float a, float b) {            // v_cvt_pk_f16_f32 (round to nearest): a -> low half
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2v{a, b}, f16x2v));
}
// x - (float) half of p: one v_fma_mix_f32 reading the half straight from the packed register (exact: |x - h| <= 2^-11 |x|)
__device__ __forceinline__ float resid_lo(unsigned p, float x) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
__device__ __forceinline__ float resid_hi(unsigned p, float x) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
// x = h1 + h2 (fp16 each), two values per call: four vector instructions
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& p0, unsigned& p1) {
    p0 = pk_f16(x0, x1);
    p1 = pk_f16(resid_lo(p0, x0), resid_hi(p0, x1));
}
// eight fp32 values (already scaled) -> two MFMA operand fragments (element j of the fragment = value j)
__device__ __forceinline__ void split8(const float (&x)[8], u32x4 (&f)[2]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned a, b;
        split_pair(x[2 * j], x[2 * j + 1], a, b);
        f[0][j] = a; f[1][j] = b;
    }
}
// the three products of one K = 16 step, small terms first
#ifdef DCA_EXP_H2_TWO      // experiment (must FAIL the parity tests: tools/gpu_heads_narrow_check.sh): the A operand's second piece dropped
#define MFMA_H3(A, Bf, ACC) { ACC = MFMAH(A[0], Bf[1], ACC); ACC = MFMAH(A[0], Bf[0], ACC); }
#else
#define MFMA_H3(A, Bf, ACC) { ACC = MFMAH(A[1], Bf[0], ACC); ACC = MFMAH(A[0], Bf[1], ACC); ACC = MFMAH(A[0], Bf[0], ACC); }
#endif

__device__ __forceinline__ float pow2i(int e) { return __uint_as_float((unsigned)(127 + e) << 23); }      // -126 <= e <= 127
// the exponent that brings a block's largest magnitude m into [2^kTop, 2^(kTop + 1)); 0 for an all-zero block
__device__ __forceinline__ int block_exp(float m) {
    if (!(m > 0.f) || !(m < INFINITY)) return 0;
    const int e = kTop + 1 - __builtin_amdgcn_frexp_expf(m);          // frexp: m = f 2^k, f in [0.5, 1)
    return e < -60 ? -60 : (e > 60 ? 60 : e);
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

struct HeadsArgs2 {
    long long* timing;
    const unsigned short* HA;         // [NT][2][32 rows][64 k] fp16 pieces of the scaled decoder output (forward A operand)
    const unsigned short* HT;         // [NT][2][64 k][32 rows, order of the MFMA row map] (dW A operand)
    const int* eH;                    // [NT] exponent of each row tile's scale
    const float* H; long ldh;         // the decoder output itself: the four-wave kernel (one row tile per workgroup) splits it on the fly
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
    float* ws_dw;  long dw_stride;   // [S][(hL + 2)][ldws], in g units
    float* ws_dh;                     // [NT][npart][32][64]: one partial per workgroup of the row tile's batch split, in g units
    int npart, nitems;                // partials per row tile (= grid / S), work items (= S x gene tiles)
    const int* tile_order;
    double* partials;
    long plane, ldws;
    int B, hL, G;
    int S, NT;
    float ridge, inv_n;
    int d_exp;                        // kD0 = kDExp0 + d_exp (<= 0: datasets with counts beyond ~8 000)
    int tile_base;                    // this launch takes the gene tiles tile_order[tile_base ..] (the tail launch: see make_heads_plan)
};

// the count behind an escape byte of the compact store (counts >= 255: rare)
__device__ __forceinline__ float escaped_count(const HeadsArgs2& p, long srow, int col) {
    float v = 255.f;
    if (p.ovf_ptr)
        for (int i = p.ovf_ptr[srow], e = p.ovf_ptr[srow + 1]; i < e; ++i)
            if (p.ovf_col[i] == col) { v = p.ovf_val[i]; break; }
    return v;
}

// Decoder output -> fp16 pieces, once per launch (B x 64 elements: ~1 us).  One 32-row tile per block: the tile's largest
// magnitude fixes its scale 2^eH[t].
__global__ __launch_bounds__(256) void heads_split_h_kernel(const float* H, long ldh, int B, int hL,
                                                            unsigned short* HA, unsigned short* HT, int* eH) {
    __shared__ unsigned short tr[2][64][32 + 2];
    __shared__ float wm[4];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int row = tid >> 3, k0 = (tid & 7) * 8;
    const long grow = (long)t * 32 + row;
    float x[8];
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        x[j] = (grow < B && k0 + j < hL) ? H[grow * ldh + k0 + j] : 0.f;
        m = fmaxf(m, fabsf(x[j]));
    }
    m = wave_max(m);
    if ((tid & 63) == 0) wm[tid >> 6] = m;
    __syncthreads();
    const int e = block_exp(fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3])));
    if (tid == 0) eH[t] = e;
    const float sc = pow2i(e);
#pragma unroll
    for (int j = 0; j < 8; ++j) x[j] *= sc;
    u32x4 f[2];
    split8(x, f);
    // row -> position inside the tile's transposed image: the order in which a lane of the 32x32 MFMA result
    // holds its rows (row bits 2 and 3 swapped), so that 8 consecutive positions are one K = 16 operand half
    const int pos = (row & 0x13) | ((row & 4) << 1) | ((row & 8) >> 1);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        *reinterpret_cast<u32x4*>(HA + (((long)t * 2 + q) * 32 + row) * 64 + k0) = f[q];
#pragma unroll
        for (int j = 0; j < 8; ++j) tr[q][k0 + j][pos] = (unsigned short)(f[q][j >> 1] >> (16 * (j & 1)));
    }
    __syncthreads();
    for (int c = tid; c < 2 * 64 * 4; c += 256) {           // 16-byte chunks of [2][64][32]
        const int q = c >> 8, i = (c >> 2) & 63, part = c & 3;
        u32x4 v;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (unsigned)tr[q][i][part * 8 + 2 * j] | ((unsigned)tr[q][i][part * 8 + 2 * j + 1] << 16);
        *reinterpret_cast<u32x4*>(HT + (((long)t * 2 + q) * 64 + i) * 32 + part * 8) = v;
    }
}

template <bool HAS_PI, bool CONST_DISP, bool YC>
__global__ __launch_bounds__(64 * kWR2) void heads_fused_h2_kernel(HeadsArgs2 p) {
    using YV = std::conditional_t<YC, unsigned, float>;        // a count as the kernel holds it: the byte code / the fp32 value
    constexpr int WR = kWR2;
    constexpr int NH = 1 + (CONST_DISP ? 0 : 1) + (HAS_PI ? 1 : 0);
    constexpr int PI_H = NH - 1;
    constexpr int KT = 64;
    constexpr int ST_PLANE = kTG * kLdS;                   // fp32 [gene][row] plane of a per-gene dispersion's gradient sums
    constexpr int QCAP = 192;                              // queue entries (16 bytes each): at most 63 left over + 2 x 64 pushed
    constexpr int CELL_LD = 144;                           // bytes per gene of a head's cell tile: 8 cells of 16 bytes + one of padding
    constexpr int HEAD_B = kTG * CELL_LD;                  // bytes of one head's cell tile
    constexpr int ROWT = 96;                               // the wave's row table: storage row of the tile's 32 rows (this tile | next) + size factors
    constexpr int PW_FLOATS = NH * HEAD_B / 4 + (CONST_DISP ? ST_PLANE : 0) + QCAP * 4 + ROWT;      // per wave
    constexpr int NTHREADS = 64 * WR;
    constexpr int W_PIECE = 64 * 64;                       // bytes of one (head, piece) weight image: 64 k x 32 genes fp16
    constexpr int W_FLOATS = NH * 2 * W_PIECE / 4;
    constexpr int NRED = NH * 2 * 16 + NH + 1;
    constexpr int BIAS_FLOATS = (NH + 1) * 32;             // biases (+ log-dispersion) of the 32 genes
    constexpr int LDS_FLOATS = W_FLOATS + WR * PW_FLOATS + BIAS_FLOATS;
    static_assert((WR / 2) * NRED * 64 <= LDS_FLOATS, "dW reduce scratch");
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ double lred[WR];
    __shared__ float wmaxs[WR];

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
    // the scale of this wave's weight-gradient accumulators: the smallest exponent among ITS row tiles (the same tiles in
    // every work item) + kD0
    const int kD0 = kDExp0 + p.d_exp;
    int eHmin = 1 << 20;
    for (int t = s_wg * WR + r; t < p.NT; t += p.S * WR) { const int e = p.eH[t]; eHmin = e < eHmin ? e : eHmin; }
    eHmin = __builtin_amdgcn_readfirstlane(eHmin == (1 << 20) ? 0 : eHmin);
#pragma unroll 1
    for (int j = 0; j < nrounds; ++j) {
    const int gb = j * p.npart + ((j & 1) ? p.npart - 1 - wq : wq);
    int s_opaque = s_wg;
    asm volatile("" : "+v"(s_opaque));       // opaque per round: what depends on it is recomputed, not kept live across rounds
    const int s = __builtin_amdgcn_readfirstlane(s_opaque);
    const int gt = p.tile_order ? p.tile_order[gb + p.tile_base] : gb + p.tile_base;
    const int g0 = gt * kTG;
    const int gene = g0 + l31;
    const bool tile_ok = g0 < p.G;
    const bool gvalid = gene < p.G;

    unsigned char* const Wimg = reinterpret_cast<unsigned char*>(lds);
    // wave-private: the cell tile [head][32 genes][8 cells of 16 bytes] -- a cell is 4 consecutive row POSITIONS of a gene
    // (positions: the order of the MFMA row map / of the transposed H image): first the four fp32 pre-activations F leaves
    // there, then, IN PLACE, the gradient's two fp16 pieces [piece 0: 4 x fp16 | piece 1: 4 x fp16] --, the fp32 plane of a
    // per-gene dispersion, the non-zero queue
    unsigned char* const Pimg = reinterpret_cast<unsigned char*>(lds + W_FLOATS + wave * PW_FLOATS);
    float* const Th = lds + W_FLOATS + wave * PW_FLOATS + NH * HEAD_B / 4;
    u32x4* const Qe = reinterpret_cast<u32x4*>(lds + W_FLOATS + wave * PW_FLOATS + NH * HEAD_B / 4 + (CONST_DISP ? ST_PLANE : 0));
    float* const Bs = lds + W_FLOATS + WR * PW_FLOATS;          // [head][32] biases, then [32] log-dispersion
    // the wave's row table: the dense pass takes the storage rows / size factors of its four consecutive rows with ONE 16-byte
    // read each (they live lane-distributed in registers: eight cross-lane reads per group before)
    int* const Rt = reinterpret_cast<int*>(lds + W_FLOATS + wave * PW_FLOATS + PW_FLOATS - ROWT);      // [2][32]
    float* const Sft = reinterpret_cast<float*>(Rt + 64);                                              // [32]

    // ---- head weights of this gene tile -> LDS as fp16 pieces scaled by 2^eW (eW from the tile's largest weight), image
    // [head][piece][k][32 genes], the four 16-byte units of a row rotated by (k >> 2): conflict-free for the direct
    // 16-byte reads of dH (lanes = rows k) and for the transposing reads of F (4 consecutive rows per lane group)
    int eW = 0;
    if (tile_ok) {
        // ONE batch of independent 16-byte loads (clamped addresses, zeroed afterwards): one memory round trip
        constexpr int UNR = (NH * 64 * 8 + NTHREADS - 1) / NTHREADS;
        constexpr int ITEMS = NH * 64 * 8;
        float4 v[UNR];
        float m = 0.f;
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = tid + u * NTHREADS;
            const int c4 = idx & 7, kq = (idx >> 3) & 63, h = (idx >> 9) < NH ? (idx >> 9) : NH - 1;
            const int kc = kq < p.hL ? kq : p.hL - 1;
            long gcol = g0 + c4 * 4;
            if (gcol > p.plane - 4) gcol = p.plane - 4;
            v[u] = *reinterpret_cast<const float4*>(p.Wh + (long)kc * p.ldw + (long)h * p.plane + gcol);
        }
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = tid + u * NTHREADS;
            const int c4 = idx & 7, kq = (idx >> 3) & 63;
            const int gcol = g0 + c4 * 4;
            const bool kv = idx < ITEMS && kq < p.hL && gcol <= p.plane - 4;
            if (!kv || gcol + 0 >= p.G) v[u].x = 0.f;
            if (!kv || gcol + 1 >= p.G) v[u].y = 0.f;
            if (!kv || gcol + 2 >= p.G) v[u].z = 0.f;
            if (!kv || gcol + 3 >= p.G) v[u].w = 0.f;
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
        }
        m = wave_max(m);
        if (lane == 0) wmaxs[wave] = m;
        __syncthreads();
        m = fmaxf(fmaxf(fmaxf(wmaxs[0], wmaxs[1]), fmaxf(wmaxs[2], wmaxs[3])), fmaxf(fmaxf(wmaxs[4], wmaxs[5]), fmaxf(wmaxs[6], wmaxs[7])));
        eW = __builtin_amdgcn_readfirstlane(block_exp(m));
        const float wsc = pow2i(eW);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = tid + u * NTHREADS;
            if (idx >= ITEMS) continue;
            const int c4 = idx & 7, kq = (idx >> 3) & 63, h = idx >> 9;
            unsigned a0, a1, b0, b1;
            split_pair(v[u].x * wsc, v[u].y * wsc, a0, a1);
            split_pair(v[u].z * wsc, v[u].w * wsc, b0, b1);
            const int off = kq * 64 + ((((c4 >> 1) + (kq >> 2)) & 3) << 4) + ((c4 & 1) << 3);
            *reinterpret_cast<u32x2*>(Wimg + (h * 2 + 0) * W_PIECE + off) = u32x2{a0, b0};
            *reinterpret_cast<u32x2*>(Wimg + (h * 2 + 1) * W_PIECE + off) = u32x2{a1, b1};
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
        // the lane's 16-byte units of a dH partial: batch row of ROW POSITION l31 (bits 2 and 3 swapped), hidden units 4 hi ..
        const int dh_lane = ((((l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1)) * KT) + 4 * hi) * 4;
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
        // ahead of their use (the first forward step of the NEXT tile during the dW products of this one)
        u32x4 ha0[2];
        auto load_ha = [&](int tt, int ks, u32x4 (&dst)[2]) {
            const int so = tt * (kHTile * 2);
#pragma unroll
            for (int q = 0; q < 2; ++q)
                dst[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ha_rs, ha_lane + q * 4096 + ks * 16, so, 0));
        };
        auto load_ht = [&](int tt, int ks, u32x4 (&dst)[2][2]) {
            const int so = tt * (kHTile * 2);
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
                    dst[q][ib] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                        ht_rs, ht_lane + q * 4096 + ib * 2048 + ks * 32, so, 0));
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
            const unsigned char* b0 = Wimg + (h * 2 + q) * W_PIECE + ks * 512;
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
            return *reinterpret_cast<const u32x4*>(Wimg + (h * 2 + q) * W_PIECE + jb * 2048 + wdr[gs]);
        };
        // Cell tile addresses.  Group g of the lane's 16 rows (accumulator elements 4 g .. 4 g + 3 = rows 8 g + 4 hi ..) holds
        // the row positions 4 c .. 4 c + 3 with c = (g & 1) + 2 hi + 4 (g >> 1): cell c of the lane's gene.  dW reads the
        // cells 2 hi + 4 ks and + 1 (its 8 positions of K step ks).  dtr: the lane's chunk of the transposing read of dH --
        // gene 8 hi + t16 / 4 (+ 16 gs + 4 half) of a K step, cell 4 (lane / 16 & 1) + (t16 & 3) (+ 8 bytes: piece 1)
        const int cell_lane = l31 * CELL_LD + 2 * hi * 16;          // + ((g & 1) + 4 (g >> 1)) * 16
        const int dtr = (8 * hi + (t16 >> 2)) * CELL_LD + c8 * 16;

        int eHt = 0;
        if (t < p.NT) {
            srow_l = load_srow(t);
            sf_l = p.sf[srow_l];
            eHt = p.eH[t];
            load_ha(t, 0, ha0);
            if (lane < 32) { Rt[l31] = srow_l; Sft[l31] = sf_l; }
            wave_sync();
            {
                const int4 r4 = *reinterpret_cast<const int4*>(Rt + 4 * hi);
                yA[0] = count_at(r4.x); yA[1] = count_at(r4.y); yA[2] = count_at(r4.z); yA[3] = count_at(r4.w);
            }
        }
        int rb = 0;                                  // which half of Rt holds the current tile's rows
#ifdef DCA_HEADS_TIMING
        long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        long long tlast = __builtin_readcyclecounter();
        t_loop0 = tlast;
#endif
        int tile_no = wave >> 2;
        for (; t < p.NT; t += tstep) {
#if defined(DCA_EXP_PHASEPRIO)
            __builtin_amdgcn_s_setprio(2);           // experiment: the matrix phases above the element-wise pass
#elif !defined(DCA_EXP_NOPRIO)
            if ((tile_no++) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0);
#endif
            TSTAMP(0)
            const int row0 = t * kTR;
            const int tn = t + tstep < p.NT ? t + tstep : t;
            // the tile's scales (wave-uniform): forward products carry 2^(eH[t] + eW), the gradient D = g 2^kDt
            eHt = __builtin_amdgcn_readfirstlane(eHt);
            const int kDt = kD0 - (eHt - eHmin);
            const float fscale = pow2i(eHt + eW), funscale = pow2i(-(eHt + eW));
            // F and Z run until the tile's gradients fit the fp16 range at the scale 2^kDe: once, but for a tile that holds a
            // count in the hundreds or a dispersion at its floor (wave-uniform, rare, exact: see Z)
            int kDe = kDt;
            float lacc;
            const float thw = CONST_DISP ? Bs[NH * 32 + l31] : 0.f;
            const int srow_n = load_srow(tn);
            if (lane < 32) Rt[(rb ^ 1) * 32 + l31] = srow_n;        // (read after the wave_sync that follows the staging stores)
            const int eHn = p.eH[tn];
            float scs = 0.f;                          // 2^kDe
            int qn = 0;                               // queue fill
            float dmax = 0.f;                         // the lane's largest |D| of this tile (matrix-product planes only)
#ifdef DCA_HEADS_TIMING
            long long tsparse = 0;
#endif
            auto z_sparse = [&](int q0, int cnt) {
                const bool act = lane < cnt;
                const u32x4 ent = Qe[q0 + (act ? lane : 0)];
                const unsigned e = ent[0];
                const int gq = (e >> 5) & 31, row = e & 31;
                const float sfr = __shfl(sf_l, row, 64);
                const int sr = __shfl(srow_l, row, 64);
                const float am = __uint_as_float(ent[1]);
                const float ad = CONST_DISP ? Bs[NH * 32 + gq] : __uint_as_float(ent[2]);
                const float ap = HAS_PI ? __uint_as_float(ent[3]) : 0.f;
                float yq = (float)(e >> 16);
                // counts that do not fit the queue's 16 bits (rare): a wave-uniform branch around the memory access, and the
                // wait for it INSIDE the branch -- left to the compiler, the join in front of the first use of yq waits for
                // vmcnt(0), i.e. for every prefetch in flight, in every batch
                if (__ballot(act && (e >> 16) == 0xFFFFu)) {
                    if (act && (e >> 16) == 0xFFFFu) yq = YC ? escaped_count(p, sr, g0 + gq) : p.y[(long)sr * p.ldy + g0 + gq];
                    __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
                }
                float o[3] = {0.f, 0.f, 0.f};
                float nll;
                if (HAS_PI) {
                    nll = zinb_nz_elem<CONST_DISP, YC>(am, ad, ap, sfr, yq, p.ridge, o[0], o[1], o[2]);
                } else {
                    float dmu = 0.f, dth = 0.f, dpi = 0.f;
                    const Heads hd = head_acts<HAS_PI, CONST_DISP>(am, ad, ap, sfr);
                    nll = nll_elem<HAS_PI, true, true, YC>(hd, yq, p.ridge, dmu, dth, dpi);
                    o[0] = dmu * hd.gm; o[1] = dth * hd.gd;
                }
                lacc += act ? nll : 0.f;
                o[0] *= scs; o[1] *= scs; o[2] *= scs;
                dmax = fmaxf(dmax, act ? fmaxf(fmaxf(fabsf(o[0]), CONST_DISP ? 0.f : fabsf(o[1])), fabsf(o[2])) : 0.f);
                if (act) {
                    const int pos = (row & 0x13) | ((row & 4) << 1) | ((row & 8) >> 1);
                    unsigned char* pa = Pimg + gq * CELL_LD + (pos >> 2) * 16 + (pos & 3) * 2;
                    auto put = [&](int h, float v) {
                        const _Float16 h1 = (_Float16)v;
                        const _Float16 h2 = (_Float16)(v - (float)h1);
                        *reinterpret_cast<_Float16*>(pa + h * HEAD_B) = h1;
                        *reinterpret_cast<_Float16*>(pa + h * HEAD_B + 8) = h2;
                    };
                    put(0, o[0]);
                    if (CONST_DISP) Th[gq * kLdS + row] = o[1]; else put(1, o[1]);
                    if (HAS_PI) put(PI_H, o[2]);
                }
            };
            auto z_flush = [&](bool last) {
                while (qn >= 64 || (last && qn > 0)) {
                    const int c = qn < 64 ? qn : 64;
                    wave_sync();
#ifdef DCA_HEADS_TIMING
                    const long long f0 = __builtin_readcyclecounter();
#endif
                    z_sparse(qn - c, c);
#ifdef DCA_HEADS_TIMING
                    tsparse += __builtin_readcyclecounter() - f0;        // (reported in slot 2; still part of slot 4's total)
#endif
                    qn -= c;
                }
            };
            auto z_dense = [&](auto fullv, int grp, const YV (&yv)[kZU]) {
                constexpr bool FULLV = decltype(fullv)::value;
                // the group's pre-activations: one 16-byte cell per head
                unsigned char* const cell = Pimg + cell_lane + ((grp & 1) + 4 * (grp >> 1)) * 16;
                float i_am[kZU], i_ad[kZU], i_ap[kZU];
                {
                    const float4 vm = *reinterpret_cast<const float4*>(cell);
                    i_am[0] = vm.x; i_am[1] = vm.y; i_am[2] = vm.z; i_am[3] = vm.w;
                    if (CONST_DISP) {
#pragma unroll
                        for (int j = 0; j < kZU; ++j) i_ad[j] = thw;
                    } else {
                        const float4 vd = *reinterpret_cast<const float4*>(cell + HEAD_B);
                        i_ad[0] = vd.x; i_ad[1] = vd.y; i_ad[2] = vd.z; i_ad[3] = vd.w;
                    }
                    if (HAS_PI) {
                        const float4 vp = *reinterpret_cast<const float4*>(cell + PI_H * HEAD_B);
                        i_ap[0] = vp.x; i_ap[1] = vp.y; i_ap[2] = vp.z; i_ap[3] = vp.w;
                    } else {
#pragma unroll
                        for (int j = 0; j < kZU; ++j) i_ap[j] = 0.f;
                    }
                }
                float o_m[kZU], o_d[kZU], o_p[kZU];
                bool o_nz[kZU];
                const float4 sf4 = *reinterpret_cast<const float4*>(Sft + 8 * grp + 4 * hi);       // rows 8 grp + 4 hi ..
                const float sfj[kZU] = {sf4.x, sf4.y, sf4.z, sf4.w};
#pragma unroll
                for (int j = 0; j < kZU; ++j) {
                    const int row = rowmap(grp * kZU + j, hi);
                    const bool valid = FULLV || ((row0 + row < p.B) && gvalid);
                    const YV yj = yv[j];
                    bool nz;
                    if constexpr (YC) nz = valid && yj != 0u;
                    else nz = valid && (HAS_PI ? !(yj < kZeroThresh) : (yj != 0.f));
                    const float sc = (valid && !nz) ? scs : 0.f;        // (a non-zero element's pieces come from the compacted pass)
                    if (HAS_PI) {
                        float gmv, gdv, gpv;
                        const float nll = zinb_zero_elem<CONST_DISP>(i_am[j], i_ad[j], i_ap[j], sfj[j], p.ridge, gmv, gdv, gpv);
                        lacc += (valid && !nz) ? nll : 0.f;
                        o_m[j] = gmv * sc;
                        o_d[j] = gdv * sc;
                        o_p[j] = gpv * sc;
                        dmax = fmaxf(dmax, fmaxf(fabsf(o_m[j]), fabsf(o_p[j])));
                    } else {
                        float gmv, gdv;
                        const float nll = nb_zero_elem<CONST_DISP>(i_am[j], i_ad[j], sfj[j], gmv, gdv);
                        lacc += (valid && !nz) ? nll : 0.f;
                        o_m[j] = gmv * sc;
                        o_d[j] = gdv * sc;
                        o_p[j] = 0.f;
                        dmax = fmaxf(dmax, fabsf(o_m[j]));
                    }
                    if (!CONST_DISP) dmax = fmaxf(dmax, fabsf(o_d[j]));
                    o_nz[j] = nz;
                }
                // the group's gradients as fp16 pieces, in place of the pre-activations: [piece 0: 4 positions | piece 1]
                {
                    auto put4 = [&](int h, const float (&v)[kZU]) {
                        unsigned a0, a1, b0, b1;
                        split_pair(v[0], v[1], a0, a1);
                        split_pair(v[2], v[3], b0, b1);
                        *reinterpret_cast<u32x4*>(cell + h * HEAD_B) = u32x4{a0, b0, a1, b1};
                    };
                    put4(0, o_m);
                    if (!CONST_DISP) put4(1, o_d);
                    if (HAS_PI) put4(PI_H, o_p);
                }
#pragma unroll
                for (int j = 0; j < kZU; ++j) {
                    const int row = rowmap(grp * kZU + j, hi);
                    const bool nz = o_nz[j];
                    const unsigned long long m = __ballot(nz);
                    const int slot = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (nz) {
                        const YV yj = yv[j];
                        unsigned y16;
                        if constexpr (YC) y16 = yj == 255u ? 0xFFFFu : yj;
                        else y16 = (yj < 65535.f && yj == floorf(yj)) ? (unsigned)yj : 0xFFFFu;
                        Qe[slot] = u32x4{(unsigned)(l31 * 32 + row) | (y16 << 16), __float_as_uint(i_am[j]), __float_as_uint(i_ad[j]),
                                         __float_as_uint(i_ap[j])};
                    } else if (CONST_DISP) {
                        Th[l31 * kLdS + row] = o_d[j];
                    }
                    qn += __popcll(m);
                    if (j == 1) z_flush(false);          // (the queue holds at most 63 + 2 x 64 entries)
                }
                z_flush(false);
            };
            auto load_y = [&](int half, int grp, YV (&yv)[kZU]) {          // half: 0 this tile's rows, 1 the next tile's
                const int4 r4 = *reinterpret_cast<const int4*>(Rt + (rb ^ half) * 32 + 8 * grp + 4 * hi);
                yv[0] = count_at(r4.x); yv[1] = count_at(r4.y); yv[2] = count_at(r4.z); yv[3] = count_at(r4.w);
            };
            auto z_loop = [&](auto fullv) {             // (ONE call site of the dense pass: its code, and the non-zero pass
#pragma unroll 1                                        //  inside it, exist once per row-range variant)
                for (int grp = 0; grp < 16 / kZU; ++grp) {
                    if (grp + 1 < 16 / kZU) load_y(0, grp + 1, yB);
                    else load_y(1, 0, yB);               // the next tile's first group
                    z_dense(fullv, grp, yA);
#pragma unroll
                    for (int j = 0; j < kZU; ++j) yA[j] = yB[j];
                }
                z_flush(true);
            };
            for (;;) {
            // ---- F: pre-activations, K = 64 as 4 steps of 16 (lane half hi covers k in [32 hi, 32 hi + 32))
            // (the accumulators start from the scaled bias of the lane's gene: one LDS read per head instead of 16 additions)
            f32x16 acc[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) {
                const float bias_h = Bs[h * 32 + l31] * fscale;
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[h][e] = bias_h;
            }
            {
                u32x4 hb[2][2];
#pragma unroll
                for (int q = 0; q < 2; ++q) hb[0][q] = ha0[q];
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    if (ks < 3) load_ha(t, ks + 1, hb[(ks + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);          // the request stays AHEAD of this step's products
#pragma unroll
                    for (int h = 0; h < NH; ++h) {
                        u32x4 bf[2] = {w_tr(h, 0, ks), w_tr(h, 1, ks)};
                        MFMA_H3(hb[ks & 1], bf, acc[h])
                    }
                }
            }
            TSTAMP(1)
            TSTAMP(2)
            // ---- stage the unscaled pre-activations: one 16-byte cell per head and group of four rows
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *reinterpret_cast<float4*>(Pimg + h * HEAD_B + cell_lane + ((g & 1) + 4 * (g >> 1)) * 16) =
                        make_float4(acc[h][4 * g] * funscale, acc[h][4 * g + 1] * funscale, acc[h][4 * g + 2] * funscale, acc[h][4 * g + 3] * funscale);
            wave_sync();
            TSTAMP(3)

            // ---- Z: element-wise likelihood and gradient (dense y = 0 pass over the staged cells; the non-zero elements compacted
            // into 64-lane batches through a queue that carries their three pre-activations).  The gradients D = g 2^kDe are
            // split ONCE, here, into their two fp16 pieces, which replace the pre-activations in their cell: dW reads its B
            // operand with 8-byte reads, dH its A operand with transposing reads.  A tile whose largest |D| would leave the fp16
            // range (a count in the hundreds, a dispersion at its floor) repeats F and this pass with the scale its maximum
            // needs.
            scs = pow2i(kDe);
            lacc = 0.f;
            qn = 0;
            dmax = 0.f;
#if defined(DCA_EXP_PHASEPRIO)
            __builtin_amdgcn_s_setprio(0);
#endif
#ifdef DCA_HEADS_TIMING
            tsparse = 0;
#endif
#ifndef DCA_EXP_X3NOZ
            if (row0 + kTR <= p.B && g0 + kTG <= p.G) z_loop(std::true_type{}); else z_loop(std::false_type{});
#endif
#ifdef DCA_HEADS_TIMING
            tacc[2] += tsparse;
#endif
            if (!__ballot(!(dmax <= kDLim))) break;   // (a NaN / inf gradient takes the slow branch once and stays what it is)
            const float mx = wave_max(dmax);
            const int need = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_frexp_expf(mx) - 1 - kTop);    // mx 2^-need < 2^(kTop + 1)
            if (!(need > 0) || kDe < kDt - 200) break;
            kDe -= need > 100 ? 100 : need;
            load_y(0, 0, yA);                        // the pass consumed the counts of its first group, F its first operands,
            load_ha(t, 0, ha0);                      // and the cells hold pieces: once more from the products
            wave_sync();
            }   // F + Z (until the scale fits)
            dacc += (double)lacc;
            const float sf_n = p.sf[srow_n];
#if defined(DCA_EXP_PHASEPRIO)
            __builtin_amdgcn_s_setprio(2);
#endif
            wave_sync();
            TSTAMP(4)
            const int shift = kDt - kDe;
            if (shift) {                             // the wave's accumulators to the scale of this tile's D
                const float down = pow2i(-shift);
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                        for (int e = 0; e < 16; ++e) dW[h][ib][e] *= down;
            }

            // ---- dH[row, i] = sum_genes D[row, gene] W[i, gene]: A = the D pieces read transposed (row position l31, 8 genes
            // per K-step half), B = the weight image read directly.  MFMA row m = row POSITION m (the order of the transposed
            // H image): batch row (m & 0x13) | bit 2 <-> bit 3.
            u32x4 htb[2][2][2];
            load_ht(t, 0, htb[0]);                   // first K step of the dW operands: in flight during the dH products
            __builtin_amdgcn_sched_barrier(0);
            {
                // the accumulators start from what the workgroup's earlier gene tiles left for this row tile, brought to
                // this tile's scale 2^(kDe + eW) (exact: a power of two); the partial itself is kept in g units
                const float fs = pow2i(kDe + eW), fu = pow2i(-(kDe + eW));
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
                        for (int eq = 0; eq < 4; ++eq) {
                            const u32x4 u = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(dh_rs, dh_lane + 128 * jb + 32 * eq, 0, 0));
#pragma unroll
                            for (int w = 0; w < 4; ++w) { const unsigned uw = u[w]; dHa[jb][4 * eq + w] = __uint_as_float(uw) * fs; }
                        }
                }
                // the products TRANSPOSED -- hidden units x row positions: a lane then holds four consecutive hidden units of its
                // row per accumulator quad, and the partial moves in 16-byte units (8 loads + 8 stores per tile instead of 32 + 32)
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int gs = 0; gs < 2; ++gs) {
                        u32x4 af[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const unsigned char* b0 = Pimg + h * HEAD_B + gs * 16 * CELL_LD + 8 * q + dtr;
                            const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b0));
                            const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(b0 + 4 * CELL_LD));
                            const u32x2 a = __builtin_bit_cast(u32x2, lo), b = __builtin_bit_cast(u32x2, hi4);
                            af[q] = u32x4{a[0], a[1], b[0], b[1]};
                        }
#pragma unroll
                        for (int jb = 0; jb < 2; ++jb) {
                            u32x4 bf[2] = {w_dr(h, 0, jb, gs), w_dr(h, 1, jb, gs)};
                            MFMA_H3(bf, af, dHa[jb])
                        }
                    }
                TSTAMP(7)
#pragma unroll
                for (int jb = 0; jb < 2; ++jb)
#pragma unroll
                    for (int eq = 0; eq < 4; ++eq) {
                        u32x4 v;
#pragma unroll
                        for (int w = 0; w < 4; ++w) v[w] = __float_as_uint(dHa[jb][4 * eq + w] * fu);
                        __builtin_amdgcn_raw_buffer_store_b128(v, dh_rs, dh_lane + 128 * jb + 32 * eq, 0, 0);
                    }
            }
            load_ht(t, 1, htb[1]);
            load_ha(tn, 0, ha0);                     // next tile's first forward step: in flight during the dW products
            __builtin_amdgcn_sched_barrier(0);
            TSTAMP(5)
            // ---- dW[i, gene] += sum_rows H[row, i] D[row, gene]: B = the lane's own D column as stored (row positions in
            // the order of the MFMA row map = the order of the transposed H image), A = H^T pieces.  The accumulators carry
            // 2^(eHmin + kD0) = 2^(eH[t] + kDt) in every tile.  The column sums (bias gradients) from the pieces: v_dot2_f32_f16.
            float tsum[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h) tsum[h] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    u32x4 bf[2];
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const unsigned char* c0 = Pimg + h * HEAD_B + cell_lane + 4 * ks * 16 + 8 * q;      // cells 2 hi + 4 ks and + 1
                        const u32x2 lo = *reinterpret_cast<const u32x2*>(c0), hi2 = *reinterpret_cast<const u32x2*>(c0 + 16);
                        bf[q] = u32x4{lo[0], lo[1], hi2[0], hi2[1]};
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            const unsigned pw = bf[q][w];        // (a bit_cast of the vector element itself takes element 0)
                            tsum[h] = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2v, pw), f16x2v{(_Float16)1.f, (_Float16)1.f}, tsum[h], false);
                        }
                    }
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib) {
                        u32x4 af[2] = {htb[ks][0][ib], htb[ks][1][ib]};
                        MFMA_H3(af, bf, dW[h][ib])
                    }
                }
            {
                const float gun = pow2i(-kDe);       // the tile's column sums back to g units
#pragma unroll
                for (int h = 0; h < NH; ++h) bsum[h] = fmaf(tsum[h], gun, bsum[h]);
                if (CONST_DISP) {
                    float ts = 0.f;
#pragma unroll
                    for (int e = 0; e < 16; ++e) ts += Th[l31 * kLdS + rowmap(e, hi)];
                    thsum = fmaf(ts, gun, thsum);
                }
            }
            if (shift) {                             // the accumulators back to the wave's scale
                const float up = pow2i(shift);
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                        for (int e = 0; e < 16; ++e) dW[h][ib][e] *= up;
            }
            srow_l = srow_n;
            sf_l = sf_n;
            eHt = eHn;
            if (lane < 32) Sft[l31] = sf_n;          // (the pass over this tile is over: its size factors are no longer read)
            rb ^= 1;
            wave_sync();
            TSTAMP(6)
        }
#ifdef DCA_HEADS_TIMING
        t_loop1 = __builtin_readcyclecounter();
        if (p.timing && lane == 0)
            for (int i = 0; i < 8; ++i) p.timing[((long)blockIdx.x * WR + wave) * 10 + i] += tacc[i];
#endif
        // the wave's weight gradient back to g units x H (exact: a power of two) before the waves are summed
        {
            const float wun = pow2i(-(eHmin + kD0));
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int e = 0; e < 16; ++e) dW[h][ib][e] *= wun;
        }
    }

    __syncthreads();                      // every wave is done with the LDS weights / staging of this item

    // ---- dW / bias-gradient sums of the WR row slots: ordered tree through LDS
    {
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
        const float fin = direct ? p.inv_n : 1.f;    // (partials stay in g units: the reduce applies 1 / n once)
        const bool cw = gene < p.plane;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
#pragma unroll
            for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = ib * 32 + rowmap(e, hi);
                    if (cw && i < p.hL) out[(long)i * ldo + (long)h * p.plane + gene] = dW[h][ib][e] * fin;
                }
            const float bv = bsum[h] + __shfl_xor(bsum[h], 32, 64);
            if (cw && hi == 0) out[(long)p.hL * ldo + (long)h * p.plane + gene] = bv * fin;
        }
        if (CONST_DISP) {
            const float tv = thsum + __shfl_xor(thsum, 32, 64);
            if (direct) {                            // ConstantDispersionLayer chain (dca/layers.py:17-21)
                if (gvalid && hi == 0) {
                    const float e = expf(p.theta_w[gene]);
                    p.g_theta[gene] = (e >= 1e-3f && e <= 1e4f) ? tv * p.inv_n * e : 0.f;
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
    float scale;                       // the partials are sums of UNSCALED gradients (g units): 1 / n is applied here, once
    // the gene tiles of the TAIL launch (make_heads_plan): their columns are summed from ws2 over S2 partials instead
    const float* ws2; int S2; const int* tail_tiles; int ntail; long plane;
    int tail_first;                    // tail_tiles == NULL (file order): the tail launch took the tiles tail_first .. tail_first + ntail - 1
};

// bid / nblk: this workgroup's index among the nblk that share the reduction (the stand-alone kernel: blockIdx / gridDim;
// the combined launch below: the workgroups behind those of the dH reduction)
__device__ __forceinline__ void reduce_dw_body(const ReduceDwArgs& q, int bid, int nblk) {
    const int hL = q.hL;
    const long ldws = q.ldws, ncols = q.ncols; float* gW = q.gW; const long ldg = q.ldg;
    const float* theta_w = q.theta_w; float* g_theta = q.g_theta; const int G = q.G;
    const float sc = q.scale;
    // which gene tiles came from the tail launch (a bit per tile; kMaxGrid tiles at most)
    __shared__ unsigned tailbm[kMaxGrid / 32];
    if (q.ntail > 0) {
        for (int i = threadIdx.x; i < kMaxGrid / 32; i += 256) tailbm[i] = 0u;
        __syncthreads();
        for (int i = threadIdx.x; i < q.ntail; i += 256) {
            const int tl = q.tail_tiles ? q.tail_tiles[i] : q.tail_first + i;
            if (tl >= 0 && tl < kMaxGrid) atomicOr(&tailbm[tl >> 5], 1u << (tl & 31));
        }
        __syncthreads();
    }
    auto is_tail = [&](long col_in_plane) { const int tl = (int)(col_in_plane / kTG); return q.ntail > 0 && ((tailbm[tl >> 5] >> (tl & 31)) & 1u); };
    if ((ncols & 3) == 0 && (ldws & 3) == 0 && (ldg & 3) == 0 && (q.stride & 3) == 0 && (q.plane & 3) == 0 &&
        ((reinterpret_cast<uintptr_t>(q.ws) | reinterpret_cast<uintptr_t>(gW) | reinterpret_cast<uintptr_t>(q.ws2)) & 15) == 0) {
        // 16 bytes per lane (the plane width is a multiple of 4: a quad never straddles two gene tiles); partial s is added in
        // order s = 0, 1, ..
        const long nq = ncols >> 2, totalq = (long)(hL + 1) * nq;
        for (long idx = (long)bid * 256 + threadIdx.x; idx < totalq; idx += (long)nblk * 256) {
            const long i = idx / nq, c = (idx - i * nq) << 2;
            const bool tl = is_tail(c % q.plane);
            const float* ws = tl ? q.ws2 : q.ws; const int S = tl ? q.S2 : q.S; const long stride = q.stride;
            float4 v = *reinterpret_cast<const float4*>(ws + i * ldws + c);
            for (int s = 1; s < S; ++s) {
                const float4 x = *reinterpret_cast<const float4*>(ws + (long)s * stride + i * ldws + c);
                v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
            }
            v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
            *reinterpret_cast<float4*>(gW + i * ldg + c) = v;
        }
    } else {
        const long total = (long)(hL + 1) * ncols;
        for (long idx = (long)bid * 256 + threadIdx.x; idx < total; idx += (long)nblk * 256) {
            const long i = idx / ncols, c = idx - i * ncols;
            const bool tl = is_tail(c % q.plane);
            const float* ws = tl ? q.ws2 : q.ws; const int S = tl ? q.S2 : q.S;
            float v = 0.f;
            for (int s = 0; s < S; ++s) v += ws[(long)s * q.stride + i * ldws + c];
            gW[i * ldg + c] = v * sc;
        }
    }
    if (g_theta) {
        for (long c = (long)bid * 256 + threadIdx.x; c < G; c += (long)nblk * 256) {
            const bool tl = is_tail(c);
            const float* ws = tl ? q.ws2 : q.ws; const int S = tl ? q.S2 : q.S; const long stride = q.stride;
            float v = 0.f;
            for (int s = 0; s < S; ++s) v += ws[(long)s * stride + (long)(hL + 1) * ldws + c];
            const float e = expf(theta_w[c]);
            g_theta[c] = (e >= 1e-3f && e <= 1e4f) ? v * sc * e : 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void heads_reduce_dw_kernel(ReduceDwArgs q) { reduce_dw_body(q, blockIdx.x, gridDim.x); }

// dH[row, i] = sum over gene tiles of ws[row tile][gt][row % 32][i]; GL threads split the gene tiles of
// one output quad, combined in fixed order through LDS.
struct ReduceDhArgs {
    const float* ws; int ntg, B, Bpad, KT, hL; float* dH; long lddh; float scale;
    const double* loss_partials; int n_partials; double loss_scale; float* loss_out;
    const float* ws2; int ntg2;        // the tail launch's partials [row tile][ntg2][32][64] (ntg2 = 0: none), added behind the others
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
        if (q.ntg2 > 0) {
            const float4* src2 = reinterpret_cast<const float4*>(q.ws2) + t * q.ntg2 * tq + within;
            for (int gt = gl; gt < q.ntg2; gt += GL) { const float4 x = src2[(long)gt * tq]; v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w; }
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
        if (i + 0 < hL) d[0] = v.x * q.scale;
        if (i + 1 < hL) d[1] = v.y * q.scale;
        if (i + 2 < hL) d[2] = v.z * q.scale;
        if (i + 3 < hL) d[3] = v.w * q.scale;
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


struct HeadsPlan {
    bool small;                          // fewer than 5 row tiles: the four-wave kernel, one workgroup per (gene tile, row tile)
    int HLB, WR, S, NT, ntg, ngb, grid;
    int nitems, npart;                   // work items (S x gene tiles), dH partials per row tile
    long ldws, dw_stride, dw_bytes, dh_bytes, hs_bytes;
    // TAIL launch (persistent kernel only; ntail = 0: none): the gene tiles a uniform plan would leave to a last, poorly filled
    // round of workgroups -- tile_order[nmain ..] -- go to a second launch of the same kernel with S2 > S batch splits, so that
    // the round they cost is a short one (25 000 genes at 4 096 rows: 782 tiles x 2 = 1 564 items = 6.1 rounds ran as 7;
    // now 6 rounds + 14 tiles x 16 splits = 224 one-tile items).  Its partials live behind the main launch's.
    int nmain, ntail, S2, grid2, npart2, nitems2;
    long dh2_bytes, dw2_bytes;
};

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// a pure function of the shape: kernel, batch splits, workspace layout
bool make_heads_plan(int B, int hL, int G, long plane, int flags, HeadsPlan* out) {
    if (flags & (DCAHIP_NLL_POISSON | DCAHIP_NLL_MSE)) return false;     // NB / ZINB family only
    if (B <= 0 || G <= 0 || hL <= 0 || hL > 64 || plane < G || (plane & 3) || plane > ((G + 31) & ~31)) return false;
    if (B > (1 << 22)) return false;                 // H is addressed through a 32-bit buffer resource (B x 64 floats)
    const bool has_pi = flags & DCAHIP_NLL_HAS_PI, cdisp = flags & DCAHIP_NLL_CONST_DISP;
    const int NH = 1 + (cdisp ? 0 : 1) + (has_pi ? 1 : 0);
    HeadsPlan p;
    p.HLB = 2;
    p.NT = (B + kTR - 1) / kTR;
    p.ntg = (G + kTG - 1) / kTG;
    p.ngb = p.ntg;
    if (p.ngb > kMaxGrid) return false;
    // the 8-wave persistent kernel from 5 row tiles on (waves beyond the batch idle): measured against the four-wave kernel
    // at G = 20 000: B = 128 0.088 / 0.086 ms (kept on the four-wave kernel), 160: 0.100 / 0.104, 192: 0.103 / 0.121,
    // 224: 0.107 / 0.142 (profiles/r02z_heads_kernel_switch.txt).  Below: at most 4 x 2 048 workgroups, one per tile pair.
    p.small = p.NT < kWr8MinNT;
    static_assert((kWr8MinNT - 1) * kMaxGrid <= kMaxSmallGrid, "loss partials of the four-wave kernel");
    p.WR = p.small ? 1 : kWR2;
    const int smax = (p.NT + p.WR - 1) / p.WR;
    double best = 1e300;
    p.S = 1;
    const double item_cost = 0.75;                   // per work item: the weight prologue and the reduce (in tile times)
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
    if (p.small) {                                   // one workgroup per (gene tile, row tile): S = NT weight-gradient partials
        p.S = p.NT;
        p.nitems = p.NT * p.ntg;
        p.grid = p.nitems;
        p.npart = p.ntg;
    } else {                                         // persistent: as many workgroups as are resident (one per CU), a multiple of S
        const int res = kCUs / p.S * p.S;
        if (p.grid > res) p.grid = res;
        p.npart = p.grid / p.S;
    }
    p.nmain = p.ngb; p.ntail = 0; p.S2 = 0; p.grid2 = 0; p.npart2 = 0; p.nitems2 = 0;
#ifndef DCA_EXP_NO_TAIL_LAUNCH
    if (!p.small && p.S >= 2) {
        const int full = p.ngb / p.npart, rem = p.ngb - full * p.npart;
        const int tiles = (p.NT + p.S * p.WR - 1) / (p.S * p.WR);
        if (full >= 1 && rem > 0) {
            const double uniform = (double)(full + 1) * (tiles + item_cost);
            double best2 = 1e300; int bestS = 0;
            for (int S2 = 2 * p.S; S2 <= smax && (long)S2 * rem <= kMaxGrid; S2 *= 2) {
                const long items2 = (long)S2 * rem;
                const long rounds2 = (items2 + kCUs - 1) / kCUs;
                const int tiles2 = (p.NT + S2 * p.WR - 1) / (S2 * p.WR);
                const double c2 = (double)rounds2 * (tiles2 + item_cost);
                if (c2 < best2 - 1e-9) { best2 = c2; bestS = S2; }
            }
            // (a second launch: its own prologue and ramp, priced at one tile)
            if (bestS && (double)full * (tiles + item_cost) + best2 + 1.0 < uniform - 0.5) {
                p.nmain = full * p.npart; p.ntail = rem; p.S2 = bestS;
                p.nitems = p.S * p.nmain;                       // the main launch: whole rounds only
                p.nitems2 = p.S2 * p.ntail;
                const int res2 = kCUs / p.S2 * p.S2;
                p.grid2 = p.nitems2 < res2 ? p.nitems2 : res2;
                p.npart2 = p.grid2 / p.S2;
            }
        }
    }
#endif
    p.ldws = (long)NH * plane;
    p.dw_stride = (long)(hL + 2) * p.ldws;
    p.dw_bytes = (long)p.S * p.dw_stride * (long)sizeof(float);
    p.dh_bytes = (long)p.npart * p.NT * kTR * (p.HLB * 32) * (long)sizeof(float);
    p.dw2_bytes = (long)p.S2 * p.dw_stride * (long)sizeof(float);
    p.dh2_bytes = (long)p.npart2 * p.NT * kTR * (p.HLB * 32) * (long)sizeof(float);
    p.hs_bytes = 2L * p.NT * kHTile * 2 + (((long)p.NT * 4 + 15) & ~15L);      // split decoder output, both layouts + the tile exponents
    *out = p;
    return true;
}

long long* g_timing = nullptr;           // (DCA_HEADS_TIMING builds only: per-phase cycle counters)

// =====================================================================================================
// K-HEADS for batches below 160 rows (B <= 32, ONE row tile, is the reference's default batch size, dca/api.py:33):
// one (gene tile, row tile) pair per workgroup, FOUR waves working on it together.  With a lone wave per tile the
// 625 tiles of G = 20 000 leave three quarters of the SIMDs idle and every phase is one wave's dependent chain:
// measured 41 us, of which 10 launch + weight prologue, 13 the likelihood pass, 18 the three products and their stores.  Here
//   F   : wave h < NH computes head h (12 MFMA),
//   Z   : wave w takes the row-slot group w (4 of the 16 slots of each lane half) -- dense pass + its own non-zero queue,
//   dW  : wave h < NH accumulates and stores head h (12 MFMA), while
//   dH  : wave 3 multiplies all heads (36 MFMA) and stores the tile's partial sum,
// all through one weight image and one staging tile in LDS; operands, arithmetic and scales are those of the persistent kernel
// (the row tile's scale from the decoder rows the product waves hold, the gene tile's from the weights, D = g 2^kD0, a tile
// with a value beyond the fp16 range scaled down as a whole); the loss partial of the tile is the sum of its four waves' in
// wave order.
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
    constexpr int W_FLOATS = NH * 2 * W_PIECE / 4;
    constexpr int BIAS_FLOATS = CONST_DISP ? (NH + 1) * 32 : 0;       // conditional dispersion: biases straight from memory
    constexpr int NW = 4;
    static_assert(NW * kZU == 16, "one Z group per wave");
    // a wave queues at most its 4 x 64 elements.  With 256 entries per queue the workgroup's LDS stays below a third
    // of the CU's 160 KB: three workgroups per CU = 768 resident, the 625 gene tiles of G = 20 000 in ONE round
    // (at two per CU the last 113 tiles ran after the first 512: 30 us instead of 17)
    constexpr int QCAP = kZU * 64;
    constexpr int LDS_FLOATS = W_FLOATS + ST_TILE + NW * QCAP + BIAS_FLOATS;
    static_assert(LDS_FLOATS * 4 + NW * 8 + 64 <= 42 * 1280, "three workgroups per CU (LDS is allocated in 1280-byte granules)");
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ double lred[NW];
    __shared__ float wmaxs[NW];
    __shared__ float dmaxs[NW];

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
    const int kD0 = kDExp0 + p.d_exp;

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

    // ---- head weights of the gene tile -> fp16 pieces in LDS (image, rotation and scale rule of the persistent kernel)
    int eW;
    {
        constexpr int UNR = 2 * NH;                  // NH * 64 * 8 sixteen-byte items over 256 threads: one batch of loads
        float4 v[UNR];
        float m = 0.f;
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
            const int c4 = idx & 7, kq = (idx >> 3) & 63;
            const int gcol = g0 + c4 * 4;
            const bool kv = kq < p.hL && gcol <= p.plane - 4;
            if (!kv || gcol + 0 >= p.G) v[u].x = 0.f;
            if (!kv || gcol + 1 >= p.G) v[u].y = 0.f;
            if (!kv || gcol + 2 >= p.G) v[u].z = 0.f;
            if (!kv || gcol + 3 >= p.G) v[u].w = 0.f;
            m = fmaxf(fmaxf(m, fmaxf(fabsf(v[u].x), fabsf(v[u].y))), fmaxf(fabsf(v[u].z), fabsf(v[u].w)));
        }
        m = wave_max(m);
        if (lane == 0) wmaxs[wave] = m;
        __syncthreads();
        eW = __builtin_amdgcn_readfirstlane(block_exp(fmaxf(fmaxf(wmaxs[0], wmaxs[1]), fmaxf(wmaxs[2], wmaxs[3]))));
        const float wsc = pow2i(eW);
#pragma unroll
        for (int u = 0; u < UNR; ++u) {
            const int idx = tid + u * 256;
            const int c4 = idx & 7, kq = (idx >> 3) & 63, h = idx >> 9;
            unsigned a0, a1, b0, b1;
            split_pair(v[u].x * wsc, v[u].y * wsc, a0, a1);
            split_pair(v[u].z * wsc, v[u].w * wsc, b0, b1);
            const int off = kq * 64 + ((((c4 >> 1) + (kq >> 2)) & 3) << 4) + ((c4 & 1) << 3);
            *reinterpret_cast<u32x2*>(Wimg + (h * 2 + 0) * W_PIECE + off) = u32x2{a0, b0};
            *reinterpret_cast<u32x2*>(Wimg + (h * 2 + 1) * W_PIECE + off) = u32x2{a1, b1};
        }
        if (CONST_DISP && tid < 32) Bs[NH * 32 + tid] = (g0 + tid < p.G) ? p.theta_w[g0 + tid] : 0.f;
    }
    // the row tile's scale, from the decoder rows the product waves hold (every product wave holds the whole tile)
    int eHt = 0;
    if (wave < NH) {
        float m = 0.f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const bool ok = row0 + l31 < p.B && 32 * hi + 8 * ks + j < p.hL;
                hx[ks][j] = ok ? hx[ks][j] : 0.f;
                m = fmaxf(m, fabsf(hx[ks][j]));
            }
        eHt = __builtin_amdgcn_readfirstlane(block_exp(wave_max(m)));
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
        const unsigned char* b0 = Wimg + (h * 2 + q) * W_PIECE + ks * 512;
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
        return *reinterpret_cast<const u32x4*>(Wimg + (h * 2 + q) * W_PIECE + jb * 2048 + wdr[gs]);
    };

    // ---- F: wave h computes the pre-activations of head h and stages them [gene][row]
    if (wave < NH) {
        const int h = wave;
        f32x16 acc;
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[e] = 0.f;
        const float hsc = pow2i(eHt);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float x[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = hx[ks][j] * hsc;
            u32x4 af[2];
            split8(x, af);
            u32x4 bf[2] = {w_tr(h, 0, ks), w_tr(h, 1, ks)};
            MFMA_H3(af, bf, acc)
        }
        const float bias = gvalid ? bias_h : 0.f;
        const float funscale = pow2i(-(eHt + eW));
#pragma unroll
        for (int e = 0; e < 16; ++e) St[h * ST_PLANE + l31 * kLdS + rowmap(e, hi)] = fmaf(acc[e], funscale, bias);
    }
    __syncthreads();

    // ---- Z: element-wise likelihood and gradient of this wave's four row slots per lane half; D = g 2^kD0
    double dacc = 0.0;
    {
        const float scs = pow2i(kD0);
        const float thw = CONST_DISP ? Bs[NH * 32 + l31] : 0.f;
        float lacc = 0.f;
        int qn = 0;
        float dmax = 0.f;                            // the lane's largest |D| (matrix-product planes only)
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
            const float sc = valid ? scs : 0.f;
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
            dmax = fmaxf(dmax, fmaxf(fmaxf(fabsf(o_m[j]), CONST_DISP ? 0.f : fabsf(o_d[j])), fabsf(o_p[j])));
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
            if (__ballot(act && (e >> 16) == 0xFFFFu)) {     // (see heads_fused_h2_kernel)
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
            o1 *= scs; o2 *= scs; o3 *= scs;
            dmax = fmaxf(dmax, act ? fmaxf(fmaxf(fabsf(o1), CONST_DISP ? 0.f : fabsf(o2)), fabsf(o3)) : 0.f);
            if (act) {
                St[idx] = o1;
                if (CONST_DISP) St[TH_P * ST_PLANE + idx] = o2; else St[ST_PLANE + idx] = o2;
                if (HAS_PI) St[PI_H * ST_PLANE + idx] = o3;
            }
            qn -= c;
        }
        dacc = (double)lacc;
        dmax = wave_max(dmax);
        if (lane == 0) dmaxs[wave] = dmax;
    }
    __syncthreads();

    // a value beyond the fp16 range somewhere in the tile: the whole tile's D is scaled down by what its maximum needs
    int shift = 0;
    {
        const float m = fmaxf(fmaxf(dmaxs[0], dmaxs[1]), fmaxf(dmaxs[2], dmaxs[3]));
        if (!(m <= kDLim)) {
            const int need = __builtin_amdgcn_frexp_expf(m) - 1 - kTop;
            shift = __builtin_amdgcn_readfirstlane(need > 0 ? (need > 100 ? 100 : need) : 0);
        }
    }
    const float down = pow2i(-shift);

    if (wave == NW - 1) {
        // ---- dH[row, i] = sum_genes D[row, gene] W[i, gene], all heads: the tile's partial sum (in g units)
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
                for (int j = 0; j < 8; ++j) dv[j] = St[h * ST_PLANE + (16 * gs + 8 * hi + j) * kLdS + l31] * down;
                u32x4 af[2];
                split8(dv, af);
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    u32x4 bf[2] = {w_dr(h, 0, jb, gs), w_dr(h, 1, jb, gs)};
                    MFMA_H3(af, bf, dHa[jb])
                }
            }
        const float fu = pow2i(-(kD0 - shift + eW));
#pragma unroll
        for (int jb = 0; jb < 2; ++jb)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float v = dHa[jb][e] * fu;
                dh_out[rowmap(e, hi) * KT + jb * 32 + l31] = v;
            }
        if (CONST_DISP) {                            // ConstantDispersionLayer chain (dca/layers.py:17-21)
            float thsum = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) thsum += St[TH_P * ST_PLANE + l31 * kLdS + rowmap(e, hi)];
            const float tv = (thsum + __shfl_xor(thsum, 32, 64)) * pow2i(-kD0);
            if (p.NT == 1) {
                if (gvalid && hi == 0) {
                    const float ex = expf(p.theta_w[gene]);
                    p.g_theta[gene] = (ex >= 1e-3f && ex <= 1e4f) ? tv * p.inv_n * ex : 0.f;
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
        const float hsc = pow2i(eHt);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            float dv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                dv[j] = St[h * ST_PLANE + l31 * kLdS + rowmap(8 * ks + j, hi)];
                bsum += dv[j];
                dv[j] *= down;
            }
            u32x4 bf[2];
            split8(dv, bf);
#pragma unroll
            for (int ib = 0; ib < 2; ++ib) {
                float x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    x[j] = (row0 + rowmap(8 * ks + j, hi) < p.B && l31 + 32 * ib < p.hL) ? htx[ks][ib][j] * hsc : 0.f;
                u32x4 af[2];
                split8(x, af);
                MFMA_H3(af, bf, dW[ib])
            }
        }
        const bool cw = gene < p.plane;
        // one row tile: straight to the gradient buffer (x 1 / n); several: row tile t's partial in g units, summed (and
        // scaled) by the dW reduce launch
        float* const out = p.NT == 1 ? p.gW : p.ws_dw + (long)t * p.dw_stride;
        const long ldo = p.NT == 1 ? p.ldg : p.ldws;
        const float fin = p.NT == 1 ? p.inv_n : 1.f;
        const float wun = pow2i(-(eHt + kD0 - shift)) * 1.f;
#pragma unroll
        for (int ib = 0; ib < 2; ++ib)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = ib * 32 + rowmap(e, hi);
                const float v = dW[ib][e] * wun * fin;
                if (cw && i < p.hL) out[(long)i * ldo + (long)h * p.plane + gene] = v;
            }
        const float bv = (bsum + __shfl_xor(bsum, 32, 64)) * pow2i(-kD0) * fin;
        if (cw && hi == 0) out[(long)p.hL * ldo + (long)h * p.plane + gene] = bv;
    }

    // ---- loss: wave -> workgroup (wave order) -> one partial per tile
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dacc += __shfl_down(dacc, off, 64);
    if (lane == 0) lred[wave] = dacc;
    __syncthreads();
    if (tid == 0) p.partials[blockIdx.x] = ((lred[0] + lred[1]) + lred[2]) + lred[3];
}

// C [32, 32] = A [32, K] B [K, 32] with the operand scaling, the two-piece fp16 split and the three products of K-HEADS, one
// wave (dcahip_x3_product_32x32: the accuracy contract of the matrix products, tested against fp64)
__global__ __launch_bounds__(64) void x3_product_kernel(const float* A, const float* B, float* C, int K) {
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    float ma = 0.f, mb = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            ma = fmaxf(ma, fabsf(A[(long)l31 * K + k0 + 8 * hi + j]));
            mb = fmaxf(mb, fabsf(B[(long)(k0 + 8 * hi + j) * 32 + l31]));
        }
    const int ea = block_exp(wave_max(ma)), eb = block_exp(wave_max(mb));
    const float sa = pow2i(ea), sb = pow2i(eb);
    f32x16 acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
        float av[8], bv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            av[j] = A[(long)l31 * K + k0 + 8 * hi + j] * sa;
            bv[j] = B[(long)(k0 + 8 * hi + j) * 32 + l31] * sb;
        }
        u32x4 af[2], bf[2];
        split8(av, af);
        split8(bv, bf);
        MFMA_H3(af, bf, acc)
    }
    const float un = pow2i(-(ea + eb));
#pragma unroll
    for (int e = 0; e < 16; ++e) C[rowmap(e, hi) * 32 + l31] = acc[e] * un;
}

template <bool P, bool C, bool YC>
void launch_fused(const HeadsPlan& pl, const HeadsArgs2& a, hipStream_t s) {
    if (pl.small) hipLaunchKernelGGL((heads_fused_small_kernel<P, C, YC>), dim3(pl.grid), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((heads_fused_h2_kernel<P, C, YC>), dim3(pl.grid), dim3(64 * kWR2), 0, s, a);
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
        HeadsPlan q;
        if (!make_heads_plan(b, hL, G, plane, flags, &q)) continue;
        const long n = q.dw_bytes + q.dh_bytes + q.hs_bytes + q.dw2_bytes + q.dh2_bytes;
        if (n > need) need = n;
    }
    return need;
}

extern "C" int dcahip_x3_product_32x32(const float* A, const float* B, float* C, int K, void* stream) {
    if (!A || !B || !C || K <= 0 || (K & 15)) return DCAHIP_EINVAL;
    hipLaunchKernelGGL(x3_product_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), A, B, C, K);
    return (int)hipGetLastError();
}

#ifdef DCA_HEADS_TIMING
extern "C" void dcahip_heads_set_timing(long long* buf) { g_timing = buf; }
#endif

// (rounded up to pairs, the unit in which the engine sorts tiles by cost; the kernels read the first ceil(G / 32))
extern "C" int dcahip_heads_tile_order_len(int G) { return G > 0 ? (((G + kTG - 1) / kTG + kWG - 1) / kWG) * kWG : 0; }

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
                                      loss_partials, n_partials_out, workspace, workspace_bytes, tile_order, loss_out, 0, stream);
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
                                          float* loss_out, int d_exp, void* stream) {
    const bool has_pi = flags & DCAHIP_NLL_HAS_PI, cdisp = flags & DCAHIP_NLL_CONST_DISP;
    HeadsPlan pl;
    if (!make_heads_plan(B, hL, G, plane, flags, &pl)) return DCAHIP_EINVAL;
    if (!H || !Wh || !bh || (!y && !yc) || !sf || !gW || !dH || !loss_partials || !workspace) return DCAHIP_EINVAL;
    if (cdisp && (!theta_w || !g_theta)) return DCAHIP_EINVAL;
    if (workspace_bytes < pl.dw_bytes + pl.dh_bytes + pl.hs_bytes + pl.dw2_bytes + pl.dh2_bytes) return DCAHIP_EINVAL;
    if (!al16(H) || !al16(Wh) || !al16(workspace) || (ldh & 3) || (ldw & 3) || ldh < ((hL + 3) & ~3))
        return DCAHIP_EINVAL;
    // the y = 0 gradients stay below the fp16 range by the likelihood's own bounds (theta <= 1e4, |d / d pi| <= 1 + ridge / 2):
    // a ridge beyond 1e3 is not a regulariser any more, and a scale beyond 2^-24 carries nothing
    if (d_exp > 0 || d_exp < -24 || !(ridge >= 0.f) || ridge > 1e3f) return DCAHIP_EINVAL;
    const int NH = 1 + (cdisp ? 0 : 1) + (has_pi ? 1 : 0);
    if (ldw < (long)NH * plane || ldg < (long)NH * plane) return DCAHIP_EINVAL;
    if (yc ? (ldc < G || ldc > 0xffffffffL) : (ldy < G || ldy > 0xffffffffL)) return DCAHIP_EINVAL;
    float* ws_dh = static_cast<float*>(workspace);
    float* ws_dw = ws_dh + pl.dh_bytes / sizeof(float);
    float* ws_dh2 = ws_dw + pl.dw_bytes / sizeof(float);             // (tail launch: behind the main launch's regions)
    float* ws_dw2 = ws_dh2 + pl.dh2_bytes / sizeof(float);
    hipStream_t s = static_cast<hipStream_t>(stream);
    bool direct_dw = false;
    {
        unsigned short* HA = reinterpret_cast<unsigned short*>(reinterpret_cast<char*>(workspace) + pl.dh_bytes + pl.dw_bytes +
                                                               pl.dh2_bytes + pl.dw2_bytes);
        unsigned short* HT = HA + (long)pl.NT * kHTile;
        int* eH = reinterpret_cast<int*>(HT + (long)pl.NT * kHTile);
        if (!pl.small) {                     // (the four-wave kernel splits its one row tile itself)
            hipLaunchKernelGGL(heads_split_h_kernel, dim3(pl.NT), dim3(256), 0, s, H, ldh, B, hL, HA, HT, eH);
            int rc0 = (int)hipGetLastError();
            if (rc0 != 0) return rc0;
        }
        direct_dw = pl.S == 1;
        HeadsArgs2 a{g_timing, HA, HT, eH, H, ldh, gW, ldg, g_theta, Wh, ldw, bh, theta_w, y, ldy, yc, ldc, ovf_ptr, ovf_col, ovf_val,
                     sf, perm, cursor, ws_dw, pl.dw_stride, ws_dh,
                     pl.npart, pl.nitems, tile_order, loss_partials, plane, pl.ldws, B, hL, G, pl.S, pl.NT, ridge, inv_n, d_exp, 0};
        auto launch = [&](const HeadsPlan& q, const HeadsArgs2& b) {
            if (yc) {
                if (has_pi && cdisp) launch_fused<true, true, true>(q, b, s);
                else if (has_pi) launch_fused<true, false, true>(q, b, s);
                else if (cdisp) launch_fused<false, true, true>(q, b, s);
                else launch_fused<false, false, true>(q, b, s);
            } else {
                if (has_pi && cdisp) launch_fused<true, true, false>(q, b, s);
                else if (has_pi) launch_fused<true, false, false>(q, b, s);
                else if (cdisp) launch_fused<false, true, false>(q, b, s);
                else launch_fused<false, false, false>(q, b, s);
            }
        };
        launch(pl, a);
        if (pl.ntail > 0) {
            // the tail launch: the same kernel over tile_order[nmain ..] with S2 batch splits, its own partial regions, its loss
            // partials behind the main launch's
            HeadsPlan q2 = pl;
            q2.grid = pl.grid2; q2.S = pl.S2;
            HeadsArgs2 b = a;
            b.ws_dw = ws_dw2; b.ws_dh = ws_dh2; b.npart = pl.npart2; b.nitems = pl.nitems2; b.S = pl.S2;
            b.partials = loss_partials + pl.grid; b.tile_base = pl.nmain;
            launch(q2, b);
        }
    }
    int rc = (int)hipGetLastError();
    if (rc != 0) return rc;
    const int n_part = pl.grid + pl.grid2;
    if (n_partials_out) *n_partials_out = n_part;
    {
        const int KT = pl.HLB * 32;
        const long nq = (long)B * (KT / 4);
        // (every partial is a sum of UNSCALED gradients: the reductions apply 1 / n, once per output element)
        const ReduceDhArgs qh{ws_dh, pl.npart, B, pl.NT * kTR, KT, hL, dH, lddh, inv_n, loss_partials, n_part, (double)inv_n, loss_out,
                              ws_dh2, pl.npart2};
        const long total = (long)(hL + 1) * pl.ldws;
        long gr = (total / 4 + 255) / 256;                  // the weight-gradient reduction moves 16 bytes per lane
        if (gr > 2048) gr = 2048;
        if (gr < 1) gr = 1;
        const ReduceDwArgs qw{ws_dw, pl.S, pl.dw_stride, hL, pl.ldws, pl.ldws, gW, ldg, cdisp ? theta_w : nullptr,
                              cdisp ? g_theta : nullptr, G, inv_n,
                              ws_dw2, pl.S2, (pl.ntail > 0 && tile_order) ? tile_order + pl.nmain : nullptr, pl.ntail, plane, pl.nmain};
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
