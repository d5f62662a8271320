// K-HEADS: the output heads of the autoencoder as ONE kernel -- forward GEMM of the three
// Dense heads, NB / ZINB negative log-likelihood + gradient, weight/bias gradient and input
// gradient -- so that the [cells x genes] pre-activation and gradient planes never exist in
// HBM.  gfx950 (MI355X), wave64, fp32 MFMA (v_mfma_f32_32x32x2_f32, exact fp32).
//
// Reference path replaced: dca/network.py:369-385 (pi / dispersion / mean Dense heads,
// ColwiseMultLayer, SliceLayer, ZINB loss closure), dca/network.py:38-39, dca/layers.py:21,85,
// dca/loss.py:72-156 and TensorFlow's autodiff of all of it.
//
// Work decomposition (gene-stationary):
//   * a wave owns one 32-gene tile of every head and a strided set of 32-row batch tiles;
//     a workgroup = kWG gene tiles x WR row slots; the [hL x 32 genes x heads] slice of the
//     head weights sits in LDS for the lifetime of the workgroup, the wave's weight-gradient
//     slice ([hL x 32] per head) in MFMA accumulators for the lifetime of the wave.
//   * per (row tile, gene tile):
//       F   pre-activations  A = H W + b        32 rows x 32 genes x heads, K = hL
//       Z   NLL + d NLL / d A   element-wise, through a wave-private LDS staging tile
//       Bk  dW += H^T D   (D read from staging in exactly the MFMA B-operand layout)
//           dH  = D W^T   (D read transposed from the same staging tile) -> partial per gene tile
//     No workgroup barrier inside the loop: waves drift apart, so one wave's transcendental
//     (VALU) phase overlaps its SIMD partner's MFMA phases.
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
constexpr int kMaxGrid = 2048; // = dcahip_zinb_max_partials()
constexpr int kCUs = 256;
constexpr int kZU = 4;        // staged rows per Z group (two groups per loop iteration)
constexpr int kQCap = 320;     // non-zero queue entries per wave (< 64 left over + 4 x 64 pushed)
constexpr int kLdH = 65;       // row stride of the H tile parked in the staging buffer

#ifdef DCA_EXP_NOMFMA
#define MFMA(a, b, c) (c)
#else
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)
#endif

#ifdef DCA_HEADS_TIMING
#define TSTAMP(i) { const long long now_ = __builtin_readcyclecounter(); tacc[i] += now_ - tlast; tlast = now_; }
#else
#define TSTAMP(i)
#endif

struct HeadsArgs {
    long long* timing;                // debug builds only: per-wave phase cycle sums
    const float* H;  long ldh;
    const float* Wh; long ldw;
    const float* bh;
    const float* theta_w;
    const float* y;  long ldy;
    const float* sf;
    const int* perm;
    const long long* cursor;
    float* ws_dw;  long dw_stride;   // [S][(hL + 2)][ldws]
    float* ws_dh;                     // [NT][ntg][32][KT]: the partials of one row tile are contiguous
    int ntg;
    const int* tile_order;            // gene tiles in the order workgroups take them (kWG consecutive entries each), or NULL
    double* partials;
    long plane, ldws;
    int B, hL, G;
    int S, NT;
    float ridge, inv_n;
};

__device__ __forceinline__ int rowmap(int e, int hi) { return (e & 3) + 8 * (e >> 2) + 4 * hi; }

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// FULLK: hL == 32 * HLB exactly and H rows densely packed (ldh == hL; the default 64-wide decoder): no k /
// hidden-unit guards at all and a compile-time row stride (H loads are one base + immediate offsets), which
// also keeps dozens of loop-invariant clamped offsets, address pairs and predicates out of the register file.
template <bool HAS_PI, bool CONST_DISP, int HLB, int WR, bool FULLK>
__global__ __launch_bounds__(64 * kWG * WR) void heads_fused_kernel(HeadsArgs p) {
    constexpr int NH = 1 + (CONST_DISP ? 0 : 1) + (HAS_PI ? 1 : 0);
    constexpr int PI_H = NH - 1;                 // plane of the pi head (when present)
    constexpr int KT = HLB * 32;                 // padded hidden width
    constexpr int KH = HLB * 16;                 // k per half-wave in the forward
    constexpr int WS_TILE = NH * KT * kLdS;      // floats of one gene tile's weights
    constexpr int ST_PLANE = kTG * kLdS;
    constexpr int NP = NH + (CONST_DISP ? 1 : 0);  // staging planes (+ d nll / d theta for const-disp)
    constexpr int TH_P = NH;                        // that extra plane
    constexpr int ST_WAVE = NP * ST_PLANE > kTR * kLdH ? NP * ST_PLANE : kTR * kLdH;
    constexpr int NTHREADS = 64 * kWG * WR;
    constexpr int NRED = NH * HLB * 16 + NH + 1; // registers a wave hands over in the dW reduce
    constexpr int LDS_FLOATS = kWG * WS_TILE + kWG * WR * (ST_WAVE + kQCap);
    static_assert(WR == 1 || (kWG * WR / 2) * NRED * 64 <= LDS_FLOATS, "dW reduce scratch");
    static_assert(kTR * kLdH <= ST_WAVE, "H tile must fit the staging buffer");
    __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
    __shared__ double lred[kWG * WR];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: tile indices live in SGPRs
#ifdef DCA_HEADS_TIMING
    const long long t_entry = __builtin_readcyclecounter();
    long long t_loop0 = t_entry, t_loop1 = t_entry;
#endif
    const int l31 = lane & 31, hi = lane >> 5;
    const int g = wave / WR, r = wave % WR;
    const int s = blockIdx.x % p.S, gb = blockIdx.x / p.S;
    // which gene tiles share a workgroup is free (every result is per gene tile): the caller can pair tiles of
    // similar non-zero load, a workgroup lasts as long as its slower tile
    const int gt = p.tile_order ? p.tile_order[gb * kWG + g] : gb * kWG + g;
    const int g0 = gt * kTG;
    const int gene = g0 + l31;
    const bool tile_ok = g0 < p.G;
    const bool gvalid = gene < p.G;
    const long long cur = p.cursor ? *p.cursor : 0;

    float* Ws = lds;
    float* Wsg = Ws + g * WS_TILE;
    float* St = lds + kWG * WS_TILE + wave * ST_WAVE;
    unsigned* Q = reinterpret_cast<unsigned*>(lds + kWG * WS_TILE + kWG * WR * ST_WAVE) + wave * kQCap;

    // ---- head weights of this workgroup's genes -> LDS, [tile][head][k][33]
    {
        constexpr int NT4 = kWG * NH * KT * (kTG / 4);
        for (int idx = tid; idx < NT4; idx += NTHREADS) {
            const int c4 = idx & 7;
            int rest = idx >> 3;
            const int k = rest % KT; rest /= KT;
            const int h = rest % NH;
            const int gg = rest / NH;
            const int gtile = p.tile_order ? p.tile_order[gb * kWG + gg] : gb * kWG + gg;
            const int gcol = gtile * kTG + c4 * 4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (k < p.hL && gcol < p.plane)
                v = *reinterpret_cast<const float4*>(p.Wh + (long)k * p.ldw + (long)h * p.plane + gcol);
            float* d = Ws + ((gg * NH + h) * KT + k) * kLdS + c4 * 4;
            d[0] = gcol + 0 < p.G ? v.x : 0.f;
            d[1] = gcol + 1 < p.G ? v.y : 0.f;
            d[2] = gcol + 2 < p.G ? v.z : 0.f;
            d[3] = gcol + 3 < p.G ? v.w : 0.f;
        }
    }
    __syncthreads();

    f32x16 dW[NH][HLB];
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
        for (int ib = 0; ib < HLB; ++ib)
#pragma unroll
            for (int e = 0; e < 16; ++e) dW[h][ib][e] = 0.f;
    float bsum[NH];
#pragma unroll
    for (int h = 0; h < NH; ++h) bsum[h] = 0.f;
    float thsum = 0.f;
    double dacc = 0.0;

    if (tile_ok) {
        float bias[NH];
#pragma unroll
        for (int h = 0; h < NH; ++h) bias[h] = gvalid ? p.bh[(long)h * p.plane + gene] : 0.f;
        const float thw = (CONST_DISP && gvalid) ? p.theta_w[gene] : 0.f;

        const int hl4 = (p.hL + 3) & ~3;
        const long LDH = FULLK ? (long)KT : p.ldh;
        const int gene_c = gvalid ? gene : p.G - 1;         // clamped: loads stay unconditional
        // count loads: storage row (non-negative) x row stride as ONE 32 x 32 -> 64-bit multiply-add instead of
        // the sign-extended 64 x 64 product (three quarter-rate multiplies per load); the plan checks ldy < 2^32
        const float* const ycol = p.y + gene_c;
        const unsigned ldy_u = (unsigned)p.ldy;
        // Software pipeline across tiles: the row indices (perm), size factors, H rows and the
        // first count groups of tile t+1 are requested while tile t is in its Z / Bk phases.
        const int tstep = p.S * WR;
        int t = s * WR + r;
        const long dh_tstride = (long)p.ntg * (kTR * KT);
        float* const dh_base = p.ws_dh + (long)gt * (kTR * KT) + l31;
        int srow_l = 0;
        float sf_l = 1.f;
        float4 hv[KH / 4];
        float yA[kZU], yB[kZU];
        auto row_clamped = [&](int tt) { const int rl = tt * kTR + l31; return rl < p.B ? rl : p.B - 1; };
        auto load_srow = [&](int tt) { const int rlc = row_clamped(tt); return p.perm ? p.perm[cur + rlc] : (int)(cur + rlc); };
        // FULLK: H through a buffer resource ([B x KT] floats): every load is (per-lane offset fixed for the whole
        // kernel) + (wave-uniform tile offset in an SGPR) + immediate, and rows beyond B read as 0 in hardware --
        // no clamps, no selects, no 64-bit address pairs in the register file
        const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.H), 0, FULLK ? p.B * KT * 4 : 0, 0x00020000);
        const int hv_lane = (l31 * KT + hi * KH) * 4;            // bytes: row l31 of a tile, this lane half's k range
        const int hd_lane = (4 * hi * KT + l31) * 4;             // bytes: row 4 hi of a tile, hidden unit l31
        auto load_hv = [&](int tt) {
            if (FULLK) {
                const int so = tt * (kTR * KT * 4);
#pragma unroll
                for (int c = 0; c < KH / 4; ++c) {
                    const auto w = __builtin_amdgcn_raw_buffer_load_b128(hrs, hv_lane + 16 * c, so, 0);
                    static_assert(sizeof(w) == 16, "128-bit buffer load");
                    hv[c] = __builtin_bit_cast(float4, w);
                }
                return;
            }
            const float* hp = p.H + (long)row_clamped(tt) * LDH;
#pragma unroll
            for (int c = 0; c < KH / 4; ++c) {
                const int k = hi * KH + 4 * c;
                const int kc = (FULLK || k < hl4) ? k : hl4 - 4;
                hv[c] = *reinterpret_cast<const float4*>(hp + kc);
            }
        };
        if (t < p.NT) {
            srow_l = load_srow(t);
            sf_l = p.sf[srow_l];
            load_hv(t);
#pragma unroll
            for (int j = 0; j < kZU; ++j) {
                const int sr = __shfl(srow_l, rowmap(j, hi), 64);
                yA[j] = ycol[(unsigned long long)(unsigned)sr * ldy_u];
            }
        }
#ifdef DCA_HEADS_TIMING
        long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        long long tlast = __builtin_readcyclecounter();
        t_loop0 = tlast;
#endif
        // Two waves share a SIMD (wave w and w + 4 of the workgroup) and one of them runs ~15 % ahead of the other
        // (issue arbitration by priority, then age); the workgroup then waits for the slower half at the end.
        // Alternating the priority per tile, in anti-phase between the halves, treats them alike.
        int tile_no = wave >> 2;
        for (; t < p.NT; t += tstep) {
            if (kWG * WR == 8) { if ((tile_no++) & 1) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(0); }
            TSTAMP(0)
            const int row0 = t * kTR;
            const bool rv = row0 + l31 < p.B;
            const int tn = t + tstep < p.NT ? t + tstep : t;     // next tile (or a harmless re-read)
            // ---- this tile's rows of H -> the (still unused) staging tile, [row][k] with an odd
            // row stride: the forward's A operands are then plain conflict-free ds_reads
#pragma unroll
            for (int c = 0; c < KH / 4; ++c) {
                const int k = hi * KH + 4 * c;
                float* d = St + l31 * kLdH + k;
                d[0] = (FULLK || (rv && k + 0 < p.hL)) ? hv[c].x : 0.f;      // FULLK: rows beyond B were loaded as 0
                d[1] = (FULLK || (rv && k + 1 < p.hL)) ? hv[c].y : 0.f;
                d[2] = (FULLK || (rv && k + 2 < p.hL)) ? hv[c].z : 0.f;
                d[3] = (FULLK || (rv && k + 3 < p.hL)) ? hv[c].w : 0.f;
            }
            wave_sync();
            TSTAMP(1)
            // ---- F: pre-activations.  Lane half hi covers k in [hi*KH, hi*KH+KH): the MFMA's
            // k order is free as long as A and B agree.
            f32x16 acc[NH];
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[h][e] = 0.f;    // bias joins at the staging store
#pragma unroll
            for (int kk = 0; kk < KH; ++kk) {
                const float a = St[l31 * kLdH + hi * KH + kk];
#pragma unroll
                for (int h = 0; h < NH; ++h) {
                    const float b = Wsg[(h * KT + hi * KH + kk) * kLdS + l31];
                    acc[h] = MFMA(a, b, acc[h]);
                }
            }
            wave_sync();
            TSTAMP(2)
            // ---- stage [gene][row] (row stride 1, gene stride 33)
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    St[h * ST_PLANE + l31 * kLdS + rowmap(e, hi)] = acc[h][e] + bias[h];
            const int srow_n = load_srow(tn);       // in flight during the Z loop
            wave_sync();
            TSTAMP(3)

            // ---- Z: element-wise likelihood and gradient.  16 staged rows = 4 groups of 4; the
            // counts of group g+1 are in flight while group g is evaluated (two named buffers,
            // no register rotation: a rotation would have to wait for the load it just issued).
            // The last request already belongs to the next tile.
            // Counts are ~93 % zeros: the dense pass evaluates the y = 0 formulas for every
            // element and queues the positions of the non-zero ones; the queue is drained 64
            // entries at a time by the NB branch (lgamma / digamma differences), which therefore
            // runs ~2x per tile instead of 16x with 5 % of its lanes alive.
            float lacc = 0.f;
            int qn = 0;
            auto z_dense = [&](auto fullv, int grp, const float (&yv)[kZU]) {
                constexpr bool FULLV = decltype(fullv)::value;      // interior tile: every row and gene of it exists
                // (1) all staged inputs of the group first: the kZU element chains below are then
                // independent (no LDS store between their loads) and interleave
                float i_am[kZU], i_ad[kZU], i_ap[kZU];
#pragma unroll
                for (int j = 0; j < kZU; ++j) {
                    const int row = rowmap(grp * kZU + j, hi);
                    const int idx = l31 * kLdS + row;
                    i_am[j] = St[idx];
                    i_ad[j] = CONST_DISP ? thw : St[ST_PLANE + idx];
                    i_ap[j] = HAS_PI ? St[PI_H * ST_PLANE + idx] : 0.f;
                }
                // (2) arithmetic
                float o_m[kZU], o_d[kZU], o_p[kZU];
                bool o_nz[kZU];
#pragma unroll
                for (int j = 0; j < kZU; ++j) {
                    const int row = rowmap(grp * kZU + j, hi);
                    const bool valid = FULLV || ((row0 + row < p.B) && gvalid);
                    const float yj = yv[j];
#if defined(DCA_EXP_NOZ)
                    const bool nz = false;
                    lacc += valid ? yj : 0.f;
                    o_m[j] = i_am[j] * i_sf[j]; o_d[j] = i_ad[j]; o_p[j] = i_ap[j];
                    o_nz[j] = nz;
#else
#if defined(DCA_EXP_NOSPARSE)
                    const bool nz = false;
#else
                    const bool nz = valid && (HAS_PI ? !(yj < kZeroThresh) : (yj != 0.f));
#endif
                    if (HAS_PI) {
                        // the y = 0 formulas for every element (the non-zero ones are redone by the sparse pass)
                        float gmv, gdv, gpv;
                        const float nll = zinb_zero_elem<CONST_DISP>(i_am[j], i_ad[j], i_ap[j], __shfl(sf_l, row, 64), p.ridge, gmv, gdv, gpv);
                        const float sc = valid ? p.inv_n : 0.f;      // pre-activations of padding are finite
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
#endif
                }
                // (3) results / queue
#pragma unroll
                for (int j = 0; j < kZU; ++j) {
                    const int row = rowmap(grp * kZU + j, hi);
                    const int idx = l31 * kLdS + row;
                    const bool nz = o_nz[j];
                    const unsigned long long m = __ballot(nz);
                    const int slot = qn + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    if (nz) {
                        // entry = staging index | count as uint16 (0xFFFF: not representable --
                        // re-read from memory by the sparse pass)
                        const float yj = yv[j];
                        const unsigned y16 = (yj < 65535.f && yj == floorf(yj)) ? (unsigned)yj : 0xFFFFu;
                        Q[slot] = (unsigned)idx | (y16 << 16);
                    } else {                 // pre-activations of queued elements stay in place
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
                const float ad = CONST_DISP ? __shfl(thw, gq, 64) : St[ST_PLANE + idx];
                const float ap = HAS_PI ? St[PI_H * ST_PLANE + idx] : 0.f;
                float yq = (float)(e >> 16);
                if ((e >> 16) == 0xFFFFu) yq = p.y[(long)sr * p.ldy + g0 + gq];
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
            };
            auto z_flush = [&](bool last) {
                while (qn >= 64 || (last && qn > 0)) {
                    const int c = qn < 64 ? qn : 64;
                    wave_sync();
                    z_sparse(qn - c, c);
                    qn -= c;
                }
            };
            auto load_y = [&](int srow_src, int grp, float (&yv)[kZU]) {
#pragma unroll
                for (int j = 0; j < kZU; ++j) {
                    const int sr = __shfl(srow_src, rowmap(grp * kZU + j, hi), 64);
                    yv[j] = ycol[(unsigned long long)(unsigned)sr * ldy_u];
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
                    else load_y(srow_n, 0, yA);              // next tile's first group
                    z_dense(fullv, 2 * it + 1, yB);
                    z_flush(last);
                }
            };
            // wave-uniform: all but the last row tile / gene tile take the path without validity selects
            if (row0 + kTR <= p.B && g0 + kTG <= p.G) z_loop(std::true_type{}); else z_loop(std::false_type{});
            dacc += (double)lacc;
            // next tile: size factors; this tile: A operands of the weight-gradient
            // product (H rows, lanes along the hidden units) -- all in flight during the dH MFMAs
            const float sf_n = p.sf[srow_n];
            wave_sync();
            TSTAMP(4)

            // ---- Bk (1): dH[row, i] = sum_genes D[row, gene] W[i, gene] (this gene tile's share;
            // D read transposed from the staging tile)
            {
                f32x16 dHa[HLB];
#pragma unroll
                for (int jb = 0; jb < HLB; ++jb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) dHa[jb][e] = 0.f;
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int kk = 0; kk < 16; ++kk) {
                        const int gl = 16 * hi + kk;
                        const float a = St[h * ST_PLANE + gl * kLdS + l31];
#pragma unroll
                        for (int jb = 0; jb < HLB; ++jb) {
                            const float b = Wsg[(h * KT + jb * 32 + l31) * kLdS + gl];
                            dHa[jb] = MFMA(a, b, dHa[jb]);
                        }
                    }
                TSTAMP(7)                              // timing build: MFMA part of the dH phase ends here
                float* dst = dh_base + (long)t * dh_tstride;
#pragma unroll
                for (int jb = 0; jb < HLB; ++jb)
#pragma unroll
                    for (int e = 0; e < 16; ++e) dst[rowmap(e, hi) * KT + jb * 32] = dHa[jb][e];
            }
            float Hd[HLB][16];
            if (FULLK) {
                const int so = row0 * (KT * 4);
#pragma unroll
                for (int ib = 0; ib < HLB; ++ib)
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        Hd[ib][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                            hrs, hd_lane + (rowmap(e, 0) * KT + ib * 32) * 4, so, 0));
            } else {
#pragma unroll
                for (int ib = 0; ib < HLB; ++ib)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = row0 + rowmap(e, hi);
                        const int i = ib * 32 + l31;
                        const int rc = row < p.B ? row : p.B - 1;
                        const int ic = i < hl4 ? i : hl4 - 1;
                        Hd[ib][e] = p.H[(long)rc * LDH + ic];
                    }
            }
            load_hv(tn);                           // next tile's H rows: in flight during the dW MFMAs
            TSTAMP(5)
            // ---- Bk (2): dW[i, gene] += sum_rows H[row, i] D[row, gene]; the staged D column
            // of a lane IS its B operand (k slot = lane half), A = H rows of this tile
#pragma unroll
            for (int h = 0; h < NH; ++h)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float b = St[h * ST_PLANE + l31 * kLdS + rowmap(e, hi)];
                    bsum[h] += b;                    // bias gradient = column sum of D
#pragma unroll
                    for (int ib = 0; ib < HLB; ++ib) {
                        // FULLK: no select -- rows beyond B were loaded as 0 (buffer bounds) and their staged D
                        // is exactly 0 as well
                        const bool ok = FULLK || ((row0 + rowmap(e, hi) < p.B) && (ib * 32 + l31 < p.hL));
                        dW[h][ib] = MFMA(ok ? Hd[ib][e] : 0.f, b, dW[h][ib]);
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
            for (int i = 0; i < 8; ++i) p.timing[((long)blockIdx.x * (kWG * WR) + wave) * 10 + i] = tacc[i];
#endif
    }

    // ---- loss: wave -> workgroup -> one partial per workgroup
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) dacc += __shfl_down(dacc, off, 64);
    if (lane == 0) lred[wave] = dacc;
    __syncthreads();                      // also: every wave is done with the LDS weights / staging
    if (tid == 0) {
        double v = 0.0;
        for (int w = 0; w < kWG * WR; ++w) v += lred[w];
        p.partials[blockIdx.x] = v;
    }

    // ---- dW / bias-gradient sums of the WR row slots of a gene tile: ordered tree through LDS
    if (WR > 1) {
        float* red = lds;
#pragma unroll
        for (int step = 1; step < WR; step *= 2) {
            const int slot = g * (WR / 2) + r / (2 * step);
            float* rs = red + (long)slot * NRED * 64 + lane;
            if (r % (2 * step) == step) {
                int n = 0;
#pragma unroll
                for (int h = 0; h < NH; ++h)
#pragma unroll
                    for (int ib = 0; ib < HLB; ++ib)
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
                    for (int ib = 0; ib < HLB; ++ib)
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
        float* out = p.ws_dw + (long)s * p.dw_stride;
        const bool cw = gene < p.plane;
#pragma unroll
        for (int h = 0; h < NH; ++h) {
#pragma unroll
            for (int ib = 0; ib < HLB; ++ib)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = ib * 32 + rowmap(e, hi);
                    if (cw && (FULLK || i < p.hL)) out[(long)i * p.ldws + (long)h * p.plane + gene] = dW[h][ib][e];
                }
            const float bv = bsum[h] + __shfl_xor(bsum[h], 32, 64);
            if (cw && hi == 0) out[(long)p.hL * p.ldws + (long)h * p.plane + gene] = bv;
        }
        if (CONST_DISP) {
            const float tv = thsum + __shfl_xor(thsum, 32, 64);
            if (cw && hi == 0) out[(long)(p.hL + 1) * p.ldws + gene] = tv;
        }
    }
#ifdef DCA_HEADS_TIMING
    if (p.timing && lane == 0) {
        long long* tp = p.timing + ((long)blockIdx.x * (kWG * WR) + wave) * 10;
        tp[8] = t_loop0 - t_entry;                                   // prologue: weights -> LDS, first requests
        tp[9] = (long long)__builtin_readcyclecounter() - t_loop1;   // epilogue: loss, dW tree, partial stores issued
    }
#endif
}

// gW[i, col] = sum_s ws[s][i][col], i = 0..hL (row hL = bias gradient), then the
// ConstantDispersionLayer chain (dca/layers.py:17-21) on the per-gene theta sums.
__global__ __launch_bounds__(256) void heads_reduce_dw_kernel(const float* ws, int S, long stride,
                                                              int hL, long ldws, long ncols,
                                                              float* gW, long ldg,
                                                              const float* theta_w, float* g_theta,
                                                              int G) {
    const long total = (long)(hL + 1) * ncols;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const long i = idx / ncols, c = idx - i * ncols;
        float v = 0.f;
        for (int s = 0; s < S; ++s) v += ws[(long)s * stride + i * ldws + c];
        gW[i * ldg + c] = v;
    }
    if (g_theta) {
        for (long c = (long)blockIdx.x * 256 + threadIdx.x; c < G; c += (long)gridDim.x * 256) {
            float v = 0.f;
            for (int s = 0; s < S; ++s) v += ws[(long)s * stride + (long)(hL + 1) * ldws + c];
            const float e = expf(theta_w[c]);
            g_theta[c] = (e >= 1e-3f && e <= 1e4f) ? v * e : 0.f;
        }
    }
}

// dH[row, i] = sum over gene tiles of ws[row tile][gt][row % 32][i]; GL threads split the gene tiles of
// one output quad, combined in fixed order through LDS.
template <int GL>
__global__ __launch_bounds__(256) void heads_reduce_dh_kernel(const float* ws, int ntg, int B, int Bpad,
                                                              int KT, int hL, float* dH, long lddh) {
    constexpr int OUT = 256 / GL;
    __shared__ float4 red[256];
    const int o = threadIdx.x % OUT, gl = threadIdx.x / OUT;
    const int q4 = KT / 4;
    const long nq = (long)B * q4;
    const long quad = (long)blockIdx.x * OUT + o;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (quad < nq) {
        const long tq = (long)kTR * q4;                      // quads of one (row tile, gene tile) partial
        const long t = quad / tq, within = quad - t * tq;
        const float4* src = reinterpret_cast<const float4*>(ws) + t * ntg * tq + within;
#pragma unroll 4
        for (int gt = gl; gt < ntg; gt += GL) {
            const float4 x = src[(long)gt * tq];
            v.x += x.x; v.y += x.y; v.z += x.z; v.w += x.w;
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
}

struct HeadsPlan {
    int HLB, WR, S, NT, ntg, ngb, grid;
    long ldws, dw_stride, dw_bytes, dh_bytes;
};

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// a pure function of the shape: row slots per workgroup, batch splits, workspace layout
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
    p.ngb = (p.ntg + kWG - 1) / kWG;
    if (p.ngb > kMaxGrid) return false;
    p.WR = p.NT >= 4 ? 4 : 1;
    const int smax = (p.NT + p.WR - 1) / p.WR;
    double best = 1e300;
    p.S = 1;
    for (int S = 1; S <= smax && (long)S * p.ngb <= kMaxGrid; ++S) {
        const long items = (long)S * p.ngb;
        const long rounds = (items + kCUs - 1) / kCUs;
        const int tiles = (p.NT + S * p.WR - 1) / (S * p.WR);
        const double cost = (double)rounds * (tiles + 0.75);
        if (cost < best - 1e-9) { best = cost; p.S = S; }
    }
    p.grid = p.S * p.ngb;
    p.ldws = (long)NH * plane;
    p.dw_stride = (long)(hL + 2) * p.ldws;
    p.dw_bytes = (long)p.S * p.dw_stride * (long)sizeof(float);
    p.dh_bytes = (long)p.ntg * p.NT * kTR * (p.HLB * 32) * (long)sizeof(float);
    *out = p;
    return true;
}

long long* g_timing = nullptr;

template <bool P, bool C>
void launch_fused(const HeadsPlan& pl, const HeadsArgs& a, hipStream_t s) {
    const bool full = a.hL == 32 * pl.HLB && a.ldh == 32 * pl.HLB;
#define DCA_LF(WRV, FK) hipLaunchKernelGGL((heads_fused_kernel<P, C, 2, WRV, FK>), dim3(pl.grid), dim3(64 * kWG * WRV), 0, s, a)
    if (pl.WR == 4) { if (full) DCA_LF(4, true); else DCA_LF(4, false); }
    else            { if (full) DCA_LF(1, true); else DCA_LF(1, false); }
#undef DCA_LF
}

}  // namespace

// sufficient for every batch of at most B rows (the plan of a smaller batch may split more)
extern "C" long dcahip_heads_fused_workspace_bytes(int B, int hL, int G, long plane, int flags) {
    HeadsPlan p;
    if (!make_heads_plan(B, hL, G, plane, flags, &p)) return 0;
    long smax = kMaxGrid / p.ngb;
    if (smax > p.NT) smax = p.NT;
    if (smax < 1) smax = 1;
    return smax * p.dw_stride * (long)sizeof(float) + p.dh_bytes;
}

#ifdef DCA_HEADS_TIMING
extern "C" void dcahip_heads_set_timing(long long* buf) { g_timing = buf; }
#endif

extern "C" int dcahip_heads_tile_order_len(int G) { return G > 0 ? (((G + kTG - 1) / kTG + kWG - 1) / kWG) * kWG : 0; }

extern "C" int dcahip_heads_fused_ordered(const float* H, long ldh, const float* Wh, long ldw,
                                          const float* bh, long plane, const float* theta_w,
                                          const float* y, long ldy, const float* sf, const int* perm,
                                          const long long* cursor, int B, int hL, int G, float ridge,
                                          float inv_n, int flags, float* gW, long ldg, float* g_theta,
                                          float* dH, long lddh, double* loss_partials, int* n_partials_out,
                                          void* workspace, long workspace_bytes, const int* tile_order,
                                          void* stream);

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
    const bool has_pi = flags & DCAHIP_NLL_HAS_PI, cdisp = flags & DCAHIP_NLL_CONST_DISP;
    HeadsPlan pl;
    if (!make_heads_plan(B, hL, G, plane, flags, &pl)) return DCAHIP_EINVAL;
    if (!H || !Wh || !bh || !y || !sf || !gW || !dH || !loss_partials || !workspace) return DCAHIP_EINVAL;
    if (cdisp && (!theta_w || !g_theta)) return DCAHIP_EINVAL;
    if (workspace_bytes < pl.dw_bytes + pl.dh_bytes) return DCAHIP_EINVAL;
    if (!al16(H) || !al16(Wh) || !al16(workspace) || (ldh & 3) || (ldw & 3) || ldh < ((hL + 3) & ~3))
        return DCAHIP_EINVAL;
    const int NH = 1 + (cdisp ? 0 : 1) + (has_pi ? 1 : 0);
    if (ldw < (long)NH * plane || ldg < (long)NH * plane) return DCAHIP_EINVAL;
    if (ldy < G || ldy > 0xffffffffL) return DCAHIP_EINVAL;
    float* ws_dh = static_cast<float*>(workspace);
    float* ws_dw = ws_dh + pl.dh_bytes / sizeof(float);
    HeadsArgs a{g_timing, H, ldh, Wh, ldw, bh, theta_w, y, ldy, sf, perm, cursor, ws_dw, pl.dw_stride, ws_dh, pl.ntg,
                tile_order, loss_partials, plane, pl.ldws, B, hL, G, pl.S, pl.NT, ridge, inv_n};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (has_pi && cdisp) launch_fused<true, true>(pl, a, s);
    else if (has_pi) launch_fused<true, false>(pl, a, s);
    else if (cdisp) launch_fused<false, true>(pl, a, s);
    else launch_fused<false, false>(pl, a, s);
    int rc = (int)hipGetLastError();
    if (rc != 0) return rc;
    if (n_partials_out) *n_partials_out = pl.grid;
    {
        const long total = (long)(hL + 1) * pl.ldws;
        long gr = (total + 255) / 256;
        if (gr > 2048) gr = 2048;
        hipLaunchKernelGGL(heads_reduce_dw_kernel, dim3((int)gr), dim3(256), 0, s, ws_dw, pl.S,
                           pl.dw_stride, hL, pl.ldws, pl.ldws, gW, ldg, cdisp ? theta_w : nullptr,
                           cdisp ? g_theta : nullptr, G);
        rc = (int)hipGetLastError();
        if (rc != 0) return rc;
    }
    {
        const int KT = pl.HLB * 32;
        const long nq = (long)B * (KT / 4);
        if (nq >= 64L * 512) {
            hipLaunchKernelGGL(heads_reduce_dh_kernel<4>, dim3((int)((nq + 63) / 64)), dim3(256), 0, s,
                               ws_dh, pl.ntg, B, pl.NT * kTR, KT, hL, dH, lddh);
        } else {
            hipLaunchKernelGGL(heads_reduce_dh_kernel<16>, dim3((int)((nq + 15) / 16)), dim3(256), 0, s,
                               ws_dh, pl.ntg, B, pl.NT * kTR, KT, hL, dH, lddh);
        }
        rc = (int)hipGetLastError();
    }
    return rc;
}
