// Small-tensor kernels of the training step: batch-norm statistics / apply / backward (+ReLU),
// column sums, clipvalue + RMSprop.  gfx950, wave64.
//
// Activations here are [B, H] with H = hidden width (64/32/64 by default): every kernel maps
// 64 consecutive columns to the 64 lanes of a wave (coalesced 256-byte row segments) and the 4
// waves of a workgroup to 4 interleaved row lanes, reduced through LDS.  All reductions have a
// fixed order (deterministic), statistics are merged with Chan's parallel formula so that
// data-parallel ranks can exchange (count, mean, M2) triples instead of raw sums.
//
// Reference semantics restated: keras BatchNormalization(center=True, scale=False), momentum
// .99, eps 1e-3, biased batch variance (dca/network.py:127-128); Activation('relu')
// (network.py:132-135); opt.RMSprop(clipvalue) (dca/train.py:54-57);
// ConstantDispersionLayer gradient (dca/layers.py:17-21).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "dcahip.h"

namespace {

constexpr int kMaxChunks = 256;
constexpr int kApplyRows = 16;   // rows per workgroup in the apply kernels (256 workgroups at B = 4096)

// Activation codes (the `act` / `relu` arguments of the C ABI): keras.activations names accepted by
// Activation(self.activation) and LeakyReLU (alpha 0.3), dca/network.py:132-135.
//   0 linear  1 relu  2 tanh  3 sigmoid  4 elu  5 selu  6 softplus  7 softsign  8 LeakyReLU(0.3)
constexpr float kSeluScale = 1.0507009873554805f, kSeluAlpha = 1.6732632423543772f;

// relu / linear inline; the other activations out of line: the small-batch kernels unroll over their rows, and eight
// inlined libm bodies per row made them instruction-fetch bound (10 000-line kernels that run once per step)
__device__ __attribute__((noinline)) float act_fwd_other(int a, float x) {
    switch (a) {
        case 2: return tanhf(x);
        case 3: { const float e = expf(-fabsf(x)); const float s = 1.f / (1.f + e); return x >= 0.f ? s : e * s; }
        case 4: return x > 0.f ? x : expm1f(x);
        case 5: return kSeluScale * (x > 0.f ? x : kSeluAlpha * expm1f(x));
        case 6: return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x)));
        case 7: return x / (1.f + fabsf(x));
        default: return x > 0.f ? x : 0.3f * x;
    }
}
__device__ __forceinline__ float act_fwd(int a, float x) {
    if (a == 1) return fmaxf(x, 0.f);
    if (a == 0) return x;
    return act_fwd_other(a, x);
}

// derivative expressed through the OUTPUT h = act(x) (what the backward pass has at hand)
__device__ __attribute__((noinline)) float act_grad_other(int a, float h) {
    switch (a) {
        case 2: return 1.f - h * h;
        case 3: return h * (1.f - h);
        case 4: return h > 0.f ? 1.f : h + 1.f;
        case 5: return h > 0.f ? kSeluScale : h + kSeluScale * kSeluAlpha;
        case 6: return -expm1f(-h);                          // sigmoid(x) = 1 - exp(-softplus(x))
        case 7: { const float t = 1.f - fabsf(h); return t * t; }
        default: return h > 0.f ? 1.f : 0.3f;
    }
}
__device__ __forceinline__ float act_grad(int a, float h) {
    if (a == 1) return h > 0.f ? 1.f : 0.f;
    if (a == 0) return 1.f;
    return act_grad_other(a, h);
}

__host__ __device__ inline int n_chunks(int B) {
    int r = (B + 63) / 64;           // 64 chunks at B = 4096: every apply workgroup re-merges them
    return r < 1 ? 1 : (r > kMaxChunks ? kMaxChunks : r);
}
__host__ __device__ inline int chunk_rows(int B, int R) { return (B + R - 1) / R; }

// sum over the 4 row lanes (waves) of a workgroup; result valid in every thread
__device__ __forceinline__ float wg_rowlane_sum(float v, float* sm /*[4][64]*/) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    __syncthreads();
    sm[ty * 64 + tx] = v;
    __syncthreads();
    return (sm[tx] + sm[64 + tx]) + (sm[128 + tx] + sm[192 + tx]);
}

// part[r][0][c] = mean of column c over the chunk's rows, part[r][1][c] = sum (x - mean)^2
__global__ __launch_bounds__(256) void col_moments_kernel(const float* Z, long ldz, int B, int H,
                                                          float* part) {
    __shared__ float sm[256];
    const int R = gridDim.x, r = blockIdx.x;
    const int cr = chunk_rows(B, R);
    const int r0 = r * cr, r1 = min(B, r0 + cr);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const float cnt = (float)max(r1 - r0, 0);
    for (int c0 = 0; c0 < H; c0 += 64) {
        const int c = c0 + tx;
        float s = 0.f;
        if (c < H) {
#pragma unroll 8
            for (int i = r0 + ty; i < r1; i += 4) s += Z[(long)i * ldz + c];
        }
        const float mean = cnt > 0.f ? wg_rowlane_sum(s, sm) / cnt : 0.f;
        float q = 0.f;
        if (c < H) {
#pragma unroll 8
            for (int i = r0 + ty; i < r1; i += 4) { const float d = Z[(long)i * ldz + c] - mean; q += d * d; }
        }
        const float m2 = wg_rowlane_sum(q, sm);
        if (c < H && ty == 0) {
            part[((long)r * 2 + 0) * H + c] = mean;
            part[((long)r * 2 + 1) * H + c] = m2;
        }
    }
}

// Chan et al. merge of E (count, mean, M2) entries for column c
__device__ __forceinline__ void merge_entries(const float* entries, const float* counts, int E,
                                              int H, int c, int Bfallback, double& n_out,
                                              double& mean_out, double& m2_out) {
    double n = 0.0, mean = 0.0, m2 = 0.0;
    const int cr = chunk_rows(Bfallback, E);
    for (int e = 0; e < E; ++e) {
        double ne;
        if (counts) ne = (double)counts[e];
        else { const int r0 = e * cr; int r1 = r0 + cr; if (r1 > Bfallback) r1 = Bfallback; ne = r1 > r0 ? (double)(r1 - r0) : 0.0; }
        if (ne <= 0.0) continue;
        const double me = (double)entries[((long)e * 2 + 0) * H + c];
        const double qe = (double)entries[((long)e * 2 + 1) * H + c];
        const double tot = n + ne;
        const double delta = me - mean;
        mean += delta * ne / tot;
        m2 += qe + delta * delta * n * ne / tot;
        n = tot;
    }
    n_out = n; mean_out = mean; m2_out = m2;
}


// Workgroup version of the merge for the apply kernel: the E entries of a column are split over
// the 4 row lanes (ty), combined through LDS.  Two passes in fp64 -- N = sum n_e,
// mean = sum n_e mean_e / N, M2 = sum (M2_e + n_e (mean_e - mean)^2) -- no division inside the
// loops (the sequential Chan merge spent ~20 us per launch on fp64 divisions at 64 entries).
__device__ __forceinline__ double entry_count(const float* counts, int e, int E, int Bfallback) {
    if (counts) return (double)counts[e];
    const int cr = chunk_rows(Bfallback, E);
    const int r0 = e * cr;
    int r1 = r0 + cr;
    if (r1 > Bfallback) r1 = Bfallback;
    return r1 > r0 ? (double)(r1 - r0) : 0.0;
}

__device__ __forceinline__ void merge_entries_wg(const float* entries, const float* counts, int E,
                                                 int H, int c, int Bfallback, double* sm /*[2][256]*/,
                                                 double& n_out, double& mean_out, double& m2_out) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    double n = 0.0, sw = 0.0;
    const int cc = c < H ? c : H - 1;
    // batches of 4 entries per thread: the loads of a batch are unconditional (clamped) and in flight together
    for (int e0 = ty; e0 < E; e0 += 16) {
        float me[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) me[u] = entries[((long)(e0 + 4 * u < E ? e0 + 4 * u : E - 1) * 2 + 0) * H + cc];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 4 * u;
            const double ne = (e < E && c < H) ? entry_count(counts, e, E, Bfallback) : 0.0;
            n += ne;
            sw += ne * (double)me[u];
        }
    }
    __syncthreads();
    sm[ty * 64 + tx] = n; sm[256 + ty * 64 + tx] = sw;
    __syncthreads();
    n = (sm[tx] + sm[64 + tx]) + (sm[128 + tx] + sm[192 + tx]);
    sw = (sm[256 + tx] + sm[320 + tx]) + (sm[384 + tx] + sm[448 + tx]);
    const double mean = n > 0.0 ? sw / n : 0.0;
    double q = 0.0;
    for (int e0 = ty; e0 < E; e0 += 16) {
        float me[4], qe[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long eo = (long)(e0 + 4 * u < E ? e0 + 4 * u : E - 1) * 2;
            me[u] = entries[(eo + 0) * H + cc]; qe[u] = entries[(eo + 1) * H + cc];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + 4 * u;
            const double ne = (e < E && c < H) ? entry_count(counts, e, E, Bfallback) : 0.0;
            const double d = (double)me[u] - mean;
            q += ne > 0.0 ? (double)qe[u] + ne * d * d : 0.0;
        }
    }
    __syncthreads();
    sm[ty * 64 + tx] = q;
    __syncthreads();
    n_out = n; mean_out = mean;
    m2_out = (sm[tx] + sm[64 + tx]) + (sm[128 + tx] + sm[192 + tx]);
}

__global__ __launch_bounds__(256) void moments_combine_kernel(const float* entries,
                                                              const float* counts, int E, int H,
                                                              float* out) {
    __shared__ double smd[512];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    double n, mean, m2;
    merge_entries_wg(entries, counts, E, H, c, 0, smd, n, mean, m2);
    if (threadIdx.x < 64 && c < H) {
        out[c] = (float)mean;
        out[H + c] = (float)m2;
    }
}

struct BnApplyArgs {
    const float* Z; long ldz; int B, H;
    const float* entries; const float* counts; int E;
    const float* beta; float* mm; float* mv;
    float momentum, eps; int relu;
    float* Hout; long ldh; float* xhat; long ldx; float* inv_std;
};

__global__ __launch_bounds__(256) void bn_relu_apply_kernel(BnApplyArgs a) {
    extern __shared__ float dyn[];               // [2][H]: mean, inv_std
    float* s_mean = dyn;
    float* s_inv = dyn + a.H;
    __shared__ double smd[512];
    for (int c0 = 0; c0 < a.H; c0 += 64) {
        const int c = c0 + (threadIdx.x & 63);
        float mean = 0.f, var = 1.f;
        if (a.entries) {
            double n, m, m2;
            merge_entries_wg(a.entries, a.counts, a.E, a.H, c, a.B, smd, n, m, m2);
            mean = (float)m;
            var = (float)(m2 / n);                // biased variance
            if (blockIdx.x == 0 && threadIdx.x < 64 && c < a.H) {
                // moving = moving - (moving - batch) * (1 - momentum)
                a.mm[c] = a.mm[c] - (a.mm[c] - mean) * (1.f - a.momentum);
                a.mv[c] = a.mv[c] - (a.mv[c] - var) * (1.f - a.momentum);
            }
        } else if (c < a.H) {
            mean = a.mm[c]; var = a.mv[c];
        }
        if (threadIdx.x < 64 && c < a.H) {
            const float inv = 1.f / sqrtf(var + a.eps);
            s_mean[c] = mean; s_inv[c] = inv;
            if (blockIdx.x == 0 && a.inv_std) a.inv_std[c] = inv;
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int r0 = blockIdx.x * kApplyRows, r1 = min(a.B, r0 + kApplyRows);
    for (int c0 = 0; c0 < a.H; c0 += 64) {
        const int c = c0 + tx;
        if (c >= a.H) continue;
        const float mean = s_mean[c], inv = s_inv[c], beta = a.beta ? a.beta[c] : 0.f;
        float z[kApplyRows / 4];
#pragma unroll
        for (int u = 0; u < kApplyRows / 4; ++u) {           // loads first: the activation may branch
            const int i = r0 + ty + 4 * u;
            z[u] = a.Z[(long)(i < r1 ? i : r1 - 1) * a.ldz + c];
        }
#pragma unroll
        for (int u = 0; u < kApplyRows / 4; ++u) {
            const int i = r0 + ty + 4 * u;
            if (i < r1) {
                const float xh = (z[u] - mean) * inv;
                if (a.xhat) a.xhat[(long)i * a.ldx + c] = xh;
                a.Hout[(long)i * a.ldh + c] = act_fwd(a.relu, xh + beta);
            }
        }
    }
}

// part[r][0][c] = sum dy, part[r][1][c] = sum dy*xhat over the chunk's rows; dy = dh*[h>0]
__global__ __launch_bounds__(256) void bn_bwd_sums_kernel(const float* dH, long ldd,
                                                          const float* Hact, long ldh,
                                                          const float* xhat, long ldx, int B, int H,
                                                          float* part, int act) {
    __shared__ float sm[256];
    const int R = gridDim.x, r = blockIdx.x;
    const int cr = chunk_rows(B, R);
    const int r0 = r * cr, r1 = min(B, r0 + cr);
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int c0 = 0; c0 < H; c0 += 64) {
        const int c = c0 + tx;
        float s1 = 0.f, s2 = 0.f;
        if (c < H)
            for (int i0 = r0 + ty; i0 < r1; i0 += 32) {      // 8 rows per batch: all 24 loads in flight, then the arithmetic
                float d[8], h[8], x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int i = i0 + 4 * u < r1 ? i0 + 4 * u : r1 - 1;
                    d[u] = dH[(long)i * ldd + c]; h[u] = Hact[(long)i * ldh + c]; x[u] = xhat[(long)i * ldx + c];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float dy = i0 + 4 * u < r1 ? d[u] * act_grad(act, h[u]) : 0.f;
                    s1 += dy; s2 += dy * x[u];
                }
            }
        const float t1 = wg_rowlane_sum(s1, sm);
        const float t2 = wg_rowlane_sum(s2, sm);
        if (c < H && ty == 0) {
            part[((long)r * 2 + 0) * H + c] = t1;
            part[((long)r * 2 + 1) * H + c] = t2;
        }
    }
}

struct BnBwdArgs {
    const float* dH; long ldd; const float* Hact; long ldh; const float* xhat; long ldx;
    const float* inv_std; const float* sums; int E; float n_total; int B, H;
    float* dZ; long ldz; float* dbeta; int act;
};

__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBwdArgs a) {
    extern __shared__ float dyn[];               // [2][H]: S1/n, S2/n
    float* s1 = dyn;
    float* s2 = dyn + a.H;
    __shared__ float smf[256];
    for (int c0 = 0; c0 < a.H; c0 += 64) {
        const int c = c0 + (threadIdx.x & 63);
        float v1 = 0.f, v2 = 0.f;
        if (c < a.H)
#pragma unroll 4
            for (int e = threadIdx.x >> 6; e < a.E; e += 4) {
                v1 += a.sums[((long)e * 2 + 0) * a.H + c];
                v2 += a.sums[((long)e * 2 + 1) * a.H + c];
            }
        v1 = wg_rowlane_sum(v1, smf);
        v2 = wg_rowlane_sum(v2, smf);
        if (threadIdx.x < 64 && c < a.H) {
            if (blockIdx.x == 0 && a.dbeta) a.dbeta[c] = v1;
            s1[c] = v1 / a.n_total; s2[c] = v2 / a.n_total;
        }
    }
    __syncthreads();
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int r0 = blockIdx.x * kApplyRows, r1 = min(a.B, r0 + kApplyRows);
    for (int c0 = 0; c0 < a.H; c0 += 64) {
        const int c = c0 + tx;
        if (c >= a.H) continue;
        const float m1 = s1[c], m2 = s2[c], inv = a.inv_std[c];
        float d[kApplyRows / 4], h[kApplyRows / 4], x[kApplyRows / 4];
#pragma unroll
        for (int u = 0; u < kApplyRows / 4; ++u) {           // loads first: the activation derivative may branch
            const int i = r0 + ty + 4 * u < r1 ? r0 + ty + 4 * u : r1 - 1;
            d[u] = a.dH[(long)i * a.ldd + c]; h[u] = a.Hact[(long)i * a.ldh + c]; x[u] = a.xhat[(long)i * a.ldx + c];
        }
#pragma unroll
        for (int u = 0; u < kApplyRows / 4; ++u) {
            const int i = r0 + ty + 4 * u;
            if (i < r1) a.dZ[(long)i * a.ldz + c] = inv * (d[u] * act_grad(a.act, h[u]) - m1 - x[u] * m2);
        }
    }
}

// ---- small batches (B <= kFusedRows = 64 rows, one GPU): statistics AND apply in one launch, one workgroup per 64
// columns -- the reference-default batch of 32 cells (dca/train.py:37) spends its step in launch gaps and dependent
// memory round trips, not in arithmetic.  Same formulas as col_moments_kernel + bn_relu_apply_kernel with one chunk.
constexpr int kFusedRows = 64;

// RPT = rows per thread (4 row lanes): the column slab is read ONCE into registers -- at 32 rows these kernels are
// a chain of dependent memory round trips, not arithmetic
template <int RPT>
__global__ __launch_bounds__(256) void bn_relu_small_kernel(BnApplyArgs a) {
    __shared__ float sm[256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const bool cv = c < a.H;
    float z[RPT];
    float s = 0.f;
    const int ccl = cv ? c : a.H - 1;
    const float mm_in = a.mm[ccl], mv_in = a.mv[ccl], beta_in = a.beta ? a.beta[ccl] : 0.f;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {                  // unconditional loads (clamped): all in flight at once
        const int i = ty + 4 * k;
        z[k] = a.Z[(long)(i < a.B ? i : a.B - 1) * a.ldz + ccl];
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        z[k] = (cv && ty + 4 * k < a.B) ? z[k] : 0.f;
        s += z[k];
    }
    const float mean = wg_rowlane_sum(s, sm) / (float)a.B;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const float d = z[k] - mean;
        q += (ty + 4 * k < a.B) ? d * d : 0.f;
    }
    const float m2 = wg_rowlane_sum(q, sm);
    if (!cv) return;
    const float var = (float)((double)m2 / (double)a.B);             // biased variance
    const float inv = 1.f / sqrtf(var + a.eps);
    if (ty == 0) {
        a.mm[c] = mm_in - (mm_in - mean) * (1.f - a.momentum);
        a.mv[c] = mv_in - (mv_in - var) * (1.f - a.momentum);
        if (a.inv_std) a.inv_std[c] = inv;
    }
    const float beta = beta_in;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int i = ty + 4 * k;
        if (i < a.B) {
            const float xh = (z[k] - mean) * inv;
            if (a.xhat) a.xhat[(long)i * a.ldx + c] = xh;
            a.Hout[(long)i * a.ldh + c] = act_fwd(a.relu, xh + beta);
        }
    }
}

template <int RPT>
__global__ __launch_bounds__(256) void bn_bwd_small_kernel(BnBwdArgs a) {
    __shared__ float sm[256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    const bool cv = c < a.H;
    float dy[RPT], xh[RPT], ha[RPT];
    float s1 = 0.f, s2 = 0.f;
    const int ccl = cv ? c : a.H - 1;
    const float inv_in = a.inv_std[ccl];
    // every load first (unconditional, clamped addresses), then the arithmetic: see dense_bn_bwd_small_kernel
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int i = ty + 4 * k;
        const int ic = i < a.B ? i : a.B - 1;
        dy[k] = a.dH[(long)ic * a.ldd + ccl];
        ha[k] = a.Hact[(long)ic * a.ldh + ccl];
        xh[k] = a.xhat[(long)ic * a.ldx + ccl];
    }
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const bool ok = cv && ty + 4 * k < a.B;
        dy[k] = ok ? dy[k] * act_grad(a.act, ha[k]) : 0.f;
        xh[k] = ok ? xh[k] : 0.f;
        s1 += dy[k]; s2 += dy[k] * xh[k];
    }
    const float t1 = wg_rowlane_sum(s1, sm);
    const float t2 = wg_rowlane_sum(s2, sm);
    if (!cv) return;
    if (ty == 0 && a.dbeta) a.dbeta[c] = t1;
    const float m1 = t1 / a.n_total, m2 = t2 / a.n_total, inv = inv_in;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int i = ty + 4 * k;
        if (i < a.B) a.dZ[(long)i * a.ldz + c] = inv * (dy[k] - m1 - xh[k] * m2);
    }
}

// ---- small batches: a whole hidden layer per launch ---------------------------------------------------------------
// At the reference-default batch of 32 cells (dca/train.py:37) every kernel of the hidden stack is a few microseconds
// of dependent memory round trips; these two do Dense -> BatchNormalization -> activation (dca/network.py:124-135) and
// its whole backward (d beta, dZ, weight / bias gradient, input gradient) in one launch each.  B <= 64 rows,
// K <= 64 inputs; the backward additionally h <= 64 units (one workgroup owns the layer).
constexpr int kSmallK = 64;

// Operands of a small layer -> LDS with EVERY load in flight at once: fixed trip counts, unconditional loads from
// clamped addresses, zeroing at the store.  (A `for (idx ...) lds[..] = cond ? g[..] : 0` loop compiles to one
// load - wait - store round trip per iteration: 24 dependent round trips per layer at the reference batch size.)
// Ws[k][cc] = W[k, c0 + cc] (k < K, c0 + cc < H), 64 x 64;  Hs[r][k] = Hp[r, k] (k < K), rows r < 4 * RPT.
template <int RPT, int LDW, int LDH>
__device__ __forceinline__ void small_operands_to_lds(const float* W, long ldw, int K, int H, int c0,
                                                       const float* Hp, long ldp, int B,
                                                       float (*Ws)[LDW], float (*Hs)[LDH]) {
    float wv[16], hv[RPT];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int idx = threadIdx.x + 256 * u, k = idx >> 6, cc = idx & 63;
        wv[u] = W[(long)(k < K ? k : K - 1) * ldw + (c0 + cc < H ? c0 + cc : H - 1)];
    }
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int idx = threadIdx.x + 256 * u, r = idx >> 6, k = idx & 63;
        hv[u] = Hp[(long)(r < B ? r : B - 1) * ldp + (k < K ? k : K - 1)];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int idx = threadIdx.x + 256 * u, k = idx >> 6, cc = idx & 63;
        Ws[k][cc] = (k < K && c0 + cc < H) ? wv[u] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < RPT; ++u) {
        const int idx = threadIdx.x + 256 * u, r = idx >> 6, k = idx & 63;
        Hs[r][k] = k < K ? hv[u] : 0.f;
    }
}

struct DenseSmallArgs {
    const float* Hp; long ldp;          // layer input [B, K]
    const float* W; long ldw;           // kernel [K, h]
    const float* bias;                  // [h]
    int B, K, H;
    int batchnorm;
    const float* beta; float* mm; float* mv; float momentum, eps; int act;
    float* Z; long ldz;                 // pre-activation (the latent code of the centre layer), may be NULL
    float* xhat; long ldx; float* Hout; long ldh; float* inv_std;
};

// RPT = rows per thread (8 for batches of up to 32 rows, 16 up to 64): small code matters more than anything else
// here -- the kernel runs once per step, from a cold instruction cache
template <int RPT>
__global__ __launch_bounds__(256) void dense_bn_small_kernel(DenseSmallArgs a) {
    __shared__ __attribute__((aligned(16))) float Hs[kFusedRows][kSmallK + 4];
    __shared__ float Ws[kSmallK][64];
    __shared__ float sm[256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 64, c = c0 + tx;
    const bool cv = c < a.H;
    const int K4 = (a.K + 3) & ~3;
    // everything the kernel reads from memory is requested here, in one batch
    const int cc_ = cv ? c : a.H - 1;
    const float b_in = a.bias[cc_];
    const float mm_in = a.batchnorm ? a.mm[cc_] : 0.f, mv_in = a.batchnorm ? a.mv[cc_] : 0.f;
    const float beta_in = (a.batchnorm && a.beta) ? a.beta[cc_] : 0.f;
    small_operands_to_lds<RPT>(a.W, a.ldw, a.K, a.H, c0, a.Hp, a.ldp, a.B, Ws, Hs);
    float z[RPT];
    const float b = cv ? b_in : 0.f;
#pragma unroll
    for (int k = 0; k < RPT; ++k) z[k] = b;
    __syncthreads();
#pragma unroll 1
    for (int kk = 0; kk < K4; kk += 4) {
        const float w0 = Ws[kk][tx], w1 = Ws[kk + 1][tx], w2 = Ws[kk + 2][tx], w3 = Ws[kk + 3][tx];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const float4 hv = *reinterpret_cast<const float4*>(&Hs[ty + 4 * k][kk]);      // rows beyond B: finite garbage, never stored
            z[k] = fmaf(hv.w, w3, fmaf(hv.z, w2, fmaf(hv.y, w1, fmaf(hv.x, w0, z[k]))));
        }
    }
    if (a.Z && cv) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) { const int i = ty + 4 * k; if (i < a.B) a.Z[(long)i * a.ldz + c] = z[k]; }
    }
    if (!a.batchnorm) {
        if (cv) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) { const int i = ty + 4 * k; if (i < a.B) a.Hout[(long)i * a.ldh + c] = act_fwd(a.act, z[k]); }
        }
        return;
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < RPT; ++k) s += (ty + 4 * k < a.B) ? z[k] : 0.f;
    const float mean = wg_rowlane_sum(s, sm) / (float)a.B;
    float q2 = 0.f;
#pragma unroll
    for (int k = 0; k < RPT; ++k) { const float d = z[k] - mean; q2 += (ty + 4 * k < a.B) ? d * d : 0.f; }
    const float m2 = wg_rowlane_sum(q2, sm);
    if (!cv) return;
    const float var = (float)((double)m2 / (double)a.B);
    const float inv = 1.f / sqrtf(var + a.eps);
    if (ty == 0) {
        a.mm[c] = mm_in - (mm_in - mean) * (1.f - a.momentum);
        a.mv[c] = mv_in - (mv_in - var) * (1.f - a.momentum);
        if (a.inv_std) a.inv_std[c] = inv;
    }
    const float beta = beta_in;
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        const int i = ty + 4 * k;
        if (i < a.B) {
            const float xh = (z[k] - mean) * inv;
            if (a.xhat) a.xhat[(long)i * a.ldx + c] = xh;
            a.Hout[(long)i * a.ldh + c] = act_fwd(a.act, xh + beta);
        }
    }
}

// ---- small batches: the hidden stack behind the first layer's product in ONE launch -------------------------------
// [batch norm + activation of layer 0] -> [Dense + batch norm + activation] x (n - 1), one workgroup, stage after
// stage (every layer at most 64 units wide: the reference's 64-32-64).  The stages are the formulas of
// bn_relu_small_kernel / dense_bn_small_kernel; what is saved is two launches and their gaps on a step that is bound
// by exactly those (12 -> 10 launches at batch 32).
constexpr int kChainMax = 4;

struct SmallLayer {                     // = dcahip_small_layer (include/dcahip.h)
    const float* W; long ldw;           // kernel [K, h]; NULL: this entry normalises its own Z (the first layer)
    const float* bias;
    int K, H;
    const float* beta; float* mm; float* mv;
    float* Z; long ldz; float* xhat; long ldx; float* Hout; long ldh; float* inv_std;
};

struct SmallChainArgs {
    SmallLayer l[kChainMax];
    const float* Hin; long ldin;        // input of the first entry when it has a kernel
    int n, B, batchnorm, act;
    float momentum, eps;
};

template <int RPT>
__global__ __launch_bounds__(256) void hidden_small_chain_kernel(SmallChainArgs a) {
    __shared__ __attribute__((aligned(16))) float Hs[kFusedRows][kSmallK + 4];
    __shared__ float Ws[2][kSmallK][64];        // kernel of the current stage / of the next one (requested a stage ahead)
    __shared__ float sm[256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = tx;
    if (a.l[0].W) small_operands_to_lds<RPT>(a.l[0].W, a.l[0].ldw, a.l[0].K, a.l[0].H, 0, a.Hin, a.ldin, a.B, Ws[0], Hs);
#pragma unroll 1
    for (int st = 0; st < a.n; ++st) {
        const SmallLayer& L = a.l[st];
        const bool cv = c < L.H;
        // everything this stage reads from memory is requested here: the NEXT stage's kernel (it does not depend on
        // this stage) and the per-column inputs; the activations travel from stage to stage through LDS
        const bool has_next = st + 1 < a.n;
        const SmallLayer& Nx = a.l[has_next ? st + 1 : st];
        float wn[16];
        if (has_next) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int idx = threadIdx.x + 256 * u, k = idx >> 6, cc = idx & 63;
                wn[u] = Nx.W[(long)(k < Nx.K ? k : Nx.K - 1) * Nx.ldw + (cc < Nx.H ? cc : Nx.H - 1)];
            }
        }
        const int cc_ = cv ? c : L.H - 1;
        const float mm_in = a.batchnorm ? L.mm[cc_] : 0.f, mv_in = a.batchnorm ? L.mv[cc_] : 0.f;
        const float beta_in = (a.batchnorm && L.beta) ? L.beta[cc_] : 0.f;
        float z[RPT];
        if (L.W) {
            const int K4 = (L.K + 3) & ~3;
            const float b = cv ? L.bias[cc_] : 0.f;
#pragma unroll
            for (int k = 0; k < RPT; ++k) z[k] = b;
            __syncthreads();            // Ws[st & 1] and Hs (the previous stage's output) are complete
            const float (*Wc)[64] = Ws[st & 1];
#pragma unroll 1
            for (int kk = 0; kk < K4; kk += 4) {
                const float w0 = Wc[kk][tx], w1 = Wc[kk + 1][tx], w2 = Wc[kk + 2][tx], w3 = Wc[kk + 3][tx];
#pragma unroll
                for (int k = 0; k < RPT; ++k) {
                    const float4 hv = *reinterpret_cast<const float4*>(&Hs[ty + 4 * k][kk]);      // rows beyond B: never stored
                    z[k] = fmaf(hv.w, w3, fmaf(hv.z, w2, fmaf(hv.y, w1, fmaf(hv.x, w0, z[k]))));
                }
            }
            if (L.Z && cv) {
#pragma unroll
                for (int k = 0; k < RPT; ++k) { const int i = ty + 4 * k; if (i < a.B) L.Z[(long)i * L.ldz + c] = z[k]; }
            }
        } else {
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int i = ty + 4 * k;
                z[k] = L.Z[(long)(i < a.B ? i : a.B - 1) * L.ldz + cc_];
            }
        }
        float hout[RPT];
        if (!a.batchnorm) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) hout[k] = act_fwd(a.act, z[k]);
            __syncthreads();            // every thread is done reading Hs
        } else {
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < RPT; ++k) s += (cv && ty + 4 * k < a.B) ? z[k] : 0.f;
            const float mean = wg_rowlane_sum(s, sm) / (float)a.B;
            float q2 = 0.f;
#pragma unroll
            for (int k = 0; k < RPT; ++k) { const float d = z[k] - mean; q2 += (cv && ty + 4 * k < a.B) ? d * d : 0.f; }
            const float m2 = wg_rowlane_sum(q2, sm);
            const float var = (float)((double)m2 / (double)a.B);             // biased variance
            const float inv = 1.f / sqrtf(var + a.eps);
            if (cv && ty == 0) {
                L.mm[c] = mm_in - (mm_in - mean) * (1.f - a.momentum);
                L.mv[c] = mv_in - (mv_in - var) * (1.f - a.momentum);
                if (L.inv_std) L.inv_std[c] = inv;
            }
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int i = ty + 4 * k;
                const float xh = (z[k] - mean) * inv;
                if (cv && i < a.B && L.xhat) L.xhat[(long)i * L.ldx + c] = xh;
                hout[k] = act_fwd(a.act, xh + beta_in);
            }
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int i = ty + 4 * k;
            if (cv && i < a.B) L.Hout[(long)i * L.ldh + c] = hout[k];
            Hs[i][c] = cv ? hout[k] : 0.f;          // the next stage's input (columns beyond this layer's width: zero)
        }
        if (has_next) {
            float (*Wn)[64] = Ws[(st + 1) & 1];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int idx = threadIdx.x + 256 * u, k = idx >> 6, cc = idx & 63;
                Wn[k][cc] = (k < Nx.K && cc < Nx.H) ? wn[u] : 0.f;
            }
        }
    }
}

struct DenseSmallBwdArgs {
    const float* dH; long ldd;          // gradient w.r.t. the layer output [B, h]
    const float* Hact; long ldh;        // layer output
    const float* xhat; long ldx; const float* inv_std;
    const float* Hp; long ldp;          // layer input [B, K]
    const float* W; long ldw;           // kernel [K, h]
    int B, K, H; int batchnorm; float n_total; int act;
    float* gW; long ldg;                // [K + 1, h]: weight gradient, row K = bias gradient
    float* dbeta;
    float* dHp; long lddp;              // gradient w.r.t. the layer input [B, K]
};

constexpr int kBwdWGs = 8;      // workgroups of the small-batch layer backward: each recomputes dZ (cheap) and takes 1/8 of the products

template <int RPT>
__global__ __launch_bounds__(256) void dense_bn_bwd_small_kernel(DenseSmallBwdArgs a) {
    __shared__ float dZs[kFusedRows][kSmallK + 1];
    __shared__ float Hs[kFusedRows][kSmallK + 1];
    __shared__ float Ws[kSmallK][kSmallK + 1];
    __shared__ float sm[256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int wg = blockIdx.x;
    // per-column scale, then the operands of the two products -> LDS: one batch of loads (see small_operands_to_lds)
    const float inv_in = a.batchnorm ? a.inv_std[tx < a.H ? tx : a.H - 1] : 1.f;
    small_operands_to_lds<RPT>(a.W, a.ldw, a.K, a.H, 0, a.Hp, a.ldp, a.B, Ws, Hs);
    // ---- phase 1 (every workgroup): dZ = batch-norm backward of dH * act'(H), or the activation derivative alone
    {
        const int c = tx;
        const bool cv = c < a.H;
        float dy[RPT], xh[RPT], ha[RPT];
        float s1 = 0.f, s2 = 0.f;
        // every load first (unconditional, clamped addresses): the activation derivative below may branch, and a branch
        // between two loads costs a memory round trip per row
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int i = ty + 4 * k;
            const int ic = i < a.B ? i : a.B - 1, ccl = cv ? c : a.H - 1;
            dy[k] = a.dH[(long)ic * a.ldd + ccl];
            ha[k] = a.Hact[(long)ic * a.ldh + ccl];
            xh[k] = a.batchnorm ? a.xhat[(long)ic * a.ldx + ccl] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const bool ok = cv && ty + 4 * k < a.B;
            dy[k] = ok ? dy[k] * act_grad(a.act, ha[k]) : 0.f;
            xh[k] = ok ? xh[k] : 0.f;
            s1 += dy[k]; s2 += dy[k] * xh[k];
        }
        float m1 = 0.f, m2 = 0.f, inv = 1.f;
        if (a.batchnorm) {
            const float t1 = wg_rowlane_sum(s1, sm);
            const float t2 = wg_rowlane_sum(s2, sm);
            if (wg == 0 && cv && ty == 0 && a.dbeta) a.dbeta[c] = t1;
            m1 = t1 / a.n_total; m2 = t2 / a.n_total; inv = cv ? inv_in : 0.f;
        }
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int i = ty + 4 * k;
            if (i < a.B) dZs[i][c] = cv ? (a.batchnorm ? inv * (dy[k] - m1 - xh[k] * m2) : dy[k]) : 0.f;
        }
    }
    __syncthreads();
    // ---- phase 2: gW[k, c] = sum_r Hp[r, k] dZ[r, c] for the input units k = wg, wg + 8, ..; wg 0 adds row K = sum_r dZ[r, c]
    {
        const int c = tx;
        if (c < a.H) {
            constexpr int NK = kSmallK / (4 * kBwdWGs);      // k's per thread: k = wg + 8 (ty + 4 j)
            float acc[NK];
#pragma unroll
            for (int j = 0; j < NK; ++j) acc[j] = 0.f;
            float cs = 0.f;
            for (int r = 0; r < a.B; ++r) {
                const float d = dZs[r][c];
                cs += d;
#pragma unroll
                for (int j = 0; j < NK; ++j) {
                    const int k = wg + kBwdWGs * (ty + 4 * j);
                    acc[j] = fmaf(k < a.K ? Hs[r][k] : 0.f, d, acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < NK; ++j) {
                const int k = wg + kBwdWGs * (ty + 4 * j);
                if (k < a.K) a.gW[(long)k * a.ldg + c] = acc[j];
            }
            if (wg == 0 && ty == 0) a.gW[(long)a.K * a.ldg + c] = cs;
        }
    }
    // ---- phase 3: dHp[r, k] = sum_c dZ[r, c] W[k, c] for the rows r = wg + 8 ty + 32 j
    if (a.dHp) {
        const int k = tx;
        if (k < a.K) {
            constexpr int NR = RPT / kBwdWGs;            // rows per thread of this workgroup
            float acc[NR];
#pragma unroll
            for (int j = 0; j < NR; ++j) acc[j] = 0.f;
            for (int c = 0; c < a.H; ++c) {
                const float w = Ws[k][c];
#pragma unroll
                for (int j = 0; j < NR; ++j) {
                    const int r = wg + kBwdWGs * (ty + 4 * j);
                    acc[j] = fmaf(r < a.B ? dZs[r][c] : 0.f, w, acc[j]);
                }
            }
#pragma unroll
            for (int j = 0; j < NR; ++j) {
                const int r = wg + kBwdWGs * (ty + 4 * j);
                if (r < a.B) a.dHp[(long)r * a.lddp + k] = acc[j];
            }
        }
    }
}

__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* dH, long ldd, const float* Hact,
                                                       long ldh, int B, int H, float* dZ, long ldz, int act) {
    const long total = (long)B * H;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / H), c = (int)(i - (long)r * H);
        dZ[(long)r * ldz + c] = dH[(long)r * ldd + c] * act_grad(act, Hact[(long)r * ldh + c]);
    }
}

__global__ __launch_bounds__(256) void relu_fwd_kernel(const float* Z, long ldz, int B, int H,
                                                       float* Hout, long ldh, int act) {
    const long total = (long)B * H;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / H), c = (int)(i - (long)r * H);
        Hout[(long)r * ldh + c] = act_fwd(act, Z[(long)r * ldz + c]);
    }
}

__global__ __launch_bounds__(256) void colsum_chain_kernel(const float* x, long ldx, int B, int N,
                                                           const float* theta_w, float* out) {
    __shared__ float sm[256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + tx;
    float s = 0.f;
    if (c < N) for (int i = ty; i < B; i += 4) s += x[(long)i * ldx + c];
    const float tot = wg_rowlane_sum(s, sm);
    if (c < N && ty == 0) {
        float chain = 1.f;
        if (theta_w) {                           // d clip(exp(w),1e-3,1e4) / dw
            const float e = expf(theta_w[c]);
            chain = (e >= 1e-3f && e <= 1e4f) ? e : 0.f;
        }
        out[c] = tot * chain;
    }
}

// the end-of-step bookkeeping (dcahip_step_end) riding on the optimizer launch: block 0 records the batch loss and
// advances the batch cursor -- nothing else in this kernel reads either
struct StepEnd {
    const float* loss; double weight; float* hist; int rows_per_slot; double* acc; long long* cursor; int advance; int on;
};

__global__ __launch_bounds__(256) void rmsprop_clip_kernel(float* w, const float* g, float* ms,
                                                           long n, const float* lrp, float rho,
                                                           float eps, float clip, StepEnd se) {
    if (se.on && blockIdx.x == 0 && threadIdx.x == 0) {
        const long long c = se.cursor ? *se.cursor : 0;
        if (se.loss) {
            const float l = *se.loss;
            if (se.hist) se.hist[se.rows_per_slot > 0 ? c / se.rows_per_slot : 0] = l;
            if (se.acc) *se.acc += (double)l * se.weight;
        }
        if (se.cursor) *se.cursor = c + se.advance;
    }
    const float lr = *lrp;
    const long nv = n >> 2;
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += stride) {
        float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 mv = reinterpret_cast<const float4*>(ms)[i];
        float4 wv = reinterpret_cast<const float4*>(w)[i];
        float* gp = &gv.x; float* mp = &mv.x; float* wp = &wv.x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float gj = gp[j];
            if (clip > 0.f) gj = fminf(fmaxf(gj, -clip), clip);
            mp[j] = rho * mp[j] + (1.f - rho) * gj * gj;
            wp[j] = wp[j] - lr * gj / sqrtf(mp[j] + eps);
        }
        reinterpret_cast<float4*>(ms)[i] = mv;
        reinterpret_cast<float4*>(w)[i] = wv;
    }
    for (long i = (nv << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float gj = g[i];
        if (clip > 0.f) gj = fminf(fmaxf(gj, -clip), clip);
        const float m = rho * ms[i] + (1.f - rho) * gj * gj;
        ms[i] = m;
        w[i] = w[i] - lr * gj / sqrtf(m + eps);
    }
}

// dst [C, R] = src [R, C]^T through 32 x 33 LDS tiles (coalesced on both sides); source row r = perm[cursor + r] when
// perm is given (the minibatch gather of the training step)
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, long lds_, int R, int C,
                                                        float* __restrict__ dst, long ldd,
                                                        const int* __restrict__ perm, const long long* __restrict__ cursor) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    const long long cur = (perm && cursor) ? *cursor : 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = r0 + ty + 8 * k, c = c0 + tx;
        float v = 0.f;
        if (r < R && c < C) {
            const long sr = perm ? (long)perm[cur + r] : (long)r;
            v = src[sr * lds_ + c];
        }
        tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, r = r0 + tx;
        if (c < C && r < R) dst[(long)c * ldd + r] = tile[tx][ty + 8 * k];
    }
}

// the same through 64 x 64 tiles with 16-byte accesses on both sides (aligned operands, leading dimensions multiples of 4)
__global__ __launch_bounds__(256) void transpose64_kernel(const float* __restrict__ src, long lds_, int R, int C,
                                                          float* __restrict__ dst, long ldd,
                                                          const int* __restrict__ perm, const long long* __restrict__ cursor) {
    __shared__ float tile[64][65];
    const int tq = threadIdx.x & 15, tr = threadIdx.x >> 4;           // 16 quads across, 16 rows per pass
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const long long cur = (perm && cursor) ? *cursor : 0;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {                                     // all four loads in flight (clamped addresses)
        const int r = r0 + tr + 16 * k, c = c0 + 4 * tq;
        const int rc = r < R ? r : R - 1;
        const long sr = perm ? (long)perm[cur + rc] : (long)rc;
        // c is a multiple of 4 and the leading dimension too: the quad at c < C lies inside the row's storage (its tail may
        // be padding, which lands in tile columns that are never written out)
        v[k] = *reinterpret_cast<const float4*>(src + sr * lds_ + (c < C ? c : 0));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = tr + 16 * k, c = 4 * tq;
        tile[r][c] = v[k].x; tile[r][c + 1] = v[k].y; tile[r][c + 2] = v[k].z; tile[r][c + 3] = v[k].w;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = tr + 16 * k, r = 4 * tq;                        // output row c0 + c, source rows r0 + r .. + 3
        if (c0 + c < C) {
            const float4 o = make_float4(tile[r][c], tile[r + 1][c], tile[r + 2][c], tile[r + 3][c]);
            float* d = dst + (long)(c0 + c) * ldd + r0 + r;
            if (r0 + r + 4 <= R) *reinterpret_cast<float4*>(d) = o;
            else {
                if (r0 + r < R) d[0] = o.x;
                if (r0 + r + 1 < R) d[1] = o.y;
                if (r0 + r + 2 < R) d[2] = o.z;
            }
        }
    }
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int dcahip_transpose_rows(const float* src, long ld_src, const int* perm, const long long* cursor, int R, int C,
                                     float* dst, long ld_dst, void* stream) {
    if (!src || !dst || R <= 0 || C <= 0 || ld_src < C || ld_dst < R) return DCAHIP_EINVAL;
    if (R >= 64 && C >= 64 && al16(src) && al16(dst) && ld_src % 4 == 0 && ld_dst % 4 == 0)
        hipLaunchKernelGGL(transpose64_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0,
                           static_cast<hipStream_t>(stream), src, ld_src, R, C, dst, ld_dst, perm, cursor);
    else
        hipLaunchKernelGGL(transpose_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0,
                           static_cast<hipStream_t>(stream), src, ld_src, R, C, dst, ld_dst, perm, cursor);
    return (int)hipGetLastError();
}

extern "C" int dcahip_transpose(const float* src, long ld_src, int R, int C, float* dst, long ld_dst, void* stream) {
    return dcahip_transpose_rows(src, ld_src, nullptr, nullptr, R, C, dst, ld_dst, stream);
}

extern "C" int dcahip_col_moments_chunks(int B) { return n_chunks(B); }

extern "C" int dcahip_col_moments(const float* Z, long ldz, int B, int H, float* part, void* stream) {
    if (!Z || !part || B <= 0 || H <= 0) return DCAHIP_EINVAL;
    hipLaunchKernelGGL(col_moments_kernel, dim3(n_chunks(B)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), Z, ldz, B, H, part);
    return (int)hipGetLastError();
}

extern "C" int dcahip_moments_combine(const float* entries, const float* counts, int E, int H,
                                      float* out, void* stream) {
    if (!entries || !counts || !out || E <= 0 || H <= 0) return DCAHIP_EINVAL;
    hipLaunchKernelGGL(moments_combine_kernel, dim3((H + 63) / 64), dim3(256), 0,
                       static_cast<hipStream_t>(stream), entries, counts, E, H, out);
    return (int)hipGetLastError();
}

extern "C" int dcahip_bn_relu_apply(const float* Z, long ldz, int B, int H, const float* entries,
                                    const float* counts, int E, const float* beta,
                                    float* moving_mean, float* moving_var, float momentum, float eps,
                                    int relu, float* Hout, long ldh, float* xhat, long ldx,
                                    float* inv_std, void* stream) {
    // B == 0 is legal: a data-parallel rank whose shard is exhausted still has to fold the
    // global batch statistics into its moving averages
    if (!Z || !Hout || !moving_mean || !moving_var || B < 0 || H <= 0) return DCAHIP_EINVAL;
    if (entries && E <= 0) return DCAHIP_EINVAL;
    BnApplyArgs a{Z, ldz, B, H, entries, counts, E, beta, moving_mean, moving_var, momentum, eps,
                  relu, Hout, ldh, xhat, ldx, inv_std};
    const int grid = B > 0 ? (B + kApplyRows - 1) / kApplyRows : 1;
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(grid), dim3(256), 2 * H * sizeof(float),
                       static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int dcahip_bn_bwd_sums(const float* dH, long ldd, const float* Hact, long ldh,
                                  const float* xhat, long ldx, int B, int H, float* part,
                                  int act, void* stream) {
    if (!dH || !Hact || !xhat || !part || B <= 0 || H <= 0) return DCAHIP_EINVAL;
    hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(n_chunks(B)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dH, ldd, Hact, ldh, xhat, ldx, B, H, part, act);
    return (int)hipGetLastError();
}

extern "C" int dcahip_bn_bwd_apply(const float* dH, long ldd, const float* Hact, long ldh,
                                   const float* xhat, long ldx, const float* inv_std,
                                   const float* sums, int E, float n_total, int B, int H, float* dZ,
                                   long ldz, float* dbeta, int act, void* stream) {
    if (!dH || !Hact || !xhat || !inv_std || !sums || !dZ || E <= 0 || B <= 0 || H <= 0)
        return DCAHIP_EINVAL;
    BnBwdArgs a{dH, ldd, Hact, ldh, xhat, ldx, inv_std, sums, E, n_total, B, H, dZ, ldz, dbeta, act};
    const int grid = (B + kApplyRows - 1) / kApplyRows;
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid), dim3(256), 2 * H * sizeof(float),
                       static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int dcahip_bn_fused_max_rows() { return kFusedRows; }

extern "C" int dcahip_bn_relu_train_small(const float* Z, long ldz, int B, int H, const float* beta,
                                          float* moving_mean, float* moving_var, float momentum, float eps,
                                          int act, float* Hout, long ldh, float* xhat, long ldx,
                                          float* inv_std, void* stream) {
    if (!Z || !Hout || !moving_mean || !moving_var || B <= 0 || B > kFusedRows || H <= 0) return DCAHIP_EINVAL;
    BnApplyArgs a{Z, ldz, B, H, nullptr, nullptr, 0, beta, moving_mean, moving_var, momentum, eps,
                  act, Hout, ldh, xhat, ldx, inv_std};
    if (B <= 32) hipLaunchKernelGGL(bn_relu_small_kernel<8>, dim3((H + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(bn_relu_small_kernel<kFusedRows / 4>, dim3((H + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int dcahip_bn_bwd_small(const float* dH, long ldd, const float* Hact, long ldh,
                                   const float* xhat, long ldx, const float* inv_std, float n_total,
                                   int B, int H, float* dZ, long ldz, float* dbeta, int act, void* stream) {
    if (!dH || !Hact || !xhat || !inv_std || !dZ || B <= 0 || B > kFusedRows || H <= 0) return DCAHIP_EINVAL;
    BnBwdArgs a{dH, ldd, Hact, ldh, xhat, ldx, inv_std, nullptr, 0, n_total, B, H, dZ, ldz, dbeta, act};
    if (B <= 32) hipLaunchKernelGGL(bn_bwd_small_kernel<8>, dim3((H + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(bn_bwd_small_kernel<kFusedRows / 4>, dim3((H + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int dcahip_dense_small_max_k() { return kSmallK; }

extern "C" int dcahip_dense_bn_small(const float* Hp, long ldp, const float* W, long ldw, const float* bias,
                                     int B, int K, int H, int batchnorm, const float* beta,
                                     float* moving_mean, float* moving_var, float momentum, float eps, int act,
                                     float* Z, long ldz, float* xhat, long ldx, float* Hout, long ldh,
                                     float* inv_std, void* stream) {
    if (!Hp || !W || !bias || !Hout || B <= 0 || B > kFusedRows || K <= 0 || K > kSmallK || H <= 0) return DCAHIP_EINVAL;
    if (batchnorm && (!moving_mean || !moving_var)) return DCAHIP_EINVAL;
    DenseSmallArgs a{Hp, ldp, W, ldw, bias, B, K, H, batchnorm, beta, moving_mean, moving_var, momentum, eps, act,
                     Z, ldz, xhat, ldx, Hout, ldh, inv_std};
    if (B <= 32) hipLaunchKernelGGL(dense_bn_small_kernel<8>, dim3((H + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(dense_bn_small_kernel<16>, dim3((H + 63) / 64), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int dcahip_hidden_small_chain(const dcahip_small_layer* layers, int n, const float* Hin, long ldin, int B,
                                         int batchnorm, float momentum, float eps, int act, void* stream) {
    if (!layers || n < 1 || n > kChainMax || B <= 0 || B > kFusedRows) return DCAHIP_EINVAL;
    SmallChainArgs a{};
    for (int i = 0; i < n; ++i) {
        const dcahip_small_layer& q = layers[i];
        if (q.H <= 0 || q.H > 64 || !q.Hout) return DCAHIP_EINVAL;
        if (q.W) {
            if (!q.bias || q.K <= 0 || q.K > kSmallK) return DCAHIP_EINVAL;
            if (i == 0 && !Hin) return DCAHIP_EINVAL;
            if (i > 0 && q.K != layers[i - 1].H) return DCAHIP_EINVAL;
        } else if (!q.Z) return DCAHIP_EINVAL;
        if (batchnorm && (!q.moving_mean || !q.moving_var)) return DCAHIP_EINVAL;
        a.l[i] = SmallLayer{q.W, q.ldw, q.bias, q.K, q.H, q.beta, q.moving_mean, q.moving_var, q.Z, q.ldz, q.xhat, q.ldx,
                            q.Hout, q.ldh, q.inv_std};
    }
    a.Hin = Hin; a.ldin = ldin; a.n = n; a.B = B; a.batchnorm = batchnorm; a.act = act; a.momentum = momentum; a.eps = eps;
    if (B <= 32) hipLaunchKernelGGL(hidden_small_chain_kernel<8>, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(hidden_small_chain_kernel<16>, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int dcahip_dense_bn_bwd_small(const float* dH, long ldd, const float* Hact, long ldh,
                                         const float* xhat, long ldx, const float* inv_std,
                                         const float* Hp, long ldp, const float* W, long ldw,
                                         int B, int K, int H, int batchnorm, float n_total, int act,
                                         float* gW, long ldg, float* dbeta, float* dHp, long lddp, void* stream) {
    if (!dH || !Hact || !Hp || !W || !gW || B <= 0 || B > kFusedRows || K <= 0 || K > kSmallK || H <= 0 || H > kSmallK)
        return DCAHIP_EINVAL;
    if (batchnorm && (!xhat || !inv_std)) return DCAHIP_EINVAL;
    DenseSmallBwdArgs a{dH, ldd, Hact, ldh, xhat, ldx, inv_std, Hp, ldp, W, ldw, B, K, H, batchnorm, n_total, act,
                        gW, ldg, dbeta, dHp, lddp};
    if (B <= 32) hipLaunchKernelGGL(dense_bn_bwd_small_kernel<8>, dim3(kBwdWGs), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    else hipLaunchKernelGGL(dense_bn_bwd_small_kernel<16>, dim3(kBwdWGs), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int dcahip_relu_bwd(const float* dH, long ldd, const float* Hact, long ldh, int B, int H,
                               float* dZ, long ldz, int act, void* stream) {
    if (!dH || !Hact || !dZ || B <= 0 || H <= 0) return DCAHIP_EINVAL;
    long g = ((long)B * H + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(relu_bwd_kernel, dim3((int)g), dim3(256), 0, static_cast<hipStream_t>(stream),
                       dH, ldd, Hact, ldh, B, H, dZ, ldz, act);
    return (int)hipGetLastError();
}

extern "C" int dcahip_relu_fwd(const float* Z, long ldz, int B, int H, float* Hout, long ldh,
                               int act, void* stream) {
    if (!Z || !Hout || B <= 0 || H <= 0) return DCAHIP_EINVAL;
    long g = ((long)B * H + 255) / 256;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(relu_fwd_kernel, dim3((int)g), dim3(256), 0, static_cast<hipStream_t>(stream),
                       Z, ldz, B, H, Hout, ldh, act);
    return (int)hipGetLastError();
}

// Shared (per-cell scalar) heads of the *-shared networks: Dense(1) output -> [B, G] plane and back.
__global__ __launch_bounds__(256) void bcast_cols_kernel(const float* __restrict__ s, long lds, int B, int G,
                                                         float* __restrict__ out, long ldo) {
    const int r = blockIdx.y;
    const float v = s[(long)r * lds];
    for (int c = blockIdx.x * 256 + threadIdx.x; c < G; c += gridDim.x * 256) out[(long)r * ldo + c] = v;
}

__global__ __launch_bounds__(256) void row_sums_strided_kernel(const float* __restrict__ x, long ldx, int B, int G,
                                                               float* __restrict__ out, long ldo) {
    // one wave per row; fp64 lane partials over a fixed column assignment, butterfly in a fixed order: deterministic
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= B) return;
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int c = lane; c < G; c += 64) acc += (double)x[(long)r * ldx + c];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off, 64);
    if (lane == 0) out[(long)r * ldo] = (float)acc;
}

// keras.layers.PReLU (network.py:132-133, advanced_activations): f(x) = max(x, 0) + alpha_c min(x, 0) with one
// trainable slope per unit (alpha_initializer zeros).  Runs as its own element-wise layer behind the batch-norm /
// bias kernels (which then apply the linear activation).
__global__ __launch_bounds__(256) void prelu_fwd_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ alpha,
                                                        int B, int h, float* __restrict__ out, long ldo) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= h) return;
    const float a = alpha[c];
    for (int r = blockIdx.y; r < B; r += gridDim.y) {
        const float v = x[(long)r * ldx + c];
        out[(long)r * ldo + c] = v > 0.f ? v : a * v;
    }
}

__global__ __launch_bounds__(256) void prelu_bwd_kernel(float* __restrict__ d, long ldd, const float* __restrict__ x, long ldx,
                                                        const float* __restrict__ alpha, int B, int h,
                                                        double* __restrict__ partial) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= h) return;
    const float a = alpha[c];
    double sa = 0.0;
    for (int r = blockIdx.y; r < B; r += gridDim.y) {
        const float v = x[(long)r * ldx + c];
        const float g = d[(long)r * ldd + c];
        if (v > 0.f) continue;                    // slope 1, no alpha term
        sa += (double)(g * v);
        d[(long)r * ldd + c] = a * g;
    }
    partial[(long)blockIdx.y * h + c] = sa;
}

__global__ __launch_bounds__(256) void prelu_finish_kernel(const double* __restrict__ partial, int R, int h,
                                                           float* __restrict__ galpha) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= h) return;
    double s = 0.0;
    for (int y = 0; y < R; ++y) s += partial[(long)y * h + c];
    galpha[c] = (float)s;
}

constexpr int kPreluSlices = 32;

extern "C" int dcahip_prelu_workspace_doubles(int h) { return h > 0 ? kPreluSlices * h : 0; }

extern "C" int dcahip_prelu_fwd(const float* x, long ldx, const float* alpha, int B, int h, float* out, long ldo,
                                void* stream) {
    if (!x || !alpha || !out || B <= 0 || h <= 0 || ldx < h || ldo < h) return DCAHIP_EINVAL;
    const int gy = B < 64 ? B : 64;
    hipLaunchKernelGGL(prelu_fwd_kernel, dim3((h + 255) / 256, gy), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, ldx, alpha, B, h, out, ldo);
    return (int)hipGetLastError();
}

extern "C" int dcahip_prelu_bwd(float* d, long ldd, const float* x, long ldx, const float* alpha, int B, int h,
                                float* galpha, double* workspace, void* stream) {
    if (!d || !x || !alpha || !galpha || !workspace || B <= 0 || h <= 0 || ldx < h || ldd < h) return DCAHIP_EINVAL;
    const int R = B < kPreluSlices ? B : kPreluSlices;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(prelu_bwd_kernel, dim3((h + 255) / 256, R), dim3(256), 0, s, d, ldd, x, ldx, alpha, B, h, workspace);
    hipLaunchKernelGGL(prelu_finish_kernel, dim3((h + 255) / 256), dim3(256), 0, s, workspace, R, h, galpha);
    return (int)hipGetLastError();
}

// zinb-elempi (ZINBAutoencoderElemPi, network.py:424-461): m = -(Dense output) feeds MeanAct, and the dropout
// logit is an element-wise affine map of m (ElementwiseDense, layers.py:50-82): a_pi = k_g m + c_g.
__global__ __launch_bounds__(256) void elempi_fwd_kernel(float* __restrict__ a_mean, long lda, const float* __restrict__ k,
                                                         const float* __restrict__ c, int B, int G,
                                                         float* __restrict__ a_pi, long ldp) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const float kg = k[g], cg = c[g];
    for (int r = blockIdx.y; r < B; r += gridDim.y) {
        const float m = -a_mean[(long)r * lda + g];
        a_mean[(long)r * lda + g] = m;
        a_pi[(long)r * ldp + g] = kg * m + cg;
    }
}

// d_mean: in = dL/dm, out = dL/d(Dense output) = -(dL/dm + k_g dL/da_pi); partial[y][0][g] = sum_r dL/da_pi m,
// partial[y][1][g] = sum_r dL/da_pi over the rows of slice y (fixed assignment: deterministic)
__global__ __launch_bounds__(256) void elempi_bwd_kernel(const float* __restrict__ m, long lda, float* __restrict__ d_mean,
                                                         const float* __restrict__ d_pi, long ldd,
                                                         const float* __restrict__ k, int B, int G,
                                                         double* __restrict__ partial) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    const float kg = k[g];
    double sk = 0.0, sc = 0.0;
    for (int r = blockIdx.y; r < B; r += gridDim.y) {
        const float dp = d_pi[(long)r * ldd + g];
        sk += (double)(dp * m[(long)r * lda + g]);
        sc += (double)dp;
        d_mean[(long)r * ldd + g] = -(d_mean[(long)r * ldd + g] + kg * dp);
    }
    partial[((long)blockIdx.y * 2 + 0) * G + g] = sk;
    partial[((long)blockIdx.y * 2 + 1) * G + g] = sc;
}

__global__ __launch_bounds__(256) void elempi_finish_kernel(const double* __restrict__ partial, int R, int G,
                                                            float* __restrict__ gk, float* __restrict__ gc) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    double sk = 0.0, sc = 0.0;
    for (int y = 0; y < R; ++y) {
        sk += partial[((long)y * 2 + 0) * G + g];
        sc += partial[((long)y * 2 + 1) * G + g];
    }
    gk[g] = (float)sk;
    gc[g] = (float)sc;
}

constexpr int kElemPiSlices = 32;

extern "C" int dcahip_elempi_workspace_doubles(int G) { return G > 0 ? kElemPiSlices * 2 * G : 0; }

extern "C" int dcahip_elempi_fwd(float* a_mean, long lda, const float* k, const float* c, int B, int G,
                                 float* a_pi, long ldp, void* stream) {
    if (!a_mean || !k || !c || !a_pi || B <= 0 || G <= 0 || lda < G || ldp < G) return DCAHIP_EINVAL;
    const int gy = B < 64 ? B : 64;
    hipLaunchKernelGGL(elempi_fwd_kernel, dim3((G + 255) / 256, gy), dim3(256), 0, static_cast<hipStream_t>(stream),
                       a_mean, lda, k, c, B, G, a_pi, ldp);
    return (int)hipGetLastError();
}

extern "C" int dcahip_elempi_bwd(const float* m, long lda, float* d_mean, const float* d_pi, long ldd, const float* k,
                                 int B, int G, float* gk, float* gc, double* workspace, void* stream) {
    if (!m || !d_mean || !d_pi || !k || !gk || !gc || !workspace || B <= 0 || G <= 0 || lda < G || ldd < G)
        return DCAHIP_EINVAL;
    const int R = B < kElemPiSlices ? B : kElemPiSlices;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(elempi_bwd_kernel, dim3((G + 255) / 256, R), dim3(256), 0, s, m, lda, d_mean, d_pi, ldd, k, B, G,
                       workspace);
    hipLaunchKernelGGL(elempi_finish_kernel, dim3((G + 255) / 256), dim3(256), 0, s, workspace, R, G, gk, gc);
    return (int)hipGetLastError();
}

extern "C" int dcahip_bcast_cols(const float* s, long lds, int B, int G, float* out, long ldo, void* stream) {
    if (!s || !out || B <= 0 || G <= 0 || ldo < G || lds < 1) return DCAHIP_EINVAL;
    int gx = (G + 255) / 256;
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(bcast_cols_kernel, dim3(gx, B), dim3(256), 0, static_cast<hipStream_t>(stream), s, lds, B, G, out, ldo);
    return (int)hipGetLastError();
}

extern "C" int dcahip_row_sums_strided(const float* x, long ldx, int B, int G, float* out, long ldo, void* stream) {
    if (!x || !out || B <= 0 || G <= 0 || ldx < G || ldo < 1) return DCAHIP_EINVAL;
    hipLaunchKernelGGL(row_sums_strided_kernel, dim3((B + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       x, ldx, B, G, out, ldo);
    return (int)hipGetLastError();
}

extern "C" int dcahip_colsum_chain(const float* x, long ldx, int B, int N, const float* theta_w,
                                   float* out, void* stream) {
    if (!x || !out || B <= 0 || N <= 0) return DCAHIP_EINVAL;
    hipLaunchKernelGGL(colsum_chain_kernel, dim3((N + 63) / 64), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, ldx, B, N, theta_w, out);
    return (int)hipGetLastError();
}

extern "C" int dcahip_rmsprop_clip(float* w, const float* g, float* ms, long n, const float* lr,
                                   float rho, float eps, float clip, void* stream) {
    if (!w || !g || !ms || !lr || n <= 0) return DCAHIP_EINVAL;
    if (!al16(w) || !al16(g) || !al16(ms)) return DCAHIP_EINVAL;
    long grid = ((n >> 2) + 255) / 256;
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(rmsprop_clip_kernel, dim3((int)grid), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, g, ms, n, lr, rho, eps, clip, StepEnd{nullptr, 0.0, nullptr, 0, nullptr, nullptr, 0, 0});
    return (int)hipGetLastError();
}

extern "C" int dcahip_rmsprop_clip_end(float* w, const float* g, float* ms, long n, const float* lr,
                                       float rho, float eps, float clip, const float* loss, double weight,
                                       float* hist, int rows_per_slot, double* acc, long long* cursor, int advance,
                                       void* stream) {
    if (!w || !g || !ms || !lr || n <= 0) return DCAHIP_EINVAL;
    if (!al16(w) || !al16(g) || !al16(ms)) return DCAHIP_EINVAL;
    long grid = ((n >> 2) + 255) / 256;
    if (grid > 2048) grid = 2048;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(rmsprop_clip_kernel, dim3((int)grid), dim3(256), 0,
                       static_cast<hipStream_t>(stream), w, g, ms, n, lr, rho, eps, clip,
                       StepEnd{loss, weight, hist, rows_per_slot, acc, cursor, advance, 1});
    return (int)hipGetLastError();
}
