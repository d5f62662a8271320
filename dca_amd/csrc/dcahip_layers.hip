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
// layers wider than 64 units: the 64-column strips of the BatchNorm kernels go to grid.y while the row blocks alone do not
// fill the chip (512 units at batch 2048: 32 row chunks x 8 strips instead of 32 workgroups walking 8 strips each)
inline int strip_blocks(int H, int row_blocks) {
    const int strips = (H + 63) / 64;
    int y = row_blocks > 0 ? 512 / row_blocks : strips;
    if (y < 1) y = 1;
    return y < strips ? y : strips;
}

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
    for (int c0 = blockIdx.y * 64; c0 < H; c0 += 64 * gridDim.y) {     // column strips spread over grid.y
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
                                                              float* out, int Bfallback = 0) {
    __shared__ double smd[512];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    double n, mean, m2;
    merge_entries_wg(entries, counts, E, H, c, Bfallback, smd, n, mean, m2);
    if (threadIdx.x < 64 && c < H) {
        out[c] = (float)mean;
        out[H + c] = (float)m2;
    }
}

// out[2][H] = sum over E partial pairs [E][2][H], in entry order (K-STACK's backward sums of one rank, ahead of the all-reduce)
// (dbeta: this rank's LOCAL share of d beta = its own sum of dy -- the gradient bucket is summed over the ranks afterwards)
__global__ __launch_bounds__(64) void stack_sums_combine_kernel(const float* part, int E, int H, float* out, float* dbeta) {
    const int c = threadIdx.x;
    if (c >= H) return;
    float v1 = 0.f, v2 = 0.f;
    for (int e0 = 0; e0 < E; e0 += 8) {
        float a1[8], a2[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long eo = (long)(e0 + u < E ? e0 + u : E - 1) * 2;
            a1[u] = part[(eo + 0) * H + c]; a2[u] = part[(eo + 1) * H + c];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) if (e0 + u < E) { v1 += a1[u]; v2 += a2[u]; }
    }
    out[c] = v1; out[H + c] = v2;
    if (dbeta) dbeta[c] = v1;
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
    for (int c0 = blockIdx.y * 64; c0 < a.H; c0 += 64 * gridDim.y) {
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
    for (int c0 = blockIdx.y * 64; c0 < a.H; c0 += 64 * gridDim.y) {
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
    for (int c0 = blockIdx.y * 64; c0 < H; c0 += 64 * gridDim.y) {
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
    for (int c0 = blockIdx.y * 64; c0 < a.H; c0 += 64 * gridDim.y) {
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
    for (int c0 = blockIdx.y * 64; c0 < a.H; c0 += 64 * gridDim.y) {
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
    __shared__ float Ws[2][kSmallK][64];        // kernels of two consecutive stages (stage st's in Ws[st & 1])
    __shared__ float Ps[kChainMax][4][64];      // per-column inputs of every stage: bias, moving mean, moving variance, beta
    __shared__ float sm[256];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int c = tx;
    // ---- EVERYTHING the launch reads that does not depend on its own results is requested here, in one batch: the kernels of
    // the first two stages that have one, every stage's per-column inputs (wave w takes stage w), the first stage's input.
    // One memory round trip in front of the chain instead of one per stage (each was exposed: a stage's arithmetic is shorter
    // than a trip to HBM) -- the launch is a chain of dependent latencies, nothing else.
    const int s0 = a.l[0].W ? 0 : 1;             // first stage with a kernel (stage 0 may only normalise the first layer's product)
    float wa[16], wb[16];
    {
        const SmallLayer& A0 = a.l[s0 < a.n ? s0 : 0];
        const SmallLayer& A1 = a.l[s0 + 1 < a.n ? s0 + 1 : 0];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = threadIdx.x + 256 * u, k = idx >> 6, cc = idx & 63;
            wa[u] = (s0 < a.n) ? A0.W[(long)(k < A0.K ? k : A0.K - 1) * A0.ldw + (cc < A0.H ? cc : A0.H - 1)] : 0.f;
            wb[u] = (s0 + 1 < a.n) ? A1.W[(long)(k < A1.K ? k : A1.K - 1) * A1.ldw + (cc < A1.H ? cc : A1.H - 1)] : 0.f;
        }
    }
    float pin[4] = {0.f, 0.f, 0.f, 0.f};
    if (ty < a.n) {
        const SmallLayer& P = a.l[ty];
        const int cc = tx < P.H ? tx : P.H - 1;
        if (P.W) pin[0] = P.bias[cc];
        if (a.batchnorm) { pin[1] = P.mm[cc]; pin[2] = P.mv[cc]; if (P.beta) pin[3] = P.beta[cc]; }
    }
    float z0[RPT];                               // stage 0 without a kernel: its Z; with one: its input rows
#pragma unroll
    for (int k = 0; k < RPT; ++k) {
        if (a.l[0].W) {
            const int idx = threadIdx.x + 256 * k, r = idx >> 6, kk = idx & 63;
            z0[k] = a.Hin[(long)(r < a.B ? r : a.B - 1) * a.ldin + (kk < a.l[0].K ? kk : a.l[0].K - 1)];
        } else {
            const int i = ty + 4 * k;
            z0[k] = a.l[0].Z[(long)(i < a.B ? i : a.B - 1) * a.l[0].ldz + (c < a.l[0].H ? c : a.l[0].H - 1)];
        }
    }
    {
        const int K0 = a.l[s0 < a.n ? s0 : 0].K, H0 = a.l[s0 < a.n ? s0 : 0].H;
        const int K1 = a.l[s0 + 1 < a.n ? s0 + 1 : 0].K, H1 = a.l[s0 + 1 < a.n ? s0 + 1 : 0].H;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int idx = threadIdx.x + 256 * u, k = idx >> 6, cc = idx & 63;
            if (s0 < a.n) Ws[s0 & 1][k][cc] = (k < K0 && cc < H0) ? wa[u] : 0.f;
            if (s0 + 1 < a.n) Ws[(s0 + 1) & 1][k][cc] = (k < K1 && cc < H1) ? wb[u] : 0.f;
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) Ps[ty][q][tx] = pin[q];
    if (a.l[0].W) {
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int idx = threadIdx.x + 256 * k, r = idx >> 6, kk = idx & 63;
            Hs[r][kk] = kk < a.l[0].K ? z0[k] : 0.f;
        }
    }
    __syncthreads();
#pragma unroll 1
    for (int st = 0; st < a.n; ++st) {
        const SmallLayer& L = a.l[st];
        const bool cv = c < L.H;
        // the kernel of the stage after the next (buffers: two): requested now, stored when this stage's products are done
        const bool has_n2 = st >= s0 && st + 2 < a.n;
        const SmallLayer& Nx = a.l[has_n2 ? st + 2 : st];
        float wn[16];
        if (has_n2) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int idx = threadIdx.x + 256 * u, k = idx >> 6, cc = idx & 63;
                wn[u] = Nx.W[(long)(k < Nx.K ? k : Nx.K - 1) * Nx.ldw + (cc < Nx.H ? cc : Nx.H - 1)];
            }
        }
        const float mm_in = Ps[st][1][tx], mv_in = Ps[st][2][tx], beta_in = Ps[st][3][tx];
        float z[RPT];
        if (L.W) {
            const int K4 = (L.K + 3) & ~3;
            const float b = cv ? Ps[st][0][tx] : 0.f;
#pragma unroll
            for (int k = 0; k < RPT; ++k) z[k] = b;
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
            for (int k = 0; k < RPT; ++k) z[k] = z0[k];
        }
        float hout[RPT];
        if (!a.batchnorm) {
#pragma unroll
            for (int k = 0; k < RPT; ++k) hout[k] = act_fwd(a.act, z[k]);
            __syncthreads();            // every thread is done reading Hs and Ws[st & 1]
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
        if (has_n2) {
            float (*Wn)[64] = Ws[st & 1];            // (this stage's kernel: every thread is past its products -- the barriers above)
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int idx = threadIdx.x + 256 * u, k = idx >> 6, cc = idx & 63;
                Wn[k][cc] = (k < Nx.K && cc < Nx.H) ? wn[u] : 0.f;
            }
        }
        __syncthreads();                // Hs (and a replaced kernel) complete before the next stage reads them
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
            wp[j] = wp[j] - lr * gj / (sqrtf(mp[j]) + eps);
        }
        reinterpret_cast<float4*>(ms)[i] = mv;
        reinterpret_cast<float4*>(w)[i] = wv;
    }
    for (long i = (nv << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float gj = g[i];
        if (clip > 0.f) gj = fminf(fmaxf(gj, -clip), clip);
        const float m = rho * ms[i] + (1.f - rho) * gj * gj;
        ms[i] = m;
        w[i] = w[i] - lr * gj / (sqrtf(m) + eps);
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

// =====================================================================================================
// K-STACK: the hidden stack behind the first layer's product at throughput batches, ONE launch per direction.
// With one launch per operation the 64-32-64 stack of the benchmark network costs 22 launches of ~5 us each per step
// (launch floor + one memory round trip: the whole activation set at 4096 rows is 2.6 MB) -- 10 % of the step.  Here a
// workgroup owns a block of batch rows for the whole pass and keeps its tiles in LDS; the only thing workgroups
// exchange are the batch-norm statistics (forward: one (mean, M2) pair per workgroup and layer; backward: the two sums
// of dcahip_bn_bwd_sums) and, at the end of the backward pass, the weight-gradient partials -- through grid-wide
// barriers (an arrival counter polled with agent-scope loads, release before / acquire after: the forms
// MI355X_MICROARCH.md lists as valid).  Every workgroup must be resident at once: at most 256 of them, launched on an
// otherwise idle device (the stream order of the training step guarantees it); every spin is bounded.
// Same formulas, same merge (merge_entries_wg) and same summation structure as the per-operation kernels above.
// =====================================================================================================
constexpr int kStackMaxLayers = 8;
constexpr int kStackRows = 64;          // most rows a workgroup owns
constexpr int kStackMaxWG = 256;
constexpr int kStackLd = 68;            // LDS row stride (floats): 16-byte aligned rows, rows 4 banks apart

struct StackBwdLayer {                  // = dcahip_stack_bwd_layer (include/dcahip.h)
    const float* W; long ldw; int K, H;
    const float* Hact; long ldh; const float* xhat; long ldx; const float* inv_std;
    const float* Hprev; long ldp;
    float* gW; long ldg; float* dbeta;
    float* dHin; long lddh;             // gradient w.r.t. this layer's output (read at the start of a launch; written for the layer below)
};

struct StackFwdArgs {
    SmallLayer l[kStackMaxLayers];
    int n, B, act, nwg, first, last;
    float momentum, eps;
    float* part;                        // [n][nwg][2][64]
    unsigned* sync;                     // [0] arrivals, [1] exits, [2] error flag
};

struct StackBwdArgs {
    StackBwdLayer l[kStackMaxLayers];
    int n, B, act, nwg, first, last;
    float n_total;
    float* dZ0; long ldz0;              // OUT: gradient w.r.t. the first layer's pre-activation
    float* part;                        // [n][nwg][2][64]
    float* gwp;                         // [n][nwg][65][64] weight (+ bias) gradient partials
    unsigned* sync;
};

__device__ __forceinline__ void grid_barrier(unsigned* sync, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) {                      // never hang the device: flag the error and go on
                __hip_atomic_store(sync + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// the last workgroup to leave puts the counters back to zero for the next launch
__device__ __forceinline__ void grid_exit(unsigned* sync, unsigned nwg) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (t == nwg - 1) {
            __hip_atomic_store(sync, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sync + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// merge_entries_wg for K-STACK: the same two-pass fp64 merge of the workgroups' (mean, M2) pairs (counts from the row
// partition), with ALL of a thread's loads of a pass in flight at once -- the entries were written by other CUs and
// come from their L2 slices: a dependent load per entry costs a fabric round trip each
__device__ __forceinline__ void stack_merge(const float* entries, int E, int H, int c, int B, double* sm /*[2][256]*/,
                                            double& n_out, double& mean_out, double& m2_out) {
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int cc = c < H ? c : H - 1;
    double n = 0.0, sw = 0.0;
    for (int e0 = ty; e0 < E; e0 += 64) {
        float me[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) me[u] = entries[((long)(e0 + 4 * u < E ? e0 + 4 * u : E - 1) * 2 + 0) * H + cc];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = e0 + 4 * u;
            const double ne = (e < E && c < H) ? entry_count(nullptr, e, E, B) : 0.0;
            n += ne; sw += ne * (double)me[u];
        }
    }
    __syncthreads();
    sm[ty * 64 + tx] = n; sm[256 + ty * 64 + tx] = sw;
    __syncthreads();
    n = (sm[tx] + sm[64 + tx]) + (sm[128 + tx] + sm[192 + tx]);
    sw = (sm[256 + tx] + sm[320 + tx]) + (sm[384 + tx] + sm[448 + tx]);
    const double mean = n > 0.0 ? sw / n : 0.0;
    double q = 0.0;
    for (int e0 = ty; e0 < E; e0 += 64) {
        float me[16], qe[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const long eo = (long)(e0 + 4 * u < E ? e0 + 4 * u : E - 1) * 2;
            me[u] = entries[(eo + 0) * H + cc]; qe[u] = entries[(eo + 1) * H + cc];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int e = e0 + 4 * u;
            const double ne = (e < E && c < H) ? entry_count(nullptr, e, E, B) : 0.0;
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

// Steps of the forward pass: 0 = partial statistics of the first layer's pre-activation; 1 + i = layer i: merge the
// statistics, normalise + activate the block's rows, next layer's pre-activation of the block and ITS partial statistics.
// A launch runs steps [first, last]: all of them (cooperative: a grid barrier between steps) or one per launch (the
// kernel boundary is the barrier; the block's tile is re-read from memory at the start of the step).
__global__ __launch_bounds__(256) void hidden_stack_fwd_kernel(StackFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float zt[kStackRows * kStackLd];     // pre-activations of the current layer
    __shared__ __attribute__((aligned(16))) float ht[kStackRows * kStackLd];     // its activations (the next layer's input)
    __shared__ __attribute__((aligned(16))) float wl[64 * kStackLd];             // the next layer's kernel
    __shared__ float smf[256];
    __shared__ double smd[512];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int wg = blockIdx.x, nwg = a.nwg;
    const int cr = chunk_rows(a.B, nwg);
    const int r0 = wg * cr;
    const int nrows = max(0, min(a.B, r0 + cr) - r0);
    const int c = tx;
    unsigned nbar = 0;
    auto block_stats = [&](int i) {                       // (mean, M2) of the block's rows of layer i, from zt
        const int H = a.l[i].H;
        float sv = 0.f;
        if (c < H) for (int r = ty; r < nrows; r += 4) sv += zt[r * kStackLd + c];
        const float tot = wg_rowlane_sum(sv, smf);
        const float bmean = nrows > 0 ? tot / (float)nrows : 0.f;
        float q = 0.f;
        if (c < H) for (int r = ty; r < nrows; r += 4) { const float d = zt[r * kStackLd + c] - bmean; q += d * d; }
        const float bm2 = wg_rowlane_sum(q, smf);
        float* part = a.part + (long)i * nwg * 2 * 64;
        if (c < H && ty == 0) {
            part[((long)wg * 2 + 0) * H + c] = bmean;
            part[((long)wg * 2 + 1) * H + c] = bm2;
        }
    };
    {   // the block's rows of the pre-activation the first step of this launch works on
        const SmallLayer& L = a.l[a.first > 0 ? a.first - 1 : 0];
        for (int r = ty; r < nrows; r += 4)
            if (tx < L.H) zt[r * kStackLd + tx] = L.Z[(long)(r0 + r) * L.ldz + tx];
        __syncthreads();
    }
    bool fresh = false;                                   // statistics written by THIS launch: a grid barrier before their merge
    if (a.first == 0) { block_stats(0); fresh = true; }
    for (int step = max(a.first, 1); step <= a.last; ++step) {
        const int i = step - 1;
        const SmallLayer& L = a.l[i];
        const int H = L.H;
        if (fresh) grid_barrier(a.sync, (unsigned)nwg * (++nbar));
        // ---- statistics of the whole batch: every workgroup merges the same entries in the same order
        const float* part = a.part + (long)i * nwg * 2 * 64;
        double n_, m_, m2_;
        stack_merge(part, nwg, H, c, a.B, smd, n_, m_, m2_);
        const float mean = (float)m_;
        const float var = (float)(m2_ / n_);
        const float inv = 1.f / sqrtf(var + a.eps);
        if (wg == 0 && ty == 0 && c < H) {
            L.mm[c] = L.mm[c] - (L.mm[c] - mean) * (1.f - a.momentum);
            L.mv[c] = L.mv[c] - (L.mv[c] - var) * (1.f - a.momentum);
            if (L.inv_std) L.inv_std[c] = inv;
        }
        // ---- normalise + activation of the block's rows
        if (c < H) {
            const float beta = L.beta ? L.beta[c] : 0.f;
            for (int r = ty; r < nrows; r += 4) {
                const float xh = (zt[r * kStackLd + c] - mean) * inv;
                const float h = act_fwd(a.act, xh + beta);
                if (L.xhat) L.xhat[(long)(r0 + r) * L.ldx + c] = xh;
                L.Hout[(long)(r0 + r) * L.ldh + c] = h;
                ht[r * kStackLd + c] = h;
            }
        }
        if (i + 1 == a.n) break;
        // ---- the next layer's pre-activations of the block's rows: zt = ht W + b
        const SmallLayer& N = a.l[i + 1];
        const int K = H, HN = N.H;
        for (int idx = tid; idx < ((K + 3) & ~3) * 64; idx += 256) {
            const int k = idx >> 6, cc = idx & 63;
            wl[k * kStackLd + cc] = (k < K && cc < HN) ? N.W[(long)k * N.ldw + cc] : 0.f;
        }
        for (int idx = tid; idx < kStackRows * 4; idx += 256) {         // zero the k padding of ht (K < 64 rounded up to 4)
            const int r = idx >> 2, kk = (K & ~3) + (idx & 3);
            if (kk >= K && kk < ((K + 3) & ~3)) ht[r * kStackLd + kk] = 0.f;
        }
        __syncthreads();
        float acc[kStackRows / 4];
#pragma unroll
        for (int j = 0; j < kStackRows / 4; ++j) acc[j] = 0.f;
        const int K4 = (K + 3) & ~3;
        for (int k = 0; k < K4; k += 4) {
            const float w0 = wl[(k + 0) * kStackLd + c], w1 = wl[(k + 1) * kStackLd + c];
            const float w2 = wl[(k + 2) * kStackLd + c], w3 = wl[(k + 3) * kStackLd + c];
#pragma unroll
            for (int j = 0; j < kStackRows / 4; ++j) {
                const int r = ty + 4 * j;
                if (r < nrows) {
                    const float4 h4 = *reinterpret_cast<const float4*>(ht + r * kStackLd + k);
                    acc[j] = fmaf(h4.w, w3, fmaf(h4.z, w2, fmaf(h4.y, w1, fmaf(h4.x, w0, acc[j]))));
                }
            }
        }
        __syncthreads();                                  // every read of zt / ht of this layer is done
        if (c < HN) {
            const float b = N.bias ? N.bias[c] : 0.f;
#pragma unroll
            for (int j = 0; j < kStackRows / 4; ++j) {
                const int r = ty + 4 * j;
                if (r < nrows) {
                    const float z = acc[j] + b;
                    zt[r * kStackLd + c] = z;
                    if (N.Z) N.Z[(long)(r0 + r) * N.ldz + c] = z;
                }
            }
        }
        __syncthreads();
        block_stats(i + 1);
        fresh = true;
    }
    if (nbar) grid_exit(a.sync, (unsigned)nwg);
}

// Steps of the backward pass (n layers): 0 = dy and the two batch sums of the last layer; 1 + j = layer i = n - 1 - j:
// merge the sums, dZ, d beta, the block's weight-gradient partial, the gradient w.r.t. the layer's input and, for the
// layer below, dy and ITS sums; n + 1 = add the weight-gradient partials of the blocks.
__global__ __launch_bounds__(256) void hidden_stack_bwd_kernel(StackBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float dyt[kStackRows * kStackLd];    // dH -> dy -> dZ of the current layer
    __shared__ __attribute__((aligned(16))) float xt[kStackRows * kStackLd];     // xhat of the block's rows
    __shared__ __attribute__((aligned(16))) float ht[kStackRows * kStackLd];     // the layer's input activations
    __shared__ __attribute__((aligned(16))) float wl[64 * kStackLd];             // the layer's kernel
    __shared__ float smf[256];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int wg = blockIdx.x, nwg = a.nwg;
    const int cr = chunk_rows(a.B, nwg);
    const int r0 = wg * cr;
    const int nrows = max(0, min(a.B, r0 + cr) - r0);
    const int c = tx;
    unsigned nbar = 0;
    // dy = dH act'(h) of layer i into dyt (+ xhat into xt); with_sums: the block's share of the two batch sums -> part
    auto prep = [&](int i, bool load_dh, bool with_sums) {
        const StackBwdLayer& L = a.l[i];
        const int H = L.H;
        if (load_dh) {
            for (int r = ty; r < kStackRows; r += 4)
                dyt[r * kStackLd + tx] = (r < nrows && tx < H) ? L.dHin[(long)(r0 + r) * L.lddh + tx] : 0.f;
        }
        float s1 = 0.f, s2 = 0.f;
        if (c < H)
            for (int r = ty; r < nrows; r += 4) {
                const float h = L.Hact[(long)(r0 + r) * L.ldh + c];
                const float xh = L.xhat[(long)(r0 + r) * L.ldx + c];
                const float dy = dyt[r * kStackLd + c] * act_grad(a.act, h);
                dyt[r * kStackLd + c] = dy;
                xt[r * kStackLd + c] = xh;
                s1 += dy; s2 += dy * xh;
            }
        if (!with_sums) { __syncthreads(); return; }
        const float t1 = wg_rowlane_sum(s1, smf);
        const float t2 = wg_rowlane_sum(s2, smf);
        float* part = a.part + (long)i * nwg * 2 * 64;
        if (c < H && ty == 0) {
            part[((long)wg * 2 + 0) * H + c] = t1;
            part[((long)wg * 2 + 1) * H + c] = t2;
        }
    };
    bool fresh = false, have_tile = false;
    if (a.first == 0) { prep(a.n - 1, true, true); fresh = true; have_tile = true; }
    for (int step = max(a.first, 1); step <= min(a.last, a.n); ++step) {
        const int i = a.n - step;
        const StackBwdLayer& L = a.l[i];
        const int H = L.H;
        if (!have_tile) { prep(i, true, false); have_tile = true; }
        if (fresh) grid_barrier(a.sync, (unsigned)nwg * (++nbar));
        const float* part = a.part + (long)i * nwg * 2 * 64;
        float v1 = 0.f, v2 = 0.f;
        {
            const int cc = c < H ? c : H - 1;
            for (int e0 = ty; e0 < nwg; e0 += 64) {           // 16 entries per thread and batch: 32 loads in flight
                float p1[16], p2[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) {
                    const long eo = (long)(e0 + 4 * u < nwg ? e0 + 4 * u : nwg - 1) * 2;
                    p1[u] = part[(eo + 0) * H + cc]; p2[u] = part[(eo + 1) * H + cc];
                }
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (e0 + 4 * u < nwg && c < H) { v1 += p1[u]; v2 += p2[u]; }
            }
        }
        v1 = wg_rowlane_sum(v1, smf);
        v2 = wg_rowlane_sum(v2, smf);
        if (wg == 0 && ty == 0 && c < H && L.dbeta) L.dbeta[c] = v1;
        if (c < H) {
            const float m1 = v1 / a.n_total, m2 = v2 / a.n_total, inv = L.inv_std[c];
            for (int r = ty; r < nrows; r += 4) {
                const float dz = inv * (dyt[r * kStackLd + c] - m1 - xt[r * kStackLd + c] * m2);
                dyt[r * kStackLd + c] = dz;
                if (i == 0) a.dZ0[(long)(r0 + r) * a.ldz0 + c] = dz;
            }
        }
        if (i == 0) { fresh = true; break; }
        // ---- this layer's kernel and input activations -> LDS
        const int K = L.K;
        for (int idx = tid; idx < K * 64; idx += 256) {
            const int k = idx >> 6, cc = idx & 63;
            wl[k * kStackLd + cc] = cc < H ? L.W[(long)k * L.ldw + cc] : 0.f;
        }
        for (int idx = tid; idx < kStackRows * 64; idx += 256) {
            const int r = idx >> 6, k = idx & 63;
            ht[r * kStackLd + k] = (r < nrows && k < K) ? L.Hprev[(long)(r0 + r) * L.ldp + k] : 0.f;
        }
        __syncthreads();
        // ---- weight-gradient partial of the block: gW[k][c] = sum_r Hprev[r][k] dz[r][c], k in [16 ty, 16 ty + 16)
        float* gw = a.gwp + ((long)i * nwg + wg) * 65 * 64;
        {
            float acc[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
            float bs = 0.f;
            for (int r = 0; r < nrows; ++r) {
                const float d = dyt[r * kStackLd + c];
                bs += d;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 h4 = *reinterpret_cast<const float4*>(ht + r * kStackLd + 16 * ty + 4 * q4);
                    acc[4 * q4 + 0] = fmaf(h4.x, d, acc[4 * q4 + 0]); acc[4 * q4 + 1] = fmaf(h4.y, d, acc[4 * q4 + 1]);
                    acc[4 * q4 + 2] = fmaf(h4.z, d, acc[4 * q4 + 2]); acc[4 * q4 + 3] = fmaf(h4.w, d, acc[4 * q4 + 3]);
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int k = 16 * ty + j;
                if (k < K) gw[(long)k * 64 + c] = acc[j];
            }
            if (ty == 0) gw[(long)K * 64 + c] = bs;            // bias gradient = column sums of dZ
        }
        // ---- gradient w.r.t. the layer's input, block rows: dHp[r][k] = sum_c dz[r][c] W[k][c]; thread: k = tx, rows ty + 4 j
        {
            float acc[kStackRows / 4];
#pragma unroll
            for (int j = 0; j < kStackRows / 4; ++j) acc[j] = 0.f;
            for (int c4 = 0; c4 < 64; c4 += 4) {
                const float4 w4 = *reinterpret_cast<const float4*>(wl + tx * kStackLd + c4);
#pragma unroll
                for (int j = 0; j < kStackRows / 4; ++j) {
                    const int r = ty + 4 * j;
                    if (r < nrows) {
                        const float4 d4 = *reinterpret_cast<const float4*>(dyt + r * kStackLd + c4);
                        acc[j] = fmaf(d4.w, w4.w, fmaf(d4.z, w4.z, fmaf(d4.y, w4.y, fmaf(d4.x, w4.x, acc[j]))));
                    }
                }
            }
            __syncthreads();                              // every read of dyt of this layer is done
            const StackBwdLayer& P = a.l[i - 1];
#pragma unroll
            for (int j = 0; j < kStackRows / 4; ++j) {
                const int r = ty + 4 * j;
                const float v = (r < nrows && tx < K) ? acc[j] : 0.f;
                dyt[r * kStackLd + tx] = v;
                if (r < nrows && tx < K && P.dHin) P.dHin[(long)(r0 + r) * P.lddh + tx] = v;     // for a later launch
            }
        }
        __syncthreads();
        prep(i - 1, false, true);                         // the layer below: dy and its sums, from the tile just made
        fresh = true;
    }
    if (a.last > a.n) {
        // ---- weight gradients: element e of the layers' [K + 1, H] blocks is the sum of the blocks' partials in order
        // (thread = element e0 + tid % 32, slice tid / 32 of the blocks: its share of the partials in flight at once, the
        // 8 slices added in order through LDS)
        if (fresh) grid_barrier(a.sync, (unsigned)nwg * (++nbar));
        for (int i = 1; i < a.n; ++i) {
            const StackBwdLayer& L = a.l[i];
            const int total = (L.K + 1) * L.H;
            const float* gw = a.gwp + (long)i * nwg * 65 * 64;
            const int el = tid & 31, sl = tid >> 5;
            const int per = (nwg + 7) / 8;                    // blocks per slice
            for (int e0 = wg * 32; e0 < total; e0 += nwg * 32) {
                const int e = e0 + el;
                const int ec = e < total ? e : total - 1;
                const int k = ec / L.H, cc = ec - k * L.H;
                const float* src = gw + (long)k * 64 + cc;
                float v = 0.f;
                for (int b0 = sl * per; b0 < min(nwg, (sl + 1) * per); b0 += 16) {
                    float pv[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) pv[u] = src[(long)min(b0 + u, nwg - 1) * 65 * 64];
#pragma unroll
                    for (int u = 0; u < 16; ++u) if (b0 + u < min(nwg, (sl + 1) * per)) v += pv[u];
                }
                __syncthreads();
                smf[tid] = v;
                __syncthreads();
                if (tid < 32 && e < total) {
                    float t = 0.f;
#pragma unroll
                    for (int q8 = 0; q8 < 8; ++q8) t += smf[q8 * 32 + tid];
                    L.gW[(long)k * L.ldg + cc] = t;
                }
            }
        }
    }
    if (nbar) grid_exit(a.sync, (unsigned)nwg);
}

// A [K <= 64, H <= 64] kernel into registers (float4 number tid + 256 q, 16 per row): every load UNCONDITIONAL on a clamped
// address, no control flow between them, so they all leave back to back; w_tile_store zeroes what lies outside.
// (VEC: 0 = decide here whether rows can be read as float4, 1 / 2 = the caller knows they can / cannot -- no branch)
__device__ __forceinline__ bool w_tile_vec(const float* W, long ldw, int H) {
    return (H & 3) == 0 && (ldw & 3) == 0 && (reinterpret_cast<uintptr_t>(W) & 15) == 0;
}
template <int VEC = 0>
__device__ __forceinline__ void w_tile_request(const float* W, long ldw, int K, int H, int tid, float4 (&wv)[4]) {
    if (VEC == 0 && (!W || K <= 0)) {                      // (uniform)
#pragma unroll
        for (int q = 0; q < 4; ++q) wv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const bool vec = VEC == 1 || (VEC == 0 && w_tile_vec(W, ldw, H));
    if (vec) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 256 * q, k = min(idx >> 4, K - 1), c4 = min((idx & 15) * 4, H - 4);
            wv[q] = *reinterpret_cast<const float4*>(W + (long)k * ldw + c4);
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int idx = tid + 256 * q, k = min(idx >> 4, K - 1), c4 = (idx & 15) * 4;
            const float* src = W + (long)k * ldw;
            wv[q].x = src[min(c4 + 0, H - 1)]; wv[q].y = src[min(c4 + 1, H - 1)];
            wv[q].z = src[min(c4 + 2, H - 1)]; wv[q].w = src[min(c4 + 3, H - 1)];
        }
    }
}
__device__ __forceinline__ void w_tile_store(float* wl, int K, int H, int tid, const float4 (&wv)[4]) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = tid + 256 * q, k = idx >> 4, c4 = (idx & 15) * 4;
        float4 v = wv[q];
        if (k >= K || c4 + 0 >= H) v.x = 0.f;
        if (k >= K || c4 + 1 >= H) v.y = 0.f;
        if (k >= K || c4 + 2 >= H) v.z = 0.f;
        if (k >= K || c4 + 3 >= H) v.w = 0.f;
        *reinterpret_cast<float4*>(wl + k * kStackLd + c4) = v;
    }
}

// ---- K-STACK, one step per launch: the same steps with every global read of a step issued UP FRONT (the block's tile,
// the other blocks' partial statistics, the layer's kernel: all independent of each other), so a launch waits for ONE
// memory round trip instead of one per phase -- at these sizes a step is nothing but latency.
constexpr int kStepRows = 32;           // rows per workgroup: 8 per row lane

struct StepFwdArgs {
    SmallLayer cur, nxt;                // layer i and layer i + 1 (nxt.W == NULL: none)
    int B, act, nwg, stats_only;
    float momentum, eps;
    const float* part_in;               // [nwg][2][H] (mean, M2) of layer i (unused with stats_only)
    float* part_out;                    // [nwg][2][H'] of layer i + 1 (stats_only: of layer i)
    // data parallel (SyncBN): the statistics of layer i come from outside instead -- one (mean, M2) entry per RANK with
    // its row count, gathered between two launches (dcahip_hidden_stack_fwd_sync); the row partition stays this rank's
    const float* ext_in; const float* ext_counts; int ext_E;
};

__global__ __launch_bounds__(256) void stack_fwd_step_kernel(StepFwdArgs a) {
    __shared__ __attribute__((aligned(16))) float ht[kStepRows * kStackLd];
    __shared__ __attribute__((aligned(16))) float wl[64 * kStackLd];
    __shared__ float smf[256];
    __shared__ double smd[1024];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int wg = blockIdx.x, E = a.nwg;
    const int cr = chunk_rows(a.B, E);
    const int r0 = wg * cr;
    const int nrows = max(0, min(a.B, r0 + cr) - r0);
    const int c = tx;
    const SmallLayer& L = a.cur;
    const int H = L.H, cc = c < H ? c : H - 1;
    // ---- every read of the step, up front
    float z[kStepRows / 4];
#pragma unroll
    for (int j = 0; j < kStepRows / 4; ++j) {
        const int r = ty + 4 * j;
        z[j] = L.Z[(long)(r0 + (r < nrows ? r : (nrows > 0 ? nrows - 1 : 0))) * L.ldz + cc];
    }
    float out_z[kStepRows / 4];
    int Hs = H;                                            // width of the layer whose block statistics this launch writes
    if (!a.stats_only) {
        const float* const ein = a.ext_in ? a.ext_in : a.part_in;      // whose statistics: every rank's, or this launch grid's
        const float* const cnt = a.ext_in ? a.ext_counts : nullptr;
        const int Em = a.ext_in ? a.ext_E : E;
        float pm[32], pq[32];
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const long eo = (long)(ty + 4 * u < Em ? ty + 4 * u : Em - 1) * 2;
            pm[u] = ein[(eo + 0) * H + cc]; pq[u] = ein[(eo + 1) * H + cc];
        }
        const SmallLayer& N = a.nxt;
        const int K = H, HN = N.W ? N.H : 0;
        float4 wv[4];
        w_tile_request(N.W, N.ldw, N.W ? K : 0, HN, tid, wv);      // the next layer's kernel [K, HN]
        const float beta = L.beta ? L.beta[cc] : 0.f;
        const float nbias = (N.W && N.bias && c < HN) ? N.bias[c] : 0.f;
        // ---- statistics of the whole batch (one pass, fp64: N, sum n m, sum (M2 + n m^2)), every block the same order
        double n = 0.0, sw = 0.0, sq = 0.0;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int e = ty + 4 * u;
            const double ne = (e < Em && c < H) ? entry_count(cnt, e, Em, a.B) : 0.0;
            n += ne; sw += ne * (double)pm[u]; sq += ne > 0.0 ? (double)pq[u] + ne * (double)pm[u] * (double)pm[u] : 0.0;
        }
        for (int e0 = ty + 128; e0 < Em; e0 += 128) {      // (more than 128 entries: further batches)
#pragma unroll 4
            for (int u = 0; u < 32; ++u) {
                const int e = e0 + 4 * u;
                if (e < Em && c < H) {
                    const double ne = entry_count(cnt, e, Em, a.B);
                    const double me = (double)ein[((long)e * 2 + 0) * H + cc], qe = (double)ein[((long)e * 2 + 1) * H + cc];
                    n += ne; sw += ne * me; sq += qe + ne * me * me;
                }
            }
        }
        smd[ty * 64 + tx] = n; smd[256 + ty * 64 + tx] = sw; smd[512 + ty * 64 + tx] = sq;
        __syncthreads();
        n = (smd[tx] + smd[64 + tx]) + (smd[128 + tx] + smd[192 + tx]);
        sw = (smd[256 + tx] + smd[320 + tx]) + (smd[384 + tx] + smd[448 + tx]);
        sq = (smd[512 + tx] + smd[576 + tx]) + (smd[640 + tx] + smd[704 + tx]);
        const double meand = n > 0.0 ? sw / n : 0.0;
        const float mean = (float)meand;
        double m2 = sq - n * meand * meand;
        if (m2 < 0.0) m2 = 0.0;
        const float var = n > 0.0 ? (float)(m2 / n) : 0.f;
        const float inv = 1.f / sqrtf(var + a.eps);
        if (wg == 0 && ty == 0 && c < H) {
            L.mm[c] = L.mm[c] - (L.mm[c] - mean) * (1.f - a.momentum);
            L.mv[c] = L.mv[c] - (L.mv[c] - var) * (1.f - a.momentum);
            if (L.inv_std) L.inv_std[c] = inv;
        }
        // ---- normalise + activation; kernel and activations -> LDS
#pragma unroll
        for (int j = 0; j < kStepRows / 4; ++j) {
            const int r = ty + 4 * j;
            float h = 0.f;
            if (r < nrows && c < H) {
                const float xh = (z[j] - mean) * inv;
                h = act_fwd(a.act, xh + beta);
                if (L.xhat) L.xhat[(long)(r0 + r) * L.ldx + c] = xh;
                L.Hout[(long)(r0 + r) * L.ldh + c] = h;
            }
            ht[r * kStackLd + c] = h;                      // zero beyond the layer's width / the block's rows
        }
        if (!N.W) return;
        w_tile_store(wl, K, HN, tid, wv);
        __syncthreads();
        float acc[kStepRows / 4];
#pragma unroll
        for (int j = 0; j < kStepRows / 4; ++j) acc[j] = 0.f;
        const int K4 = (K + 3) & ~3;
        for (int k = 0; k < K4; k += 4) {
            const float w0 = wl[(k + 0) * kStackLd + c], w1 = wl[(k + 1) * kStackLd + c];
            const float w2 = wl[(k + 2) * kStackLd + c], w3 = wl[(k + 3) * kStackLd + c];
#pragma unroll
            for (int j = 0; j < kStepRows / 4; ++j) {
                const float4 h4 = *reinterpret_cast<const float4*>(ht + (ty + 4 * j) * kStackLd + k);
                acc[j] = fmaf(h4.w, w3, fmaf(h4.z, w2, fmaf(h4.y, w1, fmaf(h4.x, w0, acc[j]))));
            }
        }
        Hs = HN;
#pragma unroll
        for (int j = 0; j < kStepRows / 4; ++j) {
            const int r = ty + 4 * j;
            out_z[j] = acc[j] + nbias;
            if (r < nrows && c < HN && N.Z) N.Z[(long)(r0 + r) * N.ldz + c] = out_z[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < kStepRows / 4; ++j) out_z[j] = z[j];
    }
    // ---- (mean, M2) of the block's rows of the layer just made
    float sv = 0.f;
#pragma unroll
    for (int j = 0; j < kStepRows / 4; ++j) if (ty + 4 * j < nrows && c < Hs) sv += out_z[j];
    const float tot = wg_rowlane_sum(sv, smf);
    const float bmean = nrows > 0 ? tot / (float)nrows : 0.f;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < kStepRows / 4; ++j) if (ty + 4 * j < nrows && c < Hs) { const float d = out_z[j] - bmean; q += d * d; }
    const float bm2 = wg_rowlane_sum(q, smf);
    if (c < Hs && ty == 0) {
        a.part_out[((long)wg * 2 + 0) * Hs + c] = bmean;
        a.part_out[((long)wg * 2 + 1) * Hs + c] = bm2;
    }
}

struct StepBwdArgs {
    StackBwdLayer cur, low;             // layer i and the layer below (low.Hact == NULL: none, i == 0)
    int B, act, nwg, sums_only;
    float n_total;
    float* dZ0; long ldz0;
    const float* part_in;               // [nwg][2][H] sums of layer i (unused with sums_only)
    float* part_out;                    // sums of the layer below (sums_only: of layer i)
    float* gwp;                         // [nwg][65][64] weight-gradient partials of layer i
    const float* ext_in;                // data parallel: [2][H] sums of layer i over every rank (all-reduced between two launches)
};

__global__ __launch_bounds__(256) void stack_bwd_step_kernel(StepBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float dyt[kStepRows * kStackLd];
    __shared__ __attribute__((aligned(16))) float ht[kStepRows * kStackLd];
    __shared__ __attribute__((aligned(16))) float wl[64 * kStackLd];
    __shared__ float smf[256];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int wg = blockIdx.x, E = a.nwg;
    const int cr = chunk_rows(a.B, E);
    const int r0 = wg * cr;
    const int nrows = max(0, min(a.B, r0 + cr) - r0);
    const int c = tx;
    const StackBwdLayer& L = a.cur;
    const int H = L.H, cc = c < H ? c : H - 1;
    // ---- every read of the step, up front
    float dh[kStepRows / 4], hv[kStepRows / 4], xv[kStepRows / 4];
#pragma unroll
    for (int j = 0; j < kStepRows / 4; ++j) {
        const int r = ty + 4 * j;
        const long row = r0 + (r < nrows ? r : (nrows > 0 ? nrows - 1 : 0));
        dh[j] = L.dHin[row * L.lddh + cc]; hv[j] = L.Hact[row * L.ldh + cc]; xv[j] = L.xhat[row * L.ldx + cc];
    }
    float dy[kStepRows / 4];
    if (a.sums_only) {
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int j = 0; j < kStepRows / 4; ++j) {
            const float d = (ty + 4 * j < nrows && c < H) ? dh[j] * act_grad(a.act, hv[j]) : 0.f;
            s1 += d; s2 += d * xv[j];
        }
        const float t1 = wg_rowlane_sum(s1, smf), t2 = wg_rowlane_sum(s2, smf);
        if (c < H && ty == 0) { a.part_out[((long)wg * 2 + 0) * H + c] = t1; a.part_out[((long)wg * 2 + 1) * H + c] = t2; }
        return;
    }
    const float* const ein = a.ext_in ? a.ext_in : a.part_in;
    const int Em = a.ext_in ? 1 : E;
    float p1[16], p2[16];                                  // the first 64 blocks' sums (the rest in further batches below)
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const long eo = (long)(ty + 4 * u < Em ? ty + 4 * u : Em - 1) * 2;
        p1[u] = ein[(eo + 0) * H + cc]; p2[u] = ein[(eo + 1) * H + cc];
    }
    const bool has_low = a.low.Hact != nullptr;
    const int K = has_low ? L.K : 0;
    float4 wv[4];
    float hp[kStepRows / 4];
    w_tile_request(has_low ? L.W : nullptr, L.ldw, K, H, tid, wv);     // this layer's kernel [K, H]
#pragma unroll
    for (int j = 0; j < kStepRows / 4; ++j) {              // the layer's input activations (= the layer below's output) and its xhat
        const int r = ty + 4 * j;
        const long row = r0 + (r < nrows ? r : (nrows > 0 ? nrows - 1 : 0));
        const int kc = tx < K ? tx : (K > 0 ? K - 1 : 0);
        hp[j] = has_low ? L.Hprev[row * L.ldp + kc] : 0.f;
    }
    const float inv = L.inv_std[cc];
    // ---- the two batch sums, every block the same order
    float v1 = 0.f, v2 = 0.f;
#pragma unroll
    for (int u = 0; u < 16; ++u) if (ty + 4 * u < Em && c < H) { v1 += p1[u]; v2 += p2[u]; }
#pragma unroll 1
    for (int e0 = ty + 64; e0 < Em; e0 += 64) {
        float q1[16], q2[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const long eo = (long)(e0 + 4 * u < Em ? e0 + 4 * u : Em - 1) * 2;
            q1[u] = ein[(eo + 0) * H + cc]; q2[u] = ein[(eo + 1) * H + cc];
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) if (e0 + 4 * u < Em && c < H) { v1 += q1[u]; v2 += q2[u]; }
    }
    v1 = wg_rowlane_sum(v1, smf);
    v2 = wg_rowlane_sum(v2, smf);
    if (wg == 0 && ty == 0 && c < H && L.dbeta && !a.ext_in) L.dbeta[c] = v1;     // (all-rank sums: d beta stays the local share)
    const float m1 = v1 / a.n_total, m2 = v2 / a.n_total;
#pragma unroll
    for (int j = 0; j < kStepRows / 4; ++j) {
        const int r = ty + 4 * j;
        float dz = 0.f;
        if (r < nrows && c < H) {
            dz = inv * (dh[j] * act_grad(a.act, hv[j]) - m1 - xv[j] * m2);
            if (!has_low) a.dZ0[(long)(r0 + r) * a.ldz0 + c] = dz;
        }
        dyt[r * kStackLd + c] = dz;
        ht[r * kStackLd + tx] = (r < nrows && tx < K) ? hp[j] : 0.f;
    }
    if (!has_low) return;
    w_tile_store(wl, K, H, tid, wv);
    __syncthreads();
    // (the layer below's output and xhat of the block's rows, column k = tx: requested now, used after the two products)
    float lh[kStepRows / 4], lx[kStepRows / 4];
#pragma unroll
    for (int j = 0; j < kStepRows / 4; ++j) {
        const int r = ty + 4 * j;
        const long row = r0 + (r < nrows ? r : (nrows > 0 ? nrows - 1 : 0));
        const int kc = tx < K ? tx : (K > 0 ? K - 1 : 0);
        lh[j] = a.low.Hact[row * a.low.ldh + kc];
        lx[j] = a.low.xhat[row * a.low.ldx + kc];
    }
    // ---- weight-gradient partial of the block: gW[k][c] = sum_r Hprev[r][k] dz[r][c], k in [16 ty, 16 ty + 16)
    float* gw = a.gwp + (long)wg * 65 * 64;
    {
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        float bs = 0.f;
#pragma unroll 2
        for (int r = 0; r < nrows; ++r) {
            const float d = dyt[r * kStackLd + c];
            bs += d;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                const float4 h4 = *reinterpret_cast<const float4*>(ht + r * kStackLd + 16 * ty + 4 * q4);
                acc[4 * q4 + 0] = fmaf(h4.x, d, acc[4 * q4 + 0]); acc[4 * q4 + 1] = fmaf(h4.y, d, acc[4 * q4 + 1]);
                acc[4 * q4 + 2] = fmaf(h4.z, d, acc[4 * q4 + 2]); acc[4 * q4 + 3] = fmaf(h4.w, d, acc[4 * q4 + 3]);
            }
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int k = 16 * ty + j;
            if (k < K) gw[(long)k * 64 + c] = acc[j];
        }
        if (ty == 0) gw[(long)K * 64 + c] = bs;
    }
    // ---- gradient w.r.t. the layer's input: dHp[r][k] = sum_c dz[r][c] W[k][c]; thread: k = tx, rows ty + 4 j
    float acc[kStepRows / 4];
#pragma unroll
    for (int j = 0; j < kStepRows / 4; ++j) acc[j] = 0.f;
#pragma unroll 2
    for (int c4 = 0; c4 < 64; c4 += 4) {
        const float4 w4 = *reinterpret_cast<const float4*>(wl + tx * kStackLd + c4);
#pragma unroll
        for (int j = 0; j < kStepRows / 4; ++j) {
            const float4 d4 = *reinterpret_cast<const float4*>(dyt + (ty + 4 * j) * kStackLd + c4);
            acc[j] = fmaf(d4.w, w4.w, fmaf(d4.z, w4.z, fmaf(d4.y, w4.y, fmaf(d4.x, w4.x, acc[j]))));
        }
    }
    // the layer below (k = its column): dH handed on through memory, dy and its block sums
    const StackBwdLayer& P = a.low;
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < kStepRows / 4; ++j) {
        const int r = ty + 4 * j;
        if (r < nrows && tx < K) {
            P.dHin[(long)(r0 + r) * P.lddh + tx] = acc[j];
            const float d = acc[j] * act_grad(a.act, lh[j]);
            s1 += d; s2 += d * lx[j];
        }
    }
    const float t1 = wg_rowlane_sum(s1, smf), t2 = wg_rowlane_sum(s2, smf);
    if (tx < K && ty == 0) { a.part_out[((long)wg * 2 + 0) * K + tx] = t1; a.part_out[((long)wg * 2 + 1) * K + tx] = t2; }
}

// ---- K-STACK, a batch that fits ONE workgroup (the reference's batch of 32): the whole backward of the stack in one launch.
// The block holds every row, so the batch sums are its own and nothing crosses blocks; the gradient w.r.t. a layer's input
// stays in registers (the product's thread layout IS the next layer's element layout).  The layer loop is unrolled (NL
// layers, 4 R rows) and EVERY operand of every layer -- activations, xhat, kernels; a layer's input is the activation tile
// of the layer below -- is requested in the first instructions, top layer first: one memory round trip for the launch.
struct ChainBwdArgs {
    StackBwdLayer l[4];
    int B, act;
    float n_total;
    float* dZ0; long ldz0;
};

template <int NL, int R, bool VEC>
__global__ __launch_bounds__(256) void stack_bwd_chain_kernel(ChainBwdArgs a) {
    __shared__ __attribute__((aligned(16))) float dyt[4 * R * kStackLd];
    __shared__ __attribute__((aligned(16))) float ht[4 * R * kStackLd];
    __shared__ __attribute__((aligned(16))) float wl[64 * kStackLd];
    __shared__ float smf[256];
    const int tid = threadIdx.x, tx = tid & 63, ty = tid >> 6;
    const int nrows = a.B, c = tx;
    float hv[NL][R], xv[NL][R], inv[NL], dh[R];
    float4 wv[NL][4];
    {
        const StackBwdLayer& T = a.l[NL - 1];
        const int cc = c < T.H ? c : T.H - 1;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int r = ty + 4 * j;
            dh[j] = T.dHin[(long)(r < nrows ? r : nrows - 1) * T.lddh + cc];
        }
    }
#pragma unroll
    for (int t = 0; t < NL; ++t) {
        const int i = NL - 1 - t;
        const StackBwdLayer& L = a.l[i];
        const int cc = c < L.H ? c : L.H - 1;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int r = ty + 4 * j;
            const long row = r < nrows ? r : nrows - 1;
            hv[i][j] = L.Hact[row * L.ldh + cc]; xv[i][j] = L.xhat[row * L.ldx + cc];
        }
        inv[i] = L.inv_std[cc];
        if (i > 0) w_tile_request<VEC ? 1 : 2>(L.W, L.ldw, L.K, L.H, tid, wv[i]);
    }
#pragma unroll
    for (int t = 0; t < NL; ++t) {
        const int i = NL - 1 - t;
        const StackBwdLayer& L = a.l[i];
        const int H = L.H, K = i > 0 ? L.K : 0;
        float dy[R], s1 = 0.f, s2 = 0.f;
        if (a.act == 1) {                                  // (one uniform branch around the rows, none per element)
#pragma unroll
            for (int j = 0; j < R; ++j) dy[j] = hv[i][j] > 0.f ? dh[j] : 0.f;
        } else {
#pragma unroll
            for (int j = 0; j < R; ++j) dy[j] = dh[j] * act_grad(a.act, hv[i][j]);
        }
#pragma unroll
        for (int j = 0; j < R; ++j) {
            dy[j] = (ty + 4 * j < nrows && c < H) ? dy[j] : 0.f;
            s1 += dy[j]; s2 += dy[j] * xv[i][j];
        }
        const float v1 = wg_rowlane_sum(s1, smf), v2 = wg_rowlane_sum(s2, smf);
        if (ty == 0 && c < H && L.dbeta) L.dbeta[c] = v1;
        const float m1 = v1 / a.n_total, m2 = v2 / a.n_total;
#pragma unroll
        for (int j = 0; j < R; ++j) {
            const int r = ty + 4 * j;
            const float dz = (r < nrows && c < H) ? inv[i] * (dy[j] - m1 - xv[i][j] * m2) : 0.f;
            if (i == 0 && r < nrows && c < H) a.dZ0[(long)r * a.ldz0 + c] = dz;
            dyt[r * kStackLd + c] = dz;
            if (i > 0) ht[r * kStackLd + tx] = (r < nrows && tx < K) ? hv[i > 0 ? i - 1 : 0][j] : 0.f;
        }
        if (i == 0) break;
        w_tile_store(wl, K, H, tid, wv[i]);
        __syncthreads();
        // ---- weight gradient, complete: gW[k][c] = sum_r Hprev[r][k] dz[r][c], k in [16 ty, 16 ty + 16); bias row = column sums
        {
            float acc[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) acc[j] = 0.f;
            float bs = 0.f;
#pragma unroll 4
            for (int r = 0; r < nrows; ++r) {
                const float d = dyt[r * kStackLd + c];
                bs += d;
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 h4 = *reinterpret_cast<const float4*>(ht + r * kStackLd + 16 * ty + 4 * q4);
                    acc[4 * q4 + 0] = fmaf(h4.x, d, acc[4 * q4 + 0]); acc[4 * q4 + 1] = fmaf(h4.y, d, acc[4 * q4 + 1]);
                    acc[4 * q4 + 2] = fmaf(h4.z, d, acc[4 * q4 + 2]); acc[4 * q4 + 3] = fmaf(h4.w, d, acc[4 * q4 + 3]);
                }
            }
            if (c < H) {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int k = 16 * ty + j;
                    if (k < K) L.gW[(long)k * L.ldg + c] = acc[j];
                }
                if (ty == 0) L.gW[(long)K * L.ldg + c] = bs;
            }
        }
        // ---- gradient w.r.t. the layer's input: dHp[r][k] = sum_c dz[r][c] W[k][c]; thread: k = tx, rows ty + 4 j --
        // the element layout of the layer below, so it never leaves the registers
#pragma unroll
        for (int j = 0; j < R; ++j) dh[j] = 0.f;
#pragma unroll 4
        for (int c4 = 0; c4 < 64; c4 += 4) {
            const float4 w4 = *reinterpret_cast<const float4*>(wl + tx * kStackLd + c4);
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const float4 d4 = *reinterpret_cast<const float4*>(dyt + (ty + 4 * j) * kStackLd + c4);
                dh[j] = fmaf(d4.w, w4.w, fmaf(d4.z, w4.z, fmaf(d4.y, w4.y, fmaf(d4.x, w4.x, dh[j]))));
            }
        }
        const StackBwdLayer& P = a.l[i > 0 ? i - 1 : 0];
        if (P.dHin) {
#pragma unroll
            for (int j = 0; j < R; ++j) {
                const int r = ty + 4 * j;
                if (r < nrows && tx < K) P.dHin[(long)r * P.lddh + tx] = dh[j];
            }
        }
        __syncthreads();                                   // the tiles are free for the layer below
    }
}

template <int R, bool VEC>
static void launch_bwd_chain(int n, const ChainBwdArgs& q, hipStream_t st) {
    switch (n) {
        case 1: hipLaunchKernelGGL((stack_bwd_chain_kernel<1, R, VEC>), dim3(1), dim3(256), 0, st, q); break;
        case 2: hipLaunchKernelGGL((stack_bwd_chain_kernel<2, R, VEC>), dim3(1), dim3(256), 0, st, q); break;
        case 3: hipLaunchKernelGGL((stack_bwd_chain_kernel<3, R, VEC>), dim3(1), dim3(256), 0, st, q); break;
        default: hipLaunchKernelGGL((stack_bwd_chain_kernel<4, R, VEC>), dim3(1), dim3(256), 0, st, q); break;
    }
}

inline int stack_workgroups(int B, int rows) {
    int n = (B + rows - 1) / rows;
    return n < 1 ? 1 : n;
}
constexpr int kStackMaxPhaseWG = 1024;      // workgroups of a one-step launch (no residency requirement)

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
    hipLaunchKernelGGL(col_moments_kernel, dim3(n_chunks(B), strip_blocks(H, n_chunks(B))), dim3(256), 0,
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
    hipLaunchKernelGGL(bn_relu_apply_kernel, dim3(grid, strip_blocks(H, grid)), dim3(256), 2 * H * sizeof(float),
                       static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int dcahip_bn_bwd_sums(const float* dH, long ldd, const float* Hact, long ldh,
                                  const float* xhat, long ldx, int B, int H, float* part,
                                  int act, void* stream) {
    if (!dH || !Hact || !xhat || !part || B <= 0 || H <= 0) return DCAHIP_EINVAL;
    hipLaunchKernelGGL(bn_bwd_sums_kernel, dim3(n_chunks(B), strip_blocks(H, n_chunks(B))), dim3(256), 0,
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
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(grid, strip_blocks(H, grid)), dim3(256), 2 * H * sizeof(float),
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

// ---- K-STACK entry points
extern "C" int dcahip_hidden_stack_max_rows(void) { return kStackMaxWG * kStackRows; }

extern "C" long dcahip_hidden_stack_workspace_bytes(int n_layers, int B) {
    if (n_layers < 1 || n_layers > kStackMaxLayers || B <= 0 || B > kStackMaxWG * kStackRows) return 0;
    const long nwg = stack_workgroups(B, 16);        // the finest row partition a caller may ask for
    return 256 + ((long)n_layers * nwg * 2 * 64 + (long)n_layers * nwg * 65 * 64) * (long)sizeof(float);
}

static int stack_plan(int n, int B, int rows_per_wg, int first, int last, int nsteps, int* nwg_out) {
    if (rows_per_wg < 16 || rows_per_wg > kStackRows || first < 0 || last < first || last >= nsteps) return DCAHIP_EINVAL;
    const int nwg = stack_workgroups(B, rows_per_wg);
    if (nwg > kStackMaxPhaseWG) return DCAHIP_EINVAL;
    if (last > first && nwg > kStackMaxWG) return DCAHIP_EINVAL;      // several steps per launch: every workgroup resident
    *nwg_out = nwg;
    return 0;
}

extern "C" int dcahip_hidden_stack_fwd(const dcahip_small_layer* layers, int n, int B, float momentum, float eps, int act,
                                       int rows_per_wg, int first_step, int last_step,
                                       void* workspace, long workspace_bytes, void* stream) {
    if (!layers || n < 1 || n > kStackMaxLayers || B <= 0 || B > kStackMaxWG * kStackRows || !workspace) return DCAHIP_EINVAL;
    if (workspace_bytes < dcahip_hidden_stack_workspace_bytes(n, B) || !al16(workspace)) return DCAHIP_EINVAL;
    StackFwdArgs a{};
    if (stack_plan(n, B, rows_per_wg, first_step, last_step, n + 1, &a.nwg)) return DCAHIP_EINVAL;
    for (int i = 0; i < n; ++i) {
        const dcahip_small_layer& q = layers[i];
        if (q.H <= 0 || q.H > 64 || !q.Hout || !q.moving_mean || !q.moving_var) return DCAHIP_EINVAL;
        if (i == 0 ? !q.Z : (!q.W || q.K != layers[i - 1].H)) return DCAHIP_EINVAL;
        if (i > 0 && !q.Z && !(first_step == 0 && last_step == n)) return DCAHIP_EINVAL;   // one step per launch: Z hands over
        a.l[i] = SmallLayer{q.W, q.ldw, q.bias, q.K, q.H, q.beta, q.moving_mean, q.moving_var, q.Z, q.ldz, q.xhat, q.ldx,
                            q.Hout, q.ldh, q.inv_std};
    }
    a.n = n; a.B = B; a.act = act; a.first = first_step; a.last = last_step; a.momentum = momentum; a.eps = eps;
    a.sync = static_cast<unsigned*>(workspace);
    a.part = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256);
    if (first_step == last_step && rows_per_wg == kStepRows) {
        // one step per launch at the step kernels' row partition: every read of the step up front
        StepFwdArgs q{};
        const int i = first_step == 0 ? 0 : first_step - 1;
        q.cur = a.l[i];
        if (first_step > 0 && i + 1 < n) q.nxt = a.l[i + 1];
        q.B = B; q.act = act; q.nwg = a.nwg; q.stats_only = first_step == 0; q.momentum = momentum; q.eps = eps;
        q.part_in = a.part + (long)i * a.nwg * 2 * 64;
        q.part_out = a.part + (long)(first_step == 0 ? 0 : i + 1) * a.nwg * 2 * 64;
        hipLaunchKernelGGL(stack_fwd_step_kernel, dim3(a.nwg), dim3(256), 0, static_cast<hipStream_t>(stream), q);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(hidden_stack_fwd_kernel, dim3(a.nwg), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

// ---- K-STACK between the exchanges of a data-parallel step (SyncBN): ONE step per call, the statistics of the step's input
// layer handed in as one entry per rank, the statistics of the layer the step makes merged over this rank's row blocks into
// `stat_out` ([2][H]: mean, M2) for the all-gather that follows.  Rows per workgroup: the step kernels' 32.
extern "C" int dcahip_hidden_stack_step_blocks(int B) { return B > 0 ? stack_workgroups(B, kStepRows) : 0; }

extern "C" int dcahip_hidden_stack_fwd_sync(const dcahip_small_layer* layers, int n, int B, float momentum, float eps, int act,
                                            int step, const float* ext_entries, const float* ext_counts, int ext_E,
                                            float* stat_out, void* workspace, long workspace_bytes, void* stream) {
    if (!layers || n < 1 || n > kStackMaxLayers || B <= 0 || B > kStackMaxWG * kStackRows || !workspace) return DCAHIP_EINVAL;
    if (workspace_bytes < dcahip_hidden_stack_workspace_bytes(n, B) || !al16(workspace) || step < 0 || step > n) return DCAHIP_EINVAL;
    if (step > 0 && (!ext_entries || !ext_counts || ext_E <= 0)) return DCAHIP_EINVAL;
    const int nwg = stack_workgroups(B, kStepRows);
    if (nwg > kStackMaxPhaseWG) return DCAHIP_EINVAL;
    SmallLayer l[kStackMaxLayers];
    for (int i = 0; i < n; ++i) {
        const dcahip_small_layer& q = layers[i];
        if (q.H <= 0 || q.H > 64 || !q.Hout || !q.moving_mean || !q.moving_var || !q.Z) return DCAHIP_EINVAL;
        if (i > 0 && (!q.W || q.K != layers[i - 1].H)) return DCAHIP_EINVAL;
        l[i] = SmallLayer{q.W, q.ldw, q.bias, q.K, q.H, q.beta, q.moving_mean, q.moving_var, q.Z, q.ldz, q.xhat, q.ldx,
                          q.Hout, q.ldh, q.inv_std};
    }
    float* part = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256);
    StepFwdArgs q{};
    const int i = step == 0 ? 0 : step - 1;
    q.cur = l[i];
    if (step > 0 && i + 1 < n) q.nxt = l[i + 1];
    q.B = B; q.act = act; q.nwg = nwg; q.stats_only = step == 0; q.momentum = momentum; q.eps = eps;
    q.part_in = part + (long)i * nwg * 2 * 64;
    const int made = step == 0 ? 0 : i + 1;                       // the layer whose block statistics this launch writes
    q.part_out = part + (long)made * nwg * 2 * 64;
    if (step > 0) { q.ext_in = ext_entries; q.ext_counts = ext_counts; q.ext_E = ext_E; }
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(stack_fwd_step_kernel, dim3(nwg), dim3(256), 0, st, q);
    if (made < n && stat_out) {
        const int Hm = l[made].H;
        hipLaunchKernelGGL(moments_combine_kernel, dim3((Hm + 63) / 64), dim3(256), 0, st,
                           (const float*)q.part_out, (const float*)nullptr, nwg, Hm, stat_out, B);
    }
    return (int)hipGetLastError();
}

extern "C" int dcahip_hidden_stack_bwd_sync(const dcahip_stack_bwd_layer* layers, int n, int B, float n_total, int act,
                                            float* dZ0, long ldz0, int step, const float* ext_sums, float* sums_out,
                                            void* workspace, long workspace_bytes, void* stream) {
    if (!layers || n < 1 || n > kStackMaxLayers || B <= 0 || B > kStackMaxWG * kStackRows || !dZ0 || !workspace) return DCAHIP_EINVAL;
    if (workspace_bytes < dcahip_hidden_stack_workspace_bytes(n, B) || !al16(workspace) || step < 0 || step > n) return DCAHIP_EINVAL;
    if (step > 0 && !ext_sums) return DCAHIP_EINVAL;
    const int nwg = stack_workgroups(B, kStepRows);
    if (nwg > kStackMaxPhaseWG) return DCAHIP_EINVAL;
    StackBwdLayer l[kStackMaxLayers];
    for (int i = 0; i < n; ++i) {
        const dcahip_stack_bwd_layer& q = layers[i];
        if (q.H <= 0 || q.H > 64 || !q.Hact || !q.xhat || !q.inv_std || !q.dH) return DCAHIP_EINVAL;
        if (i > 0 && (!q.W || !q.Hprev || !q.gW || q.K != layers[i - 1].H || q.K > 64)) return DCAHIP_EINVAL;
        l[i] = StackBwdLayer{q.W, q.ldw, q.K, q.H, q.Hact, q.ldh, q.xhat, q.ldx, q.inv_std, q.Hprev, q.ldp, q.gW, q.ldg,
                             q.dbeta, q.dH, q.lddh};
    }
    float* part = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256);
    float* gwp = part + (long)n * nwg * 2 * 64;
    StepBwdArgs q{};
    const int i = step == 0 ? n - 1 : n - step;
    q.cur = l[i];
    if (step > 0 && i > 0) q.low = l[i - 1];
    q.B = B; q.act = act; q.nwg = nwg; q.sums_only = step == 0; q.n_total = n_total;
    q.dZ0 = dZ0; q.ldz0 = ldz0;
    q.part_in = part + (long)i * nwg * 2 * 64;
    const int made = step == 0 ? i : i - 1;                       // the layer whose block sums this launch writes (-1: none)
    q.part_out = part + (long)(made >= 0 ? made : 0) * nwg * 2 * 64;
    q.gwp = gwp + (long)i * nwg * 65 * 64;
    if (step > 0) q.ext_in = ext_sums;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(stack_bwd_step_kernel, dim3(nwg), dim3(256), 0, st, q);
    if (made >= 0 && sums_out)
        hipLaunchKernelGGL(stack_sums_combine_kernel, dim3(1), dim3(64), 0, st, (const float*)q.part_out, nwg, l[made].H, sums_out,
                           l[made].dbeta);
    return (int)hipGetLastError();
}

extern "C" int dcahip_hidden_stack_bwd(const dcahip_stack_bwd_layer* layers, int n, int B, float n_total, int act,
                                       float* dZ0, long ldz0, int rows_per_wg, int first_step, int last_step,
                                       void* workspace, long workspace_bytes, void* stream) {
    if (!layers || n < 1 || n > kStackMaxLayers || B <= 0 || B > kStackMaxWG * kStackRows || !dZ0) return DCAHIP_EINVAL;
    const bool one_launch = first_step == 0 && last_step == n + 1;
    const bool chain = one_launch && B <= kStackRows && n <= 4;  // one workgroup holds the batch: no workspace
    if (!chain && (!workspace || workspace_bytes < dcahip_hidden_stack_workspace_bytes(n, B) || !al16(workspace)))
        return DCAHIP_EINVAL;
    StackBwdArgs a{};
    if (stack_plan(n, B, rows_per_wg, first_step, last_step, n + 2, &a.nwg)) return DCAHIP_EINVAL;
    for (int i = 0; i < n; ++i) {
        const dcahip_stack_bwd_layer& q = layers[i];
        if (q.H <= 0 || q.H > 64 || !q.Hact || !q.xhat || !q.inv_std) return DCAHIP_EINVAL;
        if (i > 0 && (!q.W || !q.Hprev || !q.gW || q.K != layers[i - 1].H || q.K > 64)) return DCAHIP_EINVAL;
        if (!q.dH && (i == n - 1 || !one_launch)) return DCAHIP_EINVAL;
        a.l[i] = StackBwdLayer{q.W, q.ldw, q.K, q.H, q.Hact, q.ldh, q.xhat, q.ldx, q.inv_std, q.Hprev, q.ldp, q.gW, q.ldg,
                               q.dbeta, q.dH, q.lddh};
    }
    a.n = n; a.B = B; a.act = act; a.first = first_step; a.last = last_step; a.n_total = n_total;
    a.dZ0 = dZ0; a.ldz0 = ldz0;
    if (chain) {
        // the batch fits one workgroup: the chain kernel (no partials, no workspace traffic)
        ChainBwdArgs q{};
        for (int i = 0; i < n; ++i) q.l[i] = a.l[i];
        q.B = B; q.act = act; q.n_total = n_total; q.dZ0 = dZ0; q.ldz0 = ldz0;
        bool vec = true;                                         // every kernel readable as float4 rows?
        for (int i = 1; i < n; ++i) vec = vec && (q.l[i].H & 3) == 0 && (q.l[i].ldw & 3) == 0 && al16(q.l[i].W);
        hipStream_t st = static_cast<hipStream_t>(stream);
        if (B <= 32) { if (vec) launch_bwd_chain<8, true>(n, q, st); else launch_bwd_chain<8, false>(n, q, st); }
        else { if (vec) launch_bwd_chain<16, true>(n, q, st); else launch_bwd_chain<16, false>(n, q, st); }
        return (int)hipGetLastError();
    }
    a.sync = static_cast<unsigned*>(workspace);
    a.part = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256);
    a.gwp = a.part + (long)n * a.nwg * 2 * 64;
    if (first_step == last_step && first_step <= n && rows_per_wg == kStepRows) {
        StepBwdArgs q{};
        const int i = first_step == 0 ? n - 1 : n - first_step;
        q.cur = a.l[i];
        if (first_step > 0 && i > 0) q.low = a.l[i - 1];
        q.B = B; q.act = act; q.nwg = a.nwg; q.sums_only = first_step == 0; q.n_total = n_total;
        q.dZ0 = dZ0; q.ldz0 = ldz0;
        q.part_in = a.part + (long)i * a.nwg * 2 * 64;
        q.part_out = a.part + (long)(first_step == 0 ? i : (i > 0 ? i - 1 : 0)) * a.nwg * 2 * 64;
        q.gwp = a.gwp + (long)i * a.nwg * 65 * 64;
        hipLaunchKernelGGL(stack_bwd_step_kernel, dim3(a.nwg), dim3(256), 0, static_cast<hipStream_t>(stream), q);
        return (int)hipGetLastError();
    }
    hipLaunchKernelGGL(hidden_stack_bwd_kernel, dim3(a.nwg), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}
