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
#include "dcahip.h"

namespace {

constexpr float kEps = 1e-10f;        // loss.py:65
constexpr float kThetaMax = 1e6f;     // loss.py:85
constexpr float kZeroThresh = 1e-8f;  // loss.py:138
constexpr int kMaxPartials = 2048;    // 256 CUs x 8 blocks
constexpr int kSmallY = 16;

// log(n!) for n = 0..16
__constant__ float kLogFact[kSmallY + 1] = {
    0.0f, 0.0f, 0.69314718055994531f, 1.7917594692280550f, 3.1780538303479458f,
    4.7874917427820458f, 6.5792512120101012f, 8.5251613610654147f, 10.604602902745251f,
    12.801827480081469f, 15.104412573075516f, 17.502307845873887f, 19.987214495661885f,
    22.552163853123425f, 25.191221182738680f, 27.899271383840890f, 30.671860106080672f};

// ---- elementary functions on the transcendental unit (v_exp_f32 / v_log_f32 / v_rcp_f32, 1 ulp
// each) with first-order compensation of the range step, so that no IEEE division or libm call
// sits on the per-element path.  exp: the rounding of x*log2(e) is re-applied as a correction
// term; log: log2 result times ln2 in two pieces; log1p / expm1 use Kahan's exact-ratio forms.
__device__ __forceinline__ float frcp(float x) { return __builtin_amdgcn_rcpf(x); }

__device__ __forceinline__ float fexp(float x) {
    x = fminf(fmaxf(x, -104.f), 88.7f);
    const float L = 1.44269504088896340736f, Llo = 1.92596299112661746e-8f;
    const float t = x * L;
    const float r = fmaf(x, Llo, fmaf(x, L, -t));
    const float e = __builtin_amdgcn_exp2f(t);
    return fmaf(e, r * 0.69314718055994531f, e);
}

__device__ __forceinline__ float flog(float x) {           // x > 0, normal
    const float y = __builtin_amdgcn_logf(x);              // log2(x)
    const float C = 0.693147182464599609375f, Clo = -1.90465429995776804e-9f;
    const float r = y * C;
    return r + fmaf(y, Clo, fmaf(y, C, -r));
}

__device__ __forceinline__ float flog1p(float t) {         // t >= 0
    const float u = 1.f + t;
    const float d = u - 1.f;
    const float l = flog(u);
    return d == 0.f ? t : l * (t * frcp(d));
}

__device__ __forceinline__ float fexpm1_neg(float x, float ex) {   // x <= 0, ex = fexp(x)
    const float d = ex - 1.f;
    const float k = d * x * frcp(flog(ex));
    return x < -17.f ? -1.f : (d == 0.f ? x : k);
}

__device__ __forceinline__ float digamma_pos(float x) {
    // x > 0: upward recurrence to x >= 6, then the asymptotic series (rare generic path)
    float r = 0.f;
    while (x < 6.f) { r -= 1.f / x; x += 1.f; }
    const float xi = 1.f / x, xi2 = xi * xi;
    return r + logf(x) - 0.5f * xi - xi2 * (1.f / 12.f - xi2 * (1.f / 120.f - xi2 * (1.f / 252.f)));
}

// lgamma / digamma route for non-integer or large counts: rare, kept out of line so the hot
// loop stays small (registers, instruction cache)
template <bool GRAD>
__device__ __attribute__((noinline)) float nb_t1_generic(float tp, float y, float* dpsi) {
    if (GRAD) *dpsi = digamma_pos(y + tp) - digamma_pos(tp);
    return lgammaf(tp) + lgammaf(y + 1.f) - lgammaf(y + tp);
}

struct Heads {       // activations of one element
    float mu, gm;    // mean * sf,            d mu / d a_mean          (0 outside the clip window)
    float theta, gd; // dispersion,           d theta / d a_disp
    float pi, omp;   // dropout prob, 1 - pi (computed directly, no cancellation)
};

template <bool HAS_PI, bool CONST_DISP>
__device__ __forceinline__ Heads head_acts(float am, float ad, float ap, float sf) {
    Heads h;
    const float e = fexp(am);                                   // network.py:38
    const bool mwin = (e >= 1e-5f) && (e <= 1e6f);
    h.mu = fminf(fmaxf(e, 1e-5f), 1e6f) * sf;                   // layers.py:85
    h.gm = mwin ? e * sf : 0.f;
    if (CONST_DISP) {                                           // layers.py:21 (ad = theta_w[g])
        h.theta = fminf(fmaxf(fexp(ad), 1e-3f), 1e4f);
        h.gd = 1.f;                                             // chained in dcahip_colsum_chain
    } else {                                                    // network.py:39
        const float ex = fexp(-fabsf(ad));
        const float u = 1.f + ex, d = u - 1.f;
        const float s = frcp(u);
        const float l1 = d == 0.f ? ex : flog(u) * (ex * frcp(d));      // log1p(ex)
        const float sp = fmaxf(ad, 0.f) + l1;
        const bool dwin = (sp >= 1e-4f) && (sp <= 1e4f);
        h.theta = fminf(fmaxf(sp, 1e-4f), 1e4f);
        h.gd = dwin ? (ad >= 0.f ? s : ex * s) : 0.f;
    }
    h.theta = fminf(h.theta, kThetaMax);                        // loss.py:85
    if (HAS_PI) {
        const float ex = fexp(-fabsf(ap));
        const float s = frcp(1.f + ex);
        h.pi = ap >= 0.f ? s : ex * s;
        h.omp = ap >= 0.f ? ex * s : s;
    } else {
        h.pi = 0.f; h.omp = 1.f;
    }
    return h;
}

// One element of the loss and (GRAD) its gradient w.r.t. (mu, theta, pi).
template <bool HAS_PI, bool GRAD>
__device__ __forceinline__ float nll_elem(const Heads& h, float y, float ridge,
                                          float& dmu, float& dth, float& dpi) {
    const float theta = h.theta, mu = h.mu;
    const float tp = theta + kEps;
    float nll;
    if (HAS_PI && y < kZeroThresh) {
        // zero_case = -log(pi + (1-pi) * (theta/(theta+mu+eps))^theta + eps)   loss.py:136-137
        const float den = theta + mu + kEps;
        const float rden = frcp(den);
        const float t = (mu + kEps) * frcp(theta);     // theta/den = 1/(1+t)
        const float logq = -flog1p(t);
        const float tl = theta * logq;
        const float z = fexp(tl);
        const float D = h.pi + h.omp * z + kEps;
        nll = -flog(D);
        if (GRAD) {
            const float invD = frcp(D);
            const float oz = h.omp * z * invD;
            dmu = oz * theta * rden;
            // log q + 1 - q = -log1p(t) + t/(1+t): series below t = 2^-5 (cancellation)
            const float fs = -t * t * (0.5f - t * (2.f / 3.f - t * (0.75f - t * (0.8f - t * (5.f / 6.f)))));
            const float fl = logq + (mu + kEps) * rden;
            dth = -oz * (t < 0.03125f ? fs : fl);
            dpi = fexpm1_neg(tl, z) * invD;            // -(1 - z)/D
        }
    } else {
        // NB.loss: t1 + t2, loss.py:87-88
        const float rtp = frcp(tp);
        const float l1p = flog1p(mu * rtp);
        float t1, dpsi = 0.f;
        if (y == floorf(y) && y <= (float)kSmallY) {
            // lgamma(y+tp) - lgamma(tp) = log prod_{i<y}(tp+i); psi difference = sum 1/(tp+i)
            const int n = (int)y;
            float p1 = 1.f, p2 = 1.f;
            for (int i = 0; i < n; ++i) {
                const float x = tp + (float)i;
                if (i < 8) p1 *= x; else p2 *= x;
                if (GRAD) dpsi += frcp(x);
            }
            t1 = kLogFact[n] - (flog(p1) + (n > 8 ? flog(p2) : 0.f));
        } else {
            t1 = nb_t1_generic<GRAD>(tp, y, &dpsi);
        }
        const float mue = mu + kEps;
        const float t2 = (theta + y) * l1p + y * (flog(tp) - flog(mue));
        nll = t1 + t2;
        if (HAS_PI) nll -= flog(h.omp + kEps);         // loss.py:130
        if (GRAD) {
            // (theta+y)/(tp+mu) - y/(mu+eps) and -(theta+y)mu/(tp(tp+mu)) + y/tp, combined over a
            // common denominator: identical algebra, no cancellation between O(1) terms.
            const float rtm = frcp(tp + mu);
            dmu = theta * (mue - y) * rtm * frcp(mue);
            dth = -dpsi + l1p + (y * tp - theta * mu) * rtp * rtm;
            dpi = HAS_PI ? frcp(h.omp + kEps) : 0.f;
        }
    }
    if (HAS_PI) {
        nll += ridge * h.pi * h.pi;                    // loss.py:139-140
        if (GRAD) dpi += 2.f * ridge * h.pi;
    }
    return nll;
}

struct NllArgs {
    const float *a_mean, *a_disp, *a_pi, *theta_w, *y, *sf;
    const int* perm;
    const long long* cursor;
    long lda, ldy, ldd;
    float *d_mean, *d_disp, *d_pi;
    double* partials;
    int B, G;
    float ridge, inv_n;
};

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

template <bool HAS_PI, bool CONST_DISP, bool GRAD, int V>
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
            if (!CONST_DISP) ldv<V>(a.a_disp + ao, vd);
            if (HAS_PI) ldv<V>(a.a_pi + ao, vp);
            ldv<V>(a.y + srow * a.ldy + g, vy);
            float om[V], od[V], op[V];
            float lacc = 0.f;
#pragma unroll
            for (int j = 0; j < V; ++j) {
                float dmu = 0.f, dth = 0.f, dpi = 0.f;
                const Heads h = head_acts<HAS_PI, CONST_DISP>(vm[j], vd[j], HAS_PI ? vp[j] : 0.f, sf);
                const float nll = nll_elem<HAS_PI, GRAD>(h, vy[j], a.ridge, dmu, dth, dpi);
                const bool valid = (g + j) < a.G;
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
                stv<V>(a.d_disp + dof, od);
                if (HAS_PI) stv<V>(a.d_pi + dof, op);
            }
        }
    }
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
            for (int j = 0; j < V; ++j) o[j] = fminf(fmaxf(expf(v[j]), 1e-5f), 1e6f) * sf;
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
    if (vec) hipLaunchKernelGGL((zinb_nll_kernel<HAS_PI, CONST_DISP, GRAD, 4>), grid, dim3(256), 0, s, a);
    else     hipLaunchKernelGGL((zinb_nll_kernel<HAS_PI, CONST_DISP, GRAD, 1>), grid, dim3(256), 0, s, a);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int dcahip_version(void) { return DCAHIP_VERSION; }
extern "C" int dcahip_zinb_max_partials(void) { return kMaxPartials; }

extern "C" int dcahip_zinb_nll(const float* a_mean, const float* a_disp, const float* a_pi, long lda,
                               const float* theta_w, const float* y, long ldy, const float* sf,
                               const int* perm, const long long* cursor, int B, int G, float ridge,
                               float inv_n, int flags, float* d_mean, float* d_disp, float* d_pi,
                               long ldd, double* loss_partials, int* n_partials_out, void* stream) {
    const bool has_pi = flags & DCAHIP_NLL_HAS_PI, cdisp = flags & DCAHIP_NLL_CONST_DISP;
    const bool grad = d_mean != nullptr;
    if (B <= 0 || G <= 0 || !a_mean || !y || !sf || !loss_partials) return DCAHIP_EINVAL;
    if (has_pi && !a_pi) return DCAHIP_EINVAL;
    if (cdisp ? (theta_w == nullptr) : (a_disp == nullptr)) return DCAHIP_EINVAL;
    if (grad && (!d_disp || (has_pi && !d_pi))) return DCAHIP_EINVAL;
    bool vec = (lda % 4 == 0) && (ldy % 4 == 0) && al16(a_mean) && al16(y) &&
               (cdisp ? al16(theta_w) : al16(a_disp)) && (!has_pi || al16(a_pi)) &&
               lda >= ((G + 3) & ~3) && ldy >= ((G + 3) & ~3);
    if (grad) vec = vec && (ldd % 4 == 0) && ldd >= ((G + 3) & ~3) && al16(d_mean) && al16(d_disp) && (!has_pi || al16(d_pi));
    NllArgs a{a_mean, a_disp, a_pi, theta_w, y, sf, perm, cursor, lda, ldy, ldd,
              d_mean, d_disp, d_pi, loss_partials, B, G, ridge, inv_n};
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
#define DCA_DISPATCH(P, C)                                                  \
    return grad ? launch_nll<P, C, true>(a, vec, grid, s) : launch_nll<P, C, false>(a, vec, grid, s)
    if (has_pi && cdisp) { DCA_DISPATCH(true, true); }
    if (has_pi) { DCA_DISPATCH(true, false); }
    if (cdisp) { DCA_DISPATCH(false, true); }
    DCA_DISPATCH(false, false);
#undef DCA_DISPATCH
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
                                       float* theta, float* pi, long ldo, void* stream) {
    if (B <= 0 || G <= 0 || !sf) return DCAHIP_EINVAL;
    if ((mean_sf && !a_mean) || (theta && !a_disp) || (pi && !a_pi)) return DCAHIP_EINVAL;
    const int Gp = (G + 3) & ~3;
    bool vec = (lda % 4 == 0) && (ldo % 4 == 0) && lda >= Gp && ldo >= Gp;
    const void* ps[6] = {a_mean, a_disp, a_pi, mean_sf, theta, pi};
    for (const void* p : ps) vec = vec && (p == nullptr || al16(p));
    InferArgs a{a_mean, a_disp, a_pi, sf, mean_sf, theta, pi, lda, ldo, B, G};
    const int V = vec ? 4 : 1;
    const long total = (long)B * (((G + V - 1) / V + 255) / 256);
    const int grid = (int)(total < 4096 ? total : 4096);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vec) hipLaunchKernelGGL(heads_infer_kernel<4>, dim3(grid), dim3(256), 0, s, a);
    else     hipLaunchKernelGGL(heads_infer_kernel<1>, dim3(grid), dim3(256), 0, s, a);
    return (int)hipGetLastError();
}
