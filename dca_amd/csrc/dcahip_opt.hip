// K-OPT: the non-default Keras optimizers selectable through dca/train.py:54-57
// (opt.__dict__[optimizer](lr, clipvalue)) and the l1 / l2 kernel regularisers of
// dca/network.py:114-126, 144-146, 369-380 on the flat parameter buffer.
// (RMSprop, the reference default, is dcahip_rmsprop_clip in dcahip_layers.hip.)
//
// tf.keras 2.x semantics (keras>=2.4 delegates to it), all hyper-parameters at their defaults:
//   SGD       w -= lr g                                               (momentum 0)
//   Adagrad   a += g^2 ; w -= lr g / (sqrt(a) + 1e-7)                 (a initialised to 0.1 by the host)
//   Adadelta  a = .95 a + .05 g^2 ; u = g sqrt(d + 1e-7) / sqrt(a + 1e-7) ; d = .95 d + .05 u^2 ; w -= lr u
//   Adam      m = .9 m + .1 g ; v = .999 v + .001 g^2 ; w -= lr sqrt(1-.999^t)/(1-.9^t) m / (sqrt(v) + 1e-7)
//   Adamax    m = .9 m + .1 g ; u = max(.999 u, |g|) ; w -= lr/(1-.9^t) m / (u + 1e-7)
//   Nadam     dcahip_nadam_step below (momentum schedule with a running product kept in device memory)
// g is clipped to [-clip, clip] first (Keras clipvalue).  t = *iter + 1 is read from device memory
// (captured step graphs stay valid); dcahip_counter_add advances it after the step.
#include <hip/hip_runtime.h>
#include <math.h>
#include "dcahip.h"

namespace {

struct OptArgs {
    float* w; const float* g; float* s1; float* s2;
    long n;
    const float* lr;
    const long long* iter;
    int kind;
    float clip;
};

__device__ __forceinline__ float opt_one(int kind, float w, float g, float& a, float& b, float lr,
                                         float c1, float c2) {
    switch (kind) {
        case DCAHIP_OPT_SGD:
            return w - lr * g;
        case DCAHIP_OPT_ADAGRAD:
            a += g * g;
            return w - lr * g / (sqrtf(a) + 1e-7f);
        case DCAHIP_OPT_ADADELTA: {
            a = 0.95f * a + 0.05f * g * g;
            const float u = g * sqrtf(b + 1e-7f) / sqrtf(a + 1e-7f);
            b = 0.95f * b + 0.05f * u * u;
            return w - lr * u;
        }
        case DCAHIP_OPT_ADAM:
            a = 0.9f * a + 0.1f * g;
            b = 0.999f * b + 0.001f * g * g;
            return w - lr * c1 * a / (sqrtf(b) + 1e-7f);          // c1 = sqrt(1 - b2^t) / (1 - b1^t)
        default:                                                  // DCAHIP_OPT_ADAMAX
            a = 0.9f * a + 0.1f * g;
            b = fmaxf(0.999f * b, fabsf(g));
            return w - lr * c2 * a / (b + 1e-7f);                 // c2 = 1 / (1 - b1^t)
    }
}

__global__ __launch_bounds__(256) void optimizer_kernel(OptArgs p) {
    const float lr = *p.lr;
    const double t = (double)((p.iter ? *p.iter : 0) + 1);
    const float c2 = (float)(1.0 / (1.0 - pow(0.9, t)));
    const float c1 = (float)(sqrt(1.0 - pow(0.999, t)) / (1.0 - pow(0.9, t)));
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < p.n; i += stride) {
        float g = p.g[i];
        if (p.clip > 0.f) g = fminf(fmaxf(g, -p.clip), p.clip);
        float a = p.s1 ? p.s1[i] : 0.f, b = p.s2 ? p.s2[i] : 0.f;
        p.w[i] = opt_one(p.kind, p.w[i], g, a, b, lr, c1, c2);
        if (p.s1) p.s1[i] = a;
        if (p.s2) p.s2[i] = b;
    }
}

__global__ void counter_add_kernel(long long* c, int v) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *c += v;
}

// g += l1 sign(w) + 2 l2 w over one segment; partial[blockIdx.x] = sum l1 |w| + l2 w^2
__global__ __launch_bounds__(256) void l1l2_seg_kernel(const float* w, float* g, long n, float l1, float l2,
                                                       double* partial) {
    __shared__ double red[4];
    double acc = 0.0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float x = w[i];
        if (g) g[i] += l1 * (x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f)) + 2.f * l2 * x;
        acc += (double)(l1 * fabsf(x)) + (double)(l2 * x * x);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void l1l2_finish_kernel(const double* partial, int n, float* loss) {
    __shared__ double red[256];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += partial[i];
    red[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int i = 0; i < 256; ++i) s += red[i];
        *loss = (float)((double)*loss + s);
    }
}

// tf.keras Nadam (optimizer_v2/nadam.py): momentum schedule mu_t = b1 (1 - 0.5 * 0.96^(0.004 t)), its running
// product m_schedule (a float32 variable there, device memory here), then
//   g' = g / (1 - P_t) ; m = b1 m + (1-b1) g ; m' = m / (1 - P_t mu_{t+1}) ; v = b2 v + (1-b2) g^2 ; v' = v / (1 - b2^t)
//   w -= lr ((1 - mu_t) g' + mu_{t+1} m') / (sqrt(v') + eps),   P_t = prod_{i<=t} mu_i
__device__ __forceinline__ float nadam_mu(double t) { return (float)(0.9 * (1.0 - 0.5 * pow(0.96, 0.004 * t))); }

__global__ __launch_bounds__(256) void nadam_kernel(float* w, const float* g, float* m, float* v, long n, const float* lr_p,
                                                    const long long* iter, const float* m_schedule, float clip) {
    const float lr = *lr_p;
    const double t = (double)(*iter + 1);
    const float mu_t = nadam_mu(t), mu_t1 = nadam_mu(t + 1.0);
    const float p_new = *m_schedule * mu_t, p_next = p_new * mu_t1;
    const float vden = (float)(1.0 - pow(0.999, t));
    const long stride = (long)gridDim.x * 256;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        float gi = g[i];
        if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
        const float gp = gi / (1.f - p_new);
        const float mi = 0.9f * m[i] + 0.1f * gi;
        const float vi = 0.999f * v[i] + 0.001f * gi * gi;
        const float mbar = (1.f - mu_t) * gp + mu_t1 * (mi / (1.f - p_next));
        w[i] -= lr * mbar / (sqrtf(vi / vden) + 1e-7f);
        m[i] = mi;
        v[i] = vi;
    }
}

__global__ void nadam_schedule_kernel(float* m_schedule, const long long* iter) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *m_schedule *= nadam_mu((double)(*iter + 1));
}

constexpr int kRegBlocks = 64;

}  // namespace

extern "C" int dcahip_optimizer_step(int kind, float* w, const float* g, float* slot1, float* slot2,
                                     long n, const float* lr, const long long* iter, float clip,
                                     void* stream) {
    if (!w || !g || !lr || n <= 0) return DCAHIP_EINVAL;
    if (kind < DCAHIP_OPT_SGD || kind > DCAHIP_OPT_ADAMAX || kind == DCAHIP_OPT_RMSPROP) return DCAHIP_EINVAL;
    if (kind != DCAHIP_OPT_SGD && !slot1) return DCAHIP_EINVAL;
    if ((kind == DCAHIP_OPT_ADADELTA || kind == DCAHIP_OPT_ADAM || kind == DCAHIP_OPT_ADAMAX) && !slot2) return DCAHIP_EINVAL;
    if ((kind == DCAHIP_OPT_ADAM || kind == DCAHIP_OPT_ADAMAX) && !iter) return DCAHIP_EINVAL;
    long grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    OptArgs a{w, g, slot1, slot2, n, lr, iter, kind, clip};
    hipLaunchKernelGGL(optimizer_kernel, dim3((int)grid), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}

extern "C" int dcahip_nadam_step(float* w, const float* g, float* m, float* v, long n, const float* lr,
                                 const long long* iter, float* m_schedule, float clip, void* stream) {
    if (!w || !g || !m || !v || !lr || !iter || !m_schedule || n <= 0) return DCAHIP_EINVAL;
    long grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(nadam_kernel, dim3((int)grid), dim3(256), 0, s, w, g, m, v, n, lr, iter, m_schedule, clip);
    hipLaunchKernelGGL(nadam_schedule_kernel, dim3(1), dim3(64), 0, s, m_schedule, iter);   // after every read of the old product
    return (int)hipGetLastError();
}

extern "C" int dcahip_counter_add(long long* counter, int v, void* stream) {
    if (!counter) return DCAHIP_EINVAL;
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), counter, v);
    return (int)hipGetLastError();
}

extern "C" int dcahip_l1l2_workspace_doubles(void) { return DCAHIP_REG_MAX_SEGS * kRegBlocks; }

extern "C" int dcahip_l1l2_apply(const dcahip_reg_desc* d, const float* w, float* g, float* loss_inout,
                                 double* workspace, void* stream) {
    if (!d || !w || !workspace || d->nseg < 0 || d->nseg > DCAHIP_REG_MAX_SEGS) return DCAHIP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int used = 0;
    for (int k = 0; k < d->nseg; ++k) {
        const long n = d->end[k] - d->start[k];
        if (n <= 0 || (d->l1[k] == 0.f && d->l2[k] == 0.f)) continue;
        hipLaunchKernelGGL(l1l2_seg_kernel, dim3(kRegBlocks), dim3(256), 0, s, w + d->start[k], g ? g + d->start[k] : nullptr, n,
                           d->l1[k], d->l2[k], workspace + (long)used * kRegBlocks);
        ++used;
    }
    if (used && loss_inout)
        hipLaunchKernelGGL(l1l2_finish_kernel, dim3(1), dim3(256), 0, s, workspace, used * kRegBlocks, loss_inout);
    return (int)hipGetLastError();
}
