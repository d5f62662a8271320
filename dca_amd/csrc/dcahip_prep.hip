// K-PREP: the preprocessing of dca/io.py:88-111 (scanpy's filter counts, normalize_per_cell,
// log1p, scale) on the resident count matrix, so that the training inputs are produced in HBM
// from ONE upload of the raw counts.  All kernels are memory-bound streaming passes over the
// [n_cells, n_genes] matrix (float4 per lane, coalesced), gfx950 / wave64.
//
//   row sums          n_counts per cell      (filter_cells, normalize_per_cell: exact -- counts
//                                             are integers, accumulated in fp64)
//   column pass       x = log1p(y / fac[row]) written to X, with per-gene sums of x and x*x
//                     accumulated in fp64 per row chunk (deterministic second stage); the same
//                     pass with the transform switched off yields the per-gene counts of
//                     filter_genes
//   column stats      mean, std (ddof = 1, zero std -> 1) exactly as scanpy's scale
//   scale             x = (x - mean) / std in place
// Arithmetic follows the host restatement operation by operation (fp32 division, fp32 log1p,
// fp32 square accumulated in fp64) so that host and device inputs agree to the last ulp of
// log1p.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include "dcahip.h"

namespace {

constexpr int kMaxRowChunks = 512;

__host__ __device__ inline int prep_chunks(int n) {
    int r = (n + 127) / 128;
    return r < 1 ? 1 : (r > kMaxRowChunks ? kMaxRowChunks : r);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// one wave per row
template <int V>
__global__ __launch_bounds__(256) void row_sums_kernel(const float* Y, long ldy, int n, int G, float* out) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * 256) >> 6;
    for (int r = wave; r < n; r += nwaves) {
        const float* row = Y + (long)r * ldy;
        double s = 0.0;
        if (V == 4) {
            const int nq = G >> 2;
            for (int q = lane; q < nq; q += 64) {
                const float4 v = reinterpret_cast<const float4*>(row)[q];
                s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
            }
            for (int g = (nq << 2) + lane; g < G; g += 64) s += (double)row[g];
        } else {
            for (int g = lane; g < G; g += 64) s += (double)row[g];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) out[r] = (float)s;
    }
}

struct PassArgs {
    const float* Y; long ldy;
    const float* fac;
    float* X; long ldx;
    double* part;          // [R][2][Gp]
    int n, G, Gp, R;
    int do_log;
};

// grid.x: gene segments (256 lanes x V genes), grid.y: row chunks
template <int V>
__global__ __launch_bounds__(256) void col_pass_kernel(PassArgs a) {
    const int g = (blockIdx.x * 256 + threadIdx.x) * V;
    if (g >= a.G) return;
    const int cr = (a.n + a.R - 1) / a.R;
    const int r0 = blockIdx.y * cr;
    const int r1 = min(a.n, r0 + cr);
    double s1[V], s2[V];
#pragma unroll
    for (int j = 0; j < V; ++j) { s1[j] = 0.0; s2[j] = 0.0; }
    for (int r = r0; r < r1; ++r) {
        float v[V];
        const float* src = a.Y + (long)r * a.ldy + g;
        if (V == 4) {
            const float4 t = *reinterpret_cast<const float4*>(src);
            v[0] = t.x; v[1 % V] = t.y; v[2 % V] = t.z; v[3 % V] = t.w;
        } else {
            v[0] = src[0];
        }
        const float f = a.fac ? a.fac[r] : 1.f;
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float x = v[j];
            if (a.fac) x = __fdiv_rn(x, f);
            if (a.do_log) x = log1pf(x);
            v[j] = x;
            if (g + j < a.G) {
                s1[j] += (double)x;
                s2[j] += (double)__fmul_rn(x, x);
            }
        }
        if (a.X) {
            float* dst = a.X + (long)r * a.ldx + g;
            if (V == 4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1 % V], v[2 % V], v[3 % V]);
            else dst[0] = v[0];
        }
    }
#pragma unroll
    for (int j = 0; j < V; ++j)
        if (g + j < a.G) {
            a.part[((long)blockIdx.y * 2 + 0) * a.Gp + g + j] = s1[j];
            a.part[((long)blockIdx.y * 2 + 1) * a.Gp + g + j] = s2[j];
        }
}

__global__ __launch_bounds__(256) void col_finish_kernel(const double* part, int R, int Gp, int G,
                                                         double n_total, float* sums, float* mean,
                                                         float* stdv) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    if (g >= G) return;
    double s1 = 0.0, s2 = 0.0;
    for (int r = 0; r < R; ++r) {
        s1 += part[((long)r * 2 + 0) * Gp + g];
        s2 += part[((long)r * 2 + 1) * Gp + g];
    }
    if (sums) sums[g] = (float)s1;
    if (mean) {
        const double m = s1 / n_total;
        const double msq = s2 / n_total;
        double var = n_total > 1.0 ? (msq - m * m) * (n_total / (n_total - 1.0)) : 0.0;
        if (var < 0.0) var = 0.0;
        double sd = sqrt(var);
        if (sd == 0.0) sd = 1.0;
        mean[g] = (float)m;
        stdv[g] = (float)sd;
    }
}

template <int V>
__global__ __launch_bounds__(256) void scale_kernel(float* X, long ldx, int n, int G, const float* mean,
                                                    const float* stdv) {
    const int nvec = (G + V - 1) / V;
    const long total = (long)n * nvec;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int r = (int)(i / nvec);
        const int g = (int)(i - (long)r * nvec) * V;
        float* px = X + (long)r * ldx + g;
        if (V == 4 && g + 4 <= G) {
            float4 v = *reinterpret_cast<float4*>(px);
            const float4 m = *reinterpret_cast<const float4*>(mean + g);
            const float4 s = *reinterpret_cast<const float4*>(stdv + g);
            v.x = __fdiv_rn(v.x - m.x, s.x); v.y = __fdiv_rn(v.y - m.y, s.y);
            v.z = __fdiv_rn(v.z - m.z, s.z); v.w = __fdiv_rn(v.w - m.w, s.w);
            *reinterpret_cast<float4*>(px) = v;
        } else {
            for (int j = 0; j < V && g + j < G; ++j) px[j] = __fdiv_rn(px[j] - mean[g + j], stdv[g + j]);
        }
    }
}

}  // namespace

extern "C" int dcahip_prep_chunks(int n) { return prep_chunks(n); }

extern "C" int dcahip_prep_row_sums(const float* Y, long ldy, int n, int G, float* out, void* stream) {
    if (!Y || !out || n <= 0 || G <= 0) return DCAHIP_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    int grid = (n + 3) / 4;
    if (grid > 4096) grid = 4096;
    if (al16(Y) && (ldy & 3) == 0) hipLaunchKernelGGL(row_sums_kernel<4>, dim3(grid), dim3(256), 0, s, Y, ldy, n, G, out);
    else hipLaunchKernelGGL(row_sums_kernel<1>, dim3(grid), dim3(256), 0, s, Y, ldy, n, G, out);
    return (int)hipGetLastError();
}

extern "C" int dcahip_prep_col_pass(const float* Y, long ldy, int n, int G, const float* fac,
                                    int do_log, float* X, long ldx, double* col_part, void* stream) {
    if (!Y || !col_part || n <= 0 || G <= 0) return DCAHIP_EINVAL;
    const int Gp = (G + 3) & ~3;
    const bool vec = al16(Y) && (ldy & 3) == 0 && ldy >= Gp && (!X || (al16(X) && (ldx & 3) == 0 && ldx >= Gp));
    PassArgs a{Y, ldy, fac, X, ldx, col_part, n, G, Gp, prep_chunks(n), do_log};
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vec) {
        const dim3 grid(((G + 3) / 4 + 255) / 256, a.R);
        hipLaunchKernelGGL(col_pass_kernel<4>, grid, dim3(256), 0, s, a);
    } else {
        const dim3 grid((G + 255) / 256, a.R);
        hipLaunchKernelGGL(col_pass_kernel<1>, grid, dim3(256), 0, s, a);
    }
    return (int)hipGetLastError();
}

extern "C" int dcahip_prep_col_finish(const double* col_part, int R, int G, double n_total,
                                      float* sums, float* mean, float* stdv, void* stream) {
    if (!col_part || R <= 0 || G <= 0 || (mean && !stdv)) return DCAHIP_EINVAL;
    const int Gp = (G + 3) & ~3;
    hipLaunchKernelGGL(col_finish_kernel, dim3((G + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       col_part, R, Gp, G, n_total, sums, mean, stdv);
    return (int)hipGetLastError();
}

extern "C" int dcahip_prep_scale(float* X, long ldx, int n, int G, const float* mean, const float* stdv,
                                 void* stream) {
    if (!X || !mean || !stdv || n <= 0 || G <= 0) return DCAHIP_EINVAL;
    const bool vec = al16(X) && (ldx & 3) == 0 && al16(mean) && al16(stdv);
    const long total = (long)n * ((G + 3) / 4);
    long grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (vec) hipLaunchKernelGGL(scale_kernel<4>, dim3((int)grid), dim3(256), 0, s, X, ldx, n, G, mean, stdv);
    else hipLaunchKernelGGL(scale_kernel<1>, dim3((int)grid), dim3(256), 0, s, X, ldx, n, G, mean, stdv);
    return (int)hipGetLastError();
}
