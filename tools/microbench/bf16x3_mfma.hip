// Microbenchmark behind the "what comes next" section of DESIGN.md: can the fp32 GEMM phases of
// K-HEADS move to the bf16 matrix pipe without giving up fp32 results?
//   a = a1 + a2 + a3 (three bf16 pieces, round-to-nearest residual splits), same for b; the six
//   products a1b1, a1b2, a2b1, a1b3, a2b2, a3b1 accumulated in fp32 by v_mfma_f32_32x32x16_bf16.
// Part A: error of C = A B (32 x K x 32) against fp64 for fp32 MFMA, the 6-product split, the
//         3-product split (a1b1 + a1b2 + a2b1) and plain bf16.
// Part B: issue rate of the bf16 MFMA, and whether VALU work of the partner wave on the same SIMD
//         overlaps with it (it does not with the fp32 MFMA: mfma_valu_overlap.hip).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ unsigned short bf16_rn(float x) {       // round to nearest even, finite inputs
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

struct Split { unsigned short p[3]; };
__device__ __forceinline__ Split split3(float x) {
    Split s;
    s.p[0] = bf16_rn(x);
    const float r1 = x - bf16_f(s.p[0]);
    s.p[1] = bf16_rn(r1);
    const float r2 = r1 - bf16_f(s.p[1]);
    s.p[2] = bf16_rn(r2);
    return s;
}

union Frag { bf16x8 v; unsigned short h[8]; };

// mode 0: fp32 MFMA; 1: six products; 2: three products; 3: bf16 only.  One wave, A [32][K], B [K][32].
__global__ void gemm_tile(const float* A, const float* B, float* C, int K, int mode) {
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    f32x16 acc = {0};
    if (mode == 0) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + hi], B[(k + hi) * 32 + l31], acc, 0, 0, 0);
    } else {
        for (int k0 = 0; k0 < K; k0 += 16) {
            Frag a[3], b[3];
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + 8 * hi + j;
                const Split sa = split3(A[l31 * K + k]), sb = split3(B[k * 32 + l31]);
                for (int q = 0; q < 3; ++q) { a[q].h[j] = sa.p[q]; b[q].h[j] = sb.p[q]; }
            }
            // small terms first
            if (mode == 1) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2].v, b[0].v, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[1].v, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[2].v, acc, 0, 0, 0);
            }
            if (mode == 1 || mode == 2) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1].v, b[0].v, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[1].v, acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0].v, b[0].v, acc, 0, 0, 0);
        }
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = acc[r];
}

template <int MODE_VALU>
__global__ __launch_bounds__(512) void overlap(float* out, int n_mfma, int n_valu, int which) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (which & 1) {
            f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
            Frag x, y;
            for (int j = 0; j < 8; ++j) { x.h[j] = bf16_rn(threadIdx.x * 1e-3f + j); y.h[j] = bf16_rn(1.f + threadIdx.x * 1e-4f * j); }
            for (int i = 0; i < n_mfma; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.v, y.v, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y.v, x.v, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.v, x.v, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y.v, y.v, a3, 0, 0, 0);
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else if (which & 2) {
        float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
        const float c = 0.999f, d = 1e-3f;
        for (int i = 0; i < n_valu; ++i) {
            if (MODE_VALU == 0) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { v0 = fmaf(v0, c, d); v1 = fmaf(v1, c, d); v2 = fmaf(v2, c, d); v3 = fmaf(v3, c, d); }
            } else {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    v0 = __builtin_amdgcn_exp2f(v0 * -0.5f) + d; v1 = __builtin_amdgcn_logf(v1 + 2.f);
                    v2 = __builtin_amdgcn_rcpf(v2 + 1.5f); v3 = fmaf(v3, c, v0);
                    v0 = fmaf(v0, c, v1); v1 = fmaf(v1, c, v2); v2 = fmaf(v2, d, v3); v3 = fmaf(v3, c, d);
                }
            }
        }
        r = v0 + v1 + v2 + v3;
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int M>
float run(int nm, int nv, int which) {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    overlap<M><<<256, 512>>>(out, nm, nv, which); hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < 5; ++i) overlap<M><<<256, 512>>>(out, nm, nv, which);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); hipFree(out);
    return ms / 5;
}

int main() {
    // ---- Part A
    for (int K : {64, 4096}) {
        for (int dist = 0; dist < 2; ++dist) {
            std::vector<float> A(32 * K), B(K * 32);
            srand(7 + K + dist);
            auto rnd = [&]() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
            for (auto& x : A) x = dist ? rnd() * expf(4.f * rnd()) : rnd();
            for (auto& x : B) x = dist ? rnd() * expf(4.f * rnd()) : rnd();
            std::vector<double> ref(1024), mag(1024);
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double s = 0, m = 0;
                    for (int k = 0; k < K; ++k) { const double p = (double)A[i * K + k] * B[k * 32 + j]; s += p; m += fabs(p); }
                    ref[i * 32 + j] = s; mag[i * 32 + j] = m;
                }
            float *dA, *dB, *dC;
            hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
            hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
            const char* names[4] = {"fp32 mfma", "bf16 x 6 products", "bf16 x 3 products", "bf16 x 1"};
            for (int mode = 0; mode < 4; ++mode) {
                gemm_tile<<<1, 64>>>(dA, dB, dC, K, mode);
                std::vector<float> C(1024);
                hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
                double mx = 0, rms = 0;
                for (int i = 0; i < 1024; ++i) { const double e = fabs(C[i] - ref[i]) / mag[i]; mx = fmax(mx, e); rms += e * e; }
                printf("K=%4d %-9s %-18s max err / sum|ab| = %.3e   rms = %.3e\n", K, dist ? "lognormal" : "uniform", names[mode], mx,
                       sqrt(rms / 1024));
            }
            hipFree(dA); hipFree(dB); hipFree(dC);
        }
    }
    // ---- Part B: 16000 bf16 MFMAs per wave at 32 cycles each = 0.51M cycles
    const int nm = 4000;
    for (int mode = 0; mode < 2; ++mode) {
        const int nv = mode == 0 ? 4000 : 8000;
        float a, b, c;
        if (mode == 0) { a = run<0>(nm, nv, 1); b = run<0>(nm, nv, 2); c = run<0>(nm, nv, 3); }
        else { a = run<1>(nm, nv, 1); b = run<1>(nm, nv, 2); c = run<1>(nm, nv, 3); }
        printf("bf16 mfma vs %s: mfma alone %.3f ms (%.1f cycles/mfma at 2.4 GHz), valu alone %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n",
               mode == 0 ? "fma chains" : "exp/log/rcp mix", a, a * 2.4e6 / (4.0 * nm), b, c, a + b, a > b ? a : b);
    }
    return 0;
}
