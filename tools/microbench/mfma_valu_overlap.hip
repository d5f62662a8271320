// Microbenchmark: does fp32 MFMA (v_mfma_f32_32x32x2_f32) overlap with VALU work of the
// partner wave on the same SIMD?  512-thread blocks: waves 0-3 run MFMA chains, waves 4-7 VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int MODE_VALU>   // 0: fma chain x4, 1: transcendental mix
__global__ __launch_bounds__(512) void k(float* out, int n_mfma, int n_valu, int which) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        if (which & 1) {
            f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
            float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
            for (int i = 0; i < n_mfma; ++i) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else {
        if (which & 2) {
            float v0 = threadIdx.x * 1e-3f, v1 = v0 + 1.f, v2 = v0 + 2.f, v3 = v0 + 3.f;
            const float c = 0.999f, d = 1e-3f;
            for (int i = 0; i < n_valu; ++i) {
                if (MODE_VALU == 0) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        v0 = fmaf(v0, c, d); v1 = fmaf(v1, c, d); v2 = fmaf(v2, c, d); v3 = fmaf(v3, c, d);
                    }
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
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int M>
float run(int nm, int nv, int which) {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k<M><<<256, 512>>>(out, nm, nv, which); hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < 5; ++i) k<M><<<256, 512>>>(out, nm, nv, which);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); hipFree(out);
    return ms / 5;
}

int main() {
    const int nm = 4000;            // x4 mfma = 16000 mfma per wave = 1.02M cycles
    for (int mode = 0; mode < 2; ++mode) {
        const int nv = mode == 0 ? 8000 : 16000;
        float a, b, c;
        if (mode == 0) { a = run<0>(nm, nv, 1); b = run<0>(nm, nv, 2); c = run<0>(nm, nv, 3); }
        else { a = run<1>(nm, nv, 1); b = run<1>(nm, nv, 2); c = run<1>(nm, nv, 3); }
        printf("mode %d (%s): mfma alone %.3f ms, valu alone %.3f ms, both %.3f ms  (sum %.3f, max %.3f)\n", mode,
               mode == 0 ? "fma chains" : "exp/log/rcp mix", a, b, c, a + b, a > b ? a : b);
    }
    return 0;
}
