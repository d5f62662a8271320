// Microbenchmark (round 4): how fast can one SIMD of gfx950 issue fp32 vector arithmetic, as a function of the number of
// resident waves per SIMD and of the independent chains each wave interleaves?  And what does K-HEADS' dense y = 0
// likelihood element (zinb_zero_elem, the largest block of vector work in the step) cost per element in isolation under the
// same two knobs?  The answers separate "the vector pipe is full" from "the waves wait on their own dependencies".
//   part 1: v_fma_f32 chains (inline asm), ILP 1 / 2 / 4 / 8, 1..4 waves per SIMD
//   part 2: v_exp_f32 chains, same grid
//   part 3: zinb_zero_elem<false> on register inputs, 4 or 8 elements interleaved, 1..4 waves per SIMD
//   part 4: part 3 beside a partner wave issuing v_mfma_f32_32x32x16_bf16 back to back on the same SIMD
// build: hipcc --offload-arch=gfx950 -O3 -I dca_amd/csrc -I include -o tools/_dbg/valu_rate_probe tools/microbench/valu_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include "zinb_math.hpp"

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int ILP, int KIND>       // KIND 0: fma, 1: exp
__global__ void chain_kernel(float* out, int iters, long long* cyc) {
    float v[ILP];
#pragma unroll
    for (int i = 0; i < ILP; ++i) v[i] = threadIdx.x * 1e-3f + i;
    const float c = 0.999f, d = 1e-3f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
#pragma unroll
            for (int i = 0; i < ILP; ++i) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[i]) : "v"(c), "v"(d));
                else asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ILP; ++i) s += v[i];
    if (s == 12345.678f) out[threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int ILP, bool WITH_MFMA>
__global__ void zelem_kernel(float* out, int iters, long long* cyc, int waves_valu) {
    const int wave = threadIdx.x >> 6;
    float acc = 0.f;
    long long t0 = 0, t1 = 0;
    if (!WITH_MFMA || wave < waves_valu) {
        float am[ILP], ad[ILP], ap[ILP];
#pragma unroll
        for (int i = 0; i < ILP; ++i) {
            am[i] = -2.f + 0.01f * (threadIdx.x & 63) + 0.1f * i;
            ad[i] = 0.5f - 0.02f * (threadIdx.x & 31) + 0.05f * i;
            ap[i] = -0.3f + 0.015f * (threadIdx.x & 15) + 0.02f * i;
        }
        const float sf = 1.f + 0.001f * (threadIdx.x & 7);
        t0 = clock64();
        for (int it = 0; it < iters; ++it) {
            float gm[ILP], gd[ILP], gp[ILP], nl[ILP];
#pragma unroll
            for (int i = 0; i < ILP; ++i) nl[i] = zinb_zero_elem<false>(am[i], ad[i], ap[i], sf, 0.f, gm[i], gd[i], gp[i]);
#pragma unroll
            for (int i = 0; i < ILP; ++i) {          // feed the outputs back so that nothing is hoisted or dropped
                acc += nl[i];
                am[i] += 1e-3f * gm[i]; ad[i] += 1e-3f * gd[i]; ap[i] += 1e-3f * gp[i];
            }
        }
        t1 = clock64();
    } else {
        f32x16 a0 = {0}, a1 = {0};
        const bf16x8 x = {1, 2, 3, 4, 5, 6, 7, 8};
        for (int it = 0; it < iters * ILP * 2; ++it) {       // about as many matrix cycles as the partner has vector cycles
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, x, a1, 0, 0, 0);
        }
        acc = a0[0] + a1[1];
    }
    if (acc == 12345.678f) out[threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <typename F>
static float time_ms(F launch) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    launch(); hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < 3; ++i) launch();
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    return ms / 3;
}

int main() {
    float* out; hipMalloc(&out, 1 << 16);
    long long* cyc; hipMalloc(&cyc, 8);
    long long hc;
    const int iters = 2000;
    printf("part 1/2: cycles per instruction per SIMD (s_memtime ticks of wave 0 / instructions issued by all waves of its SIMD)\n");
    for (int kind = 0; kind < 2; ++kind)
        for (int W = 1; W <= 4; ++W) {
            printf("  %s, %d wave(s) per SIMD:", kind ? "v_exp_f32" : "v_fma_f32", W);
            auto run = [&](auto ilp_c) {
                constexpr int ILP = decltype(ilp_c)::value;
                float ms;
                if (kind == 0) ms = time_ms([&] { hipLaunchKernelGGL((chain_kernel<ILP, 0>), dim3(256), dim3(256 * W), 0, 0, out, iters, cyc); });
                else ms = time_ms([&] { hipLaunchKernelGGL((chain_kernel<ILP, 1>), dim3(256), dim3(256 * W), 0, 0, out, iters, cyc); });
                hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
                const double n = (double)iters * 16 * ILP * W;
                printf("  ILP %d: %.2f (%.3f ms)", ILP, (double)hc / n, ms);
            };
            run(std::integral_constant<int, 1>{}); run(std::integral_constant<int, 2>{});
            run(std::integral_constant<int, 4>{}); run(std::integral_constant<int, 8>{});
            printf("\n");
        }
    printf("part 3: zinb_zero_elem alone, ticks per element per SIMD (and per wave)\n");
    for (int W = 1; W <= 4; ++W) {
        float ms4 = time_ms([&] { hipLaunchKernelGGL((zelem_kernel<4, false>), dim3(256), dim3(256 * W), 0, 0, out, iters, cyc, 0); });
        hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        const double e4 = (double)hc / ((double)iters * 4);
        float ms8 = time_ms([&] { hipLaunchKernelGGL((zelem_kernel<8, false>), dim3(256), dim3(256 * W), 0, 0, out, iters, cyc, 0); });
        hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        const double e8 = (double)hc / ((double)iters * 8);
        printf("  %d wave(s) per SIMD: 4 interleaved: %.1f per wave-element = %.1f per SIMD-element (%.3f ms);  8 interleaved: %.1f = %.1f (%.3f ms)\n",
               W, e4, e4 / W, ms4, e8, e8 / W, ms8);
    }
    printf("part 4: zinb_zero_elem on wave(s) 0..n-1 of every SIMD beside ONE partner wave of back-to-back bf16 MFMAs\n");
    for (int WV = 1; WV <= 2; ++WV) {
        // waves are dealt round robin to the SIMDs: waves 0..4 WV-1 evaluate elements, the last four issue MFMAs
        const int nthreads = 256 * (WV + 1);
        float msb = time_ms([&] { hipLaunchKernelGGL((zelem_kernel<4, true>), dim3(256), dim3(nthreads), 0, 0, out, iters, cyc, 4 * WV); });
        hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        const double eb = (double)hc / ((double)iters * 4);
        float msa = time_ms([&] { hipLaunchKernelGGL((zelem_kernel<4, false>), dim3(256), dim3(256 * WV), 0, 0, out, iters, cyc, 0); });
        hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
        const double ea = (double)hc / ((double)iters * 4);
        printf("  %d element wave(s) per SIMD: alone %.1f ticks per wave-element (%.3f ms); beside the MFMA wave %.1f (%.3f ms: MFMA wave alone would take %.3f ms at 32 cycles per MFMA and 2.4 GHz)\n",
               WV, ea, msa, eb, msb, iters * 4 * 2 * 2 * 32 / 2.4e6);
    }
    return 0;
}
