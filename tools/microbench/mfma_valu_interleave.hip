// Microbenchmark: how many vector instructions ride in the shadow of a v_mfma_f32_32x32x16_bf16 issued by the SAME wave,
// and what the same streams cost with one / two waves per SIMD and across waves.  Instruction order pinned with asm volatile.
//   variants: FILL = plain (v_fma_f32 on 8 independent registers), mix (v_cndmask / v_med3 / v_mul / v_cvt_pk), trans (v_exp_f32)
//             ACC  = 4 rotating accumulators (independent MFMAs) or chains of 6 MFMAs on one accumulator (MFMA_X3's pattern)
//             WAVES = 4 (one per SIMD, 512-register budget) or 8 (two per SIMD)
// Output: shader cycles per MFMA (s_memtime of wave 0) and wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;

#ifdef ACC_AGPR
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b))
#else
#define MFMA(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#endif
#define PKFMA(x, c, d) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d))
#define FMA(x, c, d) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d))
#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define MED3(x, c, d) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c), "v"(d))
#define CND(x, c) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(c))
#define MUL(x, c) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(x) : "v"(c))
#define CVT(x, y) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(x) : "v"(y))

using f32x2 = __attribute__((ext_vector_type(2))) float;
template <int MODE> __device__ __forceinline__ void filler(int j, float (&v)[8], float c, float d) {
    float& x = v[j & 7];
    if (MODE == 3) {
        f32x2 pv = {v[(2 * j) & 6], v[((2 * j) & 6) + 1]};
        const f32x2 pc = {c, c}, pd = {d, d};
        PKFMA(pv, pc, pd);
        v[(2 * j) & 6] = pv[0]; v[((2 * j) & 6) + 1] = pv[1];
        return;
    }
    if (MODE == 0) FMA(x, c, d);
    else if (MODE == 2) EXP(x);
    else {
        switch (j % 5) {
            case 0: MED3(x, c, d); break;
            case 1: CND(x, c); break;
            case 2: MUL(x, c); break;
            case 3: FMA(x, c, d); break;
            default: CVT(x, c); break;
        }
    }
}

// K fillers after every MFMA; CHAIN: 6 consecutive MFMAs share an accumulator (4 accumulators rotate per group of 6)
template <int K, int MODE, bool CHAIN, int NT>
__global__ __launch_bounds__(NT) void k_inter(float* out, long long* cyc, int iters, int mfma_on, int valu_role) {
    extern __shared__ float lds_force[];           // occupancy control
    const int wave = threadIdx.x >> 6;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    u32x4 a = {0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, b = a;
    a[0] += threadIdx.x; b[1] += threadIdx.x * 3;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 0.5f + 0.01f * (threadIdx.x + j);
    const float c = 0.999f, d = 1e-3f;
    // valu_role: 0 = every wave runs the interleaved stream; 1 = waves 0-3 MFMA only, waves 4-7 fillers only (cross-wave)
    const bool do_mfma = mfma_on && (valu_role == 0 || wave < 4);
    const bool do_valu = (valu_role == 0 || wave >= 4);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    if (do_mfma && do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 24; ++m) {
                if (CHAIN) MFMA(acc[(m / 6) & 3], a, b); else MFMA(acc[m & 3], a, b);
#pragma unroll
                for (int j = 0; j < K; ++j) filler<MODE>(m * K + j, v, c, d);
            }
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 24; ++m) { if (CHAIN) MFMA(acc[(m / 6) & 3], a, b); else MFMA(acc[m & 3], a, b); }
        }
    } else if (do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 24; ++m) {
#pragma unroll
                for (int j = 0; j < K; ++j) filler<MODE>(m * K + j, v, c, d);
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    float r = 0.f;
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][7];
    for (int j = 0; j < 8; ++j) r += v[j];
    if (r == 12345.678f) out[threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int K, int MODE, bool CHAIN, int NT>
void run(const char* label, int mfma_on, int role) {
    float* out; long long* cyc;
    hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 8);
    const int iters = 400;
    const size_t lds = 100 * 1024;                  // one workgroup per CU
    hipFuncSetAttribute((const void*)k_inter<K, MODE, CHAIN, NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    k_inter<K, MODE, CHAIN, NT><<<256, NT, lds>>>(out, cyc, iters, mfma_on, role); hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < 3; ++i) k_inter<K, MODE, CHAIN, NT><<<256, NT, lds>>>(out, cyc, iters, mfma_on, role);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); ms /= 3;
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n_mfma = 24.0 * iters;
    printf("%-34s K=%2d waves/SIMD=%d %s: %7.1f cycles per MFMA slot (wave 0), %.3f ms wall = %6.1f ns*2.4 per slot\n", label, K, NT / 256,
           CHAIN ? "chain6" : "indep ", (double)c / n_mfma, ms, ms * 1e6 * 2.4 / n_mfma);
    hipFree(out); hipFree(cyc);
}

template <int MODE, bool CHAIN, int NT> void sweep(const char* label) {
    run<0, MODE, CHAIN, NT>(label, 1, 0);
    run<2, MODE, CHAIN, NT>(label, 1, 0);
    run<4, MODE, CHAIN, NT>(label, 1, 0);
    run<5, MODE, CHAIN, NT>(label, 1, 0);
    run<6, MODE, CHAIN, NT>(label, 1, 0);
    run<8, MODE, CHAIN, NT>(label, 1, 0);
    run<13, MODE, CHAIN, NT>(label, 1, 0);
    run<13, MODE, CHAIN, NT>("  same fillers, no MFMA", 0, 0);
}

int main() {
#ifdef ACC_AGPR
    printf("==== accumulators in AGPRs\n");
#endif
    printf("== packed fma fillers (one v_pk_fma_f32 = two lanes' worth), one wave per SIMD, then two\n");
    sweep<3, false, 256>("pk_fma fillers");
    sweep<3, false, 512>("pk_fma fillers");
    printf("== plain fma fillers, two waves per SIMD in-wave, then cross-wave\n");
    sweep<0, false, 512>("fma fillers");
    run<5, 0, false, 512>("fma, cross-wave", 1, 1);
    run<8, 0, false, 512>("fma, cross-wave", 1, 1);
    run<13, 0, false, 512>("fma, cross-wave", 1, 1);
    run<13, 0, false, 512>("  fma fillers alone on waves 4-7", 0, 1);
    run<13, 2, false, 512>("exp, cross-wave", 1, 1);
    run<13, 2, false, 512>("  exp fillers alone on waves 4-7", 0, 1);
    printf("== one wave per SIMD, interleaved in the wave\n");
    sweep<0, false, 256>("fma fillers");
    sweep<1, false, 256>("mixed fillers");
    sweep<1, true, 256>("mixed fillers");
    sweep<2, false, 256>("exp fillers");
    printf("== two waves per SIMD, each interleaved in the wave\n");
    sweep<1, false, 512>("mixed fillers");
    sweep<1, true, 512>("mixed fillers");
    printf("== two waves per SIMD, MFMA wave beside filler wave (cross-wave)\n");
    run<5, 1, false, 512>("mixed, cross-wave", 1, 1);
    run<13, 1, false, 512>("mixed, cross-wave", 1, 1);
    run<13, 1, true, 512>("mixed, cross-wave", 1, 1);
    return 0;
}
