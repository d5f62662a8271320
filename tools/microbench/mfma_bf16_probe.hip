// Probe of ds_read_b64_tr_b16 (gfx950 LDS transpose read) semantics: LDS holds lds16[i] = i; lane l supplies the
// address of the 4 shorts [4l, 4l+4).  Prints, per lane and result element j, the LDS index received -- i.e. which
// (source lane, position) each result element comes from.  Expected (MI355X guide): within every 16-lane group,
// result[t][j] = data of lane 4j + (t >> 2), position t & 3.
// Part 2: issue-slot model -- cycles per v_mfma_f32_32x32x16_bf16 with 0..12 independent VALU between consecutive
// MFMAs of ONE wave (1 or 2 waves per SIMD), i.e. how much element-wise work hides in the MFMA shadow.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__global__ void probe(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int l = threadIdx.x;
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + 4 * l));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}

template <int NV>
__global__ __launch_bounds__(512) void shadow(float* out, int iters, int waves_active) {
    const int wave = threadIdx.x >> 6;
    if (wave >= waves_active) return;
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    union { bf16x8 v; unsigned short h[8]; } x, y;
    for (int j = 0; j < 8; ++j) { x.h[j] = 0x3f80 + (threadIdx.x & 7); y.h[j] = 0x3f00 + j; }
    float v[12];
    for (int j = 0; j < 12; ++j) v[j] = threadIdx.x * 1e-3f + j;
    const float c = 0.999f, d = 1e-3f;
    for (int i = 0; i < iters; ++i) {
#define STEP(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.v, y.v, acc, 0, 0, 0); \
        _Pragma("unroll") for (int j = 0; j < NV; ++j) v[j] = fmaf(v[j], c, d);
        STEP(a0) STEP(a1) STEP(a2) STEP(a3)
#undef STEP
    }
    float r = a0[0] + a1[1] + a2[2] + a3[3];
    for (int j = 0; j < 12; ++j) r += v[j];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

// Part 3: dependent accumulation -- the six products of one K step go into ONE accumulator in K-HEADS; NACC = number
// of accumulators the MFMA stream cycles over (1 = every MFMA waits for its predecessor's result)
template <int NACC>
__global__ __launch_bounds__(512) void chain(float* out, int iters, int waves_active) {
    const int wave = threadIdx.x >> 6;
    if (wave >= waves_active) return;
    f32x16 a[NACC];
    for (int k = 0; k < NACC; ++k) a[k] = f32x16{0};
    union { bf16x8 v; unsigned short h[8]; } x, y;
    for (int j = 0; j < 8; ++j) { x.h[j] = 0x3f80 + (threadIdx.x & 7); y.h[j] = 0x3f00 + j; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int rep = 0; rep < 12 / NACC; ++rep)
#pragma unroll
            for (int k = 0; k < NACC; ++k) a[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.v, y.v, a[k], 0, 0, 0);
    }
    float r = 0.f;
    for (int k = 0; k < NACC; ++k) r += a[k][k];
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int NACC>
float run_chain(int iters, int waves) {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    chain<NACC><<<256, 512>>>(out, iters, waves); hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < 5; ++i) chain<NACC><<<256, 512>>>(out, iters, waves);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); hipFree(out);
    return ms / 5;
}

template <int NV>
float run(int iters, int waves) {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    shadow<NV><<<256, 512>>>(out, iters, waves); hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < 5; ++i) shadow<NV><<<256, 512>>>(out, iters, waves);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); hipFree(out);
    return ms / 5;
}

// Part 4: packed fp32 -- does v_pk_fma_f32 (two fp32 FMAs per lane) issue at the rate of one v_fma_f32?  PK = 0: 16
// independent v_fma_f32 chains, PK = 1: 8 independent v_pk_fma_f32 chains (same flops); waves 4-7 optionally run bf16 MFMAs
// on the same SIMDs (which: 1 = VALU waves only, 3 = both).
typedef __attribute__((ext_vector_type(2))) float f32x2p;
template <int PK>
__global__ __launch_bounds__(512) void packed(float* out, int iters, int which) {
    const int wave = threadIdx.x >> 6;
    float r = 0.f;
    if (wave < 4) {
        const float c = 0.999f, d = 1e-3f;
        if (PK == 0) {
            float v[16];
            for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 1e-3f + j;
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = fmaf(v[j], c, d);
            }
            for (int j = 0; j < 16; ++j) r += v[j];
        } else {
            f32x2p v[8];
            const f32x2p cc = {c, c}, dd = {d, d};
            for (int j = 0; j < 8; ++j) v[j] = f32x2p{threadIdx.x * 1e-3f + j, threadIdx.x * 2e-3f + j};
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __builtin_elementwise_fma(v[j], cc, dd);
            }
            for (int j = 0; j < 8; ++j) r += v[j][0] + v[j][1];
        }
    } else if (which & 2) {
        f32x16 a0 = {0}, a1 = {0};
        union { bf16x8 v; unsigned short h[8]; } x, y;
        for (int j = 0; j < 8; ++j) { x.h[j] = 0x3f80 + (threadIdx.x & 7); y.h[j] = 0x3f00 + j; }
        for (int i = 0; i < iters / 4; ++i) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x.v, y.v, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(y.v, x.v, a1, 0, 0, 0);
        }
        r = a0[0] + a1[1];
    }
    if (r == 12345.678f) out[threadIdx.x] = r;
}

template <int PK>
float run_packed(int iters, int which) {
    float* out; hipMalloc(&out, 4096);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    packed<PK><<<256, 512>>>(out, iters, which); hipDeviceSynchronize();
    hipEventRecord(s);
    for (int i = 0; i < 5; ++i) packed<PK><<<256, 512>>>(out, iters, which);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e); hipFree(out);
    return ms / 5;
}

int main() {
    unsigned short* d; hipMalloc(&d, 512);
    probe<<<1, 64>>>(d);
    std::vector<unsigned short> h(256);
    hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
    int ok = 1;
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) {
            const int idx = h[l * 4 + j];
            printf("  [src lane %2d pos %d]", idx >> 2, idx & 3);
            const int t = l & 15, g = l & ~15;
            if (idx != 4 * (g + 4 * j + (t >> 2)) + (t & 3)) ok = 0;
        }
        printf("\n");
    }
    printf("tr16 semantic as expected (result[t][j] = lane 4j + t/4, pos t%%4 within the 16-lane group): %s\n", ok ? "YES" : "NO");
    const int iters = 4000;       // 16000 MFMAs per wave
#define ROW(NV) { const float t4 = run<NV>(iters, 4), t8 = run<NV>(iters, 8); \
        printf("VALU per MFMA %2d: 1 wave/SIMD %.3f ms = %.1f cyc/MFMA;  2 waves/SIMD %.3f ms = %.1f cyc per MFMA-pair-slot (per wave-MFMA %.1f)\n", \
               NV, t4, t4 * 2.4e6 / (4.0 * iters), t8, t8 * 2.4e6 / (4.0 * iters), t8 * 2.4e6 / (8.0 * iters)); }
    ROW(0) ROW(2) ROW(4) ROW(6) ROW(8) ROW(12)
#define CROW(NA) { const float t4 = run_chain<NA>(1400, 4), t8 = run_chain<NA>(1400, 8); \
        printf("accumulators %d: 1 wave/SIMD %.3f ms = %.1f cyc/MFMA;  2 waves/SIMD %.3f ms = %.1f cyc per wave-MFMA\n", \
               NA, t4, t4 * 2.4e6 / (12.0 * 1400), t8, t8 * 2.4e6 / (2 * 12.0 * 1400)); }
    CROW(1) CROW(2) CROW(3) CROW(4) CROW(6)
    {
        const int it = 20000;        // 16 fp32 FMAs per lane and iteration in both forms
        const float s1 = run_packed<0>(it, 1), p1 = run_packed<1>(it, 1), s3 = run_packed<0>(it, 3), p3 = run_packed<1>(it, 3);
        printf("16 fp32 FMA / iteration: v_fma_f32 x16 %.3f ms (%.2f cyc per instruction), v_pk_fma_f32 x8 %.3f ms (%.2f cyc per instruction);  "
               "beside a partner wave's bf16 MFMAs: %.3f / %.3f ms\n", s1, s1 * 2.4e6 / (16.0 * it), p1, p1 * 2.4e6 / (8.0 * it), s3, p3);
    }
    return 0;
}
