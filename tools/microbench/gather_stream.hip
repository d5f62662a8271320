// How fast can gathered rows of a resident [n, ld] fp32 matrix be streamed, as a function of the contiguous bytes a
// workgroup takes from a row at a time?  (The first layer's forward reads B = 4096 permuted rows of 20 000 floats per
// step: 328 MB.)  Each workgroup owns 128 rows x one K slab and walks the slab in chunks of SEG bytes per row; a wave
// reads 64 x 16 bytes per instruction = 1024 / SEG rows at once.  Sum into registers, one store per thread.
//   hipcc --offload-arch=gfx950 -O3 -o gather_stream gather_stream.hip && ./gather_stream
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

template <int SEG>   // bytes per row per step: 64 .. 1024
__global__ __launch_bounds__(256) void stream_kernel(const float* X, long ld, const int* perm, int rows, int kslab, float* out) {
    constexpr int LPR = SEG / 16;            // lanes per row segment
    constexpr int RPI = 256 / LPR;           // rows per instruction across the workgroup
    const int t = threadIdx.x;
    const int rt = blockIdx.x, ks = blockIdx.y;
    const int r0 = rt * 128;
    const long k0 = (long)ks * kslab;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const int lr = t / LPR, lu = t % LPR;
    for (long k = 0; k < kslab; k += SEG / 4) {
#pragma unroll
        for (int i = 0; i < 128 / RPI; ++i) {
            const int r = r0 + lr + i * RPI;
            const long row = perm[r < rows ? r : rows - 1];
            const float4 v = *reinterpret_cast<const float4*>(X + row * ld + k0 + k + lu * 4);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    out[(long)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + t] = acc.x + acc.y + acc.z + acc.w;
}

// the same stream through LDS-DMA (global_load_lds_dwordx4: 1 KB per wave instruction, lane-linear destination) into a
// ring of NST stages of 128 rows x SEG bytes, NST - 1 stages in flight, counted waits, nothing consumed
template <int SEG, int NST>
__global__ __launch_bounds__(256) void dma_kernel(const float* X, long ld, const int* perm, int rows, int kslab, float* out) {
    constexpr int LPR = SEG / 16, RPI = 256 / LPR, NI = 128 / RPI;      // NI DMA instructions per thread and stage
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int t = threadIdx.x, wave = t >> 6, lane = t & 63;
    const int r0 = blockIdx.x * 128;
    const long k0 = (long)blockIdx.y * kslab;
    const int lr = t / LPR, lu = t % LPR;
    const float* src[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const int r = r0 + lr + i * RPI;
        src[i] = X + (long)perm[r < rows ? r : rows - 1] * ld + k0 + lu * 4;
    }
    const int nsteps = kslab / (SEG / 4);
    auto request = [&](int st) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src[i],
                                             (__attribute__((address_space(3))) void*)(lds + st * 128 * SEG + (i * 4 + wave) * 1024), 16, 0, 0);
            src[i] += SEG / 4;
        }
    };
    for (int q = 0; q < NST - 1 && q < nsteps; ++q) request(q);
    float acc = 0.f;
    int cur = 0, nxt = NST - 1;
    for (int k = 0; k < nsteps; ++k) {
        if (k + NST - 1 <= nsteps) {
            if (NI * (NST - 2) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (NI * (NST - 2) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (NI * (NST - 2) == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (NI * (NST - 2) == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (NI * (NST - 2) == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            else if (NI * (NST - 2) == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        acc += reinterpret_cast<const float*>(lds + cur * 128 * SEG)[t];
        if (k + NST - 1 < nsteps) request(nxt);
        cur = cur == NST - 1 ? 0 : cur + 1; nxt = nxt == NST - 1 ? 0 : nxt + 1;
    }
    out[(long)(blockIdx.y * gridDim.x + blockIdx.x) * 256 + t] = acc;
}

template <int SEG, int NST>
void run_dma(const float* X, long ld, const int* perm, int rows, int K, float* out, int splits) {
    const int kslab = K / splits / (SEG / 4) * (SEG / 4);
    dim3 grid((rows + 127) / 128, splits);
    const int bytes = NST * 128 * SEG;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&dma_kernel<SEG, NST>), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((dma_kernel<SEG, NST>), grid, dim3(256), bytes, 0, X, ld, perm + 11 * rows, rows, kslab, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((dma_kernel<SEG, NST>), grid, dim3(256), bytes, 0, X, ld, perm + i * rows, rows, kslab, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    const double tot = (double)rows * kslab * splits * 4;
    printf("LDS-DMA segment %4d B, %d stages (%3d KB LDS), %2d K slabs: %.3f ms  %.2f TB/s\n", SEG, NST, bytes >> 10, splits, ms, tot / ms / 1e9);
}

template <int SEG>
void run(const float* X, long ld, const int* perm, int rows, int K, float* out, int splits) {
    const int kslab = K / splits / (SEG / 4) * (SEG / 4);
    dim3 grid((rows + 127) / 128, splits);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL(stream_kernel<SEG>, grid, dim3(256), 0, 0, X, ld, perm + 11 * rows, rows, kslab, out);
    hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(stream_kernel<SEG>, grid, dim3(256), 0, 0, X, ld, perm + i * rows, rows, kslab, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
    const double bytes = (double)rows * kslab * splits * 4;
    printf("segment %4d B, %2d K slabs: %.3f ms  %.2f TB/s\n", SEG, splits, ms, bytes / ms / 1e9);
}

int main() {
    const int n = 68579, G = 20000, rows = 4096;
    const long ld = G;
    float* X; hipMalloc(&X, (size_t)n * ld * 4); hipMemset(X, 0x3c, (size_t)n * ld * 4);
    std::vector<int> p(n); std::iota(p.begin(), p.end(), 0);
    std::mt19937 rng(1); std::shuffle(p.begin(), p.end(), rng);
    // a fresh set of rows per launch (as consecutive minibatches are): the Infinity Cache must not serve the stream
    int* perm; hipMalloc(&perm, (size_t)12 * rows * 4); hipMemcpy(perm, p.data(), (size_t)12 * rows * 4, hipMemcpyHostToDevice);
    float* out; hipMalloc(&out, 64 << 20);
    for (int splits : {16, 32}) {
        run<64>(X, ld, perm, rows, G, out, splits);
        run<128>(X, ld, perm, rows, G, out, splits);
        run<256>(X, ld, perm, rows, G, out, splits);
        run<512>(X, ld, perm, rows, G, out, splits);
        run<1024>(X, ld, perm, rows, G, out, splits);
    }
    run_dma<64, 3>(X, ld, perm, rows, G, out, 32);
    run_dma<64, 6>(X, ld, perm, rows, G, out, 32);
    run_dma<256, 3>(X, ld, perm, rows, G, out, 32);
    run_dma<256, 4>(X, ld, perm, rows, G, out, 32);
    run_dma<64, 3>(X, ld, perm, rows, G, out, 64);
    run_dma<256, 3>(X, ld, perm, rows, G, out, 64);
    return 0;
}
