// How much of a short kernel's latency is instruction fetch?  The same 4096 dependent-free v_fma per lane executed
// (a) as a 16-instruction loop, (b) straight-line (32 KB of code), (c) straight-line twice in a row inside ONE launch
// (second pass = warm instruction cache).  One wave, launched back to back; time per launch from HIP events.
//   hipcc --offload-arch=gfx950 -O3 -o icache_probe icache_probe.hip && ./icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define F1 asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
#define F4 F1 F1 F1 F1
#define F16 F4 F4 F4 F4
#define F64 F16 F16 F16 F16
#define F256 F64 F64 F64 F64
#define F1024 F256 F256 F256 F256
#define F4096 F1024 F1024 F1024 F1024

__global__ void k_loop(float* o, int n) {
    float x = threadIdx.x, y = 1.0001f;
    for (int i = 0; i < n; ++i) { F16 }
    o[threadIdx.x] = x;
}
__global__ void k_straight(float* o, long long* t) {
    float x = threadIdx.x, y = 1.0001f;
    long long t0 = __builtin_readcyclecounter();
    F4096
    long long t1 = __builtin_readcyclecounter();
    F4096
    long long t2 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { t[0] = t1 - t0; t[1] = t2 - t1; }
    o[threadIdx.x] = x;
}
__global__ void k_straight_loop2(float* o, long long* t) {
    float x = threadIdx.x, y = 1.0001f;
    long long tt[2];
    for (int r = 0; r < 2; ++r) {
        long long t0 = __builtin_readcyclecounter();
        F4096
        tt[r] = __builtin_readcyclecounter() - t0;
        asm volatile("" : "+v"(x));
    }
    if (threadIdx.x == 0) { t[0] = tt[0]; t[1] = tt[1]; }
    o[threadIdx.x] = x;
}
__global__ void k_empty(float* o) { o[threadIdx.x] = 1.f; }

template <class F> float timeit(F f, int n = 200) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a);
    for (int i = 0; i < n; ++i) f();
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / n;
}
int main() {
    float* o; long long* t; hipMalloc(&o, 4096); hipMalloc(&t, 64);
    long long h[2];
    printf("empty kernel                          %.2f us per launch\n", timeit([&] { hipLaunchKernelGGL(k_empty, 1, 64, 0, 0, o); }));
    printf("4096 fma as a 16-instruction loop     %.2f us per launch\n", timeit([&] { hipLaunchKernelGGL(k_loop, 1, 64, 0, 0, o, 256); }));
    printf("8192 fma as a 16-instruction loop     %.2f us per launch\n", timeit([&] { hipLaunchKernelGGL(k_loop, 1, 64, 0, 0, o, 512); }));
    printf("2 x 4096 fma straight-line (2 x 32 KB) %.2f us per launch\n", timeit([&] { hipLaunchKernelGGL(k_straight, 1, 64, 0, 0, o, t); }));
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("   cycles (s_memtime): first 4096 %lld, second 4096 (other addresses, also cold) %lld\n", h[0], h[1]);
    printf("2 passes over ONE 32 KB block          %.2f us per launch\n", timeit([&] { hipLaunchKernelGGL(k_straight_loop2, 1, 64, 0, 0, o, t); }));
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("   cycles (s_memtime): cold pass %lld, warm pass %lld\n", h[0], h[1]);
    return 0;
}
