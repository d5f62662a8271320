// PMC calibration: known-byte streaming kernels with the access widths K-HEADS uses
// (dword loads / stores, 128 B per half-wave) and dwordx4 for reference.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy_dword(const float* __restrict__ a, float* __restrict__ b, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i] + 1.f;
}
__global__ void copy_dwordx4(const float4* __restrict__ a, float4* __restrict__ b, long n4) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        float4 v = a[i]; v.x += 1.f; b[i] = v;
    }
}
__global__ void read_dword(const float* __restrict__ a, float* __restrict__ out, long n) {
    float s = 0.f;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s += a[i];
    if (s == 123.456f) out[0] = s;
}
// K-HEADS' count access: every half-wave reads one 128-byte segment (32 genes) of a different row
__global__ void read_rowseg(const float* __restrict__ a, float* __restrict__ out, long rows, long ld) {
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
    const long segs = ld / 32;
    float s = 0.f;
    // wave w walks tiles (32 rows x 32 genes) like the kernel: 16 loads, rows e + 4*hi pattern
    for (long t = wave; t < (rows / 32) * segs; t += nwaves) {
        const long rt = t / segs, seg = t - rt * segs;
        for (int e = 0; e < 16; ++e) {
            const long row = rt * 32 + (e & 3) + 8 * (e >> 2) + 4 * hi;
            s += a[row * ld + seg * 32 + l31];
        }
    }
    if (s == 123.456f) out[0] = s;
}
int main() {
    const long n = 1L << 29;   // 2 GiB per buffer (> 256 MiB Infinity Cache)
    float *a, *b; hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
    for (int r = 0; r < 2; ++r) {
        copy_dword<<<4096, 256>>>(a, b, n);
        copy_dwordx4<<<4096, 256>>>((const float4*)a, (float4*)b, n / 4);
        read_dword<<<4096, 256>>>(a, b, n);
        read_rowseg<<<4096, 256>>>(a, b, n / 20000 / 32 * 32, 20000);
    }
    hipDeviceSynchronize();
    printf("bytes per buffer %ld, rowseg bytes %ld\n", n * 4, (n / 20000 / 32 * 32) * 20000 * 4);
    return 0;
}
