// K-DROP: Keras Dropout (dca/network.py:98-99, 137-138) with a counter-based generator, so the
// backward pass recomputes the mask instead of storing it and a data-parallel run draws the masks of
// the single-process run (include/dcahip.h).  One thread per group of 4 columns = one Philox block.
#include <hip/hip_runtime.h>
#include "dcahip.h"

namespace {

struct DropArgs {
    const float* x; long ldx;
    const int* perm; const long long* cursor;
    int B, h, hq;
    float rate, scale;
    unsigned k0, k1;
    const long long* step;
    unsigned layer;
    long row0;
    float* out; long ldo;
};

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                               unsigned (&o)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned lo0 = 0xD2511F53u * c0, hi0 = __umulhi(0xD2511F53u, c0);
        const unsigned lo1 = 0xCD9E8D57u * c2, hi1 = __umulhi(0xCD9E8D57u, c2);
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

__global__ __launch_bounds__(256) void dropout_kernel(DropArgs p) {
    const long total = (long)p.B * p.hq;
    const unsigned step = (unsigned)(p.step ? *p.step : 0);
    const long cur = (p.perm && p.cursor) ? (long)*p.cursor : 0;
    for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
        const int r = (int)(t / p.hq), q = (int)(t - (long)r * p.hq);
        const unsigned long long grp = (unsigned long long)(p.row0 + r) * (unsigned long long)p.hq + (unsigned long long)q;
        unsigned o[4];
        philox4x32_10((unsigned)grp, (unsigned)(grp >> 32), step, p.layer, p.k0, p.k1, o);
        const long src = p.perm ? (long)p.perm[cur + r] : (long)r;
        const float* xr = p.x + src * p.ldx + 4 * q;
        float* orow = p.out + (long)r * p.ldo + 4 * q;
        const int nc = min(4, p.h - 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j < nc) {
                const float u = (float)(o[j] >> 8) * 5.9604644775390625e-08f;      // 2^-24, exact
                orow[j] = (u >= p.rate) ? xr[j] * p.scale : 0.f;
            }
        }
    }
}

}  // namespace

extern "C" int dcahip_dropout_apply(const float* x, long ldx, const int* perm, const long long* cursor, int B, int h,
                                    float rate, unsigned long long seed, const long long* step, int layer, long row0,
                                    float* out, long ldo, void* stream) {
    if (!x || !out || B < 0 || h <= 0 || ldx < h || ldo < h || !(rate >= 0.f) || !(rate < 1.f) || row0 < 0) return DCAHIP_EINVAL;
    if (perm && !cursor) return DCAHIP_EINVAL;
    if (perm && x == out) return DCAHIP_EINVAL;
    if (B == 0) return 0;
    DropArgs a;
    a.x = x; a.ldx = ldx; a.perm = perm; a.cursor = cursor; a.B = B; a.h = h; a.hq = (h + 3) / 4;
    a.rate = rate; a.scale = 1.f / (1.f - rate);
    a.k0 = (unsigned)seed; a.k1 = (unsigned)(seed >> 32);
    a.step = step; a.layer = (unsigned)layer; a.row0 = row0; a.out = out; a.ldo = ldo;
    long blocks = ((long)B * a.hq + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(dropout_kernel, dim3((int)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), a);
    return (int)hipGetLastError();
}
