// Does the fp16 matrix pipe of gfx950 preserve fp16 DENORMAL inputs?  (v_mfma_f32_32x32x16_f16; also v_dot2_f32_f16 and the
// fp32 -> fp16 conversions.)  Decides whether an fp32 operand can be carried as TWO fp16 pieces (x 2^e = h1 + h2, |h2| <= 2^-11
// |h1|: h2 is denormal whenever |x 2^e| < 2^-3) instead of three bf16 pieces.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_f16_denorm tools/microbench/mfma_f16_denorm.hip && /tmp/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdio.h>
#include <math.h>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using h8 = __attribute__((ext_vector_type(8))) _Float16;
using h2v = __attribute__((ext_vector_type(2))) _Float16;
using f2v = __attribute__((ext_vector_type(2))) float;

__global__ void k(const float* in, float* out) {
    const float a = in[0], b = in[1];               // a: a value that is an fp16 denormal, b: a normal fp16
    h8 A, B;
    for (int j = 0; j < 8; ++j) { A[j] = (_Float16)a; B[j] = (_Float16)b; }
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, acc, 0, 0, 0);      // every output element = 16 a b
    f32x16 acc2;
    for (int e = 0; e < 16; ++e) acc2[e] = 0.f;
    acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(B, A, acc2, 0, 0, 0);    // denormal on the B side
    if (threadIdx.x == 0) {
        out[0] = acc[0]; out[1] = acc2[0];
        out[2] = (float)(_Float16)a;                                       // the conversion itself keeps the denormal?
        h2v pa = __builtin_convertvector(f2v{a, a * 3.f}, h2v);           // packed conversion (v_cvt_pk_f16_f32 on gfx950?)
        out[3] = (float)pa[0]; out[4] = (float)pa[1];
        h2v ones = {(_Float16)1.f, (_Float16)1.f};
        out[5] = __builtin_amdgcn_fdot2(pa, ones, 0.f, false);             // v_dot2_f32_f16 with denormal inputs
        // residual through a mixed fma: x - (float)h
        const float x = in[2];
        const _Float16 h = (_Float16)x;
        out[6] = fmaf((float)h, -1.f, x);
        out[7] = (float)(_Float16)out[6];
    }
}

int main() {
    float h_in[3] = {ldexpf(1.f, -20), 1024.f, 0.1234567f}, h_out[8];
    float *d_in, *d_out;
    hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d_in, d_out);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    printf("a = 2^-20 (fp16 denormal), b = 1024: expected 16 a b = %g\n", 16.0 * ldexp(1.0, -20) * 1024.0);
    printf("  mfma, denormal in A: %g   denormal in B: %g   -> %s\n", h_out[0], h_out[1],
           (h_out[0] != 0.f && h_out[1] != 0.f) ? "PRESERVED" : "FLUSHED");
    printf("  (float)(half)a = %g (want %g); packed convert: %g %g (want %g %g); dot2(pa, ones) = %g (want %g)\n", h_out[2], h_in[0], h_out[3], h_out[4],
           h_in[0], 3 * h_in[0], h_out[5], 4 * h_in[0]);
    printf("  x = %.9g: x - (float)(half)x = %.9g, its half = %.9g\n", h_in[2], h_out[6], h_out[7]);
    return 0;
}
