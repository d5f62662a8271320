// fp32 values as TWO fp16 pieces after a power-of-two block scale -- the arithmetic of K-HEADS (dcahip_heads.hip, which states
// it in full) for the plane GEMMs and the likelihood kernel that writes gradient planes.  x 2^e = h1 + h2 (round to nearest):
// 2^-22 relative, 2^-25 absolute where h2 is an fp16 denormal (preserved by the matrix pipe of gfx950); three products
// a1 b1 + a1 b2 + a2 b1 per fp32 product.  No reference counterpart (the reference multiplies in fp32 on the CPU).
#pragma once
#include <hip/hip_runtime.h>

namespace {

using h2_f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using h2_f32x2 = __attribute__((ext_vector_type(2))) float;
using h2_f16x8 = __attribute__((ext_vector_type(8))) _Float16;

constexpr int kH2Top = 13;              // a scaled block's largest magnitude lies in [2^13, 2^14)

__device__ __forceinline__ unsigned h2_pk(float a, float b) {             // v_cvt_pk_f16_f32 (round to nearest): a -> low half
    return __builtin_bit_cast(unsigned, __builtin_convertvector(h2_f32x2{a, b}, h2_f16x2));
}
__device__ __forceinline__ float h2_resid_lo(unsigned p, float x) {       // x - (float) low half of p, exact
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
__device__ __forceinline__ float h2_resid_hi(unsigned p, float x) {
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(x));
    return r;
}
__device__ __forceinline__ void h2_split_pair(float x0, float x1, unsigned& p0, unsigned& p1) {
    p0 = h2_pk(x0, x1);
    p1 = h2_pk(h2_resid_lo(p0, x0), h2_resid_hi(p0, x1));
}
__device__ __forceinline__ float h2_pow2i(int e) { return __uint_as_float((unsigned)(127 + e) << 23); }      // -126 <= e <= 127
// the exponent that brings a block's largest magnitude m into [2^13, 2^14); 0 for an all-zero (or non-finite) block
__device__ __forceinline__ int h2_block_exp(float m) {
    if (!(m > 0.f) || !(m < INFINITY)) return 0;
    const int e = kH2Top + 1 - __builtin_amdgcn_frexp_expf(m);
    return e < -60 ? -60 : (e > 60 ? 60 : e);
}

}  // namespace
