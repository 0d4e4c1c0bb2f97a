// fp32 -> three bf16 planes ("bf16x3" operand format), shared by every kernel that produces GEMM operands.
//
//   x = x0 + x1 + x2,   x0 = bf16_rn(x),  x1 = bf16_rn(x - x0),  x2 = bf16_rn(x - x0 - x1)
//
// Both subtractions are exact in fp32 (the residual of a rounding to fewer bits is representable), |x1| <= 2^-8 |x|,
// |x2| <= 2^-16 |x|, and x2 is the exact remainder (24 - 16 = 8 significant bits left): (x2 + x1) + x0 == x bit for bit,
// so the planes REPLACE the fp32 matrix (bf3_join below), they are not an approximation of it.
// A GEMM forms a * b as a0b2 + a1b1 + a2b0 + a0b1 + a1b0 + a0b0 on the bf16 matrix pipe (fp32 accumulate); the dropped
// cross terms are below 2^-24 |ab| (dense.hip has the derivation and the accuracy tests).
// Out of range: +-inf or a value whose bf16 rounding overflows gives inf - inf = NaN in x1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bf3 {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// two fp32 values -> their three bf16 terms (v_cvt_pk_bf16_f32 + exact fp32 subtractions)
__device__ __forceinline__ void split2(float v0, float v1, bf16x2& p0, bf16x2& p1, bf16x2& p2) {
    const f32x2 a = {v0, v1};
    p0 = __builtin_convertvector(a, bf16x2);
    const f32x2 r1 = a - __builtin_convertvector(p0, f32x2);
    p1 = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(p1, f32x2);
    p2 = __builtin_convertvector(r2, bf16x2);
}

__device__ __forceinline__ void split4(float v0, float v1, float v2, float v3, bf16x4& p0, bf16x4& p1, bf16x4& p2) {
    bf16x2 a0, a1, a2, b0, b1, b2;
    split2(v0, v1, a0, a1, a2);
    split2(v2, v3, b0, b1, b2);
    p0 = bf16x4{a0[0], a0[1], b0[0], b0[1]};
    p1 = bf16x4{a1[0], a1[1], b1[0], b1[1]};
    p2 = bf16x4{a2[0], a2[1], b2[0], b2[1]};
}

// exact inverse: (x2 + x1) + x0
__device__ __forceinline__ float join(__bf16 p0, __bf16 p1, __bf16 p2) { return ((float)p2 + (float)p1) + (float)p0; }

__device__ __forceinline__ float4 join4(bf16x4 p0, bf16x4 p1, bf16x4 p2) {
    return make_float4(join(p0[0], p1[0], p2[0]), join(p0[1], p1[1], p2[1]), join(p0[2], p1[2], p2[2]),
                       join(p0[3], p1[3], p2[3]));
}

// 4 consecutive elements of row-major planes at element offset `off` (off % 4 == 0, planes 8-byte aligned)
__device__ __forceinline__ void store4(__bf16* __restrict__ planes, int64_t plane_stride, int64_t off, float v0, float v1,
                                       float v2, float v3) {
    bf16x4 p0, p1, p2;
    split4(v0, v1, v2, v3, p0, p1, p2);
    *reinterpret_cast<bf16x4*>(planes + off) = p0;
    *reinterpret_cast<bf16x4*>(planes + plane_stride + off) = p1;
    *reinterpret_cast<bf16x4*>(planes + 2 * plane_stride + off) = p2;
}

__device__ __forceinline__ float4 load4(const __bf16* __restrict__ planes, int64_t plane_stride, int64_t off) {
    const bf16x4 p0 = *reinterpret_cast<const bf16x4*>(planes + off);
    const bf16x4 p1 = *reinterpret_cast<const bf16x4*>(planes + plane_stride + off);
    const bf16x4 p2 = *reinterpret_cast<const bf16x4*>(planes + 2 * plane_stride + off);
    return join4(p0, p1, p2);
}

}  // namespace bf3
