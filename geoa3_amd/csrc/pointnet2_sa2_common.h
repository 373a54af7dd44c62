// Helpers shared by the PointNet++ level-2 kernels (pointnet2_sa2.hip: forward, pre-transform, the centre-major backward;
// pointnet2_sa2b.hip: the destination-ordered backward): tile geometry, power-of-two operand scales, the two-instruction
// fp16 hi / lo operand split.
#pragma once
#include "pointnet_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int S2_K = 128;                // channels of a0 and a1
constexpr int S2_C = 256;                // pooled channels
constexpr int S2_PT = S2_K + 4;          // floats per row (= sample) of a tile
constexpr int S2_PF = 32;                // W2 rows in flight per lane in phase 1 (one wave per SIMD: L2 latency is exposed)

__device__ __forceinline__ unsigned s2_exp(float m) {
  const unsigned E = (__float_as_uint(m) >> 23) & 0xffu;
  return E < 14u ? 14u : (E > 254u ? 254u : E);
}
__device__ __forceinline__ float s2_scale(unsigned E) { return __uint_as_float((267u - E) << 23); }     // max -> [2^13, 2^14)
__device__ __forceinline__ float s2_unscale(unsigned E) { return __uint_as_float((E - 13u) << 23); }
__device__ __forceinline__ void s2_swap32(float& a, float& b) {   // a[32..63] <-> b[0..31]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
// (hi, lo) fp16 images of x0 * s and x1 * s, packed: v_pk_mul_f32 + v_cvt_pk_f16_f32 for the hi pieces, one v_fma_mixlo /
// mixhi_f16 per lo piece (the residual x * s - hi as one exact fma): two instructions per element where the scalar form
// took five (the file is compiled with -fno-slp-vectorize: the SLP vectoriser turns the residual pair into three).  The
// same bits as hi = rn16(x s), lo = rn16(x s - hi): s is a power of two.
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void s2_split2(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
  const float2v x = {x0, x1};
  // (the scale as a VECTOR register operand of the packed multiply: packed-FP32 instructions with SGPR-pair operands at
  // two waves per SIMD are what computed wrong values in conv_bwd_chain_kernel -- NOTEBOOK 5a; none are formed here)
  float sv = s;
  asm volatile("" : "+v"(sv));
  const half2v h = __builtin_convertvector(x * sv, half2v);
  const half2v l = {(_Float16)__builtin_fmaf(x0, s, -(float)h[0]), (_Float16)__builtin_fmaf(x1, s, -(float)h[1])};
  hi = __builtin_bit_cast(unsigned, h);
  lo = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ void s2_split8(const float (&x)[8], float s, half8& oh, half8& ol) {
  uint4v H, Lw;
#pragma unroll
  for (int j2 = 0; j2 < 4; ++j2) {
    unsigned a, b;
    s2_split2(x[2 * j2], x[2 * j2 + 1], s, a, b);
    H[j2] = a;
    Lw[j2] = b;
  }
  oh = __builtin_bit_cast(half8, H);
  ol = __builtin_bit_cast(half8, Lw);
}
__device__ __forceinline__ int s2_rl(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float s2_rlf(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

}  // namespace
