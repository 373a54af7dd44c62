// PointNet++ SSG, first set-abstraction level, fused (gfx950): ball-grouped xyz -> shared MLP 3->64->64->128 (Conv2d 1x1 +
// eval-mode BatchNorm2d folded + ReLU) -> max over the 64 samples, forward and input-gradient.
// Reference: pointnet2_modules.py:29-74 (_PointnetSAModuleBase.forward), pointnet2_utils.py:296-333 (QueryAndGroup),
// PointNetPP_ssg.py:58-66 (npoint 512, radius 0.2, nsample 64, mlp [0(+3), 64, 64, 128]).
//
// The reference materialises [B,64,512,64] and [B,128,512,64] activations (2.1 + 2.1 + 4.2 GB at B = 250) and walks
// them once per layer and once more per layer in backward.  Here ONE WAVEFRONT owns one centroid and takes its two
// 32-sample column blocks one after the other; EVERY layer runs on the f16 matrix core with SPLIT fp32 operands (the
// arithmetic of pointnet_wide_split.hip: v = hi + lo, a*w = a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, fp32 accumulation;
// per-block power-of-two activation scales, weights scaled and split into fp16 LDS images when they are staged):
//   * layer 1 (K = 3 + bias) is ONE k-step: the hi / lo pieces of the point and of the weights share the K dimension
//     (round 5; on the VALU it and its relu masks were 700 of the backward's 3300 vector instructions per centroid);
//   * its accumulator registers 8s..8s+7 ARE the B operand of a k-step of layer 2 -- channels in the accumulator's row
//     order, which the W2 image in LDS follows (position ks*16 + 8h + j <-> channel 32(ks>>1) + 16(ks&1) + 4h + (j&3) +
//     8(j>>2)); the bias of layer 2 enters as the accumulators' initial value, so the relu gate is their sign;
//   * layer 3 TRANSPOSED (activations as A: rows = samples; weights as B: columns = channels), so a lane ends with ONE
//     channel and 16 of the block's samples: the max over samples is lane-local + one exchange;
//   no activation ever leaves the registers, an operand split is two v_fma_mix per element.
// Backward recomputes the two hidden layers (cheaper than reading 6 GB; the gates are the signs of the recomputed
// accumulators: no masks), routes the pooled gradient through a one-hot B operand against a W3^T image, chains the
// accumulator registers again as B operands against a W2^T image, and finishes with the K = 3 contraction and the
// scatter to the points.
// History: fp32 MFMA 1.67 / 2.61 ms (forward / backward, B = 250); split fp16 with layer 1 on the VALU 0.68 / 1.13 ms.
#include "pointnet_kernels.h"
#include "profile.h"

namespace {

// One 8-wave workgroup per CU = two waves per SIMD, forward and backward.  Measured (tools/bench_sa1.py, B = 250): three
// and four waves per SIMD (12-wave workgroups at 168 VGPRs, two 8-wave workgroups at 128) are SLOWER (backward 0.81 ->
// 0.93 ms, forward 0.65 -> 0.75 ms), one wave per SIMD much slower (1.10 / 0.71 ms); s_setprio around the matrix phases
// and sched_barrier-pinned LDS prefetch do not help either.
constexpr int SA_WF = 8;
constexpr int SA_WB = 8;
constexpr int SA_S = 64;           // samples per centroid
constexpr int SA_PH = 64 * 2 + 16;    // bytes per row of a 64-k fp16 image (conflict-free 16-byte reads)
constexpr int SA_PH2 = 128 * 2 + 16;  // bytes per row of the 128-k image (W3^T)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef unsigned uint4v __attribute__((ext_vector_type(4)));

struct Sa1Lds {
  float* w1;            // backward: [64][3] fp32 rows for the K = 3 contraction
  float* b2;            // [64]
  float* b3;            // forward: [128]
  float* red;           // [48] reduction scratch of the staging pass
  int* kw;              // [3]: log2 of the power-of-two scales of the W1 / W2 / W3 images
  unsigned char* w1a;   // [64 out][16 k] layer 1 as ONE k-step: (wh.xyz, bh | wh.xyz, bl | wl.xyz, 0 | wl.xyz, 0)
  unsigned char* w2h;   // [64 out][64 positions] hi, lo at + 64 * SA_PH          (A operand of layer 2)
  unsigned char* w3h;   // forward: [128 out][64 positions] hi, lo at + 128 * SA_PH   (B operand of layer 3)
                        // backward: W3^T [64 i][128 ch] hi, lo at + 64 * SA_PH2      (A operand of d h2)
  unsigned char* w2t;   // backward: W2^T [64 i][64 positions] hi, lo at + 64 * SA_PH (A operand of d h1)
  unsigned* pa;         // backward: [waves][384] one-hot records of the wave's centroid (below)
};
constexpr int sa1_lds_bytes(int waves, bool bwd) {
  return (64 * 4 + 64 + 128 + 48 + 16) * 4 + 64 * 32 + 2 * 64 * SA_PH +
         (bwd ? 2 * 64 * SA_PH2 + 2 * 64 * SA_PH + waves * 1536 : 2 * 128 * SA_PH);
}

__device__ __forceinline__ Sa1Lds sa1_carve(float* sm, bool bwd) {
  Sa1Lds L;
  L.w1 = sm;
  L.b2 = L.w1 + 64 * 4;
  L.b3 = L.b2 + 64;
  L.red = L.b3 + 128;
  L.kw = reinterpret_cast<int*>(L.red + 48);
  L.w1a = reinterpret_cast<unsigned char*>(L.kw + 16);
  L.w2h = L.w1a + 64 * 32;
  L.w3h = L.w2h + 2 * 64 * SA_PH;
  L.w2t = L.w3h + (bwd ? 2 * 64 * SA_PH2 : 2 * 128 * SA_PH);
  L.pa = reinterpret_cast<unsigned*>(L.w2t + 2 * 64 * SA_PH);
  return L;
}

// channel held at position p = ks*16 + 8h + j of a k-permuted image: the row order of the 32x32 accumulator
__device__ __forceinline__ int sa_perm(int p) {
  const int ks = p >> 4, h = (p >> 3) & 1, j = p & 7;
  return 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h + (j & 3) + 8 * (j >> 2);
}
// Every scale of this file is a power of two 2^k carried as its integer k (wave-uniform: SALU arithmetic):
// sa_k(m) puts a maximum m into [2^13, 2^14); lo / hi clamp the biased exponent (tiny maxima keep a finite scale).
__device__ __forceinline__ int sa_k(float m, int lo = 110, int hi = 254) {
  int E = (int)((__float_as_uint(m) >> 23) & 0xffu);
  E = E < lo ? lo : (E > hi ? hi : E);
  return 140 - E;
}
__device__ __forceinline__ float sa_pow2(int k) {
  k = k < -126 ? -126 : (k > 127 ? 127 : k);
  return __uint_as_float((unsigned)(k + 127) << 23);
}
__device__ __forceinline__ void sa_put(unsigned char* img, int lo_off, int off, float v) {
  const _Float16 h = (_Float16)v;
  *reinterpret_cast<_Float16*>(img + off) = h;
  *reinterpret_cast<_Float16*>(img + lo_off + off) = (_Float16)(v - (float)h);
}
__device__ __forceinline__ float sa_relu(float x) { return __builtin_amdgcn_fmed3f(x, 0.f, __builtin_huge_valf()); }

template <int SA_T, bool BWD>
__device__ __forceinline__ void sa1_stage(const geoa3_sa1_weights& w, const Sa1Lds& L) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float m1 = 0.f, m2 = 0.f, m3 = 0.f;
  for (int e = tid; e < 64 * 3; e += SA_T) m1 = fmaxf(m1, __builtin_fabsf(w.w1[e]));
  for (int e = tid; e < 64; e += SA_T) m1 = fmaxf(m1, __builtin_fabsf(w.b1[e]));
  for (int e = tid; e < 64 * 64; e += SA_T) m2 = fmaxf(m2, __builtin_fabsf(w.w2[e]));
  for (int e = tid; e < 128 * 64; e += SA_T) m3 = fmaxf(m3, __builtin_fabsf(w.w3[e]));
  m1 = wave_max(m1);
  m2 = wave_max(m2);
  m3 = wave_max(m3);
  if (lane == 0) {
    L.red[wave] = m1;
    L.red[16 + wave] = m2;
    L.red[32 + wave] = m3;
  }
  __syncthreads();
  float t1 = 0.f, t2 = 0.f, t3 = 0.f;
  for (int i = 0; i < SA_T / 64; ++i) {
    t1 = fmaxf(t1, L.red[i]);
    t2 = fmaxf(t2, L.red[16 + i]);
    t3 = fmaxf(t3, L.red[32 + i]);
  }
  const int k1 = sa_k(t1), k2 = sa_k(t2), k3 = sa_k(t3);
  const float s1 = sa_pow2(k1), s2 = sa_pow2(k2), s3 = sa_pow2(k3);
  if (tid == 0) {
    L.kw[0] = k1;
    L.kw[1] = k2;
    L.kw[2] = k3;
  }
  // Row (and, for the backward, column) l1 norms of the weights: |W x|_inf <= |W|_l1rows |x|_inf bounds every hidden
  // activation from the points' largest coordinate, and every back-propagated gradient from the pooled gradient's largest
  // entry -- RIGOROUSLY, so the power-of-two scales of the operand splits come from scalar integer arithmetic per centroid
  // instead of a measured maximum per block (16 v_max3 + a 12-step wave reduction per split, a tenth of the kernels'
  // vector instructions).  An l1 bound over 64 terms is loose by ~2^3 per layer; fp16's exponent range takes 2^10 of slack
  // before a lo piece turns subnormal.  kw[3..7] = E(.) with x < 2^E(x): W1 rows, b1, W2 rows, b2, then W3 columns /
  // W2 columns (backward).
  {
    float r1 = 0.f, rb1 = 0.f, r2 = 0.f, rb2 = 0.f, c3 = 0.f, c2 = 0.f;
    if (tid < 64) {
      r1 = __builtin_fabsf(w.w1[3 * tid]) + __builtin_fabsf(w.w1[3 * tid + 1]) + __builtin_fabsf(w.w1[3 * tid + 2]);
      rb1 = __builtin_fabsf(w.b1[tid]);
      rb2 = __builtin_fabsf(w.b2[tid]);
      for (int kk = 0; kk < 64; ++kk) r2 += __builtin_fabsf(w.w2[tid * 64 + kk]);
      if (BWD) {
        for (int ch = 0; ch < 128; ++ch) c3 += __builtin_fabsf(w.w3[ch * 64 + tid]);
        for (int o = 0; o < 64; ++o) c2 += __builtin_fabsf(w.w2[o * 64 + tid]);
      }
    }
    if (tid < 64) {     // (the first wavefront holds all 64 rows)
      r1 = wave_max(r1);
      rb1 = wave_max(rb1);
      r2 = wave_max(r2);
      rb2 = wave_max(rb2);
      c3 = wave_max(c3);
      c2 = wave_max(c2);
      auto E = [](float x) { int e = (int)((__float_as_uint(x) >> 23) & 0xffu) - 126; return e < -100 ? -100 : e; };
      if (tid == 0) {
        L.kw[3] = E(r1);
        L.kw[4] = E(rb1);
        L.kw[5] = E(r2);
        L.kw[6] = E(rb2);
        L.kw[7] = E(c3);
        L.kw[8] = E(c2);
      }
    }
  }
  for (int e = tid; e < 64; e += SA_T) {
    // layer 1 as one k-step of the matrix core (K = 16): against the operand (ph.xyz, one | pl.xyz, one) of the lanes of k
    // half 0 and (ph.xyz, 0 | pl.xyz, 0) of k half 1 -- wh.ph + bh.one + wh.pl + bl.one + wl.ph + wl.pl, fp32 accumulation
    _Float16 hh[4], ll[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float v = (d < 3 ? w.w1[3 * e + d] : w.b1[e]) * s1;
      hh[d] = (_Float16)v;
      ll[d] = (_Float16)(v - (float)hh[d]);
    }
    const half8 r0 = {hh[0], hh[1], hh[2], hh[3], hh[0], hh[1], hh[2], ll[3]};
    const half8 r1 = {ll[0], ll[1], ll[2], (_Float16)0.f, ll[0], ll[1], ll[2], (_Float16)0.f};
    *reinterpret_cast<half8*>(L.w1a + e * 32) = r0;
    *reinterpret_cast<half8*>(L.w1a + e * 32 + 16) = r1;
    L.b2[e] = w.b2[e];
    if (BWD) {
      L.w1[3 * e + 0] = w.w1[3 * e + 0];
      L.w1[3 * e + 1] = w.w1[3 * e + 1];
      L.w1[3 * e + 2] = w.w1[3 * e + 2];
    }
  }
  if (!BWD)
    for (int e = tid; e < 128; e += SA_T) L.b3[e] = w.b3[e];
  for (int e = tid; e < 64 * 64; e += SA_T) {   // W2 [out][position]: A operand of layer 2; the positions follow the
    const int o = e >> 6, p = e & 63;           // row order of the layer-1 accumulators, which ARE its B operand
    sa_put(L.w2h, 64 * SA_PH, o * SA_PH + p * 2, w.w2[o * 64 + sa_perm(p)] * s2);
  }
  if (!BWD) {
    for (int e = tid; e < 128 * 64; e += SA_T) {   // W3 [out][position]: B operand of the transposed layer 3
      const int o = e >> 6, p = e & 63;
      sa_put(L.w3h, 128 * SA_PH, o * SA_PH + p * 2, w.w3[o * 64 + sa_perm(p)] * s3);
    }
  } else {
    for (int e = tid; e < 128 * 64; e += SA_T) {   // W3^T [i][ch]: A operand of d h2 = W3^T dz3
      const int ch = e >> 6, i = e & 63;
      sa_put(L.w3h, 64 * SA_PH2, i * SA_PH2 + ch * 2, w.w3[e] * s3);
    }
    for (int e = tid; e < 64 * 64; e += SA_T) {    // W2^T [i][position]: A operand of d h1 = W2^T dz2
      const int i = e >> 6, p = e & 63;
      sa_put(L.w2t, 64 * SA_PH, i * SA_PH + p * 2, w.w2[sa_perm(p) * 64 + i] * s2);
    }
  }
  __syncthreads();
}

// (hi, lo) fp16 images of x0 * s and x1 * s, packed: v_pk_mul_f32 + v_cvt_pk_f16_f32 for the pair of hi pieces, one
// v_fma_mixlo / mixhi_f16 per lo piece (the residual x * s - hi as ONE exact fma, converted and packed by the same
// instruction): two instructions per element, all of them visible to the compiler's scheduler and hazard recogniser.
// The file is compiled with -fno-slp-vectorize (geoa3_amd/build.py): the SLP vectoriser turns the pair of residuals into
// v_cvt_f32_f16 x 2 + v_pk_fma_f32 + v_cvt_pk_f16_f32 (three per element).
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float float2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sa_split2(float x0, float x1, float s, unsigned& hi, unsigned& lo) {
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
// two accumulator tiles (rows 32 t + mfma_row(r, lane)) -> the operands of the four k-steps that consume them: registers
// 8 s .. 8 s + 7 of tile t are k-step 2 t + s in the accumulator's row order (the weight images follow it: sa_perm)
__device__ __forceinline__ void sa_split_tiles(const f32x16 (&v)[2], float s, half8 (&oh)[4], half8 (&ol)[4]) {
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    uint4v H, Lw;
#pragma unroll
    for (int j2 = 0; j2 < 4; ++j2) {
      unsigned a, b;
      sa_split2(v[ks >> 1][8 * (ks & 1) + 2 * j2], v[ks >> 1][8 * (ks & 1) + 2 * j2 + 1], s, a, b);
      H[j2] = a;
      Lw[j2] = b;
    }
    oh[ks] = __builtin_bit_cast(half8, H);
    ol[ks] = __builtin_bit_cast(half8, Lw);
  }
}

// instruction-stream shaping: a matrix instruction holds the SIMD's issue port for 8 of its 32 cycles; five plain vector
// instructions fit into the rest (MI355X_MICROARCH.md, cycle constants).  SA_MIX(n): n x {1 MFMA, 5 VALU} in this order.
#define SA_SB() __builtin_amdgcn_sched_barrier(0)
#define SA_MIX(n, v)                                         \
  _Pragma("unroll") for (int sg_ = 0; sg_ < (n); ++sg_) {     \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);       \
    __builtin_amdgcn_sched_group_barrier(0x002, (v), 0);     \
  }

#define SA_ZERO16 f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}

// the B operand of layer 1 for one column block: p * 2^kp split, `one` = 2^kp in the lanes of k half 0 (bias), 0 in k half 1
__device__ __forceinline__ half8 sa1_bop(float px, float py, float pz, float sp, _Float16 one) {
  const float x = px * sp, y = py * sp, z = pz * sp;
  const _Float16 hx = (_Float16)x, hy = (_Float16)y, hz = (_Float16)z;
  const half8 b = {hx, hy, hz, one, (_Float16)(x - (float)hx), (_Float16)(y - (float)hy), (_Float16)(z - (float)hz), one};
  return b;
}
// layer 1 of one column block on the matrix core: z[t][r] = 2^(k1 + kp) (w1 p + b1), channel 32 t + mfma_row(r, lane),
// sample = lane & 31 of the block
__device__ __forceinline__ void sa1_layer1(const Sa1Lds& L, half8 bop, int lane, f32x16 (&z)[2]) {
  const unsigned char* ar = L.w1a + (lane & 31) * 32 + (lane >> 5) * 16;
  const half8 a0 = *reinterpret_cast<const half8*>(ar), a1 = *reinterpret_cast<const half8*>(ar + 32 * 32);
  z[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0, bop, SA_ZERO16, 0, 0, 0);
  z[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, bop, SA_ZERO16, 0, 0, 0);
}
// scales from bounds (sa1_stage): with |p| 2^kp < 2^14,  h1 < 2^e1, e1 = max(EW1 + 14 - kp, EB1) + 1, and the layer-1
// accumulators (2^(k1 + kp) h1) times 2^kx stay below 2^14 for kx = 14 - e1 - k1 - kp; h2 < 2^e2, e2 = max(EW2 + e1, EB2) + 1
struct Sa1Scales {
  int kx, ka, ky;   // operand scale of h1; scale of the layer-2 accumulators (k1 + kp + kx + k2); operand scale of h2
};
__device__ __forceinline__ Sa1Scales sa1_scales(int k1, int k2, int kp, int EW1, int EB1, int EW2, int EB2) {
  const int a = EW1 + 14 - kp, e1 = (a > EB1 ? a : EB1) + 1;
  Sa1Scales S;
  S.kx = 14 - e1 - k1 - kp;
  S.ka = k1 + kp + S.kx + k2;
  const int b = EW2 + e1, e2 = (b > EB2 ? b : EB2) + 1;
  S.ky = 14 - e2 - S.ka;
  return S;
}
__device__ __forceinline__ void sa_relu(f32x16 (&z)[2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) z[t][r] = sa_relu(z[t][r]);
}
// relu in place, k of the block's maximum (measured: the pooled gradient's scale still is)
__device__ __forceinline__ int sa_relu_k(f32x16 (&z)[2]) {
  float m = 0.f;
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      z[t][r] = sa_relu(z[t][r]);
      m = fmaxf(m, z[t][r]);
    }
  return sa_k(wave_max(m));
}
// the bias of layer 2 in the accumulators' layout, times the accumulators' scale (exact: a power of two): once per centroid,
// the C operand of the first matrix instruction of BOTH column blocks' chains (the blocks share their scales)
__device__ __forceinline__ void sa1_bias2(const Sa1Lds& L, float bscale, int lane, f32x16 (&bias)[2]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 b = *reinterpret_cast<const float4*>(L.b2 + 32 * t + 8 * g + 4 * (lane >> 5));
      bias[t][4 * g + 0] = b.x * bscale;
      bias[t][4 * g + 1] = b.y * bscale;
      bias[t][4 * g + 2] = b.z * bscale;
      bias[t][4 * g + 3] = b.w * bscale;
    }
}
// layer 2 of one column block: acc[t][r] = 2^kacc (W2 h1 + b2), rows 32 t + mfma_row(r, lane); the bias enters as the
// accumulators' initial value, so the relu gate of the backward is the accumulator's sign
__device__ __forceinline__ void sa1_layer2(const Sa1Lds& L, const half8 (&xh)[4], const half8 (&xl)[4], const f32x16 (&bias)[2],
                                           int lane, f32x16 (&acc)[2]) {
  const unsigned char* wr = L.w2h + (lane & 31) * SA_PH + (lane >> 5) * 16;
  half8 fr[2][4];
  auto fetch = [&](int ks, half8 (&f)[4]) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f[2 * t] = *reinterpret_cast<const half8*>(wr + t * 32 * SA_PH + ks * 32);
      f[2 * t + 1] = *reinterpret_cast<const half8*>(wr + t * 32 * SA_PH + ks * 32 + 64 * SA_PH);
    }
  };
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) {
    fetch(ks, fr[ks & 1]);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const half8 wh = fr[ks & 1][2 * t], wl = fr[ks & 1][2 * t + 1];
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[ks], ks == 0 ? bias[t] : acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[ks], acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[ks], acc[t], 0, 0, 0);
    }
  }
}

// a centroid's gather for the lane (n, h) = (lane & 31, lane >> 5): the points of samples n and 32 + n relative to the
// centroid (both halves of the wave hold both column blocks' operands), requested one centroid ahead
struct Sa1Pts {
  int ia, ib;
  float ax, ay, az, bx, by, bz;
};
__device__ __forceinline__ void sa1_load(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                         const int32_t* __restrict__ idx, int c, int M, int N, int lane, Sa1Pts& P) {
  const int b = c / M;   // (wave-uniform: scalar arithmetic)
  P.ia = idx[(size_t)c * SA_S + (lane & 31)];
  P.ib = idx[(size_t)c * SA_S + 32 + (lane & 31)];
  const float* qa = xyz + ((size_t)b * N + P.ia) * 3;
  const float* qb = xyz + ((size_t)b * N + P.ib) * 3;
  const float* ctr = new_xyz + (size_t)c * 3;
  const float cx = ctr[0], cy = ctr[1], cz = ctr[2];
  P.ax = qa[0] - cx;
  P.ay = qa[1] - cy;
  P.az = qa[2] - cz;
  P.bx = qb[0] - cx;
  P.by = qb[1] - cy;
  P.bz = qb[2] - cz;
}
__device__ __forceinline__ int sa1_kp(const Sa1Pts& P) {   // 2^kp |p| < 2^14 for both blocks; 2^kp itself an fp16 normal
  const float m = fmaxf(fmaxf(fmaxf(__builtin_fabsf(P.ax), __builtin_fabsf(P.ay)), fmaxf(__builtin_fabsf(P.az), __builtin_fabsf(P.bx))),
                        fmaxf(__builtin_fabsf(P.by), __builtin_fabsf(P.bz)));
  return sa_k(wave_max(m), 126, 154);
}

template <int SA_T>
__global__ __launch_bounds__(SA_T) void sa1_fwd_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                      const int32_t* __restrict__ idx, geoa3_sa1_weights w, int B, int N,
                                                      int M, float* __restrict__ out, uint8_t* __restrict__ arg, int m0, int Mc) {
  extern __shared__ __attribute__((aligned(16))) float sa_sm[];
  const Sa1Lds L = sa1_carve(sa_sm, false);
  sa1_stage<SA_T, false>(w, L);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5, l31 = lane & 31;
  const int k1 = __builtin_amdgcn_readfirstlane(L.kw[0]), k2 = __builtin_amdgcn_readfirstlane(L.kw[1]),
            k3 = __builtin_amdgcn_readfirstlane(L.kw[2]);
  const int EW1 = __builtin_amdgcn_readfirstlane(L.kw[3]), EB1 = __builtin_amdgcn_readfirstlane(L.kw[4]),
            EW2 = __builtin_amdgcn_readfirstlane(L.kw[5]), EB2 = __builtin_amdgcn_readfirstlane(L.kw[6]);
  // centroids m0 .. m0 + Mc - 1 of every cloud: the walk runs over j in [0, B * Mc), centroid c = (j / Mc) * M + m0 + j % Mc
  const int total = B * Mc, stride = (int)gridDim.x * (SA_T / 64);   // B * M < 2^31 / 128 (checked by the launcher)
  auto cof = [&](int j) { return (j / Mc) * M + m0 + j % Mc; };       // (wave-uniform: scalar arithmetic)
  int jc = (int)blockIdx.x * (SA_T / 64) + wave;
  Sa1Pts P;
  if (jc < total) sa1_load(xyz, new_xyz, idx, cof(jc), M, N, lane, P);
  for (; jc < total; jc += stride) {
    asm volatile("" ::: "memory");   // the weights stay in LDS: no hoisting of their loads out of the centroid loop
    const int c = cof(jc);
    const Sa1Pts Q = P;
    if (jc + stride < total) sa1_load(xyz, new_xyz, idx, cof(jc + stride), M, N, lane, P);
    const int kp = sa1_kp(Q);
    const float sp = sa_pow2(kp);
    const _Float16 one = h ? (_Float16)0.f : (_Float16)sp;
    const Sa1Scales SC = sa1_scales(k1, k2, kp, EW1, EB1, EW2, EB2);   // the same for both column blocks
    f32x16 bias2[2];
    sa1_bias2(L, sa_pow2(SC.ka), lane, bias2);
    float bv[4];   // best (value, sample) of channel 32 t3 + l31 over this lane's samples
    int bs[4];
    // The two 32-sample column blocks A and B go through the layers SKEWED by one stage, so that every matrix phase of one
    // block has the other block's vector phase (relu, maximum, operand split, arg-max) beside it in ONE wave's instruction
    // stream -- an in-order wave cannot fill its own dependent MFMA chain's gaps otherwise, and two waves per SIMD running
    // the same program overlapped only 17 % of the matrix pipe's busy cycles (SQ_VALU_MFMA_COEXEC_CYCLES):
    //   L1(A) L1(B) | split(A) | L2(A) + split(B) | L2(B) + split2(A) | L3(A) + split2(B) + argmax(A) | L3(B) + argmax | tail
    half8 xhA[4], xlA[4], xhB[4], xlB[4];
    f32x16 zA[2], zB[2];
    sa1_layer1(L, sa1_bop(Q.ax, Q.ay, Q.az, sp, one), lane, zA);
    sa1_layer1(L, sa1_bop(Q.bx, Q.by, Q.bz, sp, one), lane, zB);
    SA_SB();
    sa_relu(zA);
    sa_split_tiles(zA, sa_pow2(SC.kx), xhA, xlA);
    SA_SB();
    sa1_layer2(L, xhA, xlA, bias2, lane, zA);                   // (zA: layer-2 accumulators of A from here)
    sa_relu(zB);
    sa_split_tiles(zB, sa_pow2(SC.kx), xhB, xlB);
    SA_MIX(24, 4)
    SA_SB();
    sa1_layer2(L, xhB, xlB, bias2, lane, zB);
    sa_relu(zA);
    sa_split_tiles(zA, sa_pow2(SC.ky), xhA, xlA);              // the A fragments of the transposed layer 3
    SA_MIX(24, 5)
    SA_SB();
    auto tile = [&](const half8 (&xh)[4], const half8 (&xl)[4], int t3, f32x16& a3) {
      const unsigned char* wr = L.w3h + (t3 * 32 + l31) * SA_PH + h * 16;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const half8 bh = *reinterpret_cast<const half8*>(wr + ks * 32);
        const half8 bl = *reinterpret_cast<const half8*>(wr + ks * 32 + 128 * SA_PH);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[ks], bh, ks == 0 ? SA_ZERO16 : a3, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[ks], bl, a3, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[ks], bh, a3, 0, 0, 0);
      }
    };
    // a3[r]: channel t3*32 + l31, sample cb*32 + mfma_row(r, lane); ascending sample order: the first maximal sample wins,
    // as F.max_pool2d.  The blocks carry their own power-of-two scales: compared unscaled (exact).
    auto argmax = [&](const f32x16& a3, float un3, int cb, int t3) {
      float v = a3[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) v = fmaxf(v, a3[r]);
      int smp = 3 + 8 * 3;                                     // mfma_row(r, lane) - 4 h
#pragma unroll
      for (int r = 14; r >= 0; --r) smp = a3[r] == v ? (r & 3) + 8 * (r >> 2) : smp;
      v *= un3;
      if (cb == 0 || v > bv[t3]) {
        bv[t3] = v;
        bs[t3] = cb * 32 + 4 * h + smp;
      }
    };
    f32x16 t0, t1;
    {
      const float un3 = sa_pow2(-(SC.ka + SC.ky + k3));
      tile(xhA, xlA, 0, t0);
      sa_relu(zB);
      sa_split_tiles(zB, sa_pow2(SC.ky), xhB, xlB);
      tile(xhA, xlA, 1, t1);
      argmax(t0, un3, 0, 0);
      tile(xhA, xlA, 2, t0);
      argmax(t1, un3, 0, 1);
      tile(xhA, xlA, 3, t1);
      argmax(t0, un3, 0, 2);
      SA_MIX(48, 5)
      SA_SB();
      const float un3b = un3;
      tile(xhB, xlB, 0, t0);
      argmax(t1, un3, 0, 3);
      tile(xhB, xlB, 1, t1);
      argmax(t0, un3b, 1, 0);
      tile(xhB, xlB, 2, t0);
      argmax(t1, un3b, 1, 1);
      tile(xhB, xlB, 3, t1);
      argmax(t0, un3b, 1, 2);
      SA_MIX(48, 4)
      SA_SB();
      argmax(t1, un3b, 1, 3);
    }
#pragma unroll
    for (int t3 = 0; t3 < 4; ++t3) {
      float v = bv[t3];
      int smp = bs[t3];
      const float ov = __shfl_xor(v, 32, 64);
      const int os = __shfl_xor(smp, 32, 64);
      const bool take = ov > v || (ov == v && os < smp);
      v = take ? ov : v;
      smp = take ? os : smp;
      if (lane < 32) {   // 128 contiguous bytes of the centroid's row of out_t [B,M,128]
        const int ch = t3 * 32 + lane;
        out[(size_t)c * 128 + ch] = fmaxf(v + L.b3[ch], 0.f);
        arg[(size_t)c * 128 + ch] = (uint8_t)smp;
      }
    }
  }
}

template <int SA_T>
__global__ __launch_bounds__(SA_T) __attribute__((amdgpu_waves_per_eu(SA_WB / 4, SA_WB / 4))) void sa1_bwd_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                      const int32_t* __restrict__ idx, geoa3_sa1_weights w, int B, int N,
                                                      int M, const float* __restrict__ out,
                                                      const uint8_t* __restrict__ arg, const float* __restrict__ g,
                                                      float* __restrict__ dxyz, float* __restrict__ dnew,
                                                      float* __restrict__ dpbuf) {
  extern __shared__ __attribute__((aligned(16))) float sa_sm[];
  const Sa1Lds L = sa1_carve(sa_sm, true);
  sa1_stage<SA_T, true>(w, L);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), h = lane >> 5, l31 = lane & 31;
  const int k1 = __builtin_amdgcn_readfirstlane(L.kw[0]), k2 = __builtin_amdgcn_readfirstlane(L.kw[1]),
            k3 = __builtin_amdgcn_readfirstlane(L.kw[2]);
  const int EW1 = __builtin_amdgcn_readfirstlane(L.kw[3]), EB1 = __builtin_amdgcn_readfirstlane(L.kw[4]),
            EW2 = __builtin_amdgcn_readfirstlane(L.kw[5]), EB2 = __builtin_amdgcn_readfirstlane(L.kw[6]),
            EW3T = __builtin_amdgcn_readfirstlane(L.kw[7]);
  // d h2 = W3^T dz3 with |dz3| 2^kg < 2^14: the accumulators (2^(k3 + kg) d h2) stay below 2^(k3 + EW3T + 14), so the
  // operand scale of the next product is the same for every centroid
  const int kd = -k3 - EW3T;
  // one-hot records of the wave's centroid: channel PAIRS (2 q, 2 q + 1) as packed fp16 images of the scaled pooled gradient
  // (hi pieces | lo pieces), and per channel the arg-max sample as a one-hot bit over samples 0-31 | 32-63
  unsigned* s_hp = L.pa + wave * 384;   // [64]
  unsigned* s_lp = s_hp + 64;           // [64]
  unsigned* s_m = s_lp + 64;            // [2 blocks][128]
  const int total = B * M, stride = (int)gridDim.x * (SA_T / 64);   // B * M < 2^31 / 128 (checked by the launcher)
  int c = (int)blockIdx.x * (SA_T / 64) + wave;
  Sa1Pts P;
  if (c < total) sa1_load(xyz, new_xyz, idx, c, M, N, lane, P);
  for (; c < total; c += stride) {
    asm volatile("" ::: "memory");   // the weights stay in LDS: no hoisting of their loads out of the centroid loop
    const Sa1Pts Q = P;
    int kg;
    {
      // every channel's scaled gradient (through the output relu) is split ONCE by the lane that loads it; the one-hot
      // operand construction below is then a bit-field extract per channel (0 / -1: is this lane's sample the channel's
      // arg-max?), one byte permute per pair to join two of them, and an AND per packed pair -- no compare through an SGPR
      // (two wait states between a v_cmp and the v_cndmask that reads it on this chip)
      const float2 ov = *reinterpret_cast<const float2*>(out + (size_t)c * 128 + 2 * lane);
      const float2 gv = *reinterpret_cast<const float2*>(g + (size_t)c * 128 + 2 * lane);
      const unsigned av = *reinterpret_cast<const unsigned short*>(arg + (size_t)c * 128 + 2 * lane);
      const float g0 = ov.x > 0.f ? gv.x : 0.f, g1 = ov.y > 0.f ? gv.y : 0.f;
      kg = sa_k(wave_max(fmaxf(__builtin_fabsf(g0), __builtin_fabsf(g1))));
      const float sg = sa_pow2(kg);
      const float z0 = g0 * sg, z1 = g1 * sg;
      const _Float16 h0 = (_Float16)z0, h1 = (_Float16)z1;
      const _Float16 l0 = (_Float16)(z0 - (float)h0), l1 = (_Float16)(z1 - (float)h1);
      s_hp[lane] = (unsigned)__builtin_bit_cast(unsigned short, h0) | (unsigned)__builtin_bit_cast(unsigned short, h1) << 16;
      s_lp[lane] = (unsigned)__builtin_bit_cast(unsigned short, l0) | (unsigned)__builtin_bit_cast(unsigned short, l1) << 16;
      const unsigned a0 = av & 0x3fu, a1 = (av >> 8) & 0x3fu;
      const unsigned b0 = 1u << (a0 & 31u), b1 = 1u << (a1 & 31u);
      *reinterpret_cast<uint2*>(s_m + 2 * lane) = make_uint2(a0 < 32u ? b0 : 0u, a1 < 32u ? b1 : 0u);
      *reinterpret_cast<uint2*>(s_m + 128 + 2 * lane) = make_uint2(a0 < 32u ? 0u : b0, a1 < 32u ? 0u : b1);
    }
    if (c + stride < total) sa1_load(xyz, new_xyz, idx, c + stride, M, N, lane, P);
    const int kp = sa1_kp(Q);
    const float sp = sa_pow2(kp);
    const _Float16 one = h ? (_Float16)0.f : (_Float16)sp;
    const Sa1Scales SC = sa1_scales(k1, k2, kp, EW1, EB1, EW2, EB2);
    f32x16 bias2[2];
    sa1_bias2(L, sa_pow2(SC.ka), lane, bias2);
    float dpx = 0.f, dpy = 0.f, dpz = 0.f;   // of sample `lane`
    // The two column blocks (samples 0-31, 32-63) go through the forward recomputation and the three backward products ONE
    // AFTER THE OTHER, each with its own power-of-two scales: half the live accumulators, three waves per SIMD.
#ifdef GEOA3_SA1_UNROLL_CB
#pragma unroll
#else
#pragma unroll 1
#endif
    for (int cb = 0; cb < 2; ++cb) {
#ifndef GEOA3_SA1_UNROLL_CB
      asm volatile("" ::: "memory");   // (the weight fragments are re-read per block: no hoisting out of this loop)
#endif
      // d h2 [64 x 32 samples] = W3^T dz3, dz3 one-hot per channel (only the arg-max sample carries gradient): element j
      // of lane (sample, k half) for k-step ks is channel ch = 16 ks + 8 h + j, non-zero only in the lane of the channel's
      // arg-max sample; the A operand is the W3^T image.  FIRST (it needs nothing of the forward): the recomputed layers'
      // registers are not live beside the k-steps' operands, which are read from LDS one k-step ahead.
      f32x16 d2[2];
      {
        const unsigned char* wr = L.w3h + l31 * SA_PH2 + h * 16;
        const unsigned* mrow = s_m + cb * 128 + 8 * h;
        uint4v rec[2][4];   // (hi pairs, lo pairs, 2 x 4 masks) of a k-step
        half8 fr[2][4];
        auto fetch = [&](int ks, uint4v (&r)[4], half8 (&f)[4]) {
          r[0] = *reinterpret_cast<const uint4v*>(s_hp + 8 * ks + 4 * h);
          r[1] = *reinterpret_cast<const uint4v*>(s_lp + 8 * ks + 4 * h);
          r[2] = *reinterpret_cast<const uint4v*>(mrow + 16 * ks);
          r[3] = *reinterpret_cast<const uint4v*>(mrow + 16 * ks + 4);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            f[2 * t] = *reinterpret_cast<const half8*>(wr + t * 32 * SA_PH2 + ks * 32);
            f[2 * t + 1] = *reinterpret_cast<const half8*>(wr + t * 32 * SA_PH2 + ks * 32 + 64 * SA_PH2);
          }
        };
        fetch(0, rec[0], fr[0]);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          if (ks + 1 < 8) fetch(ks + 1, rec[(ks + 1) & 1], fr[(ks + 1) & 1]);
          uint4v vh0, vl0;
#pragma unroll
          for (int j2 = 0; j2 < 4; ++j2) {
            const unsigned ma = (unsigned)__builtin_amdgcn_sbfe((int)rec[ks & 1][2 + (j2 >> 1)][2 * (j2 & 1)], (unsigned)l31, 1u);
            const unsigned mb = (unsigned)__builtin_amdgcn_sbfe((int)rec[ks & 1][2 + (j2 >> 1)][2 * (j2 & 1) + 1], (unsigned)l31, 1u);
            const unsigned mp = __builtin_amdgcn_perm(mb, ma, 0x07060100u);   // ma's low half | mb's high half
            vh0[j2] = rec[ks & 1][0][j2] & mp;
            vl0[j2] = rec[ks & 1][1][j2] & mp;
          }
          const half8 bh = __builtin_bit_cast(half8, vh0), bl = __builtin_bit_cast(half8, vl0);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const half8 wh = fr[ks & 1][2 * t], wl = fr[ks & 1][2 * t + 1];
            d2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, ks == 0 ? SA_ZERO16 : d2[t], 0, 0, 0);
            d2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, d2[t], 0, 0, 0);
            d2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, d2[t], 0, 0, 0);
          }
        }
      }
      // the forward of the block: layers 1 and 2; the sign of the layer-2 accumulators (bias included) is the relu gate
      const half8 bop = sa1_bop(cb ? Q.bx : Q.ax, cb ? Q.by : Q.ay, cb ? Q.bz : Q.az, sp, one);
      half8 bh[4], bl[4];
      {
        f32x16 a2[2];
        {
          f32x16 z[2];
          sa1_layer1(L, bop, lane, z);
          half8 xh[4], xl[4];
          sa_relu(z);
          sa_split_tiles(z, sa_pow2(SC.kx), xh, xl);
          sa1_layer2(L, xh, xl, bias2, lane, a2);
        }
        // through relu 2; the gated accumulator registers are the B operands of d h1 = W2^T dz2 (the W2^T image follows
        // their row order)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) d2[t][r] = a2[t][r] > 0.f ? d2[t][r] : 0.f;
        sa_split_tiles(d2, sa_pow2(kd), bh, bl);
      }
      f32x16 d1[2];
      {
        const unsigned char* wr = L.w2t + l31 * SA_PH + h * 16;
        half8 fr[2][4];
        auto fetch = [&](int ks, half8 (&f)[4]) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            f[2 * t] = *reinterpret_cast<const half8*>(wr + t * 32 * SA_PH + ks * 32);
            f[2 * t + 1] = *reinterpret_cast<const half8*>(wr + t * 32 * SA_PH + ks * 32 + 64 * SA_PH);
          }
        };
        fetch(0, fr[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          if (ks + 1 < 4) fetch(ks + 1, fr[(ks + 1) & 1]);
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const half8 wh = fr[ks & 1][2 * t], wl = fr[ks & 1][2 * t + 1];
            d1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh[ks], ks == 0 ? SA_ZERO16 : d1[t], 0, 0, 0);
            d1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl[ks], d1[t], 0, 0, 0);
            d1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh[ks], d1[t], 0, 0, 0);
          }
        }
      }
      // through relu 1 (layer 1 once more on the matrix core: its sign is the gate, in the layout of d1) and the K = 3 layer
      {
        f32x16 z[2];
        sa1_layer1(L, bop, lane, z);
        float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            // rows 32 t + 8 g4 + 4 h + (0..3): twelve consecutive floats of the [64][3] image, three 16-byte reads
            const float4* wq = reinterpret_cast<const float4*>(L.w1 + 3 * (32 * t + 8 * g4 + 4 * h));
            const float4 q0 = wq[0], q1 = wq[1], q2 = wq[2];
            const float wr3[12] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w, q2.x, q2.y, q2.z, q2.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float v = z[t][4 * g4 + i] > 0.f ? d1[t][4 * g4 + i] : 0.f;
              sx = __builtin_fmaf(wr3[3 * i], v, sx);
              sy = __builtin_fmaf(wr3[3 * i + 1], v, sy);
              sz = __builtin_fmaf(wr3[3 * i + 2], v, sz);
            }
          }
        const float f1 = sa_pow2(-(k2 + kd + k3 + kg));
        sx = (sx + __shfl_xor(sx, 32, 64)) * f1;
        sy = (sy + __shfl_xor(sy, 32, 64)) * f1;
        sz = (sz + __shfl_xor(sz, 32, 64)) * f1;
        if (h == cb) {
          dpx = sx;
          dpy = sy;
          dpz = sz;
        }
      }
    }
    // scatter to the gathered points (entries repeating the row's first index -- the ball query's padding -- leave as
    // one add) and minus the sum to the centroid
    const int b = c / M;
    const int i = h ? Q.ib : Q.ia;
    const int i0 = __shfl(i, 0, 64);
    const bool dup = lane > 0 && i == i0;
    const float ex = wave_sum(dup ? dpx : 0.f), ey = wave_sum(dup ? dpy : 0.f), ez = wave_sum(dup ? dpz : 0.f);
    if (dpbuf) {   // deterministic mode: the (centroid, sample) contributions are summed by sa1_scatter_kernel
      float* dq = dpbuf + ((size_t)c * SA_S + lane) * 3;
      dq[0] = lane == 0 ? dpx + ex : (dup ? 0.f : dpx);
      dq[1] = lane == 0 ? dpy + ey : (dup ? 0.f : dpy);
      dq[2] = lane == 0 ? dpz + ez : (dup ? 0.f : dpz);
    } else {
      float* dq = dxyz + ((size_t)b * N + i) * 3;
      if (lane == 0) {
        atomicAdd(dq + 0, dpx + ex);
        atomicAdd(dq + 1, dpy + ey);
        atomicAdd(dq + 2, dpz + ez);
      } else if (!dup) {
        atomicAdd(dq + 0, dpx);
        atomicAdd(dq + 1, dpy);
        atomicAdd(dq + 2, dpz);
      }
    }
    const float tx = wave_sum(dpx), ty = wave_sum(dpy), tz = wave_sum(dpz);
    if (lane == 0) {
      dnew[(size_t)c * 3 + 0] = -tx;
      dnew[(size_t)c * 3 + 1] = -ty;
      dnew[(size_t)c * 3 + 2] = -tz;
    }
  }
}

// dxyz[b][n] = sum over the (centroid m, sample s) with idx[b][m][s] == n of dp[b][m][s]: the scatter-add of the grouping
// gradient (group_points_gpu.cu:60 behind pointnet2_utils.py:318), one workgroup per instance, as ORDER-FREE sums: every
// contribution is converted to 64-bit fixed point at the instance's own scale (2^40 / its largest |contribution|: a first
// pass over the instance's entries) and added by LDS integer atomics -- integer addition is associative, so the result
// does not depend on the order: deterministic and batch-independent without reverse lists.  Round 4 built per-owner
// reverse lists in LDS (histogram, scan, fill, per-owner sort) and summed in ascending (m, s): 104 us per launch on the
// ellipsoid clouds, 530 us on clouds with a dense cluster (one point referenced by hundreds of balls: a single thread
// heap-sorting its list).  Padded entries (repeats of a ball's first index) were merged into sample 0 by the producer and
// are skipped.  A NaN / infinite contribution makes the destination (the whole instance, if the scale itself is not
// finite) NaN: never a silent clamp.
constexpr int SCAT_BLOCK = 1024;
__global__ __launch_bounds__(SCAT_BLOCK) void sa1_scatter_kernel(const float* __restrict__ dp, const int32_t* __restrict__ idx,
                                                                float* __restrict__ dxyz, int N, int M) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long sc_acc[];   // [3][N]
  float* s_red = reinterpret_cast<float*>(sc_acc + 3 * (size_t)N);              // [16]
  unsigned* s_bad = reinterpret_cast<unsigned*>(s_red + 16);                    // [ceil(N / 32)]
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int32_t* I = idx + (size_t)b * M * SA_S;
  const float* D = dp + (size_t)b * M * SA_S * 3;
  const int nent = M * SA_S;
  for (int i = tid; i < 3 * N; i += SCAT_BLOCK) sc_acc[i] = 0ull;
  for (int i = tid; i < (N + 31) / 32; i += SCAT_BLOCK) s_bad[i] = 0u;
  float m = 0.f;
  for (int e = tid; e < 3 * nent; e += SCAT_BLOCK) {   // largest magnitude: as an unsigned maximum of the bit patterns, NaNs skipped
    const unsigned u = __float_as_uint(D[e]) & 0x7fffffffu;
    m = __uint_as_float(u > 0x7f800000u ? __float_as_uint(m) : (u > __float_as_uint(m) ? u : __float_as_uint(m)));
  }
  m = wave_max(m);
  if (lane == 0) s_red[wave] = m;
  __syncthreads();
  m = 0.f;
#pragma unroll
  for (int w = 0; w < SCAT_BLOCK / 64; ++w) m = fmaxf(m, s_red[w]);
  const bool finite = (__float_as_uint(m) & 0x7fffffffu) < 0x7f800000u;   // (an infinite contribution; NaNs do not reach m: fmaxf)
  int Ex = (int)((__float_as_uint(m) >> 23) & 0xffu) - 126;        // m < 2^Ex
  Ex = Ex < -80 ? -80 : (Ex > 80 ? 80 : Ex);
  const float to = __uint_as_float((unsigned)(40 - Ex + 127) << 23), from = __uint_as_float((unsigned)(Ex - 40 + 127) << 23);
  for (int e = tid; e < nent; e += SCAT_BLOCK) {
    const int q = I[e];
    if ((e & (SA_S - 1)) != 0 && q == I[e & ~(SA_S - 1)]) continue;              // padding: merged into sample 0
    const float x = D[3 * e], y = D[3 * e + 1], z = D[3 * e + 2];
    // NaN / infinity by their BIT patterns: this file is compiled with -fno-honor-nans, a floating-point test may be folded
    const unsigned bx = __float_as_uint(x) & 0x7fffffffu, by = __float_as_uint(y) & 0x7fffffffu, bz = __float_as_uint(z) & 0x7fffffffu;
    if (bx >= 0x7f800000u || by >= 0x7f800000u || bz >= 0x7f800000u) {
      atomicOr(&s_bad[q >> 5], 1u << (q & 31));
      continue;
    }
    atomicAdd(&sc_acc[q], (unsigned long long)__float2ll_rn(x * to));
    atomicAdd(&sc_acc[N + q], (unsigned long long)__float2ll_rn(y * to));
    atomicAdd(&sc_acc[2 * N + q], (unsigned long long)__float2ll_rn(z * to));
  }
  __syncthreads();
  float* G = dxyz + (size_t)b * N * 3;
  for (int i = tid; i < N; i += SCAT_BLOCK) {
    const bool bad = !finite || ((s_bad[i >> 5] >> (i & 31)) & 1u);
    // (the NaN as a BIT pattern through an integer store: a floating-point NaN constant is undefined under -fno-honor-nans)
    unsigned* Gu = reinterpret_cast<unsigned*>(G);
    Gu[3 * i + 0] = bad ? 0x7fc00000u : __float_as_uint(__ll2float_rn((long long)sc_acc[i]) * from);
    Gu[3 * i + 1] = bad ? 0x7fc00000u : __float_as_uint(__ll2float_rn((long long)sc_acc[N + i]) * from);
    Gu[3 * i + 2] = bad ? 0x7fc00000u : __float_as_uint(__ll2float_rn((long long)sc_acc[2 * N + i]) * from);
  }
}

int sa1_grid(int B, int M, int waves, int per_cu) {   // persistent workgroups
  const long groups = ((long)B * M + waves - 1) / waves;
  return (int)(groups < 256 * per_cu ? groups : 256 * per_cu);
}

}  // namespace

int launch_sa1_forward_range(const float* xyz, const float* new_xyz, const int32_t* idx, const geoa3_sa1_weights* w, int B, int N,
                             int M, int m0, int m1, float* out, uint8_t* arg, hipStream_t s) {
  if (!xyz || !new_xyz || !idx || !w || !out || !arg || B <= 0 || N <= 0 || M <= 0 || m0 < 0 || m1 <= m0 || m1 > M) return GEOA3_EINVAL;
  if ((long)B * M > (1L << 24)) return GEOA3_ENOSUPPORT;
  const size_t lds = (size_t)sa1_lds_bytes(SA_WF, false);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa1_fwd_kernel<SA_WF * 64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  geoa3_prof_begin(GEOA3_PROF_SA1_FWD, s);
  hipLaunchKernelGGL(sa1_fwd_kernel<SA_WF * 64>, dim3(sa1_grid(B, m1 - m0, SA_WF, 1)), dim3(SA_WF * 64), lds, s, xyz, new_xyz, idx, *w,
                     B, N, M, out, arg, m0, m1 - m0);
  geoa3_prof_end(GEOA3_PROF_SA1_FWD, s);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_pn2_sa1_forward(const float* xyz, const float* new_xyz, const int32_t* idx,
                                     const geoa3_sa1_weights* w, int B, int N, int M, float* out, uint8_t* arg,
                                     void* stream) {
  return launch_sa1_forward_range(xyz, new_xyz, idx, w, B, N, M, 0, M, out, arg, geoa3_stream(stream));
}

extern "C" int64_t geoa3_pn2_sa1_scratch_bytes(int B, int M) {
  if (B <= 0 || M <= 0) return -1;
  return (int64_t)B * M * SA_S * 3 * (int64_t)sizeof(float);
}

// (after_bwd: recorded behind sa1_bwd_kernel, in front of the scatter -- grad_new_xyz is complete there)
int launch_sa1_backward(const float* xyz, const float* new_xyz, const int32_t* idx, const geoa3_sa1_weights* w, int B, int N, int M,
                        const float* out, const uint8_t* arg, const float* grad_out, float* grad_xyz, float* grad_new_xyz,
                        float* scratch, hipEvent_t after_bwd, void* stream);

extern "C" int geoa3_pn2_sa1_backward(const float* xyz, const float* new_xyz, const int32_t* idx,
                                      const geoa3_sa1_weights* w, int B, int N, int M, const float* out,
                                      const uint8_t* arg, const float* grad_out, float* grad_xyz, float* grad_new_xyz,
                                      float* scratch, void* stream) {
  return launch_sa1_backward(xyz, new_xyz, idx, w, B, N, M, out, arg, grad_out, grad_xyz, grad_new_xyz, scratch, nullptr, stream);
}

int launch_sa1_backward(const float* xyz, const float* new_xyz, const int32_t* idx, const geoa3_sa1_weights* w, int B, int N, int M,
                        const float* out, const uint8_t* arg, const float* grad_out, float* grad_xyz, float* grad_new_xyz,
                        float* scratch, hipEvent_t after_bwd, void* stream) {
  if (!xyz || !new_xyz || !idx || !w || !out || !arg || !grad_out || !grad_xyz || !grad_new_xyz || B <= 0 || N <= 0 ||
      M <= 0)
    return GEOA3_EINVAL;
  if ((long)B * M > (1L << 24)) return GEOA3_ENOSUPPORT;
  hipStream_t s = geoa3_stream(stream);
  // deterministic scatter (scratch given and the owner's accumulators + at least one centroid's lists fit LDS)
  const size_t l2 = (size_t)3 * N * sizeof(unsigned long long) + 16 * sizeof(float) + ((size_t)(N + 31) / 32) * sizeof(unsigned);
  const bool det = scratch && l2 <= 160 * 1024 - 512;     // (N <= ~6800 points: the sums of one instance in LDS)
  if (!det && hipMemsetAsync(grad_xyz, 0, (size_t)B * N * 3 * sizeof(float), s) != hipSuccess) return GEOA3_ELAUNCH;
  const size_t lds = (size_t)sa1_lds_bytes(SA_WB, true);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa1_bwd_kernel<SA_WB * 64>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  geoa3_prof_begin(GEOA3_PROF_SA1_BWD, s);
  hipLaunchKernelGGL(sa1_bwd_kernel<SA_WB * 64>, dim3(sa1_grid(B, M, SA_WB, 1)), dim3(SA_WB * 64), lds, s, xyz, new_xyz, idx, *w, B, N, M, out, arg,
                     grad_out, grad_xyz, grad_new_xyz, det ? scratch : nullptr);
  geoa3_prof_end(GEOA3_PROF_SA1_BWD, s);
  if (after_bwd && hipEventRecord(after_bwd, s) != hipSuccess) return GEOA3_ELAUNCH;
  if (det) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa1_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)l2);
    hipLaunchKernelGGL(sa1_scatter_kernel, dim3(B), dim3(SCAT_BLOCK), l2, s, scratch, idx, grad_xyz, N, M);
  }
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
