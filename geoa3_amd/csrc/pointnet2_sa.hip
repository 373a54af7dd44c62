// PointNet++ SSG, first set-abstraction level, fused (gfx950): ball-grouped xyz -> shared MLP 3->64->64->128 (Conv2d 1x1 +
// eval-mode BatchNorm2d folded + ReLU) -> max over the 64 samples, forward and input-gradient.
// Reference: pointnet2_modules.py:29-74 (_PointnetSAModuleBase.forward), pointnet2_utils.py:296-333 (QueryAndGroup),
// PointNetPP_ssg.py:58-66 (npoint 512, radius 0.2, nsample 64, mlp [0(+3), 64, 64, 128]).
//
// The reference materialises [B,64,512,64] and [B,128,512,64] activations (2.1 + 2.1 + 4.2 GB at B = 250) and walks
// them once per layer and once more per layer in backward.  Here ONE WAVEFRONT owns one centroid: lane = sample.
//   layer 1 (K = 3) on the VALU, lane-local; layers 2 and 3 on the f16 matrix core with SPLIT fp32 operands (the
//   arithmetic of pointnet_wide_split.hip: v = hi + lo, a*w = a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, fp32 accumulation;
//   per-wave power-of-two activation scales, weights scaled and split into fp16 LDS images when they are staged):
//   * layer 2: a chunk of 16 layer-1 channels is one k-step; v_permlane32_swap of rows (j, 8 + j) yields the B operands
//     of both 32-sample column blocks (as in pointnet_conv_split.hip);
//   * layer 3 TRANSPOSED (activations as A: rows = samples; weights as B: columns = channels), so a lane ends with ONE
//     channel and 32 of its samples: the max over samples is lane-local + one exchange.  The layer-2 accumulator
//     registers 8s..8s+7 of a lane ARE its A fragment of a k-step -- channels in the accumulator's row order, which
//     the weight image in LDS follows (position ks*16 + 8h + j <-> channel 32(ks>>1) + 16(ks&1) + 4h + (j&3) + 8(j>>2));
//   no activation ever leaves the registers.
// Backward recomputes the two hidden layers (cheaper than reading 6 GB), routes the pooled gradient through a one-hot
// B operand against a W3^T image, chains the accumulator registers again as B operands against a W2^T image (same
// row-order trick), and finishes with the K = 3 contraction and the scatter to the points.
// The fp32-MFMA form of this file ran 1.67 ms forward / 2.61 ms backward at B = 250 (MFMA floors 1.4 / 1.9 ms).
#include "pointnet_kernels.h"
#include "profile.h"

namespace {

constexpr int SA_TF = 256;         // forward: 4 wavefronts per workgroup (6 waves / 3 per SIMD measured slower)
constexpr int SA_TB = 256;         // backward: 4 wavefronts (256 VGPRs: 2 waves/SIMD)
constexpr int SA_S = 64;           // samples per centroid
constexpr int SA_PH = 64 * 2 + 16;    // bytes per row of a 64-k fp16 image (conflict-free 16-byte reads)
constexpr int SA_PH2 = 128 * 2 + 16;  // bytes per row of the 128-k image (W3^T)

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

struct Sa1Lds {
  float* w1;            // [64][4]  (w0, w1, w2, shift)
  float* b2;            // [64]
  float* b3;            // [128]
  float* scal;          // [2] 1 / scale of the w2 and w3 images; [2..9] reduction scratch
  unsigned char* w2h;   // [64 out][64 k] hi, lo at + 64 * SA_PH       (A operand of layer 2)
  unsigned char* w3h;   // forward: [128 out][64 positions] hi, lo at + 128 * SA_PH   (B operand of layer 3)
                        // backward: W3^T [64 i][128 ch] hi, lo at + 64 * SA_PH2      (A operand of d h2)
  unsigned char* w2t;   // backward: W2^T [64 i][64 positions] hi, lo at + 64 * SA_PH (A operand of d h1)
  float* scratch;       // [waves][256]
};
constexpr int sa1_lds_bytes(int threads, bool bwd) {
  return (64 * 4 + 64 + 128 + 16) * 4 + 2 * 64 * SA_PH + (bwd ? 2 * 64 * SA_PH2 + 2 * 64 * SA_PH : 2 * 128 * SA_PH) +
         (threads / 64) * 256 * 4;
}

__device__ __forceinline__ Sa1Lds sa1_carve(float* sm, bool bwd) {
  Sa1Lds L;
  L.w1 = sm;
  L.b2 = L.w1 + 64 * 4;
  L.b3 = L.b2 + 64;
  L.scal = L.b3 + 128;
  L.w2h = reinterpret_cast<unsigned char*>(L.scal + 16);
  L.w3h = L.w2h + 2 * 64 * SA_PH;
  L.w2t = L.w3h + (bwd ? 2 * 64 * SA_PH2 : 2 * 128 * SA_PH);
  L.scratch = reinterpret_cast<float*>(bwd ? L.w2t + 2 * 64 * SA_PH : L.w2t);
  return L;
}

// channel held at position p = ks*16 + 8h + j of a k-permuted image: the row order of the 32x32 accumulator
__device__ __forceinline__ int sa_perm(int p) {
  const int ks = p >> 4, h = (p >> 3) & 1, j = p & 7;
  return 32 * (ks >> 1) + 16 * (ks & 1) + 4 * h + (j & 3) + 8 * (j >> 2);
}
__device__ __forceinline__ unsigned sa_exp(float m) {
  const unsigned E = (__float_as_uint(m) >> 23) & 0xffu;
  return E < 14u ? 14u : (E > 254u ? 254u : E);
}
__device__ __forceinline__ float sa_scale(unsigned E) { return __uint_as_float((267u - E) << 23); }     // max -> [2^13, 2^14)
__device__ __forceinline__ float sa_unscale(unsigned E) { return __uint_as_float((E - 13u) << 23); }
__device__ __forceinline__ void sa_put(unsigned char* img, int lo_off, int off, float v) {
  const _Float16 h = (_Float16)v;
  *reinterpret_cast<_Float16*>(img + off) = h;
  *reinterpret_cast<_Float16*>(img + lo_off + off) = (_Float16)(v - (float)h);
}

template <int SA_T, bool BWD>
__device__ __forceinline__ void sa1_stage(const geoa3_sa1_weights& w, const Sa1Lds& L) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int e = tid; e < 64; e += SA_T) {
    L.w1[4 * e + 0] = w.w1[3 * e + 0];
    L.w1[4 * e + 1] = w.w1[3 * e + 1];
    L.w1[4 * e + 2] = w.w1[3 * e + 2];
    L.w1[4 * e + 3] = w.b1[e];
    L.b2[e] = w.b2[e];
  }
  for (int e = tid; e < 128; e += SA_T) L.b3[e] = w.b3[e];
  float m2 = 0.f, m3 = 0.f;
  for (int e = tid; e < 64 * 64; e += SA_T) m2 = fmaxf(m2, __builtin_fabsf(w.w2[e]));
  for (int e = tid; e < 128 * 64; e += SA_T) m3 = fmaxf(m3, __builtin_fabsf(w.w3[e]));
  m2 = wave_max(m2);
  m3 = wave_max(m3);
  if (lane == 0) {
    L.scal[2 + wave] = m2;
    L.scal[6 + wave] = m3;
  }
  __syncthreads();
  float t2 = 0.f, t3 = 0.f;
  for (int i = 0; i < SA_T / 64; ++i) {
    t2 = fmaxf(t2, L.scal[2 + i]);
    t3 = fmaxf(t3, L.scal[6 + i]);
  }
  const unsigned E2 = sa_exp(t2), E3 = sa_exp(t3);
  const float s2 = sa_scale(E2), s3 = sa_scale(E3);
  if (tid == 0) {
    L.scal[0] = sa_unscale(E2);
    L.scal[1] = sa_unscale(E3);
  }
  for (int e = tid; e < 64 * 64; e += SA_T) {   // W2 [out][k]: A operand of layer 2
    const int o = e >> 6, k = e & 63;
    sa_put(L.w2h, 64 * SA_PH, o * SA_PH + k * 2, w.w2[e] * s2);
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

__device__ __forceinline__ void sa_swap32(float& a, float& b) {   // a[32..63] <-> b[0..31]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

// Layers 1 and 2 for the wave's 64 samples (lane = sample, p = xyz[sample] - centroid).
// h2[cb][t][r]: relu'd layer-2 output, MFMA D layout: channel t*32 + (r&3) + 8*(r>>2) + 4*(lane>>5), sample cb*32 + (lane&31).
// m1lo / m1hi: bit k set = layer-1 channel k of THIS LANE'S sample is active (z > 0).
// TWICE (the backward kernel): layer 1 is evaluated a first time for the relu masks and the wave's maximum (the power-of-
// two scale of the fp16 split needs it before any operand is formed), then sixteen channels at a time right in front of
// the k-step that consumes them -- 48 live registers less than holding all of h1, for 192 more FMAs against ~3500 vector
// instructions per centroid (the backward spilled at its 256-register budget).  Same values, same scale, same bits.
template <bool TWICE = false>
__device__ __forceinline__ void sa1_hidden(const Sa1Lds& L, float px, float py, float pz, int lane, f32x16 (&h2)[2][2],
                                           unsigned& m1lo, unsigned& m1hi) {
  float h1[TWICE ? 16 : 64];
  m1lo = 0u;
  m1hi = 0u;
  float m = 0.f;
#pragma unroll
  for (int k = 0; k < 64; ++k) {
    const float4 w = *reinterpret_cast<const float4*>(L.w1 + 4 * k);
    const float z = w.x * px + w.y * py + w.z * pz + w.w;
    if (!TWICE) h1[k] = fmaxf(z, 0.f);
    m = fmaxf(m, z);
    if (k < 32) m1lo |= (z > 0.f ? 1u : 0u) << k;
    else m1hi |= (z > 0.f ? 1u : 0u) << (k - 32);
    if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
  }
  m = fmaxf(m, 0.f);                                          // max of relu(z)
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) h2[c][t][r] = 0.f;
  const unsigned E = sa_exp(wave_max(m));
  const float sx = sa_scale(E);
  const unsigned char* wr = L.w2h + (lane & 31) * SA_PH + (lane >> 5) * 16;
#pragma unroll
  for (int c = 0; c < 4; ++c) {                                // 16 layer-1 channels = one k-step
    if (TWICE) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const float4 w = *reinterpret_cast<const float4*>(L.w1 + 4 * (16 * c + k));
        h1[k] = fmaxf(w.x * px + w.y * py + w.z * pz + w.w, 0.f);
      }
    }
    half8 xh[2], xl[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v0 = h1[(TWICE ? 0 : 16 * c) + j] * sx, v1 = h1[(TWICE ? 0 : 16 * c) + 8 + j] * sx;
      sa_swap32(v0, v1);                                       // v0: samples 0..31, v1: 32..63; lanes (sample, k half)
      const _Float16 a0 = (_Float16)v0, a1 = (_Float16)v1;
      xh[0][j] = a0;
      xl[0][j] = (_Float16)(v0 - (float)a0);
      xh[1][j] = a1;
      xl[1][j] = (_Float16)(v1 - (float)a1);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const half8 wh = *reinterpret_cast<const half8*>(wr + t * 32 * SA_PH + c * 32);
      const half8 wl = *reinterpret_cast<const half8*>(wr + t * 32 * SA_PH + c * 32 + 64 * SA_PH);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        h2[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[cb], h2[cb][t], 0, 0, 0);
        h2[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[cb], h2[cb][t], 0, 0, 0);
        h2[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[cb], h2[cb][t], 0, 0, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  const float un = sa_unscale(E) * L.scal[0];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float bias = L.b2[t * 32 + mfma_row(r, lane)];
      h2[0][t][r] = fmaxf(h2[0][t][r] * un + bias, 0.f);
      h2[1][t][r] = fmaxf(h2[1][t][r] * un + bias, 0.f);
    }
}

__global__ __launch_bounds__(SA_TF) __attribute__((amdgpu_waves_per_eu(2, 2))) void sa1_fwd_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                       const int32_t* __restrict__ idx, geoa3_sa1_weights w, int B, int N,
                                                       int M, float* __restrict__ out, uint8_t* __restrict__ arg) {
  extern __shared__ __attribute__((aligned(16))) float sa_sm[];
  const Sa1Lds L = sa1_carve(sa_sm, false);
  sa1_stage<SA_TF, false>(w, L);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long total = (long)B * M;
  for (long c = (long)blockIdx.x * (SA_TF / 64) + wave; c < total; c += (long)gridDim.x * (SA_TF / 64)) {
    asm volatile("" ::: "memory");   // the weights stay in LDS: no hoisting of their loads out of the centroid loop
    const int b = (int)(c / M);
    const int i = idx[c * SA_S + lane];
    const float* q = xyz + ((size_t)b * N + i) * 3;
    const float* ctr = new_xyz + (size_t)c * 3;
    const float px = q[0] - ctr[0], py = q[1] - ctr[1], pz = q[2] - ctr[2];
    f32x16 h2[2][2];
    unsigned m1lo, m1hi;
    sa1_hidden(L, px, py, pz, lane, h2, m1lo, m1hi);
    // layer-2 activations -> the A fragments of the transposed layer 3: registers 8s..8s+7 of tile t are k-step
    // ks = 2t + s (channels in the accumulator's row order, which the W3 image follows); one scale for the wave
    half8 ah[2][4], al[2][4];
    float un3;
    {
      float m = 0.f;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) m = fmaxf(m, h2[cb][t][r]);     // h2 >= 0
      const unsigned E = sa_exp(wave_max(m));
      const float sx = sa_scale(E);
      un3 = sa_unscale(E) * L.scal[1];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float v = h2[cb][ks >> 1][8 * (ks & 1) + j] * sx;
            const _Float16 a = (_Float16)v;
            ah[cb][ks][j] = a;
            al[cb][ks][j] = (_Float16)(v - (float)a);
          }
    }
#pragma unroll 1
    for (int t3 = 0; t3 < 4; ++t3) {
      // layer 3 TRANSPOSED: the layer-2 registers go in as the A operand (rows = samples), the weights as B (columns =
      // channels), so a lane ends up with ONE channel and 32 of its samples in registers: the max over samples is
      // lane-local plus one exchange between the register halves instead of a 32-lane shuffle reduction per register
      f32x16 a3[2];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        a3[0][r] = 0.f;
        a3[1][r] = 0.f;
      }
      const unsigned char* wr = L.w3h + (t3 * 32 + (lane & 31)) * SA_PH + (lane >> 5) * 16;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const half8 bh = *reinterpret_cast<const half8*>(wr + ks * 32);
        const half8 bl = *reinterpret_cast<const half8*>(wr + ks * 32 + 128 * SA_PH);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          a3[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb][ks], bh, a3[cb], 0, 0, 0);
          a3[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[cb][ks], bl, a3[cb], 0, 0, 0);
          a3[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[cb][ks], bh, a3[cb], 0, 0, 0);
        }
      }
      // a3[cb][r]: channel t3*32 + (lane&31), sample cb*32 + (r&3) + 8*(r>>2) + 4*(lane>>5); ascending sample order,
      // strict > : the first maximal sample wins, as F.max_pool2d
      float v = -__builtin_inff();
      int smp = 0;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool gt = a3[cb][r] > v;
          v = gt ? a3[cb][r] : v;
          smp = gt ? cb * 32 + mfma_row(r, lane) : smp;
        }
      const float ov = __shfl_xor(v, 32, 64);
      const int os = __shfl_xor(smp, 32, 64);
      const bool take = ov > v || (ov == v && os < smp);
      v = take ? ov : v;
      smp = take ? os : smp;
      if (lane < 32) {   // 128 contiguous bytes of the centroid's row of out_t [B,M,128]
        const int ch = t3 * 32 + lane;
        out[(size_t)c * 128 + ch] = fmaxf(v * un3 + L.b3[ch], 0.f);   // the (positive) scale commutes with the max
        arg[(size_t)c * 128 + ch] = (uint8_t)smp;
      }
    }
  }
}

__global__ __launch_bounds__(SA_TB) __attribute__((amdgpu_waves_per_eu(2, 2))) void sa1_bwd_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                       const int32_t* __restrict__ idx, geoa3_sa1_weights w, int B, int N,
                                                       int M, const float* __restrict__ out,
                                                       const uint8_t* __restrict__ arg, const float* __restrict__ g,
                                                       float* __restrict__ dxyz, float* __restrict__ dnew,
                                                       float* __restrict__ dpbuf) {
  extern __shared__ __attribute__((aligned(16))) float sa_sm[];
  const Sa1Lds L = sa1_carve(sa_sm, true);
  sa1_stage<SA_TB, true>(w, L);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
  float* s_gz = L.scratch + wave * 256;                       // [128] pooled gradient through the output relu
  int* s_arg = reinterpret_cast<int*>(s_gz + 128);            // [128] arg-max sample of every channel
  const long total = (long)B * M;
  for (long c = (long)blockIdx.x * (SA_TB / 64) + wave; c < total; c += (long)gridDim.x * (SA_TB / 64)) {
    asm volatile("" ::: "memory");   // the weights stay in LDS: no hoisting of their loads out of the centroid loop
    const int b = (int)(c / M);
    const int i = idx[c * SA_S + lane];
    const float* q = xyz + ((size_t)b * N + i) * 3;
    const float* ctr = new_xyz + (size_t)c * 3;
    const float px = q[0] - ctr[0], py = q[1] - ctr[1], pz = q[2] - ctr[2];
    {
      const float2 ov = *reinterpret_cast<const float2*>(out + (size_t)c * 128 + 2 * lane);
      const float2 gv = *reinterpret_cast<const float2*>(g + (size_t)c * 128 + 2 * lane);
      s_gz[2 * lane] = ov.x > 0.f ? gv.x : 0.f;
      s_gz[2 * lane + 1] = ov.y > 0.f ? gv.y : 0.f;
      s_arg[2 * lane] = arg[(size_t)c * 128 + 2 * lane];
      s_arg[2 * lane + 1] = arg[(size_t)c * 128 + 2 * lane + 1];
    }
    unsigned m1lo, m1hi, m2[2] = {0u, 0u};   // m2[cb] bit t*16 + r: layer-2 activation (D layout) is positive
    {
      f32x16 h2[2][2];
      sa1_hidden<true>(L, px, py, pz, lane, h2, m1lo, m1hi);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) m2[cb] |= (h2[cb][t][r] > 0.f ? 1u : 0u) << (t * 16 + r);
    }
    float dpx_, dpy_, dpz_;
    // d h2 [64 x 64 samples] = W3^T dz3, dz3 one-hot per channel (only the arg-max sample carries gradient).
    // The two column blocks (samples 0-31, 32-63) go through the three backward products ONE AFTER THE OTHER: with both
    // in flight the accumulators alone are 128 registers (d2, d1: 2 x 2 x 16 each) and the kernel spilled 61 dwords per
    // lane at its 256-register budget -- 1.7 GB of scratch traffic per launch (PMC: 1.97 GB against 0.28 GB algorithmic).
    // Sequentially the weight fragments are read from LDS twice and each block takes its own power-of-two scales.
    {
      // every channel's scaled gradient is split ONCE (by the lane that staged it) and kept with its arg-max sample as
      // {hi | lo << 16, sample}: the operand construction below is then one 8-byte LDS read, a compare and a select
      // per (channel, lane half) plus byte permutes
      const float g0 = s_gz[2 * lane], g1 = s_gz[2 * lane + 1];
      const int a0 = s_arg[2 * lane], a1 = s_arg[2 * lane + 1];
      const unsigned E = sa_exp(wave_max(fmaxf(__builtin_fabsf(g0), __builtin_fabsf(g1))));
      const float sg = sa_scale(E);
      const float f2 = sa_unscale(E) * L.scal[1];     // d2 holds (d h2) / f2
      unsigned* s_pa = reinterpret_cast<unsigned*>(s_gz);   // [128][2], over s_gz / s_arg (this wave's own 1 KB)
      {
        const float z0 = g0 * sg, z1 = g1 * sg;
        const _Float16 h0 = (_Float16)z0, h1 = (_Float16)z1;
        const _Float16 l0 = (_Float16)(z0 - (float)h0), l1 = (_Float16)(z1 - (float)h1);
        const unsigned p0 = (unsigned)__builtin_bit_cast(unsigned short, h0) | (unsigned)__builtin_bit_cast(unsigned short, l0) << 16;
        const unsigned p1 = (unsigned)__builtin_bit_cast(unsigned short, h1) | (unsigned)__builtin_bit_cast(unsigned short, l1) << 16;
        typedef unsigned uint4v __attribute__((ext_vector_type(4)));
        *reinterpret_cast<uint4v*>(s_pa + 4 * lane) = uint4v{p0, (unsigned)a0, p1, (unsigned)a1};
      }
      float part[2][3];
#pragma unroll 1
      for (int cb = 0; cb < 2; ++cb) {
        asm volatile("" ::: "memory");   // (the weight fragments are re-read per block: no hoisting out of this loop)
        f32x16 d2[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) d2[t][r] = 0.f;
        {
          // one-hot B operand: element j of lane (sample, k half) for k-step ks is channel ch = 16 ks + 8h + j, non-zero
          // only in the lane of the channel's arg-max sample; the A operand is the W3^T image
          const unsigned char* wr = L.w3h + l31 * SA_PH2 + h * 16;
          const int mysample = cb * 32 + l31;
#pragma unroll 2
          for (int ks = 0; ks < 8; ++ks) {
            unsigned s0[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const int ch = 16 * ks + 8 * h + j;
              const uint2 pa = *reinterpret_cast<const uint2*>(s_pa + 2 * ch);
              s0[j] = (int)pa.y == mysample ? pa.x : 0u;
            }
            typedef unsigned uint4v __attribute__((ext_vector_type(4)));
            uint4v vh0, vl0;
#pragma unroll
            for (int j2 = 0; j2 < 4; ++j2) {
              vh0[j2] = __builtin_amdgcn_perm(s0[2 * j2 + 1], s0[2 * j2], 0x05040100u);   // low halves: hi pieces
              vl0[j2] = __builtin_amdgcn_perm(s0[2 * j2 + 1], s0[2 * j2], 0x07060302u);   // high halves: lo pieces
            }
            const half8 bh = __builtin_bit_cast(half8, vh0), bl = __builtin_bit_cast(half8, vl0);
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const half8 wh = *reinterpret_cast<const half8*>(wr + t * 32 * SA_PH2 + ks * 32);
              const half8 wl = *reinterpret_cast<const half8*>(wr + t * 32 * SA_PH2 + ks * 32 + 64 * SA_PH2);
              d2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, d2[t], 0, 0, 0);
              d2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, d2[t], 0, 0, 0);
              d2[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, d2[t], 0, 0, 0);
            }
          }
        }
        // through relu 2, then d h1 = W2^T dz2 with the accumulator registers as B operands (registers 8s..8s+7 of tile
        // t = k-step 2t + s in the accumulator's row order; the W2^T image follows it)
        f32x16 d1[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) d1[t][r] = 0.f;
        float f1;   // d1 holds (d h1) / f1
        {
          const unsigned mm = cb ? m2[1] : m2[0];
          float m = 0.f;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              d2[t][r] = ((mm >> (t * 16 + r)) & 1u) ? d2[t][r] : 0.f;
              m = fmaxf(m, __builtin_fabsf(d2[t][r]));
            }
          const unsigned E1 = sa_exp(wave_max(m));
          const float sx = sa_scale(E1);
          f1 = f2 * sa_unscale(E1) * L.scal[0];
          const unsigned char* wr = L.w2t + l31 * SA_PH + h * 16;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            half8 bh, bl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float v = d2[ks >> 1][8 * (ks & 1) + j] * sx;
              const _Float16 a = (_Float16)v;
              bh[j] = a;
              bl[j] = (_Float16)(v - (float)a);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
              const half8 wh = *reinterpret_cast<const half8*>(wr + t * 32 * SA_PH + ks * 32);
              const half8 wl = *reinterpret_cast<const half8*>(wr + t * 32 * SA_PH + ks * 32 + 64 * SA_PH);
              d1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh, d1[t], 0, 0, 0);
              d1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl, d1[t], 0, 0, 0);
              d1[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh, d1[t], 0, 0, 0);
            }
          }
        }
        // through relu 1 (mask of sample cb*32 + l31 lives in that lane) and the K = 3 layer
        {
          const unsigned mlo = __shfl(m1lo, cb * 32 + l31, 64), mhi = __shfl(m1hi, cb * 32 + l31, 64);
          float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int kk = (r & 3) + 8 * (r >> 2) + 4 * h;     // bit within the word of tile t
              const bool on = (((t == 0 ? mlo : mhi) >> kk) & 1u) != 0u;
              const float z = on ? d1[t][r] * f1 : 0.f;
              const float4 wv = *reinterpret_cast<const float4*>(L.w1 + 4 * (t * 32 + kk));
              sx += wv.x * z;
              sy += wv.y * z;
              sz += wv.z * z;
            }
          part[cb][0] = sx + __shfl_xor(sx, 32, 64);
          part[cb][1] = sy + __shfl_xor(sy, 32, 64);
          part[cb][2] = sz + __shfl_xor(sz, 32, 64);
        }
      }
      dpx_ = h ? part[1][0] : part[0][0];
      dpy_ = h ? part[1][1] : part[0][1];
      dpz_ = h ? part[1][2] : part[0][2];
    }
    const float dpx = dpx_, dpy = dpy_, dpz = dpz_;
    // scatter to the gathered points (entries repeating the row's first index -- the ball query's padding -- leave as
    // one add) and minus the sum to the centroid
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

// dxyz[b][n] = sum over the (centroid m, sample s) with idx[b][m][s] == n of dp[b][m][s], in ascending (m, s): the
// scatter-add of the grouping gradient (group_points_gpu.cu:60 behind pointnet2_utils.py:318) as an OWNER-side sum over
// reverse lists built per launch in LDS (histogram, scan, unordered fill, per-owner sort), chunked over the centroids so
// that a chunk's lists fit.  Padded entries (repeats of a ball's first index) were merged into sample 0 by the producer
// and are skipped.  One workgroup per instance.  Deterministic, no atomics on floats.
constexpr int SCAT_BLOCK = 512;
__global__ __launch_bounds__(SCAT_BLOCK) void sa1_scatter_kernel(const float* __restrict__ dp, const int32_t* __restrict__ idx,
                                                                float* __restrict__ dxyz, int N, int M, int rcap) {
  extern __shared__ __attribute__((aligned(16))) int sc_sm[];
  int* s_cnt = sc_sm;                                         // [N + 1]
  float* s_g = reinterpret_cast<float*>(sc_sm + N + 1);       // [3 N]
  int* rlist = reinterpret_cast<int*>(s_g + 3 * N);           // [rcap]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int32_t* I = idx + (size_t)b * M * SA_S;
  const float* D = dp + (size_t)b * M * SA_S * 3;
  for (int i = tid; i < 3 * N; i += SCAT_BLOCK) s_g[i] = 0.f;
  const int cm = max(1, rcap / SA_S);                         // centroids per chunk
  for (int m0 = 0; m0 < M; m0 += cm) {
    const int nent = (min(M, m0 + cm) - m0) * SA_S, e0 = m0 * SA_S;
    for (int i = tid; i <= N; i += SCAT_BLOCK) s_cnt[i] = 0;
    __syncthreads();
    for (int e = tid; e < nent; e += SCAT_BLOCK) {
      const int ge = e0 + e, q = I[ge];
      if ((ge & (SA_S - 1)) == 0 || q != I[ge & ~(SA_S - 1)]) atomicAdd(&s_cnt[q], 1);
    }
    __syncthreads();
    if (tid < 64) {
      const int per = (N + 1 + 63) / 64, a0 = tid * per, a1 = min(a0 + per, N + 1);
      int sum = 0;
      for (int i = a0; i < a1; ++i) sum += s_cnt[i];
      int incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (tid >= o) incl += v;
      }
      int run = incl - sum;
      for (int i = a0; i < a1; ++i) {
        const int c = s_cnt[i];
        s_cnt[i] = run;
        run += c;
      }
    }
    __syncthreads();
    for (int e = tid; e < nent; e += SCAT_BLOCK) {
      const int ge = e0 + e, q = I[ge];
      if ((ge & (SA_S - 1)) == 0 || q != I[ge & ~(SA_S - 1)]) rlist[atomicAdd(&s_cnt[q], 1)] = ge;
    }
    __syncthreads();
    for (int i = tid; i < N; i += SCAT_BLOCK) {
      const int st = i ? s_cnt[i - 1] : 0, n = s_cnt[i] - st;
      if (n == 0) continue;
      int* L = rlist + st;
      if (n <= 24) {
        for (int a = 1; a < n; ++a) {
          const int v = L[a];
          int c = a - 1;
          while (c >= 0 && L[c] > v) {
            L[c + 1] = L[c];
            --c;
          }
          L[c + 1] = v;
        }
      } else {
        auto sift = [&](int start, int end) {
          int root = start;
          for (;;) {
            int child = 2 * root + 1;
            if (child > end) break;
            if (child + 1 <= end && L[child] < L[child + 1]) ++child;
            if (L[root] >= L[child]) break;
            const int tmp = L[root];
            L[root] = L[child];
            L[child] = tmp;
            root = child;
          }
        };
        for (int h0 = (n - 2) / 2; h0 >= 0; --h0) sift(h0, n - 1);
        for (int end = n - 1; end > 0; --end) {
          const int tmp = L[0];
          L[0] = L[end];
          L[end] = tmp;
          sift(0, end - 1);
        }
      }
      float gx = s_g[3 * i], gy = s_g[3 * i + 1], gz = s_g[3 * i + 2];
      for (int e = 0; e < n; ++e) {
        const float* d = D + (size_t)L[e] * 3;
        gx += d[0];
        gy += d[1];
        gz += d[2];
      }
      s_g[3 * i] = gx;
      s_g[3 * i + 1] = gy;
      s_g[3 * i + 2] = gz;
    }
    __syncthreads();
  }
  float* G = dxyz + (size_t)b * N * 3;
  for (int i = tid; i < 3 * N; i += SCAT_BLOCK) G[i] = s_g[i];
}

int sa1_grid(int B, int M, int waves) {
  const long groups = ((long)B * M + waves - 1) / waves;
  return (int)(groups < 256 * 2 ? groups : 256 * 2);   // persistent: 2 workgroups per CU (62 / 78 KB LDS each)
}

}  // namespace

extern "C" int geoa3_pn2_sa1_forward(const float* xyz, const float* new_xyz, const int32_t* idx,
                                     const geoa3_sa1_weights* w, int B, int N, int M, float* out, uint8_t* arg,
                                     void* stream) {
  if (!xyz || !new_xyz || !idx || !w || !out || !arg || B <= 0 || N <= 0 || M <= 0) return GEOA3_EINVAL;
  const size_t lds = (size_t)sa1_lds_bytes(SA_TF, false);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa1_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  geoa3_prof_begin(GEOA3_PROF_SA1_FWD, geoa3_stream(stream));
  hipLaunchKernelGGL(sa1_fwd_kernel, dim3(sa1_grid(B, M, SA_TF / 64)), dim3(SA_TF), lds, geoa3_stream(stream), xyz, new_xyz, idx, *w,
                     B, N, M, out, arg);
  geoa3_prof_end(GEOA3_PROF_SA1_FWD, geoa3_stream(stream));
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int64_t geoa3_pn2_sa1_scratch_bytes(int B, int M) {
  if (B <= 0 || M <= 0) return -1;
  return (int64_t)B * M * SA_S * 3 * (int64_t)sizeof(float);
}

extern "C" int geoa3_pn2_sa1_backward(const float* xyz, const float* new_xyz, const int32_t* idx,
                                      const geoa3_sa1_weights* w, int B, int N, int M, const float* out,
                                      const uint8_t* arg, const float* grad_out, float* grad_xyz, float* grad_new_xyz,
                                      float* scratch, void* stream) {
  if (!xyz || !new_xyz || !idx || !w || !out || !arg || !grad_out || !grad_xyz || !grad_new_xyz || B <= 0 || N <= 0 ||
      M <= 0)
    return GEOA3_EINVAL;
  hipStream_t s = geoa3_stream(stream);
  // deterministic scatter (scratch given and the owner's accumulators + at least one centroid's lists fit LDS)
  const size_t fixed = ((size_t)N + 1) * sizeof(int) + (size_t)3 * N * sizeof(float), cap = 160 * 1024 - 512;
  const bool det = scratch && fixed + SA_S * sizeof(int) <= cap;
  if (!det && hipMemsetAsync(grad_xyz, 0, (size_t)B * N * 3 * sizeof(float), s) != hipSuccess) return GEOA3_ELAUNCH;
  const size_t lds = (size_t)sa1_lds_bytes(SA_TB, true);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa1_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  geoa3_prof_begin(GEOA3_PROF_SA1_BWD, s);
  hipLaunchKernelGGL(sa1_bwd_kernel, dim3(sa1_grid(B, M, SA_TB / 64)), dim3(SA_TB), lds, s, xyz, new_xyz, idx, *w, B, N, M, out, arg,
                     grad_out, grad_xyz, grad_new_xyz, det ? scratch : nullptr);
  geoa3_prof_end(GEOA3_PROF_SA1_BWD, s);
  if (det) {
    size_t rcap = (cap - fixed) / sizeof(int);
    if (rcap > (size_t)M * SA_S) rcap = (size_t)M * SA_S;
    const size_t l2 = fixed + rcap * sizeof(int);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa1_scatter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)l2);
    hipLaunchKernelGGL(sa1_scatter_kernel, dim3(B), dim3(SCAT_BLOCK), l2, s, scratch, idx, grad_xyz, N, M, (int)rcap);
  }
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
