// PointNet++ SSG, first set-abstraction level, fused (gfx950): ball-grouped xyz -> shared MLP 3->64->64->128 (Conv2d 1x1 +
// eval-mode BatchNorm2d folded + ReLU) -> max over the 64 samples, forward and input-gradient.
// Reference: pointnet2_modules.py:29-74 (_PointnetSAModuleBase.forward), pointnet2_utils.py:296-333 (QueryAndGroup),
// PointNetPP_ssg.py:58-66 (npoint 512, radius 0.2, nsample 64, mlp [0(+3), 64, 64, 128]).
//
// The reference materialises [B,64,512,64] and [B,128,512,64] activations (2.1 + 2.1 + 4.2 GB at B = 250) and walks
// them once per layer and once more per layer in backward.  Here ONE WAVEFRONT owns one centroid: lane = sample.
//   layer 1 (K = 3) on the VALU, lane-local; its 64 outputs per lane become the B operands of layer 2 by swapping
//   register halves (v_permlane32_swap: two 64-sample rows -> the operands of the two 32-sample column blocks);
//   layer 2 and 3 on the fp32 matrix core (32x32x2); the D registers of layer 2 ARE the B operands of layer 3 (a D
//   register holds rows k and k+4 in its two halves, which is exactly one k-step when the A operand is read with the
//   same pairing), so no activation ever leaves the registers; the max over samples is a shuffle reduction.
// Backward recomputes the two hidden layers (cheaper than reading 6 GB), routes the pooled gradient through a one-hot
// B operand, chains D->B again through W2^T, and finishes with the K = 3 contraction and the scatter to the points.
#include "pointnet_kernels.h"

namespace {

constexpr int SA_TF = 256;         // forward: 4 wavefronts per workgroup (6 waves / 3 per SIMD measured slower)
constexpr int SA_TB = 256;         // backward: 4 wavefronts (256 VGPRs: 2 waves/SIMD)
constexpr int SA_P = 65;           // LDS pitch of the 64-wide weight rows (bank-conflict-free column reads)
constexpr int SA_S = 64;           // samples per centroid

struct Sa1Lds {
  float* w1;   // [64][4]  (w0, w1, w2, shift)
  float* w2;   // [64][SA_P]
  float* b2;   // [64]
  float* w3;   // [128][SA_P]
  float* b3;   // [128]
  float* scratch;   // [waves][256]
};
constexpr int sa1_lds_floats(int threads) { return 64 * 4 + 64 * SA_P + 64 + 128 * SA_P + 128 + (threads / 64) * 256; }

__device__ __forceinline__ Sa1Lds sa1_carve(float* sm) {
  Sa1Lds L;
  L.w1 = sm;
  L.w2 = L.w1 + 64 * 4;
  L.b2 = L.w2 + 64 * SA_P;
  L.w3 = L.b2 + 64;
  L.b3 = L.w3 + 128 * SA_P;
  L.scratch = L.b3 + 128;
  return L;
}

template <int SA_T>
__device__ __forceinline__ void sa1_stage(const geoa3_sa1_weights& w, const Sa1Lds& L) {
  const int tid = threadIdx.x;
  for (int e = tid; e < 64; e += SA_T) {
    L.w1[4 * e + 0] = w.w1[3 * e + 0];
    L.w1[4 * e + 1] = w.w1[3 * e + 1];
    L.w1[4 * e + 2] = w.w1[3 * e + 2];
    L.w1[4 * e + 3] = w.b1[e];
    L.b2[e] = w.b2[e];
  }
  for (int e = tid; e < 128; e += SA_T) L.b3[e] = w.b3[e];
  for (int e = tid; e < 64 * 64; e += SA_T) L.w2[(e >> 6) * SA_P + (e & 63)] = w.w2[e];
  for (int e = tid; e < 128 * 64; e += SA_T) L.w3[(e >> 6) * SA_P + (e & 63)] = w.w3[e];
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
__device__ __forceinline__ void sa1_hidden(const Sa1Lds& L, float px, float py, float pz, int lane, f32x16 (&h2)[2][2],
                                           unsigned& m1lo, unsigned& m1hi) {
  float h1[64];
  m1lo = 0u;
  m1hi = 0u;
#pragma unroll
  for (int k = 0; k < 64; ++k) {
    const float4 w = *reinterpret_cast<const float4*>(L.w1 + 4 * k);
    const float z = w.x * px + w.y * py + w.z * pz + w.w;
    h1[k] = fmaxf(z, 0.f);
    if (k < 32) m1lo |= (z > 0.f ? 1u : 0u) << k;
    else m1hi |= (z > 0.f ? 1u : 0u) << (k - 32);
    if ((k & 7) == 7) __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) h2[c][t][r] = 0.f;
  const float* wr = L.w2 + (lane & 31) * SA_P + (lane >> 5);
#pragma unroll
  for (int s = 0; s < 32; ++s) {
    sa_swap32(h1[2 * s], h1[2 * s + 1]);
    const float a0 = wr[2 * s], a1 = wr[32 * SA_P + 2 * s];
    h2[0][0] = mfma32(a0, h1[2 * s], h2[0][0]);
    h2[1][0] = mfma32(a0, h1[2 * s + 1], h2[1][0]);
    h2[0][1] = mfma32(a1, h1[2 * s], h2[0][1]);
    h2[1][1] = mfma32(a1, h1[2 * s + 1], h2[1][1]);
    if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // keep the operand loads next to their MFMAs (registers)
  }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float bias = L.b2[t * 32 + mfma_row(r, lane)];
      h2[0][t][r] = fmaxf(h2[0][t][r] + bias, 0.f);
      h2[1][t][r] = fmaxf(h2[1][t][r] + bias, 0.f);
    }
}

__global__ __launch_bounds__(SA_TF) __attribute__((amdgpu_waves_per_eu(2, 2))) void sa1_fwd_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                       const int32_t* __restrict__ idx, geoa3_sa1_weights w, int B, int N,
                                                       int M, float* __restrict__ out, uint8_t* __restrict__ arg) {
  extern __shared__ __attribute__((aligned(16))) float sa_sm[];
  const Sa1Lds L = sa1_carve(sa_sm);
  sa1_stage<SA_TF>(w, L);
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
      const float* wr = L.w3 + (t3 * 32 + (lane & 31)) * SA_P + 4 * (lane >> 5);
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a = wr[t * 32 + (r & 3) + 8 * (r >> 2)];
          a3[0] = mfma32(h2[0][t][r], a, a3[0]);
          a3[1] = mfma32(h2[1][t][r], a, a3[1]);
          if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
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
        out[(size_t)c * 128 + ch] = fmaxf(v + L.b3[ch], 0.f);
        arg[(size_t)c * 128 + ch] = (uint8_t)smp;
      }
    }
  }
}

__global__ __launch_bounds__(SA_TB) __attribute__((amdgpu_waves_per_eu(2, 2))) void sa1_bwd_kernel(const float* __restrict__ xyz, const float* __restrict__ new_xyz,
                                                       const int32_t* __restrict__ idx, geoa3_sa1_weights w, int B, int N,
                                                       int M, const float* __restrict__ out,
                                                       const uint8_t* __restrict__ arg, const float* __restrict__ g,
                                                       float* __restrict__ dxyz, float* __restrict__ dnew) {
  extern __shared__ __attribute__((aligned(16))) float sa_sm[];
  const Sa1Lds L = sa1_carve(sa_sm);
  sa1_stage<SA_TB>(w, L);
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
      sa1_hidden(L, px, py, pz, lane, h2, m1lo, m1hi);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) m2[cb] |= (h2[cb][t][r] > 0.f ? 1u : 0u) << (t * 16 + r);
    }
    // d h2 [64 x 64 samples] = W3^T dz3, dz3 one-hot per channel (only the arg-max sample carries gradient)
    f32x16 d2[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) d2[cb][t][r] = 0.f;
#pragma unroll 8
    for (int s = 0; s < 64; ++s) {
      const int ch = 2 * s + h;
      const float gz = s_gz[ch];
      const int am = s_arg[ch];
      const float b0 = am == l31 ? gz : 0.f, b1 = am == 32 + l31 ? gz : 0.f;
      const float a0 = L.w3[ch * SA_P + l31], a1 = L.w3[ch * SA_P + 32 + l31];
      d2[0][0] = mfma32(a0, b0, d2[0][0]);
      d2[1][0] = mfma32(a0, b1, d2[1][0]);
      d2[0][1] = mfma32(a1, b0, d2[0][1]);
      d2[1][1] = mfma32(a1, b1, d2[1][1]);
      if ((s & 3) == 3) __builtin_amdgcn_sched_barrier(0);
    }
    // through relu 2, then d h1 = W2^T dz2 with the D registers as B operands (rows k, k+4 in the two halves)
    f32x16 d1[2][2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) d1[cb][t][r] = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float z0 = ((m2[0] >> (t * 16 + r)) & 1u) ? d2[0][t][r] : 0.f;
        const float z1 = ((m2[1] >> (t * 16 + r)) & 1u) ? d2[1][t][r] : 0.f;
        const int k = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        const float a0 = L.w2[k * SA_P + l31], a1 = L.w2[k * SA_P + 32 + l31];
        d1[0][0] = mfma32(a0, z0, d1[0][0]);
        d1[1][0] = mfma32(a0, z1, d1[1][0]);
        d1[0][1] = mfma32(a1, z0, d1[0][1]);
        d1[1][1] = mfma32(a1, z1, d1[1][1]);
        if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
      }
    // through relu 1 (mask of sample cb*32 + l31 lives in that lane) and the K = 3 layer
    float part[2][3];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      const unsigned mlo = __shfl(m1lo, cb * 32 + l31, 64), mhi = __shfl(m1hi, cb * 32 + l31, 64);
      float sx = 0.f, sy = 0.f, sz = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int kk = (r & 3) + 8 * (r >> 2) + 4 * h;     // bit within the word of tile t
          const bool on = (((t == 0 ? mlo : mhi) >> kk) & 1u) != 0u;
          const float z = on ? d1[cb][t][r] : 0.f;
          const float4 wv = *reinterpret_cast<const float4*>(L.w1 + 4 * (t * 32 + kk));
          sx += wv.x * z;
          sy += wv.y * z;
          sz += wv.z * z;
          if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
      part[cb][0] = sx + __shfl_xor(sx, 32, 64);
      part[cb][1] = sy + __shfl_xor(sy, 32, 64);
      part[cb][2] = sz + __shfl_xor(sz, 32, 64);
    }
    const float dpx = h ? part[1][0] : part[0][0], dpy = h ? part[1][1] : part[0][1], dpz = h ? part[1][2] : part[0][2];
    // scatter to the gathered points (entries repeating the row's first index -- the ball query's padding -- leave as
    // one add) and minus the sum to the centroid
    const int i0 = __shfl(i, 0, 64);
    const bool dup = lane > 0 && i == i0;
    const float ex = wave_sum(dup ? dpx : 0.f), ey = wave_sum(dup ? dpy : 0.f), ez = wave_sum(dup ? dpz : 0.f);
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
    const float tx = wave_sum(dpx), ty = wave_sum(dpy), tz = wave_sum(dpz);
    if (lane == 0) {
      dnew[(size_t)c * 3 + 0] = -tx;
      dnew[(size_t)c * 3 + 1] = -ty;
      dnew[(size_t)c * 3 + 2] = -tz;
    }
  }
}

int sa1_grid(int B, int M, int waves) {
  const long groups = ((long)B * M + waves - 1) / waves;
  return (int)(groups < 256 * 2 ? groups : 256 * 2);   // persistent: 2 workgroups per CU (56-58 KB LDS each)
}

}  // namespace

extern "C" int geoa3_pn2_sa1_forward(const float* xyz, const float* new_xyz, const int32_t* idx,
                                     const geoa3_sa1_weights* w, int B, int N, int M, float* out, uint8_t* arg,
                                     void* stream) {
  if (!xyz || !new_xyz || !idx || !w || !out || !arg || B <= 0 || N <= 0 || M <= 0) return GEOA3_EINVAL;
  const size_t lds = (size_t)sa1_lds_floats(SA_TF) * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa1_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  hipLaunchKernelGGL(sa1_fwd_kernel, dim3(sa1_grid(B, M, SA_TF / 64)), dim3(SA_TF), lds, geoa3_stream(stream), xyz, new_xyz, idx, *w,
                     B, N, M, out, arg);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_pn2_sa1_backward(const float* xyz, const float* new_xyz, const int32_t* idx,
                                      const geoa3_sa1_weights* w, int B, int N, int M, const float* out,
                                      const uint8_t* arg, const float* grad_out, float* grad_xyz, float* grad_new_xyz,
                                      void* stream) {
  if (!xyz || !new_xyz || !idx || !w || !out || !arg || !grad_out || !grad_xyz || !grad_new_xyz || B <= 0 || N <= 0 ||
      M <= 0)
    return GEOA3_EINVAL;
  hipStream_t s = geoa3_stream(stream);
  if (hipMemsetAsync(grad_xyz, 0, (size_t)B * N * 3 * sizeof(float), s) != hipSuccess) return GEOA3_ELAUNCH;
  const size_t lds = (size_t)sa1_lds_floats(SA_TB) * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa1_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  hipLaunchKernelGGL(sa1_bwd_kernel, dim3(sa1_grid(B, M, SA_TB / 64)), dim3(SA_TB), lds, s, xyz, new_xyz, idx, *w, B, N, M, out, arg,
                     grad_out, grad_xyz, grad_new_xyz);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
