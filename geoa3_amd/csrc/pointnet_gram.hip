// P[b][i][o] = sum_n A[b][i][n] * G[b][o][n]   (64 x 64 per instance, contraction over the N points): the product
// h2 G64a^T behind dT64 = (h2 G64a^T) W3 in the PointNet backward (pointnet.hip).  Through the generic FC kernel
// (32 x 32 tiles, 16-byte k-strided loads, fp32 MFMA) it took 74 us for 131 MB (1.8 TB/s): both operands are read
// twice and in 32-byte pieces.  Here ONE workgroup owns an instance: 128-column chunks of both matrices arrive as
// coalesced 512-byte row pieces, are scaled by a power of two from the chunk's own maximum, split into two fp16 images
// in LDS (row-major in n: exactly the k-contiguous fragment both MFMA operands need, one ds_read_b128 each), each
// wave multiplies one 32 x 32 quadrant on the f16 matrix pipe (a*g = a_hi*g_hi + a_hi*g_lo + a_lo*g_hi, fp32
// accumulation), and the chunk's sum is scaled back and added in fp32.  The next chunk's rows are in flight in
// registers meanwhile.  A chunk passes through the LDS images in two 64-column halves (same scales, same order of the
// k-steps: same bits): 35 KB per workgroup, four workgroups per CU -- 51 -> 30.5 us at B = 250, N = 1024 (5.3 TB/s).  One workgroup per instance is latency bound (4 waves per CU: 90 us); four workgroups per
// instance take every fourth chunk and a second kernel adds their partial sums in a fixed order.  Deterministic.
#include "pointnet_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int GR_CHK = 128;                  // columns per chunk (one power-of-two scale per matrix and chunk)
constexpr int GR_HALF = GR_CHK / 2;          // columns in LDS at a time: a chunk goes through the images in two halves --
                                             // 35 KB per workgroup instead of 70, four workgroups per CU instead of two
                                             // (the kernel is bound by HBM latency: 51 -> 30.5 us at B = 250, N = 1024)
constexpr int GR_PITCH = GR_HALF * 2 + 16;   // bytes per LDS row (conflict-free 16-byte reads: 36 banks = 4 mod 32)
constexpr int GR_IMG = 64 * GR_PITCH;        // one piece of one matrix

__device__ __forceinline__ unsigned gr_exp(float m) {
  const unsigned E = (__float_as_uint(m) >> 23) & 0xffu;
  return E < 14u ? 14u : (E > 254u ? 254u : E);
}

__global__ __launch_bounds__(256) void gram64_kernel(const float* __restrict__ A, const float* __restrict__ G, int N,
                                                     float* __restrict__ P, int parts) {
  extern __shared__ __attribute__((aligned(16))) unsigned char gr_smem[];   // [A hi][A lo][G hi][G lo]
  __shared__ float s_red[2][4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* Ab = A + (size_t)b * 64 * N;
  const float* Gb = G + (size_t)b * 64 * N;
  const int chunks = (N + GR_CHK - 1) / GR_CHK;
  const int part = blockIdx.y;   // this workgroup: chunks part, part + parts, .. (partial sums: gram64_reduce_kernel)
  // wave w stages rows 16w .. 16w+15 of both matrices; a lane the columns `lane` and 64 + `lane` of the chunk (.x / .y)
  float2 ra[16], rg[16];
  auto load_chunk = [&](int c) {
    const int n = c * GR_CHK + lane;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const size_t off = (size_t)(16 * wave + r) * N + n;
      ra[r].x = n < N ? Ab[off] : 0.f;
      ra[r].y = n + GR_HALF < N ? Ab[off + GR_HALF] : 0.f;
      rg[r].x = n < N ? Gb[off] : 0.f;
      rg[r].y = n + GR_HALF < N ? Gb[off + GR_HALF] : 0.f;
    }
  };
  load_chunk(part);
  const int qi = wave >> 1, qo = wave & 1;        // this wave's quadrant: rows 32 qi .. (A), columns 32 qo .. (G rows)
  const unsigned char* pa = gr_smem + (32 * qi + (lane & 31)) * GR_PITCH + (lane >> 5) * 16;
  const unsigned char* pg = gr_smem + 2 * GR_IMG + (32 * qo + (lane & 31)) * GR_PITCH + (lane >> 5) * 16;
  f32x16 sum;
#pragma unroll
  for (int i = 0; i < 16; ++i) sum[i] = 0.f;
  for (int c = part; c < chunks; c += parts) {
    float ma = 0.f, mg = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      ma = fmaxf(ma, fmaxf(__builtin_fabsf(ra[r].x), __builtin_fabsf(ra[r].y)));
      mg = fmaxf(mg, fmaxf(__builtin_fabsf(rg[r].x), __builtin_fabsf(rg[r].y)));
    }
    ma = wave_max(ma);
    mg = wave_max(mg);
    __syncthreads();   // the previous chunk's images and maxima have been consumed
    if (lane == 0) {
      s_red[0][wave] = ma;
      s_red[1][wave] = mg;
    }
    __syncthreads();
    const unsigned Ea = gr_exp(fmaxf(fmaxf(s_red[0][0], s_red[0][1]), fmaxf(s_red[0][2], s_red[0][3])));
    const unsigned Eg = gr_exp(fmaxf(fmaxf(s_red[1][0], s_red[1][1]), fmaxf(s_red[1][2], s_red[1][3])));
    const float sa = __uint_as_float((267u - Ea) << 23), sg = __uint_as_float((267u - Eg) << 23);
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
      if (hf) __syncthreads();   // the first half's images have been consumed
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        unsigned char* d = gr_smem + (16 * wave + r) * GR_PITCH + lane * 2;   // column `lane` of this half
        const float a0 = (hf ? ra[r].y : ra[r].x) * sa, g0 = (hf ? rg[r].y : rg[r].x) * sg;
        const _Float16 ah0 = (_Float16)a0, gh0 = (_Float16)g0;
        *reinterpret_cast<_Float16*>(d) = ah0;
        *reinterpret_cast<_Float16*>(d + GR_IMG) = (_Float16)(a0 - (float)ah0);
        *reinterpret_cast<_Float16*>(d + 2 * GR_IMG) = gh0;
        *reinterpret_cast<_Float16*>(d + 3 * GR_IMG) = (_Float16)(g0 - (float)gh0);
      }
      if (hf && c + parts < chunks) load_chunk(c + parts);   // in flight under the MFMAs (the registers are free now)
      __syncthreads();
#pragma unroll
      for (int s = 0; s < GR_HALF / 16; ++s) {
        const half8 ah = *reinterpret_cast<const half8*>(pa + s * 32);
        const half8 al = *reinterpret_cast<const half8*>(pa + s * 32 + GR_IMG);
        const half8 gh = *reinterpret_cast<const half8*>(pg + s * 32);
        const half8 gl = *reinterpret_cast<const half8*>(pg + s * 32 + GR_IMG);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, gh, acc, 0, 0, 0);
      }
    }
    const float un = __uint_as_float((Ea - 13u) << 23) * __uint_as_float((Eg - 13u) << 23);
#pragma unroll
    for (int i = 0; i < 16; ++i) sum[i] += acc[i] * un;
  }
  // D[i][o]: o = 32 qo + (lane & 31), i = 32 qi + (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
  float* Pb = P + ((size_t)b * parts + part) * 4096;
#pragma unroll
  for (int r = 0; r < 16; ++r) Pb[(32 * qi + mfma_row(r, lane)) * 64 + 32 * qo + (lane & 31)] = sum[r];
}

__global__ __launch_bounds__(256) void gram64_reduce_kernel(const float* __restrict__ part, int parts, int total,
                                                            float* __restrict__ P) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int b = e >> 12, i = e & 4095;
  float v = 0.f;
  for (int p = 0; p < parts; ++p) v += part[((size_t)b * parts + p) * 4096 + i];   // fixed order
  P[e] = v;
}

}  // namespace

// scratch: B * parts * 4096 floats with parts = gram64_parts(N) (1: P is written directly, scratch unused).  A function
// of N only (never of the batch): an instance's sums must not depend on which other instances share the launch.  Eight
// parts from 8 chunks on: one chunk per workgroup at N = 1024 -- the same 53 us as four at 250 instances, 13.6 against
// 20.4 us on a 32-instance shard
int gram64_parts(int N) {
  const int chunks = (N + GR_CHK - 1) / GR_CHK;
  return chunks >= 8 ? 8 : (chunks >= 4 ? 4 : 1);
}

int launch_gram64(const float* A, const float* G, int B, int N, float* P, float* scratch, hipStream_t s) {
  auto kern = gram64_kernel;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * GR_IMG);
  const int parts = scratch ? gram64_parts(N) : 1;
  hipLaunchKernelGGL(kern, dim3(B, parts), dim3(256), 4 * GR_IMG, s, A, G, N, parts > 1 ? scratch : P, parts);
  if (parts > 1)
    hipLaunchKernelGGL(gram64_reduce_kernel, dim3((B * 4096 + 255) / 256), dim3(256), 0, s, scratch, parts, B * 4096, P);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
