// fp32 MFMA GEMMs for the narrow PointNet layers (gfx950): the channel-major 1x1 convolutions /
// per-instance feature transforms (Model/PointNet.py:79-80,138-145) and the fully connected heads
// (:82-84,150-152), forward and input-gradient.  v_mfma_f32_32x32x2_f32 is an exact k-ordered fmaf
// chain, so results differ from the CPU reference only by summation order.
#include "pointnet_kernels.h"
#include <cstdlib>

namespace {

// ------------------------------------------------------------------------------------------
// conv_cm: one wavefront owns 32 columns (points) and ALL Co output channels; the weight matrix of the
// instance sits in LDS (pitch K+1: bank-conflict-free column reads) and is staged ONCE per workgroup for
// CONV_TILES x 128 points; the activations stream straight from HBM into the B operand (each element
// is read once and reused Co/32 times from the register), 16 k-steps of loads in flight per wave.
// (Measured: 8- and 16-byte-per-lane column vectorisation is slower here -- it quadruples the
// accumulator registers and the kernel is latency-, not instruction-bound.)
// ------------------------------------------------------------------------------------------
constexpr int CONV_TILES = 1;   // 128-point tiles per workgroup (more, longer workgroups measured slower)

template <int CT>  // Co = 32*CT
__global__ __launch_bounds__(256) void conv_cm_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];  // [Co][K+1]
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int K = a.K, Co = 32 * CT, pitch = K + 1;
  const float* W = a.W + (size_t)b * a.sWb;
  if (a.sWk == 1 && (a.sWco & 3) == 0 && (a.sWb & 3) == 0) {   // rows are k-contiguous: 16-byte loads
    for (int e = tid; e < Co * K / 4; e += 256) {
      const int co = e / (K / 4), k = (e - co * (K / 4)) * 4;
      const float4 w = *reinterpret_cast<const float4*>(W + (size_t)co * a.sWco + k);
      float* d = s_w + co * pitch + k;
      d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
    }
  } else if (a.sWk == 1) {
    for (int e = tid; e < Co * K; e += 256) {
      const int co = e / K, k = e - co * K;
      s_w[co * pitch + k] = W[(size_t)co * a.sWco + k];
    }
  } else {  // transposed storage: co is the contiguous index
    for (int e = tid; e < Co * K; e += 256) {
      const int k = e / Co, co = e - k * Co;
      s_w[co * pitch + k] = W[(size_t)co * a.sWco + (size_t)k * a.sWk];
    }
  }
  __syncthreads();
  const int kh = lane >> 5;
  const float* wrow = s_w + (lane & 31) * pitch + kh;

  for (int tile = 0; tile < CONV_TILES; ++tile) {
    const int col = (blockIdx.x * CONV_TILES + tile) * 128 + wave * 32 + (lane & 31);
    if ((blockIdx.x * CONV_TILES + tile) * 128 >= a.N) break;   // workgroup-uniform
    const bool live = col < a.N;
    const int colc = live ? col : a.N - 1;
    const float* X = a.X + (size_t)b * a.sXb + colc;

    f32x16 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    constexpr int U = 32;  // k-steps whose loads are in flight together (all of K = 64)
    for (int s0 = 0; s0 < K / 2; s0 += U) {
      float xr[U];
#pragma unroll
      for (int u = 0; u < U; ++u) xr[u] = X[(size_t)(2 * (s0 + u) + kh) * a.ldX];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = 2 * (s0 + u);
#pragma unroll
        for (int t = 0; t < CT; ++t) acc[t] = mfma32(wrow[t * 32 * pitch + k], xr[u], acc[t]);
      }
    }

    if (!live) continue;
    float* Y = a.Y + (size_t)b * a.sYb + col;
    const float* Z = a.Z ? a.Z + (size_t)b * a.sZb + col : nullptr;
#pragma unroll
    for (int t = 0; t < CT; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = t * 32 + mfma_row(r, lane);
        float v = acc[t][r];
        if (a.bias) v += a.bias[co];
        if (a.relu) v = fmaxf(v, 0.f);
        if (a.accumulate) v += Y[(size_t)co * a.ldY];
        if (Z) v = Z[(size_t)co * a.ldZ] > 0.f ? v : 0.f;  // gate AFTER accumulation (sum of branches, then relu')
        Y[(size_t)co * a.ldY] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// fc: Y[m][o] = epi(sum_k X[m][k] W[o][k] + bias[o]).  Tile = 32 rows x 32 outputs per workgroup; the
// S wavefronts of the workgroup split K and are summed through LDS.  Both operands are k-contiguous,
// so each lane pulls 4 consecutive k with one 16-byte load straight into MFMA operand registers
// (k is consumed in the permuted order 8j + 4*(lane>>5) + i, identical for A and B).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void load4(const float* p, int k, int kend, bool vec, float v[4]) {
  if (vec && k + 3 < kend) {
    const float4 t = *reinterpret_cast<const float4*>(p + k);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (k + i < kend) ? p[k + i] : 0.f;
  }
}

template <int S>
__global__ __launch_bounds__(64 * S) void fc_kernel(FcArgs a) {
  __shared__ float s_red[S > 1 ? (S - 1) * 16 * 64 : 1];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int o0 = blockIdx.x * 32, m0 = blockIdx.y * 32;
  const int r = lane & 31, h = lane >> 5;
  const int m = min(m0 + r, a.M - 1), o = min(o0 + r, a.Nout - 1);
  const int bz = blockIdx.z;
  const float* xp = a.X + (size_t)bz * a.sXb + (size_t)m * a.ldX;
  const float* wp = a.W + (size_t)bz * a.sWb + (size_t)o * a.ldW;
  const bool xvec = ((a.ldX | a.sXb) & 3) == 0 && ((uintptr_t)a.X & 15) == 0;
  const bool wvec = ((a.ldW | a.sWb) & 3) == 0 && ((uintptr_t)a.W & 15) == 0;
  // this wave's K range, in units of 8
  const int k8 = (a.K + 7) / 8;
  const int per = (k8 + S - 1) / S;
  const int kb = wave * per * 8, ke = min(a.K, (wave + 1) * per * 8);

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  constexpr int U = 4;
  for (int k0 = kb; k0 < ke; k0 += 8 * U) {
    float xa[U][4], wb[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + 8 * u + 4 * h;
      load4(xp, k, ke, xvec, xa[u]);
      load4(wp, k, ke, wvec, wb[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = mfma32(xa[u][i], wb[u][i], acc);
  }
  if (S > 1) {
    if (wave > 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s_red[((wave - 1) * 16 + i) * 64 + lane] = acc[i];
    }
    __syncthreads();
    if (wave > 0) return;
#pragma unroll
    for (int w = 0; w < S - 1; ++w)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] += s_red[(w * 16 + i) * 64 + lane];
  }
  const int oc = o0 + r;
  if (oc >= a.Nout) return;
  const float bias = a.bias ? a.bias[oc] : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int mr = m0 + mfma_row(i, lane);
    if (mr < a.M) {
      float v = acc[i] + bias;
      if (a.relu) v = fmaxf(v, 0.f);
      if (a.Z) v = a.Z[(size_t)mr * a.ldZ + oc] > 0.f ? v : 0.f;
      a.Y[(size_t)bz * a.sYb + (size_t)mr * a.ldY + oc] = v;
    }
  }
}

}  // namespace

int launch_conv_cm(const ConvArgs& a, hipStream_t s) {
  if (a.K % 32 != 0 || (a.Co != 64 && a.Co != 128)) return GEOA3_ENOSUPPORT;
  const size_t lds = (size_t)a.Co * (a.K + 1) * sizeof(float);
  dim3 grid((a.N + 128 * CONV_TILES - 1) / (128 * CONV_TILES), a.B);
  if (a.Co == 64)
    hipLaunchKernelGGL(conv_cm_kernel<2>, grid, dim3(256), lds, s, a);
  else
    hipLaunchKernelGGL(conv_cm_kernel<4>, grid, dim3(256), lds, s, a);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

int launch_fc(const FcArgs& a, hipStream_t s) {
  dim3 grid((a.Nout + 31) / 32, (a.M + 31) / 32, a.batch > 1 ? a.batch : 1);
  if (a.ksplit == 8 || (a.ksplit == 0 && a.K >= 2048))
    hipLaunchKernelGGL(fc_kernel<8>, grid, dim3(512), 0, s, a);
  else if (a.K >= 256)
    hipLaunchKernelGGL(fc_kernel<4>, grid, dim3(256), 0, s, a);
  else
    hipLaunchKernelGGL(fc_kernel<1>, grid, dim3(64), 0, s, a);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
