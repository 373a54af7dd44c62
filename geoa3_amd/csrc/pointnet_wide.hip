// The three 1024-wide PointNet layers fused with their max-over-points (gfx950, fp32 MFMA):
// transform_net.conv3+bn3+relu+max (Model/PointNet.py:81-82) and conv5+bn5+relu+max (:146-147; conv5 is
// a kernel-3, pad-1 convolution over the POINT INDEX, :110).  [B,1024,N] is never written: a workgroup
// owns (instance, 128 output channels), walks all point tiles, keeps a running (max, arg-max) per
// accumulator element, and reduces it across lanes at the end.  relu and the folded BN bias commute
// with the max, so they are applied once to the 1024 maxima.
// The backward is sparse -- only the arg-max column of every channel carries gradient -- and is a
// deterministic gather-by-owner accumulation in LDS (no atomics).
#include "pointnet_kernels.h"
#include "profile.h"

namespace {

constexpr int WM_CO = 128;     // output channels per workgroup
constexpr int WM_COLS = 128;   // points per tile
constexpr int WM_THREADS = 512;
constexpr int WM_CI = 128;     // input channels of every wide layer
constexpr int WM_HALO = 4;     // left halo (float4 aligned); right halo is 4 as well

template <int TAPS>
struct WideCfg {
  static constexpr int CI_CHUNK = TAPS == 1 ? 32 : 16;     // input channels staged per pass
  static constexpr int KC = CI_CHUNK * TAPS;               // k extent of one pass
  static constexpr int WPITCH = KC + 1;                    // LDS pitch of the weight chunk
  static constexpr int XPITCH = WM_COLS + 2 * WM_HALO;     // LDS pitch of the activation chunk
  static constexpr int LDS_FLOATS = WM_CO * WPITCH + CI_CHUNK * XPITCH;
};

template <int TAPS>
__global__ __launch_bounds__(WM_THREADS) void wide_max_kernel(WideArgs a) {
  using Cfg = WideCfg<TAPS>;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_w = smem;                              // [128 co][KC+1]
  float* s_x = smem + WM_CO * Cfg::WPITCH;        // [CI_CHUNK][XPITCH]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y, co0 = blockIdx.x * WM_CO;
  const int wco = (wave & 3) * 32;                // this wave's 32 channels inside the tile
  const int wcol = (wave >> 2) * 64;              // and its 64 columns
  const int N = a.N, kh = lane >> 5, l31 = lane & 31;
  const float* X = a.X + (size_t)b * a.sXb;
  const int KTOT = TAPS * WM_CI;

  float rmax[16];
  int rarg[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    rmax[r] = -__builtin_inff();
    rarg[r] = 0;
  }

  for (int n0 = 0; n0 < N; n0 += WM_COLS) {
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    for (int ci0 = 0; ci0 < WM_CI; ci0 += Cfg::CI_CHUNK) {
      __syncthreads();  // previous pass has finished reading LDS
      // weight chunk: s_w[co][tap*CI_CHUNK + c] = W[co0+co][tap*128 + ci0 + c]
      for (int e = tid; e < WM_CO * Cfg::KC / 4; e += WM_THREADS) {
        const int q = e % (Cfg::KC / 4), co = e / (Cfg::KC / 4);
        const int kk = q * 4, tap = kk / Cfg::CI_CHUNK, c = kk - tap * Cfg::CI_CHUNK;
        const float4 w = *reinterpret_cast<const float4*>(a.W + (size_t)(co0 + co) * KTOT + tap * WM_CI + ci0 + c);
        float* d = s_w + co * Cfg::WPITCH + kk;
        d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
      }
      // activation chunk with halo: s_x[c][j] = X[ci0+c][n0 - HALO + j], zero outside [0,N)
      for (int e = tid; e < Cfg::CI_CHUNK * (Cfg::XPITCH / 4); e += WM_THREADS) {
        const int q = e % (Cfg::XPITCH / 4), c = e / (Cfg::XPITCH / 4);
        const int n = n0 - WM_HALO + q * 4;
        const float* src = X + (size_t)(ci0 + c) * a.ldX;
        float4 v;
        if (n >= 0 && n + 3 < N && (a.ldX & 3) == 0) {
          v = *reinterpret_cast<const float4*>(src + n);
        } else {
          v.x = (n >= 0 && n < N) ? src[n] : 0.f;
          v.y = (n + 1 >= 0 && n + 1 < N) ? src[n + 1] : 0.f;
          v.z = (n + 2 >= 0 && n + 2 < N) ? src[n + 2] : 0.f;
          v.w = (n + 3 >= 0 && n + 3 < N) ? src[n + 3] : 0.f;
        }
        *reinterpret_cast<float4*>(s_x + c * Cfg::XPITCH + q * 4) = v;
      }
      __syncthreads();
      const float* wp = s_w + (wco + l31) * Cfg::WPITCH + kh;
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        // column of the B operand inside the haloed row: HALO + col + tap - TAPS/2
        const float* xp = s_x + kh * Cfg::XPITCH + WM_HALO + wcol + l31 + tap - TAPS / 2;
#pragma unroll 8
        for (int c = 0; c < Cfg::CI_CHUNK; c += 2) {
          const float av = wp[tap * Cfg::CI_CHUNK + c];
          const float b0 = xp[c * Cfg::XPITCH];
          const float b1 = xp[c * Cfg::XPITCH + 32];
          acc[0] = mfma32(av, b0, acc[0]);
          acc[1] = mfma32(av, b1, acc[1]);
        }
      }
    }
    // fold this tile into the running maximum (strict >: the lowest point index wins a tie)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int col = n0 + wcol + t * 32 + l31;
      const bool ok = col < N;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool gt = ok && acc[t][r] > rmax[r];
        rmax[r] = gt ? acc[t][r] : rmax[r];
        rarg[r] = gt ? col : rarg[r];
      }
    }
  }

  // reduce over the 32 lanes that share (reg, lane>>5), i.e. over this wave's columns
#pragma unroll
  for (int r = 0; r < 16; ++r) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) {
      const float v2 = __shfl_xor(rmax[r], o, 64);
      const int i2 = __shfl_xor(rarg[r], o, 64);
      const bool take = v2 > rmax[r] || (v2 == rmax[r] && i2 < rarg[r]);
      rmax[r] = take ? v2 : rmax[r];
      rarg[r] = take ? i2 : rarg[r];
    }
  }
  // combine the two column-halves (waves w and w+4) through LDS, then bias + relu
  __syncthreads();
  float* s_v = smem;
  int* s_i = reinterpret_cast<int*>(smem + WM_CO);
  if (wave >= 4 && l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = wco + mfma_row(r, lane);
      s_v[co] = rmax[r];
      s_i[co] = rarg[r];
    }
  }
  __syncthreads();
  if (wave < 4 && l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = wco + mfma_row(r, lane);
      float v = rmax[r];
      int i = rarg[r];
      const float v2 = s_v[co];
      const int i2 = s_i[co];
      if (v2 > v || (v2 == v && i2 < i)) {
        v = v2;
        i = i2;
      }
      const size_t o = (size_t)b * a.Co + co0 + co;
      a.out[o] = fmaxf(v + a.bias[co0 + co], 0.f);
      a.arg[o] = i;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Sparse backward.  One workgroup per (instance, 128-point tile), split in two 64-column halves of 128
// threads; thread (ci, half) owns row ci of its half, so no two threads ever touch the same accumulator
// and the summation order is fixed (deterministic, no atomics).
// Per block of WB_BLOCK output channels: the first wave of each half compacts the (channel, tap) pairs
// whose arg-max column falls into its half into an LDS hit list (ballot + popcount, in (co, tap) order);
// then the 128 threads of the half walk ONLY the hits, 16 independent weight-row loads in flight.
// ------------------------------------------------------------------------------------------
constexpr int WB_COLS = 128;
constexpr int WB_BLOCK = 512;   // output channels per compaction round
constexpr int WB_BATCH = 16;

template <int TAPS>
__global__ __launch_bounds__(256) void wide_max_bwd_kernel(WideBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* s_acc = smem;                                                        // [128 ci][WB_COLS + 1]
  int* s_hit = reinterpret_cast<int*>(smem + WM_CI * (WB_COLS + 1));          // [2][WB_BLOCK*TAPS]
  int* s_cnt = s_hit + 2 * WB_BLOCK * TAPS;                                   // [2]
  const int tid = threadIdx.x, b = blockIdx.y, m0 = blockIdx.x * WB_COLS;
  const int ci = tid & 127, half = tid >> 7, lane = tid & 63;
  const bool builder = (tid & 127) < 64;                                      // first wave of each half
  const float* gb = a.g + (size_t)b * a.Co;
  const int* argb = a.arg + (size_t)b * a.Co;
  int* hits = s_hit + half * WB_BLOCK * TAPS;
  float* row = s_acc + ci * (WB_COLS + 1);
  for (int j = half * 64; j < half * 64 + 64; ++j) row[j] = 0.f;
  const int lo = m0 + half * 64, hi = lo + 64;

  for (int cb = 0; cb < a.Co; cb += WB_BLOCK) {
    __syncthreads();  // the previous round's list has been consumed
    if (builder) {
      int cnt = 0;
      for (int c0 = cb; c0 < min(cb + WB_BLOCK, a.Co); c0 += 64) {
        const int co = c0 + lane;
        const bool in = co < a.Co;
        const float g = in ? gb[co] : 0.f;
        const int base = (in ? argb[co] : 0) - TAPS / 2;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
          const int m = base + tap;
          const bool hit = g != 0.f && m >= lo && m < hi;
          const unsigned long long mask = __ballot(hit);
          if (hit) hits[cnt + __popcll(mask & ((1ull << lane) - 1ull))] = (co * TAPS + tap) | ((m - m0) << 16);
          cnt += __popcll(mask);
        }
      }
      if (lane == 0) s_cnt[half] = cnt;
    }
    __syncthreads();
    const int cnt = s_cnt[half];
    for (int h0 = 0; h0 < cnt; h0 += WB_BATCH) {
      float w[WB_BATCH], gg[WB_BATCH];
      int mm[WB_BATCH];
#pragma unroll
      for (int u = 0; u < WB_BATCH; ++u) {
        const bool ok = h0 + u < cnt;
        const int e = hits[ok ? h0 + u : cnt - 1];
        const int kt = e & 0xffff;                      // co*TAPS + tap: row of the [Co*TAPS][128] weight view
        w[u] = a.W[(size_t)kt * WM_CI + ci];
        gg[u] = ok ? gb[kt / TAPS] : 0.f;
        mm[u] = e >> 16;
      }
#pragma unroll
      for (int u = 0; u < WB_BATCH; ++u) row[mm[u]] += w[u] * gg[u];
    }
  }
  __syncthreads();
  // write out with the relu gate of the layer input; consecutive threads -> consecutive points
  float* dX = a.dX + (size_t)b * a.sXb;
  const float* Z = a.Z + (size_t)b * a.sZb;
  for (int e = tid; e < WM_CI * WB_COLS; e += 256) {
    const int c = e / WB_COLS, j = e - c * WB_COLS;
    const int m = m0 + j;
    if (m < a.N) {
      const float v = s_acc[c * (WB_COLS + 1) + j];
      dX[(size_t)c * a.ldX + m] = Z[(size_t)c * a.ldZ + m] > 0.f ? v : 0.f;
    }
  }
}

}  // namespace

int launch_wide_max(const WideArgs& a, hipStream_t s) {
  if (a.Co % WM_CO != 0 || (a.taps != 1 && a.taps != 3)) return GEOA3_ENOSUPPORT;
  dim3 grid(a.Co / WM_CO, a.B);
  const int tag = a.taps == 3 ? GEOA3_PROF_CONV5 : GEOA3_PROF_TNETWIDE;
  geoa3_prof_begin(tag, s);
  if (a.taps == 1) {
    const size_t lds = WideCfg<1>::LDS_FLOATS * sizeof(float);
    hipLaunchKernelGGL(wide_max_kernel<1>, grid, dim3(WM_THREADS), lds, s, a);
  } else {
    const size_t lds = WideCfg<3>::LDS_FLOATS * sizeof(float);
    hipLaunchKernelGGL(wide_max_kernel<3>, grid, dim3(WM_THREADS), lds, s, a);
  }
  geoa3_prof_end(tag, s);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

int launch_wide_max_bwd(const WideBwdArgs& a, hipStream_t s) {
  if (a.taps != 1 && a.taps != 3) return GEOA3_ENOSUPPORT;
  dim3 grid((a.N + WB_COLS - 1) / WB_COLS, a.B);
  const size_t lds = ((size_t)WM_CI * (WB_COLS + 1) + 2 * (size_t)WB_BLOCK * a.taps + 4) * sizeof(float);
  if (a.taps == 1) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wide_max_bwd_kernel<1>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(wide_max_bwd_kernel<1>, grid, dim3(256), lds, s, a);
  } else {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wide_max_bwd_kernel<3>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(wide_max_bwd_kernel<3>, grid, dim3(256), lds, s, a);
  }
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
