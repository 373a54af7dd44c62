// PointNet++ SSG, second set-abstraction level, backward of the two hidden layers in ONE kernel (gfx950).
// Reference: pointnet2_modules.py:57-70 (shared MLP (128+3) -> 128 -> 128 -> 256, max over the 64 samples of a ball),
// PointNetPP_ssg.py:68-76; the autograd of F.max_pool2d / Conv2d / ReLU behind it.
//
// With z = W2 a1 max-pooled over a centre's samples, the pooled gradient reaches exactly ONE sample per channel:
//   d a1[k][s] = [a1[k][s] > 0] * sum_{ch : arg[ch] == s} g[ch] W2[ch][k]                      (256 x 128 MACs per centre)
//   d a0[i][s] = [a0[i][s] > 0] * sum_k W1[k][i] d a1[k][s]                                     (dense 128 x 128 x 64)
// The unfused form ran the first line as a dense K = 256 convolution over a one-hot operand (1.04 ms at B = 250) and
// wrote / re-read the [B,128,8192] tensor d a1 and both activations as relu gates (0.58 ms for the second layer).
// Here a workgroup takes two centres at a time:
//   phase 1 (VALU, 2 wavefronts per centre, lane = k): the centre's 256 (channel, sample) pairs arrive SORTED by sample
//     (sa2_sort_kernel: stable counting sort, so the channels of a sample stay in ascending order); a lane accumulates
//     g * W2[ch][k] in a register over the channels of one sample -- W2 rows prefetched 32 entries ahead, they are the
//     only memory traffic -- and writes the gated sum once per sample into a sample-major fp32 tile in LDS;
//   phase 2 (matrix core, wave = 64 output rows x one centre): d a0 = W1^T tile with split-fp16 operands (the arithmetic
//     of pointnet_conv_split.hip; W1^T scaled + split once per workgroup into LDS, the tile scaled by a power of two from
//     its own maximum and split as it is read), gated by bits and written as 256-byte rows.
// Both relu gates come as bit masks ([B * centres][128] 64-bit words, bit s = sample s; ConvArgs::Ymask layout): d a1
// never exists in memory and neither activation is read.  Deterministic: fixed summation orders, no atomics.
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
__device__ __forceinline__ int s2_rl(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float s2_rlf(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }

// One wavefront per centre: entries (g, channel) sorted by the channel's arg-max sample, ascending channel inside a
// sample.  ent_c = channel | sample << 16 | (last entry of its sample) << 31.
__global__ __launch_bounds__(256) void sa2_sort_kernel(const float* __restrict__ gz, const int32_t* __restrict__ argt,
                                                       float* __restrict__ ent_g, int32_t* __restrict__ ent_c, long centres) {
  const long c = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= centres) return;
  int a[4], rank[4];
  float g[4];
  bool last[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    a[q] = argt[c * S2_C + q * 64 + lane] & 63;
    g[q] = gz[c * S2_C + q * 64 + lane];
    rank[q] = q * 64 + lane;
    last[q] = true;
  }
  const unsigned long long below = (1ull << lane) - 1ull;
  int base = 0;
  for (int s = 0; s < 64; ++s) {
    unsigned long long m[4];
    int pre[4], n = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      m[q] = __ballot(a[q] == s);
      pre[q] = n;
      n += (int)__builtin_popcountll(m[q]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (a[q] == s) {
        rank[q] = base + pre[q] + (int)__builtin_popcountll(m[q] & below);
        last[q] = rank[q] == base + n - 1;
      }
    base += n;
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    ent_c[c * S2_C + rank[q]] = (q * 64 + lane) | (a[q] << 16) | (last[q] ? (int)0x80000000 : 0);
    ent_g[c * S2_C + rank[q]] = g[q];
  }
}

struct Sa2BwdArgs {
  const float* ent_g;              // [centres][256]
  const int32_t* ent_c;            // [centres][256]
  const float* W2;                 // [256][128]
  const _Float16* w1img;           // sa2_prep_kernel: W1^T as A fragments, [hf][c][t][hi / lo][lane][8]
  const float* w1un;               // [1]: 1 / the image's power-of-two scale
  const unsigned long long* m1;    // [centres][128]: a1 > 0
  const unsigned long long* m0;    // [centres][128]: a0 > 0
  float* da0;                      // [B][128][M * 64]
  int B, M;                        // M centres per instance (even)
};

// W1^T [128 i][128 k] -> power-of-two scale from its maximum, split, stored in the order the matrix-core loop reads it:
// fragment (hf, c, t, piece) = rows 64 hf + 32 t .. + 31, k = 16 c .. + 15, one 16-byte element per lane (row r = lane & 31,
// k half lane >> 5).  One workgroup; 64 KB, read by every wave of sa2_bwd_kernel from L2.
__global__ __launch_bounds__(256) void sa2_prep_kernel(const float* __restrict__ W1t, _Float16* __restrict__ img,
                                                       float* __restrict__ un) {
  __shared__ float s_m[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float wv[64];
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    wv[i] = W1t[tid + 256 * i];
    m = fmaxf(m, __builtin_fabsf(wv[i]));
  }
  m = wave_max(m);
  if (lane == 0) s_m[wave] = m;
  __syncthreads();
  const unsigned Ew = s2_exp(fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])));
  const float sw = s2_scale(Ew);
  if (tid == 0) un[0] = s2_unscale(Ew);
#pragma unroll
  for (int i = 0; i < 64; ++i) {
    const int e = tid + 256 * i, row = e >> 7, k = e & 127;
    const int hf = row >> 6, t = (row >> 5) & 1, r = row & 31, c = k >> 4, h = (k >> 3) & 1, j = k & 7;
    const int frag = (hf * 8 + c) * 2 + t;
    const float v = wv[i] * sw;
    const _Float16 hi = (_Float16)v;
    img[((size_t)(frag * 2 + 0) * 64 + h * 32 + r) * 8 + j] = hi;
    img[((size_t)(frag * 2 + 1) * 64 + h * 32 + r) * 8 + j] = (_Float16)(v - (float)hi);
  }
}

constexpr int sa2_bwd_lds() { return 2 * 64 * S2_PT * 4 + 16 * 4; }

template <int MODE>   // 0 = shipped; 1 / 2 / 3: without phase 1 / phase 2 / the stores (tools/ub/sa2_ub.hip)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void sa2_bwd_kernel(Sa2BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s2_sm[];
  float* s_tile = reinterpret_cast<float*>(s2_sm);   // [2 centres][64 samples][S2_PT]
  float* s_red = s_tile + 2 * 64 * S2_PT;            // [4] tile maxima of the waves
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float unW = a.w1un[0];
  const int ci = wave >> 1, hf = wave & 1;
  float* tile = s_tile + ci * 64 * S2_PT;
  const long pairs = (long)a.B * a.M / 2;
  const int ldY = a.M * 64;
  const int k = 64 * hf + lane;

  // Two workgroups per CU (68 KB of LDS each): one is on the VALU (phase 1) while the other is on the matrix core.
  // Everything a pair needs from memory is requested ahead: its entries and gate words one iteration early, its first
  // S2_PF rows of W2 while the pair before it is in phase 2, the W1^T fragments one k-step ahead.
  struct PairRegs {
    int vc[4];
    float vg[4];
    unsigned long long mk, gm;   // gates: a1 of row k (phase 1), a0 of row 64 hf + lane (epilogue)
  };
  auto load_pair = [&](long p, PairRegs& r) {
    const long centre = 2 * p + ci;
    r.mk = a.m1[centre * S2_K + k];
    r.gm = a.m0[centre * S2_K + k];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      r.vc[q] = a.ent_c[centre * S2_C + q * 64 + lane];
      r.vg[q] = a.ent_g[centre * S2_C + q * 64 + lane];
    }
  };
  auto w2row = [&](int cw) { return (a.W2 + (size_t)(cw & 0xffff) * S2_K)[k]; };   // uniform row + lane offset
  const half8* img = reinterpret_cast<const half8*>(a.w1img) + (size_t)hf * 8 * 2 * 2 * 64 + lane;
  auto load_a = [&](int c, half8 (&f)[4]) {   // fragments (t, piece) of k-step c
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = img[(size_t)(c * 4 + i) * 64];
  };

  PairRegs cur, nxt;
  float wv[S2_PF];
  int cwr[S2_PF];   // the entries whose W2 rows are in flight (uniform)
  if ((long)blockIdx.x < pairs) {
    load_pair(blockIdx.x, cur);
#pragma unroll
    for (int u = 0; u < S2_PF; ++u) {
      cwr[u] = s2_rl(cur.vc[0], u);
      wv[u] = w2row(cwr[u]);
    }
  }

  for (long p = blockIdx.x; p < pairs; p += gridDim.x) {
    const long centre = 2 * p + ci;
    load_pair(p + gridDim.x < pairs ? p + gridDim.x : p, nxt);
    half8 af[2][4];
    load_a(0, af[0]);
    // ---- phase 1: this wave's 64 k of the centre's tile
    if (MODE != 1) {
#pragma unroll 8
      for (int s = 0; s < 64; ++s) tile[s * S2_PT + k] = 0.f;     // samples no channel points at
      float acc = 0.f, mx = 0.f;
#pragma unroll
      for (int e = 0; e < S2_C; ++e) {
        const int u = e % S2_PF;
        const int cw = cwr[u];
        const float g = s2_rlf(cur.vg[e >> 6], e & 63);
        acc = __builtin_fmaf(g, wv[u], acc);
        if (e + S2_PF < S2_C) {
          cwr[u] = s2_rl(cur.vc[(e + S2_PF) >> 6], (e + S2_PF) & 63);
          wv[u] = w2row(cwr[u]);
        }
        if (__builtin_expect(cw < 0, 0)) {   // wave-uniform: the sample's last channel
          const int col = (cw >> 16) & 63;
          const float val = (cur.mk >> col) & 1ull ? acc : 0.f;
          tile[col * S2_PT + k] = val;
          mx = fmaxf(mx, __builtin_fabsf(val));
          acc = 0.f;
        }
      }
      mx = wave_max(mx);
      if (lane == 0) s_red[wave] = mx;
    }
    // the next pair's first rows of W2: in flight across phase 2
#pragma unroll
    for (int u = 0; u < S2_PF; ++u) {
      cwr[u] = s2_rl(nxt.vc[0], u);
      wv[u] = w2row(cwr[u]);
    }
    __syncthreads();
    // ---- phase 2: rows 64 hf .. 64 hf + 63 of d a0 for the centre's 64 samples
    if (MODE != 2) {
      const unsigned Ex = s2_exp(fmaxf(s_red[2 * ci], s_red[2 * ci + 1]));
      const float sx = s2_scale(Ex);
      f32x16 acc[2][2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[cb][t][r] = 0.f;
      const float* brow = tile + (lane & 31) * S2_PT + (lane >> 5) * 8;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c + 1 < 8) load_a(c + 1, af[(c + 1) & 1]);
        half8 xh[2], xl[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const float4 b0 = *reinterpret_cast<const float4*>(brow + cb * 32 * S2_PT + c * 16);
          const float4 b1 = *reinterpret_cast<const float4*>(brow + cb * 32 * S2_PT + c * 16 + 4);
          const float x[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float v = x[j] * sx;
            const _Float16 h = (_Float16)v;
            xh[cb][j] = h;
            xl[cb][j] = (_Float16)(v - (float)h);
          }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const half8 wh = af[c & 1][2 * t], wl = af[c & 1][2 * t + 1];
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[cb], acc[cb][t], 0, 0, 0);
            acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[cb], acc[cb][t], 0, 0, 0);
            acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[cb], acc[cb][t], 0, 0, 0);
          }
        }
      }
      const float unscale = s2_unscale(Ex) * unW;
      const int b = (int)(centre / a.M), m = (int)(centre - (long)b * a.M);
      float* Y = a.da0 + (size_t)b * S2_K * ldY + (size_t)m * 64 + lane;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = acc[0][t][4 * g + i];
            v[4 + i] = acc[1][t][4 * g + i];
            s2_swap32(v[i], v[4 + i]);    // v[i]: row base + i, v[4 + i]: row base + 4 + i, lane = sample
          }
          const int row0 = 64 * hf + 32 * t + 8 * g;
#pragma unroll
          for (int i = 0; i < 8; ++i) {   // the row's gate word sits in lane 32 t + 8 g + i
            const unsigned glo = (unsigned)s2_rl((int)(unsigned)cur.gm, 32 * t + 8 * g + i);
            const unsigned ghi = (unsigned)s2_rl((int)(unsigned)(cur.gm >> 32), 32 * t + 8 * g + i);
            const bool on = ((lane < 32 ? glo >> lane : ghi >> (lane - 32)) & 1u) != 0u;
            if (MODE != 3) Y[(size_t)(row0 + i) * ldY] = on ? v[i] * unscale : 0.f;
          }
        }
    }
    __syncthreads();   // the tiles are rewritten by the next pair
    cur = nxt;
  }
}

}  // namespace

int launch_sa2_sort(const float* gz, const int32_t* argt, float* ent_g, int32_t* ent_c, long centres, hipStream_t s) {
  hipLaunchKernelGGL(sa2_sort_kernel, dim3((unsigned)((centres + 3) / 4)), dim3(256), 0, s, gz, argt, ent_g, ent_c, centres);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

// scratch: 64 KB + 256 B (the W1^T fragment image and its scale)
int launch_sa2_bwd(const float* ent_g, const int32_t* ent_c, const float* W2, const float* W1t, const unsigned long long* m1,
                   const unsigned long long* m0, float* da0, int B, int M, void* scratch, hipStream_t s) {
  if (M % 2 != 0) return GEOA3_ENOSUPPORT;
  _Float16* img = static_cast<_Float16*>(scratch);
  float* un = reinterpret_cast<float*>(static_cast<char*>(scratch) + 65536);
  hipLaunchKernelGGL(sa2_prep_kernel, dim3(1), dim3(256), 0, s, W1t, img, un);
  Sa2BwdArgs a{ent_g, ent_c, W2, img, un, m1, m0, da0, B, M};
  const long pairs = (long)B * M / 2;
  const int lds = sa2_bwd_lds();
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa2_bwd_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const unsigned grid = (unsigned)(pairs < 512 ? pairs : 512);   // two workgroups per CU, persistent
  hipLaunchKernelGGL(sa2_bwd_kernel<0>, dim3(grid), dim3(256), lds, s, a);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
