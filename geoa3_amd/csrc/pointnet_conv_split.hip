// conv_cm64 of pointnet_gemm.hip on the f16 matrix pipe with split fp32 operands (the arithmetic of
// pointnet_wide_split.hip: v = hi + lo, a*w = a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, fp32 accumulation).
//
// Why: against a plain copy of the same bytes (5.1 TB/s) the fp32-MFMA form runs at 2.4-3.9 TB/s -- its 64-cycle
// fp32 MFMAs (14 / 27 us of matrix time for the 64- / 128-wide layers at peak) do not hide behind the memory phases
// of a kernel whose workgroups are all resident at once.  Three 32-cycle f16 MFMAs per 16 k replace eight fp32 ones.
//
// Same structure as conv_cm64_kernel: one wavefront = 64 columns (lane = point) x 64 output channels, every global
// access one 256-byte row, K in chunks of 16 rows double-buffered in registers.  Differences:
//   * the 64 x K weight block is scaled by a power of two from its own maximum, split, and kept in LDS as two fp16
//     images [64][K] (rows padded by 16 B: the A operand of a k-step is one conflict-free ds_read_b128);
//   * a chunk of 16 rows is one k-step: v_permlane32_swap of rows (j, 8 + j) yields, for both 32-column blocks, the
//     B-operand element j of lanes (column, k half) -- the same swap that builds the fp32 operands;
//   * the activation scale is a RUNNING per-wave power of two: when a later chunk's maximum exceeds the range of the
//     current scale, the accumulators are rescaled (exact) and the scale shrinks; a chunk of smaller values keeps
//     the scale (its error is relative to the wave's largest values, as in the wide kernels).  Deterministic, and a
//     function of the wave's own columns only: rows of a batched run stay bit-identical to batch-1 runs.
#include "pointnet_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void cs_swap32(float& a, float& b) {   // a[32..63] <-> b[0..31]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ float cs_first_layer(const float4 w, float p0, float p1, float p2) {
  return w.x * p0 + w.y * p1 + w.z * p2 + w.w;   // the expression of pointnet_gemm.hip first_layer (same bits)
}
// biased exponent E of m clamped to [14, 254]; scale 2^(140 - E) puts m into [2^13, 2^14); unscale 2^(E - 140)
__device__ __forceinline__ unsigned cs_exp(float m) {
  const unsigned E = (__float_as_uint(m) >> 23) & 0xffu;
  return E < 14u ? 14u : (E > 254u ? 254u : E);
}
__device__ __forceinline__ float cs_scale(unsigned E) { return __uint_as_float((267u - E) << 23); }
__device__ __forceinline__ float cs_unscale(unsigned E) { return __uint_as_float((E - 13u) << 23); }

template <int NCH, bool FIRST, bool GFIRST, bool BWD3, int PN2 = 0, bool IMG = false, bool DEEP = false>  // K = 16 * NCH; BWD3 needs GFIRST;
                                                                  // PN2: 1 = pooled output, 2 = one-hot input,
                                                                  // 3 = output pooled over the instance's 128 columns;
                                                                  // IMG: weights from a pre-split fragment image;
                                                                  // DEEP: four chunks in flight per wave, two waves per
                                                                  // SIMD (launches with at most two workgroups per CU)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NCH <= 8 && !DEEP ? 4 : 2, NCH <= 8 && !DEEP ? 4 : 2))) void conv_cm64s_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cs_smem[];
  constexpr int CH = 16, K = CH * NCH;
  constexpr int PITCH = K * 2 + 16;                      // bytes per weight row and piece
  unsigned char* s_wh = cs_smem;                         // [64][PITCH] hi
  unsigned char* s_wl = cs_smem + 64 * PITCH;            // [64][PITCH] lo
  float4* s_w1 = reinterpret_cast<float4*>(cs_smem + (IMG ? 0 : 2 * 64 * PITCH));   // [64] (w1 row, b1) of the folded first layer
  float* s_part = reinterpret_cast<float*>(s_w1 + 64);   // [4 waves][9]
  float* s_red = s_part + 40;                            // [4]
  // grid (row blocks, column blocks, instances); the row blocks of one column tile read the same input
  // a.pack2 (N <= 128, shared weights): two instances per workgroup, waves 0-1 the columns of instance 2 z, waves 2-3
  // those of 2 z + 1 -- a [B][C][128] tensor (PointNet++ level 3) keeps all four waves busy
  // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so neighbours in dispatch order do NOT share
  // an L2: the R row blocks of a column tile are therefore given the linear ids base + xcd + 8 rb -- one XCD, a few
  // dispatch slots apart -- instead of base + rb (speed only; any placement computes the same values).
  int rb, cblk, zblk;
  {
    const unsigned R = gridDim.x, tiles = gridDim.y * gridDim.z;
    const unsigned L = blockIdx.x + R * (blockIdx.y + gridDim.y * blockIdx.z);
    const unsigned grp = L / (8 * R), rem = L - grp * 8 * R;
    unsigned tile;
    if ((grp + 1) * 8 <= tiles) {
      tile = grp * 8 + (rem & 7);
      rb = (int)(rem >> 3);
    } else {   // the last, partial group of tiles: dispatch order
      tile = L / R;
      rb = (int)(L - tile * R);
    }
    zblk = (int)(tile / gridDim.y);
    cblk = (int)(tile - zblk * gridDim.y);
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bz = a.pack2 ? 2 * zblk + (wave >> 1) : zblk;
  const bool b_ok = bz < a.B;
  const int b = b_ok ? bz : a.B - 1;
  const int col = a.pack2 ? (wave & 1) * 64 + lane : cblk * 256 + wave * 64 + lane;
  const bool live = b_ok && col < a.N;
  const float* X = (FIRST || PN2 == 2) ? nullptr : a.X + (size_t)b * a.sXb + (live ? col : a.N - 1);
  // PN2: this wave's centre (its 64 columns are the centre's samples)
  const int centres = a.N >> 6, centre = cblk * 4 + wave;
  // one-hot input: oh_g / oh_arg are [B][centres][K] (k contiguous): lane l holds k = 4l .. 4l+3 of the wave's centre
  float4 ohg = make_float4(0.f, 0.f, 0.f, 0.f);
  int4 oha = make_int4(-1, -1, -1, -1);
  if (PN2 == 2 && centre < centres) {
    const size_t e = ((size_t)b * centres + centre) * (16 * NCH) + 4 * lane;
    ohg = *reinterpret_cast<const float4*>(a.oh_g + e);
    oha = *reinterpret_cast<const int4*>(a.oh_arg + e);
  }
  // Per-row operands of the epilogue (bias, gate words): ONE coalesced load per wave here -- in flight across the K loop,
  // read back with v_readlane -- instead of 64 uniform loads inside the epilogue, each a dependent round trip (measured:
  // a 64 -> 64 layer on 2 instances 9.2 us without bias, 16.9 us with).  Lane l holds row rb * 64 + l.
  const size_t mword = ((size_t)b * ((a.N + 63) >> 6) + (size_t)(cblk * 4 + wave)) * a.Co;   // bit masks [B][column block][row]
  const bool wave_live = b_ok && (a.pack2 ? (wave & 1) * 64 : cblk * 256 + wave * 64) < a.N;
  const float biasv = (PN2 == 0 && a.bias) ? a.bias[rb * 64 + lane] : 0.f;
  const unsigned long long zmv = (PN2 == 0 && a.Zmask && wave_live) ? a.Zmask[mword + rb * 64 + lane] : 0ull;
  float p0 = 0.f, p1 = 0.f, p2 = 0.f;                    // T^T x of this lane's point
  float x0 = 0.f, x1 = 0.f, x2 = 0.f;
  if (FIRST || GFIRST) {
    const float* xp = a.x3 + (size_t)b * 3 * a.N + (live ? col : a.N - 1);
    x0 = xp[0];
    x1 = xp[a.N];
    x2 = xp[2 * (size_t)a.N];
    p0 = x0;
    p1 = x1;
    p2 = x2;
    if (a.T3) {   // bmm(pc^T, T)^T : x'[c] = sum_d x[d] T[d][c]   (Model/PointNet.py:138)
      const float* t = a.T3 + (size_t)b * 9;
      p0 = x0 * t[0] + x1 * t[3] + x2 * t[6];
      p1 = x0 * t[1] + x1 * t[4] + x2 * t[7];
      p2 = x0 * t[2] + x1 * t[5] + x2 * t[8];
    }
    if (tid < 64) s_w1[tid] = make_float4(a.w1[3 * tid], a.w1[3 * tid + 1], a.w1[3 * tid + 2], a.b1[tid]);
  }
  // chunks in flight: NB - 1 ahead of the one in the matrix core (a chunk = 16 rows of 256 bytes per wave).  One ahead
  // leaves a launch with few workgroups (small shards: nothing else on the CU) waiting on NCH dependent round trips
  constexpr int NB = DEEP && !FIRST ? 4 : 2;
  float xb[NB][CH];
  auto load_rows = [&](int c, float (&d)[CH]) {
    if (PN2 == 2) {
#pragma unroll
      for (int u = 0; u < CH; ++u) {
        const int src = 4 * c + (u >> 2);     // the lane that holds k = 16 c + u (component u & 3)
        const float gsel = (u & 3) == 0 ? ohg.x : ((u & 3) == 1 ? ohg.y : ((u & 3) == 2 ? ohg.z : ohg.w));
        const int asel = (u & 3) == 0 ? oha.x : ((u & 3) == 1 ? oha.y : ((u & 3) == 2 ? oha.z : oha.w));
        const float gk = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(gsel), src));
        const int ak = __builtin_amdgcn_readlane(asel, src);
        d[u] = ak == lane ? gk : 0.f;
      }
    } else {
#pragma unroll
      for (int u = 0; u < CH; ++u) d[u] = X[(size_t)(CH * c + u) * a.ldX];
    }
  };
  if (!FIRST) {
#pragma unroll
    for (int i = 0; i + 1 < NB; ++i)
      if (i < NCH) load_rows(i, xb[i]);
  }

  // IMG: the 64 x K block is read as ready A fragments (hi / lo pieces, 16 bytes per lane) from an image in fragment
  // order, [tile = row >> 5][k >> 4][piece][lane][8] (geoa3_pn2ssg_pack_images): nothing to stage, no LDS for weights
  unsigned Ew = 14u;
  const half8* wimg = nullptr;
  if (IMG) {
    wimg = reinterpret_cast<const half8*>(a.Wimg) + ((size_t)(2 * rb) * a.img_kc + a.img_c0) * 2 * 64 + lane;
  } else {
    // ---- weights: 64 x K values, K / 4 per thread (element e = tid + 256 i: row e / K, k = e % K), maximum, split.
    // K <= 128: the values stay in registers between the two passes; K = 256: they are read again (L2)
    constexpr int WPT = 64 * K / 256;
    constexpr bool WKEEP = NCH <= 8;
    float wv[WKEEP ? WPT : 1];
    const float* W = a.W + (size_t)b * a.sWb + (size_t)rb * 64 * a.sWco;
    {
      float m = 0.f;
  #pragma unroll
      for (int i = 0; i < WPT; ++i) {
        const int e = tid + 256 * i;
        int co, k;
        if (a.sWk == 1) {     // rows are k-contiguous
          co = e / K;
          k = e - co * K;
        } else {              // transposed storage: co is the contiguous index
          k = e / 64;
          co = e - k * 64;
        }
        const float v = W[(size_t)co * a.sWco + (size_t)k * a.sWk];
        if (WKEEP) wv[i] = v;
        m = fmaxf(m, __builtin_fabsf(v));
      }
      m = wave_max(m);
      if (lane == 0) s_red[wave] = m;
    }
    __syncthreads();
    Ew = cs_exp(fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])));
    {
      const float sw = cs_scale(Ew);
  #pragma unroll
      for (int i = 0; i < WPT; ++i) {
        const int e = tid + 256 * i;
        int co, k;
        if (a.sWk == 1) {
          co = e / K;
          k = e - co * K;
        } else {
          k = e / 64;
          co = e - k * 64;
        }
        const float v = (WKEEP ? wv[i] : W[(size_t)co * a.sWco + (size_t)k * a.sWk]) * sw;
        const _Float16 h = (_Float16)v;
        *reinterpret_cast<_Float16*>(s_wh + co * PITCH + k * 2) = h;
        *reinterpret_cast<_Float16*>(s_wl + co * PITCH + k * 2) = (_Float16)(v - (float)h);
      }
    }
    __syncthreads();
  }
  const unsigned char* arow = s_wh + (lane & 31) * PITCH + (lane >> 5) * 16;   // A operand: row r, k = 8h + j

  f32x16 acc[2][2];   // [column block][row tile]
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;

  unsigned Ex = 14u;   // running exponent of the wave's activation scale
  constexpr int CU = NCH < 16 ? NCH : 16;   // chunks unrolled together (K = 512: two rounds of sixteen)
#pragma unroll 1
  for (int c0 = 0; c0 < NCH; c0 += CU)
#pragma unroll
  for (int cc = 0; cc < CU; ++cc) {
    const int c = c0 + cc;
    if (FIRST) {
#pragma unroll
      for (int u = 0; u < CH; ++u) xb[cc % NB][u] = fmaxf(cs_first_layer(s_w1[CH * c + u], p0, p1, p2), 0.f);
    } else if (c + NB - 1 < NCH) {
      load_rows(c + NB - 1, xb[(cc + NB - 1) % NB]);
    }
    float* x = xb[cc % NB];
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < CH; ++u) m = fmaxf(m, __builtin_fabsf(live ? x[u] : 0.f));
    const unsigned E = cs_exp(wave_max(m));
    if (E > Ex) {   // wave-uniform: larger values than any chunk before -- shrink the scale, rescale the sums (exact)
      if (c > 0) {
        const unsigned d = E - Ex;
        const float f = d > 126u ? 0.f : __uint_as_float((127u - d) << 23);   // 2^-(E - Ex)
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][t][r] *= f;
      }
      Ex = E;
    }
    const float sx = cs_scale(Ex);
    half8 xh[2], xl[2];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float v0 = x[j] * sx, v1 = x[8 + j] * sx;
      cs_swap32(v0, v1);    // v0: column block 0, v1: column block 1; lanes (column, k half)
      const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
      xh[0][j] = h0;
      xl[0][j] = (_Float16)(v0 - (float)h0);
      xh[1][j] = h1;
      xl[1][j] = (_Float16)(v1 - (float)h1);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const half8 wh = IMG ? wimg[((size_t)t * a.img_kc + c) * 2 * 64]
                           : *reinterpret_cast<const half8*>(arow + t * 32 * PITCH + c * 32);
      const half8 wl = IMG ? wimg[(((size_t)t * a.img_kc + c) * 2 + 1) * 64]
                           : *reinterpret_cast<const half8*>(arow + t * 32 * PITCH + c * 32 + 64 * PITCH);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        if (PN2 == 1 || PN2 == 3) {   // transposed: rows = samples, columns = channels (the max over samples becomes lane-local)
          acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[cb], wh, acc[cb][t], 0, 0, 0);
          acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[cb], wh, acc[cb][t], 0, 0, 0);
          acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[cb], wl, acc[cb][t], 0, 0, 0);
        } else {
          acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[cb], acc[cb][t], 0, 0, 0);
          acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[cb], acc[cb][t], 0, 0, 0);
          acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[cb], acc[cb][t], 0, 0, 0);
        }
      }
    }
  }
  const float unscale = cs_unscale(Ex) * (IMG ? a.Wun[0] : cs_unscale(Ew));

  if (PN2 == 3) {
    // pack2 layout, N = 128: waves 2 i, 2 i + 1 hold columns 0-63 / 64-127 of instance 2 z + i.  Per wave as PN2 == 1
    // (lane-local max over 32 registers + one exchange), then the two waves of an instance meet in LDS: the max over
    // the instance's 128 points (GroupAll + max_pool2d, pointnet2_modules.py:57-70), first maximal point on ties.
    float* s_pv = s_red + 4;                                  // [4 waves][64 channels]
    int* s_ps = reinterpret_cast<int*>(s_pv + 256);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float v = -__builtin_inff();
      int smp = 0;
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const bool gt = acc[cb][t][r] > v;
          v = gt ? acc[cb][t][r] : v;
          smp = gt ? 32 * cb + mfma_row(r, lane) : smp;
        }
      const float ov = __shfl_xor(v, 32, 64);
      const int os = __shfl_xor(smp, 32, 64);
      const bool take = ov > v || (ov == v && os < smp);
      v = take ? ov : v;
      smp = take ? os : smp;
      if (lane < 32) {
        s_pv[wave * 64 + 32 * t + lane] = v * unscale;      // the (positive, per-wave) scale before the waves compare
        s_ps[wave * 64 + 32 * t + lane] = smp + 64 * (wave & 1);
      }
    }
    __syncthreads();
    if ((wave & 1) == 0 && b_ok) {
      const float v0 = s_pv[wave * 64 + lane], v1 = s_pv[(wave + 1) * 64 + lane];
      const bool take = v1 > v0;                              // ties: the lower point index (this wave's half)
      const int co = rb * 64 + lane;
      const size_t e = (size_t)bz * a.Co + co;
      a.pool_out[e] = fmaxf((take ? v1 : v0) + a.pool_bias[co], 0.f);
      a.pool_arg[e] = take ? s_ps[(wave + 1) * 64 + lane] : s_ps[wave * 64 + lane];
    }
    return;
  }
  if (PN2 == 1) {
    // acc[cb][t][r]: channel rb*64 + 32t + (lane & 31), sample 32 cb + (r&3) + 8(r>>2) + 4(lane>>5) of the wave's centre:
    // max over the 64 samples = 32 registers + one exchange between the lane halves; first maximal sample on ties
    if (centre < centres) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        float v = -__builtin_inff();
        int smp = 0;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool gt = acc[cb][t][r] > v;
            v = gt ? acc[cb][t][r] : v;
            smp = gt ? 32 * cb + mfma_row(r, lane) : smp;
          }
        const float ov = __shfl_xor(v, 32, 64);
        const int os = __shfl_xor(smp, 32, 64);
        const bool take = ov > v || (ov == v && os < smp);
        v = take ? ov : v;
        smp = take ? os : smp;
        if (lane < 32) {
          const int co = rb * 64 + 32 * t + lane;
          const size_t e = ((size_t)b * a.Co + co) * centres + centre;
          a.pool_out[e] = fmaxf(v * unscale + a.pool_bias[co], 0.f);
          a.pool_arg[e] = smp;
        }
      }
    }
    return;
  }
  // epilogue: every pair of accumulator registers becomes two 64-column rows (row, row + 4)
  float* Y = BWD3 ? nullptr : a.Y + (size_t)b * a.sYb + col;
  const float* Z = a.Z ? a.Z + (size_t)b * a.sZb + col : nullptr;
  unsigned long long mymask = 0ull;
  float q0 = 0.f, q1 = 0.f, q2 = 0.f;   // BWD3: d/d(T^T x) of this lane's point, summed over the 64 rows
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[8], z[8], y[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = acc[0][t][4 * g + i];
        v[4 + i] = acc[1][t][4 * g + i];
        cs_swap32(v[i], v[4 + i]);    // v[i]: row base+i, v[4+i]: row base+4+i, lane = column
      }
      const int row0 = rb * 64 + t * 32 + 8 * g;   // rows row0 .. row0+7 in the order of v[]
      if (a.Zmask && wave_live) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int rl = t * 32 + 8 * g + i;   // the lane that holds this row's gate word
          const unsigned mlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)zmv, rl);
          const unsigned mhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(zmv >> 32), rl);
          z[i] = ((lane < 32 ? mlo >> lane : mhi >> (lane - 32)) & 1u) ? 1.f : 0.f;
        }
      }
      if (live) {
        if (Z) {
#pragma unroll
          for (int i = 0; i < 8; ++i) z[i] = Z[(size_t)(row0 + i) * a.ldZ];
        }
        if (a.accumulate) {
#pragma unroll
          for (int i = 0; i < 8; ++i) y[i] = Y[(size_t)(row0 + i) * a.ldY];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float o = v[i] * unscale;
          if (a.bias) o += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(biasv), t * 32 + 8 * g + i));
          if (a.relu) o = fmaxf(o, 0.f);
          if (a.accumulate) o += y[i];
          if (Z || a.Zmask) o = z[i] > 0.f ? o : 0.f;  // gate AFTER accumulation (sum of branches, then relu')
          if (GFIRST) o = cs_first_layer(s_w1[row0 + i], p0, p1, p2) > 0.f ? o : 0.f;
          if (BWD3) {
            const float4 w = s_w1[row0 + i];
            q0 = fmaf(w.x, o, q0);
            q1 = fmaf(w.y, o, q1);
            q2 = fmaf(w.z, o, q2);
          } else {
            Y[(size_t)(row0 + i) * a.ldY] = o;
          }
          v[i] = o;
        }
      }
      if (a.Ymask && wave_live) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned long long mk = __ballot(live && v[i] > 0.f);
          if (lane == t * 32 + 8 * g + i) mymask = mk;
        }
      }
    }
  }
  if (a.Ymask && wave_live) a.Ymask[mword + rb * 64 + lane] = mymask;
  if (BWD3) {
    // x' = T^T x  =>  dx[d] = sum_c T[d][c] q[c];  dT[d][c] = sum_n x[d][n] q[c][n]
    if (!live) {
      q0 = 0.f;
      q1 = 0.f;
      q2 = 0.f;
    }
    float d0 = q0, d1 = q1, d2 = q2;
    if (a.T3) {
      const float* t = a.T3 + (size_t)b * 9;
      d0 = fmaf(t[2], q2, fmaf(t[1], q1, t[0] * q0));   // explicit: the chain kernel must form the same bits
      d1 = fmaf(t[5], q2, fmaf(t[4], q1, t[3] * q0));
      d2 = fmaf(t[8], q2, fmaf(t[7], q1, t[6] * q0));
    }
    if (live) {
      float* dxp = a.dx3 + (size_t)b * 3 * a.N + col;
      if (a.accumulate) {
        d0 += dxp[0];
        d1 += dxp[a.N];
        d2 += dxp[2 * (size_t)a.N];
      }
      dxp[0] = d0;
      dxp[a.N] = d1;
      dxp[2 * (size_t)a.N] = d2;
    }
    if (a.dTpart) {   // workgroup-uniform
      const float prod[9] = {x0 * q0, x0 * q1, x0 * q2, x1 * q0, x1 * q1, x1 * q2, x2 * q0, x2 * q1, x2 * q2};
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const float v = wave_sum(prod[i]);
        if (lane == 0) s_part[wave * 9 + i] = v;
      }
      __syncthreads();
      if (tid < 9) {
        a.dTpart[((size_t)b * gridDim.y + cblk) * GEOA3_DT_PITCH + tid] =
            s_part[tid] + s_part[9 + tid] + s_part[18 + tid] + s_part[27 + tid];
      }
    }
  }
}

}  // namespace

int launch_conv_cm_split(const ConvArgs& a, hipStream_t s) {
  if ((a.K != 64 && a.K != 128 && a.K != 256 && !(a.K == 512 && a.Wimg)) || a.Co <= 0 || a.Co % 64 != 0) return GEOA3_ENOSUPPORT;
  const size_t lds = (size_t)2 * 64 * (a.K * 2 + 16) + 64 * 16 + 44 * 4;
  if (a.pack2 && (a.N > 128 || a.sWb != 0 || a.Ymask || a.Zmask || (a.pool_out && !a.Wimg) || a.oh_g || a.produce_first || a.gate_first))
    return GEOA3_EINVAL;
  dim3 grid(a.Co / 64, a.pack2 ? 1 : (a.N + 255) / 256, a.pack2 ? (a.B + 1) / 2 : a.B);
  if (a.Wimg && a.pool_out) {   // K = 512, the output pooled over each instance's 128 points (PointNet++ level 3)
    if (!a.pack2 || a.N != 128 || a.K != 512 || !a.pool_arg || !a.pool_bias || !a.Wun || a.oh_g) return GEOA3_ENOSUPPORT;
    const size_t l = 64 * 16 + 44 * 4 + 2048;
    hipLaunchKernelGGL((conv_cm64s_kernel<32, false, false, false, 3, true>), grid, dim3(256), l, s, a);
    GEOA3_CHECK_LAUNCH();
    return GEOA3_OK;
  }
  if (a.pool_out || a.oh_g) {   // PointNet++ forms: a wave = one centre
    if (a.N % 64 != 0 || a.produce_first || a.gate_first) return GEOA3_ENOSUPPORT;
    if (a.pool_out && (!a.pool_arg || !a.pool_bias || a.K != 128 || a.oh_g)) return GEOA3_ENOSUPPORT;
    if (a.oh_g && (!a.oh_arg || a.K != 256)) return GEOA3_ENOSUPPORT;
    if (a.pool_out) {
      hipLaunchKernelGGL((conv_cm64s_kernel<8, false, false, false, 1>), grid, dim3(256), lds, s, a);
    } else {
      auto kern = conv_cm64s_kernel<16, false, false, false, 2>;
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    }
    GEOA3_CHECK_LAUNCH();
    return GEOA3_OK;
  }
  if (a.Wimg) {   // pre-split weights: plain layers with shared weights only
    if (a.produce_first || a.gate_first || a.sWb != 0 || (a.K != 128 && a.K != 256) || !a.Wun) return GEOA3_ENOSUPPORT;
    const size_t l = 64 * 16 + 44 * 4;
    if (a.K == 128) hipLaunchKernelGGL((conv_cm64s_kernel<8, false, false, false, 0, true>), grid, dim3(256), l, s, a);
    else hipLaunchKernelGGL((conv_cm64s_kernel<16, false, false, false, 0, true>), grid, dim3(256), l, s, a);
    GEOA3_CHECK_LAUNCH();
    return GEOA3_OK;
  }
  if (a.K == 256) {   // PointNet++ level 2 (geoa3_conv1x1): plain layers only; 68.8 KB of LDS
    if (a.produce_first || a.gate_first) return GEOA3_ENOSUPPORT;
    auto kern = conv_cm64s_kernel<16, false, false, false>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    GEOA3_CHECK_LAUNCH();
    return GEOA3_OK;
  }
  // few workgroups (at most two per CU: the 32-instance shards of a multi-GPU run): latency-bound, deep prefetch
  const bool deep = (size_t)grid.x * grid.y * grid.z <= 512;
  if (a.produce_first)
    hipLaunchKernelGGL((conv_cm64s_kernel<4, true, false, false>), grid, dim3(256), lds, s, a);
  else if (a.gate_first && a.dx3 && a.K == 64) {
    if (deep) hipLaunchKernelGGL((conv_cm64s_kernel<4, false, true, true, 0, false, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv_cm64s_kernel<4, false, true, true>), grid, dim3(256), lds, s, a);
  } else if (a.gate_first && a.dx3) {
    if (deep) hipLaunchKernelGGL((conv_cm64s_kernel<8, false, true, true, 0, false, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv_cm64s_kernel<8, false, true, true>), grid, dim3(256), lds, s, a);
  } else if (a.gate_first && a.K == 64)
    hipLaunchKernelGGL((conv_cm64s_kernel<4, false, true, false>), grid, dim3(256), lds, s, a);
  else if (a.gate_first)
    hipLaunchKernelGGL((conv_cm64s_kernel<8, false, true, false>), grid, dim3(256), lds, s, a);
  else if (a.K == 64) {
    if (deep) hipLaunchKernelGGL((conv_cm64s_kernel<4, false, false, false, 0, false, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv_cm64s_kernel<4, false, false, false>), grid, dim3(256), lds, s, a);
  } else {
    if (deep) hipLaunchKernelGGL((conv_cm64s_kernel<8, false, false, false, 0, false, true>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((conv_cm64s_kernel<8, false, false, false>), grid, dim3(256), lds, s, a);
  }
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
