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
#include <cstdlib>

namespace {

constexpr int WM_CO = 128;     // output channels per workgroup (4 waves x 32)
constexpr int WM_THREADS = 256;
constexpr int WM_CI = 128;     // input channels of every wide layer
constexpr int WM_HALO = 4;     // halo on both sides of a staged activation row (float4 aligned)

// Weights arrive in MFMA A-fragment order (host: geoa3_amd/pointnet.py pack_wide_fragments):
//   Wp[((T*TAPS + tap)*16 + j)*64 + lane][i] = W[32*T + (lane&31)][tap*128 + 8*j + 4*(lane>>5) + i]
// so that one 16-byte load per lane (1 KiB per wave, fully coalesced, served by L2) yields the A operands
// of four k-steps; k is consumed in the order 8j + 4*(lane>>5) + i, and the B operand is read from the
// LDS activation tile with the same k.  The weights never touch LDS.
//
// Activations: chunk of CHUNK input channels x (64 + halo) points, double buffered in LDS, staged through
// registers one pass ahead (global loads of pass p+1 fly under the MFMAs of pass p); ONE barrier per pass.
template <int TAPS, int CHUNK, int CT, int OCC>   // CT = 32-column sub-tiles per wave: the tile is 32*CT points wide
__global__ __launch_bounds__(WM_THREADS, OCC) void wide_max_kernel(WideArgs a) {
  constexpr int WM_COLS = 32 * CT;
  constexpr int WM_XP = WM_COLS + 2 * WM_HALO;                  // LDS pitch of an activation row
  constexpr int NXTOT = CHUNK * (WM_XP / 4);                    // float4 slots of one activation chunk
  constexpr int NX = (NXTOT + WM_THREADS - 1) / WM_THREADS;
  constexpr int NG = TAPS * CHUNK / 8;                          // weight fragment groups per pass (8 k each)
  constexpr int PASSES = WM_CI / CHUNK;
  constexpr int PF = 3;                                         // weight groups in flight
  extern __shared__ __attribute__((aligned(16))) float smem[];  // [2][CHUNK][WM_XP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware mapping: workgroup ids are dealt round-robin over the 8 XCDs, so the 8 channel tiles of one
  // instance get ids that are equal mod 8 -> one XCD, one L2 copy of the instance's activations.
  const int L = blockIdx.x, grp = L >> 6, rem = L & 63;
  const int b = grp * 8 + (rem & 7);
  if (b >= a.B) return;
  const int co0 = (rem >> 3) * WM_CO + wave * 32;               // this wave's 32 output channels
  const int N = a.N, kh = lane >> 5, l31 = lane & 31;
  const float* X = a.X + (size_t)b * a.sXb;
  const bool xvec = (a.ldX & 3) == 0;
  const float4* Wp = reinterpret_cast<const float4*>(a.W) + (size_t)(co0 / 32) * TAPS * 16 * 64 + lane;

  float4 xreg[NX];
  auto load_x = [&](int n0, int ci0) {
    const bool interior = xvec && n0 >= WM_HALO && n0 + WM_COLS + WM_HALO <= N;   // workgroup-uniform
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + i * WM_THREADS;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e < NXTOT) {
        const int q = e % (WM_XP / 4), c = e / (WM_XP / 4);
        const int n = n0 - WM_HALO + q * 4;
        const float* src = X + (size_t)(ci0 + c) * a.ldX;
        if (interior) {
          v = *reinterpret_cast<const float4*>(src + n);
        } else if (n >= 0 && n + 3 < N && xvec) {
          v = *reinterpret_cast<const float4*>(src + n);
        } else {
          v.x = (n >= 0 && n < N) ? src[n] : 0.f;
          v.y = (n + 1 >= 0 && n + 1 < N) ? src[n + 1] : 0.f;
          v.z = (n + 2 >= 0 && n + 2 < N) ? src[n + 2] : 0.f;
          v.w = (n + 3 >= 0 && n + 3 < N) ? src[n + 3] : 0.f;
        }
      }
      xreg[i] = v;
    }
  };
  auto store_x = [&](float* buf) {
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const int e = tid + i * WM_THREADS;
      if (e < NXTOT) {
        const int q = e % (WM_XP / 4), c = e / (WM_XP / 4);
        *reinterpret_cast<float4*>(buf + c * WM_XP + q * 4) = xreg[i];
      }
    }
  };
  // fragment group g of the pass starting at input channel ci0: (tap, jj) = (g / (CHUNK/8), g % (CHUNK/8))
  auto load_w = [&](int ci0, int g) -> float4 {
    const int tap = g / (CHUNK / 8), jj = g - tap * (CHUNK / 8);
    return Wp[(size_t)((tap * 16) + (ci0 >> 3) + jj) * 64];
  };

  float rmax[16];
  int rarg[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    rmax[r] = -__builtin_inff();
    rarg[r] = 0;
  }

  load_x(0, 0);
  store_x(smem);
  __syncthreads();
  int pass = 0;
  for (int n0 = 0; n0 < N; n0 += WM_COLS) {
    f32x16 acc[CT];
#pragma unroll
    for (int t = 0; t < CT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

#pragma unroll 1
    for (int p = 0; p < PASSES; ++p, ++pass) {
      const int ci0 = p * CHUNK;
      const bool last = p + 1 == PASSES;
      const int nn0 = last ? n0 + WM_COLS : n0, nci0 = last ? 0 : ci0 + CHUNK;
      const bool more = nn0 < N;
      float4 wf[PF];
#pragma unroll
      for (int g = 0; g < PF; ++g) wf[g] = load_w(ci0, g);
      if (more) load_x(nn0, nci0);
      const float* xb = smem + (pass & 1) * (CHUNK * WM_XP) + kh * 4 * WM_XP + WM_HALO + l31 - TAPS / 2;
      // B operands of one fragment group: rows 8jj + 4*kh + i, columns col+tap-TAPS/2 and +32
      float bq[4 * CT], bn[4 * CT];
      auto read_b = [&](int g, float* d) {
        const int tap = g / (CHUNK / 8), jj = g - tap * (CHUNK / 8);
        const float* xp = xb + jj * 8 * WM_XP + tap;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < CT; ++t) d[CT * i + t] = xp[i * WM_XP + 32 * t];
      };
      read_b(0, bq);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float4 w = wf[g % PF];
        // issue the memory traffic of later groups BEFORE this group's MFMAs and pin it there: the matrix
        // pipe then covers the L2 / LDS latency instead of the compiler sinking the loads next to their use
        if (g + PF < NG) wf[g % PF] = load_w(ci0, g + PF);
        if (g + 1 < NG) read_b(g + 1, bn);
        __builtin_amdgcn_sched_barrier(0);
        const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < CT; ++t) acc[t] = mfma32(wv[i], bq[CT * i + t], acc[t]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 4 * CT; ++i) bq[i] = bn[i];
      }
      if (more) store_x(smem + ((pass + 1) & 1) * (CHUNK * WM_XP));
      __syncthreads();
    }
    // fold this tile into the running maximum (strict >: the lowest point index wins a tie)
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const int col = n0 + t * 32 + l31;
      const bool ok = col < N;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const bool gt = ok && acc[t][r] > rmax[r];
        rmax[r] = gt ? acc[t][r] : rmax[r];
        rarg[r] = gt ? col : rarg[r];
      }
    }
  }

  // reduce over the 32 lanes that share (reg, lane>>5), i.e. over the columns; then bias + relu
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
  if (l31 == 0) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + mfma_row(r, lane);
      const size_t o = (size_t)b * a.Co + co;
      a.out[o] = fmaxf(rmax[r] + a.bias[co], 0.f);
      a.arg[o] = rarg[r];
    }
  }
}

// ------------------------------------------------------------------------------------------
// Column-major form of the same layer.  A work unit is (instance, 64-point tile, half of the 1024 channels): the
// activation tile [128 ci][64 + halo] is staged ONCE into LDS and serves four channel groups (each wave: 32 channels
// per group, all of K in one sweep, no barrier inside), instead of being re-staged in K-chunks for every channel
// group.  The MFMA runs with its operands swapped (activations as A, weights as B), so a lane ends up with ONE channel
// and 32 of the tile's points in registers: the max over the tile is lane-local plus one exchange between the
// register halves.  Tiles are combined through a 64-bit atomic max on (order-preserving value bits, ~point index) --
// order independent, hence deterministic -- and a tiny second kernel decodes, adds the bias and applies the relu.
// 8000 equal units over 768 resident workgroups balance to 97 % (the row-major form: 2000 units, 87 %).
// ------------------------------------------------------------------------------------------
constexpr int W2_COLS = 64;
constexpr int W2_XP = W2_COLS + 2 * WM_HALO;     // 72
constexpr int W2_GROUPS = 4;                     // channel groups of 128 per unit (half of the 1024 channels)

__device__ __forceinline__ unsigned long long wide_key(float v, int col) {
  unsigned u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);            // order-preserving map of the float
  return ((unsigned long long)u << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)col);   // ties: the LOWER point index wins
}

template <int TAPS>
__global__ __launch_bounds__(WM_THREADS, 3) void wide_max2_kernel(WideArgs a, int slots_per_xcd) {
  constexpr int NGT = TAPS * 16;                 // fragment groups (8 k each) of one channel tile
  constexpr int PF = 3;
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [128][W2_XP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, l31 = lane & 31;
  const int N = a.N, tiles = (N + W2_COLS - 1) / W2_COLS;
  const int per_inst = tiles * 2;                                   // units of one instance
  // XCD-aware static schedule: workgroup ids are dealt round-robin over the 8 XCDs; XCD x owns the instances
  // b = x, x+8, ... and its slots walk that list of units with stride slots_per_xcd
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int inst_x = (a.B - xcd + 7) / 8;                           // instances owned by this XCD
  const int units = inst_x * per_inst;
  const bool xvec = (a.ldX & 3) == 0;
  for (int u = slot; u < units; u += slots_per_xcd) {
    const int q = u / per_inst, r = u - q * per_inst;
    const int b = xcd + 8 * q, tile = r >> 1, half = r & 1;
    const int n0 = tile * W2_COLS;
    const float* X = a.X + (size_t)b * a.sXb;
    __syncthreads();   // every wave is done with the previous tile
    {
      const bool interior = xvec && n0 >= WM_HALO && n0 + W2_COLS + WM_HALO <= N;
      for (int e = tid; e < WM_CI * (W2_XP / 4); e += WM_THREADS) {
        const int qd = e % (W2_XP / 4), c = e / (W2_XP / 4);
        const int n = n0 - WM_HALO + qd * 4;
        const float* src = X + (size_t)c * a.ldX;
        float4 v;
        if (interior || (n >= 0 && n + 3 < N && xvec)) {
          v = *reinterpret_cast<const float4*>(src + n);
        } else {
          v.x = (n >= 0 && n < N) ? src[n] : 0.f;
          v.y = (n + 1 >= 0 && n + 1 < N) ? src[n + 1] : 0.f;
          v.z = (n + 2 >= 0 && n + 2 < N) ? src[n + 2] : 0.f;
          v.w = (n + 3 >= 0 && n + 3 < N) ? src[n + 3] : 0.f;
        }
        *reinterpret_cast<float4*>(smem + c * W2_XP + qd * 4) = v;
      }
    }
    __syncthreads();
    const float* xb = smem + kh * 4 * W2_XP + WM_HALO + l31 - TAPS / 2;
#pragma unroll 1
    for (int g = 0; g < W2_GROUPS; ++g) {
      const int co0 = (half * W2_GROUPS + g) * WM_CO + wave * 32;
      const float4* Wp = reinterpret_cast<const float4*>(a.W) + (size_t)(co0 / 32) * NGT * 64 + lane;
      f32x16 acc[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[0][i] = 0.f;
        acc[1][i] = 0.f;
      }
      float4 wf[PF];
#pragma unroll
      for (int f = 0; f < PF; ++f) wf[f] = Wp[(size_t)f * 64];
      float bq[8], bn[8];
      auto read_b = [&](int f, float* d) {      // fragment group f = tap*16 + jj: rows 8jj + 4kh + i, column shift tap
        const int tap = f >> 4, jj = f & 15;
        const float* xp = xb + jj * 8 * W2_XP + tap;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          d[2 * i] = xp[i * W2_XP];
          d[2 * i + 1] = xp[i * W2_XP + 32];
        }
      };
      read_b(0, bq);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int f = 0; f < NGT; ++f) {
        const float4 w = wf[f % PF];
        if (f + PF < NGT) wf[f % PF] = Wp[(size_t)(f + PF) * 64];
        if (f + 1 < NGT) read_b(f + 1, bn);
        __builtin_amdgcn_sched_barrier(0);
        const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[0] = mfma32(bq[2 * i], wv[i], acc[0]);        // operands swapped: rows = points, columns = channels
          acc[1] = mfma32(bq[2 * i + 1], wv[i], acc[1]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 8; ++i) bq[i] = bn[i];
      }
      // lane: channel co0 + l31; acc[t][r]: point n0 + 32t + (r&3) + 8(r>>2) + 4kh.  Ascending point order, strict >
      float v = -__builtin_inff();
      int col = 0;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int n = n0 + 32 * t + mfma_row(r, lane);
          const bool gt = n < N && acc[t][r] > v;
          v = gt ? acc[t][r] : v;
          col = gt ? n : col;
        }
      const float ov = __shfl_xor(v, 32, 64);
      const int oc = __shfl_xor(col, 32, 64);
      const bool take = ov > v || (ov == v && oc < col);
      v = take ? ov : v;
      col = take ? oc : col;
      if (lane < 32) atomicMax(a.keys + (size_t)b * a.Co + co0 + lane, wide_key(v, col));
    }
  }
}

__global__ __launch_bounds__(256) void wide_finalize_kernel(const unsigned long long* __restrict__ keys,
                                                            const float* __restrict__ bias, int Co, int total,
                                                            float* __restrict__ out, int* __restrict__ arg) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const unsigned long long k = keys[e];
  const unsigned u = (unsigned)(k >> 32);
  const float v = __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
  out[e] = fmaxf(v + bias[e % Co], 0.f);
  arg[e] = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
}

// ------------------------------------------------------------------------------------------
// Sparse backward.  One workgroup per (instance, 128-point tile), split in NP parts of 128/NP columns with 128
// threads each; thread (ci, part) owns row ci of its part, so no two threads ever touch the same accumulator
// and the summation order is fixed (deterministic, no atomics).
// Per block of WB_BLOCK output channels: the first wave of each part compacts the (channel, tap) pairs whose
// arg-max column falls into its part into an LDS hit list (ballot + popcount, in (co, tap) order, the upstream
// gradient stored next to it); then the 128 threads of the part walk ONLY the hits, WB_BATCH independent
// weight-row loads in flight.  The walk is L2-latency bound, so the parts (NP = 4: 16 waves per workgroup) are
// what keeps enough loads in flight: LDS (66 KB of accumulators) allows only two workgroups per CU.
// ------------------------------------------------------------------------------------------
constexpr int WB_COLS = 128;
constexpr int WB_BATCH = 16;

template <int TAPS, int NP, int WB_BLOCK>   // WB_BLOCK: output channels per compaction round
__global__ __launch_bounds__(128 * NP) void wide_max_bwd_kernel(WideBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int PC = WB_COLS / NP;                                            // columns per part
  float* s_acc = smem;                                                        // [128 ci][WB_COLS + 1]
  int* s_hit = reinterpret_cast<int*>(smem + WM_CI * (WB_COLS + 1));          // [NP][WB_BLOCK*TAPS]
  float* s_g = reinterpret_cast<float*>(s_hit + NP * WB_BLOCK * TAPS);        // [NP][WB_BLOCK*TAPS]
  int* s_cnt = reinterpret_cast<int*>(s_g + NP * WB_BLOCK * TAPS);            // [NP]
  const int tid = threadIdx.x, b = blockIdx.y, m0 = blockIdx.x * WB_COLS;
  const int ci = tid & 127, part = tid >> 7, lane = tid & 63;
  const bool builder = (tid & 127) < 64;                                      // first wave of each part
  const float* gb = a.g + (size_t)b * a.Co;
  const int* argb = a.arg + (size_t)b * a.Co;
  int* hits = s_hit + part * WB_BLOCK * TAPS;
  float* hitg = s_g + part * WB_BLOCK * TAPS;
  float* row = s_acc + ci * (WB_COLS + 1);
  for (int j = part * PC; j < part * PC + PC; ++j) row[j] = 0.f;
  const int lo = m0 + part * PC, hi = lo + PC;

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
          if (hit) {
            const int slot = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            hits[slot] = (co * TAPS + tap) | ((m - m0) << 16);
            hitg[slot] = g;
          }
          cnt += __popcll(mask);
        }
      }
      if (lane == 0) s_cnt[part] = cnt;
    }
    __syncthreads();
    const int cnt = s_cnt[part];
    for (int h0 = 0; h0 < cnt; h0 += WB_BATCH) {
      float w[WB_BATCH], gg[WB_BATCH];
      int mm[WB_BATCH];
#pragma unroll
      for (int u = 0; u < WB_BATCH; ++u) {
        const bool ok = h0 + u < cnt;
        const int e = hits[ok ? h0 + u : cnt - 1];
        w[u] = a.W[(size_t)(e & 0xffff) * WM_CI + ci];   // row co*TAPS + tap of the [Co*TAPS][128] weight view
        gg[u] = ok ? hitg[h0 + u] : 0.f;
        mm[u] = e >> 16;
      }
#pragma unroll
      for (int u = 0; u < WB_BATCH; ++u) row[mm[u]] += w[u] * gg[u];
    }
  }
  __syncthreads();
  // write out with the relu gate of the layer input: 4 consecutive points per thread (16-byte loads / stores)
  float* dX = a.dX + (size_t)b * a.sXb;
  const float* Z = a.Z + (size_t)b * a.sZb;
  const bool vec = ((a.ldX | a.ldZ) & 3) == 0;
#pragma unroll 4
  for (int e = tid; e < WM_CI * (WB_COLS / 4); e += 128 * NP) {
    const int c = e / (WB_COLS / 4), j = (e - c * (WB_COLS / 4)) * 4;
    const int m = m0 + j;
    const float* sa = s_acc + c * (WB_COLS + 1) + j;
    if (vec && m + 3 < a.N) {
      const float4 z = *reinterpret_cast<const float4*>(Z + (size_t)c * a.ldZ + m);
      float4 v;
      v.x = z.x > 0.f ? sa[0] : 0.f;
      v.y = z.y > 0.f ? sa[1] : 0.f;
      v.z = z.z > 0.f ? sa[2] : 0.f;
      v.w = z.w > 0.f ? sa[3] : 0.f;
      *reinterpret_cast<float4*>(dX + (size_t)c * a.ldX + m) = v;
    } else {
      for (int i = 0; i < 4; ++i)
        if (m + i < a.N) dX[(size_t)c * a.ldX + m + i] = Z[(size_t)c * a.ldZ + m + i] > 0.f ? sa[i] : 0.f;
    }
  }
}

}  // namespace

template <int TAPS, int CHUNK, int CT, int OCC>
static void launch_wide_variant(const WideArgs& a, hipStream_t s) {
  const size_t lds = (size_t)2 * CHUNK * (32 * CT + 2 * WM_HALO) * sizeof(float);
  const int groups = (a.B + 7) / 8;   // 8 instances x 8 channel tiles per group of 64 workgroups
  auto kern = wide_max_kernel<TAPS, CHUNK, CT, OCC>;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  hipLaunchKernelGGL(kern, dim3(groups * 64), dim3(WM_THREADS), lds, s, a);
}

int launch_wide_max(const WideArgs& a, hipStream_t s) {
  if (a.Co != 8 * WM_CO || (a.taps != 1 && a.taps != 3)) return GEOA3_ENOSUPPORT;
  const int tag = a.taps == 3 ? GEOA3_PROF_CONV5 : GEOA3_PROF_TNETWIDE;
  static const bool colmajor = getenv("GEOA3_WIDE_ROWMAJOR") == nullptr;
  if (colmajor && a.keys) {
    geoa3_prof_begin(tag, s);
    if (hipMemsetAsync(a.keys, 0, (size_t)a.B * a.Co * sizeof(unsigned long long), s) != hipSuccess) return GEOA3_ELAUNCH;
    const size_t lds = (size_t)WM_CI * W2_XP * sizeof(float);
    const int slots = 96;   // per XCD: 768 resident workgroups of 4 waves
    if (a.taps == 1) hipLaunchKernelGGL(wide_max2_kernel<1>, dim3(slots * 8), dim3(WM_THREADS), lds, s, a, slots);
    else hipLaunchKernelGGL(wide_max2_kernel<3>, dim3(slots * 8), dim3(WM_THREADS), lds, s, a, slots);
    const int total = a.B * a.Co;
    hipLaunchKernelGGL(wide_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, s, a.keys, a.bias, a.Co, total,
                       a.out, a.arg);
    geoa3_prof_end(tag, s);
    GEOA3_CHECK_LAUNCH();
    return GEOA3_OK;
  }
  geoa3_prof_begin(tag, s);
  // chunk sizes / occupancy picked on hardware (profiles/): 3 workgroups of 4 waves per CU
  if (a.taps == 1) launch_wide_variant<1, 32, 2, 3>(a, s);
  else launch_wide_variant<3, 16, 2, 3>(a, s);
  geoa3_prof_end(tag, s);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_debug_wide_bwd(const float* g, const int32_t* arg, const float* W, const float* Z, float* dX, int B,
                                    int N, int taps, void* stream) {
  WideBwdArgs a{};
  a.g = g; a.arg = arg; a.W = W;
  a.Z = Z; a.sZb = (long)128 * N; a.ldZ = N;
  a.dX = dX; a.sXb = (long)128 * N; a.ldX = N;
  a.Co = 1024; a.N = N; a.B = B; a.taps = taps;
  return launch_wide_max_bwd(a, geoa3_stream(stream));
}

template <int TAPS, int NP, int WB_BLOCK>
static void launch_wide_bwd_variant(const WideBwdArgs& a, hipStream_t s) {
  dim3 grid((a.N + WB_COLS - 1) / WB_COLS, a.B);
  const size_t lds = ((size_t)WM_CI * (WB_COLS + 1) + 2 * (size_t)NP * WB_BLOCK * TAPS + NP) * sizeof(float);
  auto kern = wide_max_bwd_kernel<TAPS, NP, WB_BLOCK>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, grid, dim3(128 * NP), lds, s, a);
}

int launch_wide_max_bwd(const WideBwdArgs& a, hipStream_t s) {
  if (a.taps != 1 && a.taps != 3) return GEOA3_ENOSUPPORT;
  // parts / compaction block picked on hardware (tools/bench_widebwd.py): 93 us / 164 us at B=250, N=1024
  if (a.taps == 1) launch_wide_bwd_variant<1, 4, 256>(a, s);
  else launch_wide_bwd_variant<3, 4, 128>(a, s);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
