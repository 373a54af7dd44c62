// The three 1024-wide PointNet layers fused with their max-over-points (gfx950, fp32 MFMA):
// transform_net.conv3+bn3+relu+max (Model/PointNet.py:81-82) and conv5+bn5+relu+max (:146-147; conv5 is
// a kernel-3, pad-1 convolution over the POINT INDEX, :110).  [B,1024,N] is never written: the maxima of
// 64-point tiles are combined through an atomic max on packed (value, point) keys.  relu and the folded BN
// bias commute with the max, so they are applied once to the 1024 maxima.
// The backward is sparse -- only the arg-max column of every channel carries gradient -- and is a
// deterministic gather-by-owner accumulation in LDS (no atomics).
#include "pointnet_kernels.h"
#include "profile.h"

namespace {

constexpr int WM_CO = 128;     // output channels per workgroup (4 waves x 32)
constexpr int WM_THREADS = 256;
constexpr int WM_CI = 128;     // input channels of every wide layer
constexpr int WM_HALO = 4;     // halo on both sides of a staged activation row (float4 aligned)

// Weights arrive in MFMA fragment order (host: geoa3_amd/pointnet.py pack_wide_fragments):
//   Wp[((T*TAPS + tap)*16 + j)*64 + lane][i] = W[32*T + (lane&31)][tap*128 + 8*j + 4*(lane>>5) + i]
// so that one 16-byte load per lane (1 KiB per wave, fully coalesced, served by L2) yields the weight operands
// of four k-steps; k is consumed in the order 8j + 4*(lane>>5) + i, and the activation operand is read from the
// LDS tile with the same k.  The weights never touch LDS.

// ------------------------------------------------------------------------------------------
// A work unit is (instance, 64-point tile, W2_GROUPS x 128 channels): the activation tile [128 ci][64 + halo] is
// staged ONCE into LDS and serves all channel groups of the unit (each wave: 32 channels per group, all of K in one
// sweep, no barrier inside; the weight fragments stream from L2 through a register ring that runs across the
// groups).  The first version of this kernel walked (instance, 128 channels) over all point tiles with the tile
// re-staged in K-chunks per channel group and a running arg-max in registers: 1.75 / 0.68 ms against 1.56 / 0.57 ms.
// The MFMA runs with its operands swapped (activations as A, weights as B), so a lane ends up with ONE channel
// and 32 of the tile's points in registers: the max over the tile is lane-local plus one exchange between the
// register halves.  Tiles are combined through a 64-bit atomic max on (order-preserving value bits, ~point index) --
// order independent, hence deterministic -- and a tiny second kernel decodes, adds the bias and applies the relu.
// 4000-8000 equal units over 768 resident workgroups balance to 95-97 % (2000 row-major units: 87 %).
// ------------------------------------------------------------------------------------------
constexpr int W2_COLS = 64;
constexpr int W2_XP = W2_COLS + 2 * WM_HALO;     // 72

template <int TAPS, int OCC, int W2_GROUPS>   // W2_GROUPS: channel groups of 128 per unit (8 / W2_GROUPS units per tile)
__global__ __launch_bounds__(WM_THREADS, OCC) void wide_max2_kernel(WideArgs a, int slots_per_xcd) {
  constexpr int NGT = TAPS * 16;                 // fragment groups (8 k each) of one channel tile
  constexpr int PF = TAPS == 1 ? 8 : 4;          // fragment loads in flight; NGT % PF == 0 (measured: 8 helps K = 128 only)
  extern __shared__ __attribute__((aligned(16))) float smem[];   // [128][W2_XP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, l31 = lane & 31;
  const int N = a.N, tiles = (N + W2_COLS - 1) / W2_COLS;
  constexpr int SPLIT = 8 / W2_GROUPS;
  const int per_inst = tiles * SPLIT;                               // units of one instance
  // XCD-aware static schedule: workgroup ids are dealt round-robin over the 8 XCDs; XCD x owns the instances
  // b = x, x+8, ... and its slots walk that list of units with stride slots_per_xcd
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int inst_x = (a.B - xcd + 7) / 8;                           // instances owned by this XCD
  const int units = inst_x * per_inst;
  const bool xvec = (a.ldX & 3) == 0;
  for (int u = slot; u < units; u += slots_per_xcd) {
    const int q = u / per_inst, r = u - q * per_inst;
    const int b = xcd + 8 * q, tile = r / SPLIT, half = r - tile * SPLIT;
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
    auto wbase = [&](int g) {
      const int co = (half * W2_GROUPS + g) * WM_CO + wave * 32;
      return reinterpret_cast<const float4*>(a.W) + (size_t)(co / 32) * NGT * 64 + lane;
    };
    // the weight fragments stream through a ring of PF registers that runs ACROSS the channel groups of the unit:
    // the first fragments of group g+1 are requested while group g still computes (NGT % PF == 0 keeps the slots)
    float4 wf[PF];
    {
      const float4* W0 = wbase(0);
#pragma unroll
      for (int f = 0; f < PF; ++f) wf[f] = W0[(size_t)f * 64];
    }
#pragma unroll 1
    for (int g = 0; g < W2_GROUPS; ++g) {
      const int co0 = (half * W2_GROUPS + g) * WM_CO + wave * 32;
      const float4* Wp = wbase(g);
      const float4* Wn = wbase(g + 1 < W2_GROUPS ? g + 1 : g);
      f32x16 acc[2];
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[0][i] = 0.f;
        acc[1][i] = 0.f;
      }
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
        else wf[f % PF] = Wn[(size_t)(f + PF - NGT) * 64];
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

__global__ __launch_bounds__(256) void wide_finalize_kernel(unsigned long long* __restrict__ keys,
                                                            const float* __restrict__ bias, int Co, int total,
                                                            float* __restrict__ out, int* __restrict__ arg) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const unsigned long long k = keys[e];
  keys[e] = 0ull;   // left clean for the next wide layer (WideArgs::keys_clean)
  const unsigned u = (unsigned)(k >> 32);
  const float v = __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
  out[e] = v != v ? v : fmaxf(v + bias[e % Co], 0.f);   // NaN (poisoned by the split kernel's range check) stays NaN
  arg[e] = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
}

// The same, and the backward's hit lists with it (Co = 1024, N <= WIDE_HITS_MAXN): one 1024-thread workgroup per instance,
// thread = channel.  The sparse backward (pointnet_wide_bwdconv.hip) needs, per 64-point tile, the (channel, tap) pairs whose
// arg-max column falls into the tile, by column, each column's list in (chunk of 64 channels, tap, channel) order.  It used
// to build them in every workgroup (250 x 16 workgroups each scanning the instance's 1024 channels); here they are built
// once per instance, all columns: hits [B][1024 TAPS] sorted by (column, chunk, tap, channel) and the columns' start
// offsets [B][N + 1], so that a tile's lists are ONE contiguous segment.  A channel is listed when its pooled value is
// positive -- the upstream gradient is gated by exactly that (zero elsewhere; a zero gradient that slips in adds +0).
//   rank of a hit among the lanes of its wave with the same column: ballots over the column's bits;
//   per-(chunk, column) counts in LDS (each wave owns its chunk's row: no atomics), prefix over the chunks per column,
//   block scan over the columns, placement.
constexpr int WIDE_HITS_MAXN = 4096;
template <int TAPS>
__global__ __launch_bounds__(1024) void wide_finalize_hits_kernel(unsigned long long* __restrict__ keys,
                                                                  const float* __restrict__ bias, int N,
                                                                  float* __restrict__ out, int* __restrict__ arg,
                                                                  int* __restrict__ hits, int* __restrict__ hoff) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fsm[];
  const int NP = (N + 1) & ~1;
  uint16_t* s_cnt = reinterpret_cast<uint16_t*>(fsm);               // [16][NP]
  int* s_off = reinterpret_cast<int*>(fsm + (size_t)32 * NP);       // [N + 1]
  int* s_wsum = s_off + N + 1;                                      // [17]
  const int b = blockIdx.x, co = threadIdx.x, lane = co & 63, chunk = co >> 6;
  const size_t e = (size_t)b * 1024 + co;
  const unsigned long long k = keys[e];
  keys[e] = 0ull;
  const unsigned u = (unsigned)(k >> 32);
  const float v = __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
  const float o = v != v ? v : fmaxf(v + bias[co], 0.f);
  const int ar = (int)(0xFFFFFFFFu - (unsigned)(k & 0xFFFFFFFFull));
  out[e] = o;
  arg[e] = ar;
  for (int i = co; i < 8 * NP; i += 1024) reinterpret_cast<int*>(s_cnt)[i] = 0;
  __syncthreads();
  const bool live = !(o <= 0.f);   // (NaN counts: the poison must reach the gradient)
  const unsigned long long lt = (1ull << lane) - 1ull;
  uint16_t* my = s_cnt + (size_t)chunk * NP;
  int local[TAPS], colm[TAPS];
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
    const int m = ar - TAPS / 2 + tap;
    const bool valid = live && m >= 0 && m < N;
    unsigned long long peers = __ballot(valid);
#pragma unroll
    for (int bit = 0; bit < 12; ++bit) {
      const bool one = (m >> bit) & 1;
      const unsigned long long bb = __ballot(one);
      peers &= one ? bb : ~bb;
    }
    const int base = valid ? (int)my[m] : 0;
    __builtin_amdgcn_wave_barrier();
    if (valid && (peers & lt) == 0ull) my[m] = (uint16_t)(base + __popcll(peers));
    __builtin_amdgcn_wave_barrier();
    local[tap] = base + __popcll(peers & lt);
    colm[tap] = valid ? m : -1;
  }
  __syncthreads();
  for (int m = co; m < N; m += 1024) {   // counts of the chunks -> exclusive prefixes; the column's total
    int run = 0;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      const int t = s_cnt[(size_t)c * NP + m];
      s_cnt[(size_t)c * NP + m] = (uint16_t)run;
      run += t;
    }
    s_off[m] = run;
  }
  __syncthreads();
  {   // exclusive scan of the N totals: thread t owns columns [t per, (t + 1) per)
    const int per = (N + 1023) >> 10, m_lo = co * per;
    int sum = 0;
    for (int j = 0; j < per; ++j) sum += m_lo + j < N ? s_off[m_lo + j] : 0;
    int incl = sum;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int t = __shfl_up(incl, d, 64);
      if (lane >= d) incl += t;
    }
    if (lane == 63) s_wsum[chunk] = incl;
    __syncthreads();
    if (co == 0) {
      int run = 0;
      for (int w = 0; w < 16; ++w) {
        const int t = s_wsum[w];
        s_wsum[w] = run;
        run += t;
      }
      s_wsum[16] = run;
    }
    __syncthreads();
    int run = s_wsum[chunk] + incl - sum;
    for (int j = 0; j < per; ++j)
      if (m_lo + j < N) {
        const int t = s_off[m_lo + j];
        s_off[m_lo + j] = run;
        run += t;
      }
    if (co == 0) s_off[N] = s_wsum[16];
  }
  __syncthreads();
  int* ho = hoff + (size_t)b * (N + 1);
  for (int m = co; m <= N; m += 1024) ho[m] = s_off[m];
  int* hl = hits + (size_t)b * 1024 * TAPS;
#pragma unroll
  for (int tap = 0; tap < TAPS; ++tap) {
    const int m = colm[tap];
    if (m >= 0) hl[s_off[m] + s_cnt[(size_t)chunk * NP + m] + local[tap]] = (co * TAPS + tap) | (m << 16);
  }
}

// ------------------------------------------------------------------------------------------
// Sparse backward.  One workgroup per (instance, WB_COLS-point tile), split in NP parts of WB_COLS/NP columns with 128
// threads each; thread (ci, part) owns row ci of its part, so no two threads ever touch the same accumulator
// and the summation order is fixed (deterministic, no atomics).
// Per block of WB_BLOCK output channels: the first wave of each part compacts the (channel, tap) pairs whose
// arg-max column falls into its part into an LDS hit list (ballot + popcount, in (co, tap) order, the upstream
// gradient stored next to it); then the 128 threads of the part walk ONLY the hits, WB_BATCH independent
// weight-row loads in flight.  The walk is L2-latency bound and the write-out bandwidth bound; they do not overlap
// inside a workgroup, so the tile is kept small (64 columns, 33 KB of accumulators): three workgroups per CU.
// ------------------------------------------------------------------------------------------
constexpr int WB_BATCH = 16;

template <int TAPS, int NP, int WB_BLOCK, int WB_COLS>   // WB_BLOCK: output channels per compaction round
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
  if (a.Zmask && WB_COLS == 64) {   // the gate as one bit per element: one 64-bit word per (channel, this tile)
    const unsigned long long* mk = a.Zmask + ((size_t)b * ((a.N + 63) >> 6) + blockIdx.x) * WM_CI;   // [B][tile][ci]
    const bool vec = (a.ldX & 3) == 0;
#pragma unroll 4
    for (int e = tid; e < WM_CI * (WB_COLS / 4); e += 128 * NP) {
      const int c = e / (WB_COLS / 4), j = (e - c * (WB_COLS / 4)) * 4;
      const int m = m0 + j;
      const float* sa = s_acc + c * (WB_COLS + 1) + j;
      const unsigned bits = (unsigned)(mk[c] >> j) & 15u;
      if (vec && m + 3 < a.N) {
        float4 v;
        v.x = bits & 1u ? sa[0] : 0.f;
        v.y = bits & 2u ? sa[1] : 0.f;
        v.z = bits & 4u ? sa[2] : 0.f;
        v.w = bits & 8u ? sa[3] : 0.f;
        *reinterpret_cast<float4*>(dX + (size_t)c * a.ldX + m) = v;
      } else {
        for (int i = 0; i < 4; ++i)
          if (m + i < a.N) dX[(size_t)c * a.ldX + m + i] = (bits >> i) & 1u ? sa[i] : 0.f;
      }
    }
    return;
  }
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

// ------------------------------------------------------------------------------------------
// Sparse backward, second form (the default): the same sums in the same order, accumulated in REGISTERS.
// One workgroup per (instance, 64-point tile), 8 wavefronts.  (1) the (channel, tap) pairs whose arg-max lands in the
// tile are compacted into a flat list in a fixed order (ballot + popcount); (2) a stable placement groups them by
// column (wave w owns 8 columns and passes over the flat list in order: no sort, no order left to chance); (3) the
// grouped list is cut into equal shares, one per wave: eight weight rows (512 B each, lane = two input channels) are in
// flight per step, the running column's sum lives in two registers and is stored to the LDS tile when the column
// changes -- no read-modify-write chain through LDS, which is what bounded the first form (97 / 151 us for 64 / 192
// hits per tile); (4) the tile is written out with the relu gate of the layer input as before.
// ------------------------------------------------------------------------------------------
constexpr int BW2_WAVES = 8, BW2_THREADS = 64 * BW2_WAVES;
template <int TAPS>
__global__ __launch_bounds__(BW2_THREADS) void wide_max_bwd2_kernel(WideBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int COLS = 64, PITCH = COLS + 1;
  float* s_acc = smem;                                                  // [128 ci][65]; before the walk: the flat list
  int* s_flat = reinterpret_cast<int*>(smem);                           // [Co * TAPS] hits in (channel chunk, tap, channel) order
  float* s_fg = smem + a.Co * TAPS;                                     // [Co * TAPS] their upstream gradients
  const int region = max(WM_CI * PITCH, 2 * a.Co * TAPS);
  int* s_list = reinterpret_cast<int*>(smem + region);                  // [Co * TAPS] (co * TAPS + tap) | (column << 16), by column
  // (a hit's upstream gradient is read again from g in the walk -- an L2 hit beside the weight row it multiplies --
  //  instead of being carried in a second list: one more workgroup per CU)
  int* s_off = s_list + a.Co * TAPS;                                    // [COLS + 1] column counts -> start offsets
  int* s_wcnt = s_off + COLS + 1;                                       // [BW2_WAVES + 1] hits found by each wave -> offsets
  int* s_sidecol = s_wcnt + BW2_WAVES + 1;                              // [BW2_WAVES]
  float* s_side = reinterpret_cast<float*>(s_sidecol + BW2_WAVES);      // [BW2_WAVES][128]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, b = blockIdx.y, m0 = blockIdx.x * COLS;
  const float* gb = a.g + (size_t)b * a.Co;
  const int* argb = a.arg + (size_t)b * a.Co;
  if (tid <= COLS) s_off[tid] = 0;
  __syncthreads();
  // (1) ordered compaction.  Wave w looks at channels [w Co/8, (w+1) Co/8) in chunks of 64; a hit's place in the flat
  // list follows (chunk, tap, channel): ballot + popcount, no ordering left to chance.
  const int cpw = (a.Co + BW2_WAVES - 1) / BW2_WAVES, c_lo = wave * cpw, c_hi = min(a.Co, c_lo + cpw);
  const unsigned long long lt = (1ull << lane) - 1ull;
  {
    int cnt = 0;
    for (int c0 = c_lo; c0 < c_hi; c0 += 64) {
      const int co = c0 + lane;
      const bool in = co < c_hi;
      const float g = in ? gb[co] : 0.f;
      const int base = (in ? argb[co] : 0) - TAPS / 2 - m0;
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int c = base + tap;
        const bool hit = g != 0.f && c >= 0 && c < COLS && m0 + c < a.N;
        cnt += __popcll(__ballot(hit));
        if (hit) atomicAdd(&s_off[c], 1);
      }
    }
    if (lane == 0) s_wcnt[wave] = cnt;
  }
  __syncthreads();
  if (tid == 0) {
    int run = 0;
    for (int w = 0; w < BW2_WAVES; ++w) {
      const int c = s_wcnt[w];
      s_wcnt[w] = run;
      run += c;
    }
    s_wcnt[BW2_WAVES] = run;
  }
  if (tid >= 64 && tid < 128) {   // exclusive scan of the 64 column counts (wave 1)
    const int cnt = s_off[lane];
    int incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o, 64);
      if (lane >= o) incl += v;
    }
    s_off[lane] = incl - cnt;
    if (lane == 63) s_off[COLS] = incl;
  }
  __syncthreads();
  {
    int pos = s_wcnt[wave];
    for (int c0 = c_lo; c0 < c_hi; c0 += 64) {
      const int co = c0 + lane;
      const bool in = co < c_hi;
      const float g = in ? gb[co] : 0.f;
      const int base = (in ? argb[co] : 0) - TAPS / 2 - m0;
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int c = base + tap;
        const bool hit = g != 0.f && c >= 0 && c < COLS && m0 + c < a.N;
        const unsigned long long mask = __ballot(hit);
        if (hit) {
          const int slot = pos + __popcll(mask & lt);
          s_flat[slot] = (co * TAPS + tap) | (c << 16);
          s_fg[slot] = g;
        }
        pos += __popcll(mask);
      }
    }
  }
  __syncthreads();
  // (2) stable placement by column: wave w owns columns 8 w .. 8 w + 7 and passes over the flat list in order, so
  // every column's list keeps the flat order
  const int total = s_off[COLS];
  {
    int basec[COLS / BW2_WAVES];
#pragma unroll
    for (int j = 0; j < COLS / BW2_WAVES; ++j) basec[j] = s_off[(COLS / BW2_WAVES) * wave + j];
    for (int i0 = 0; i0 < total; i0 += 64) {
      const int i = i0 + lane;
      const int e = i < total ? s_flat[i] : -1;
      const int col = e >> 16;     // -1 for the padding lanes
#pragma unroll
      for (int j = 0; j < COLS / BW2_WAVES; ++j) {
        const bool mine = col == (COLS / BW2_WAVES) * wave + j;
        const unsigned long long mask = __ballot(mine);
        if (mine) {
          const int slot = basec[j] + __popcll(mask & lt);
          s_list[slot] = e;
        }
        basec[j] += __popcll(mask);
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < WM_CI * PITCH; e += BW2_THREADS) s_acc[e] = 0.f;   // the flat list is done with: the tile
  __syncthreads();
  // (3) walk: the flat list is cut into equal shares, one per wave -- arg-max columns cluster on a few "critical" points,
  // so a split by columns leaves most waves idle.  A column that continues from the previous wave's share is summed
  // into the wave's side row and added to the tile afterwards, in wave order (fixed order: deterministic).
  {
    const int per = (total + BW2_WAVES - 1) / BW2_WAVES;
    const int lo = min(total, wave * per), hi = min(total, lo + per);
    constexpr int U = 8;
    const int ldW = a.ldW ? a.ldW : WM_CI;   // (co * TAPS + tap) -th row of 128 input channels
    int cur = -1;
    float acc0 = 0.f, acc1 = 0.f;
    // the column of hit lo started in an earlier share <=> lo is not the first entry of that column's list
    int side_col = -1;
    if (lo < hi) {
      const int c0 = s_list[lo] >> 16;
      if (lo > s_off[c0]) side_col = c0;
    }
    bool in_side = side_col >= 0;
    auto flush = [&]() {
      if (cur < 0) return;
      if (in_side) {
        s_side[wave * WM_CI + 2 * lane] = acc0;
        s_side[wave * WM_CI + 2 * lane + 1] = acc1;
        in_side = false;
      } else {
        s_acc[(2 * lane) * PITCH + cur] = acc0;
        s_acc[(2 * lane + 1) * PITCH + cur] = acc1;
      }
    };
    for (int h0 = lo; h0 < hi; h0 += U) {
      float2 w[U];
      float gg[U];
      int cc[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool ok = h0 + u < hi;
        const int e = s_list[ok ? h0 + u : hi - 1];
        w[u] = *reinterpret_cast<const float2*>(a.W + (size_t)(e & 0xffff) * ldW + 2 * lane);
        gg[u] = ok ? gb[(e & 0xffff) / TAPS] : 0.f;
        cc[u] = ok ? e >> 16 : -2;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (cc[u] == -2) break;
        if (cc[u] != cur) {
          flush();
          cur = cc[u];
          acc0 = 0.f;
          acc1 = 0.f;
        }
        acc0 += w[u].x * gg[u];
        acc1 += w[u].y * gg[u];
      }
    }
    flush();
    if (lane == 0) s_sidecol[wave] = side_col;
  }
  __syncthreads();
  if (tid < WM_CI) {
    for (int w = 1; w < BW2_WAVES; ++w) {
      const int c = s_sidecol[w];
      if (c >= 0) s_acc[tid * PITCH + c] += s_side[w * WM_CI + tid];
    }
  }
  __syncthreads();
  // (4) write out with the relu gate of the layer input: 4 consecutive points per thread (16-byte stores)
  float* dX = a.dX + (size_t)b * a.sXb;
  const bool vecx = (a.ldX & 3) == 0;
  if (a.Zmask) {
    const unsigned long long* mk = a.Zmask + ((size_t)b * ((a.N + 63) >> 6) + blockIdx.x) * WM_CI;   // [B][tile][ci]
#pragma unroll 4
    for (int e = tid; e < WM_CI * (COLS / 4); e += BW2_THREADS) {
      const int c = e / (COLS / 4), j = (e - c * (COLS / 4)) * 4, m = m0 + j;
      const float* sa = s_acc + c * PITCH + j;
      const unsigned bits = (unsigned)(mk[c] >> j) & 15u;
      if (vecx && m + 3 < a.N) {
        float4 v;
        v.x = bits & 1u ? sa[0] : 0.f;
        v.y = bits & 2u ? sa[1] : 0.f;
        v.z = bits & 4u ? sa[2] : 0.f;
        v.w = bits & 8u ? sa[3] : 0.f;
        *reinterpret_cast<float4*>(dX + (size_t)c * a.ldX + m) = v;
      } else {
        for (int i = 0; i < 4; ++i)
          if (m + i < a.N) dX[(size_t)c * a.ldX + m + i] = (bits >> i) & 1u ? sa[i] : 0.f;
      }
    }
    return;
  }
  const float* Z = a.Z + (size_t)b * a.sZb;
  const bool vec = vecx && (a.ldZ & 3) == 0;
#pragma unroll 4
  for (int e = tid; e < WM_CI * (COLS / 4); e += BW2_THREADS) {
    const int c = e / (COLS / 4), j = (e - c * (COLS / 4)) * 4, m = m0 + j;
    const float* sa = s_acc + c * PITCH + j;
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

void launch_wide_finalize(const WideArgs& a, hipStream_t s) {
  if (a.hits && a.hoff && a.Co == 1024 && a.N <= WIDE_HITS_MAXN && (a.taps == 1 || a.taps == 3)) {
    const int NP = (a.N + 1) & ~1;
    const size_t lds = (size_t)32 * NP + ((size_t)a.N + 1 + 17) * sizeof(int);
    if (a.taps == 1) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wide_finalize_hits_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(wide_finalize_hits_kernel<1>, dim3(a.B), dim3(1024), lds, s, a.keys, a.bias, a.N, a.out, a.arg,
                         a.hits, a.hoff);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wide_finalize_hits_kernel<3>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(wide_finalize_hits_kernel<3>, dim3(a.B), dim3(1024), lds, s, a.keys, a.bias, a.N, a.out, a.arg,
                         a.hits, a.hoff);
    }
    return;
  }
  const int total = a.B * a.Co;
  hipLaunchKernelGGL(wide_finalize_kernel, dim3((total + 255) / 256), dim3(256), 0, s, a.keys, a.bias, a.Co, total,
                     a.out, a.arg);
}

int launch_wide_max(const WideArgs& a, hipStream_t s) {
  if (a.Wh16 && !a.W2h) return launch_wide_max_split16(a, s);
  if (a.Wh) return launch_wide_max_split(a, s);
  if (a.Co != 8 * WM_CO || (a.taps != 1 && a.taps != 3) || !a.keys) return GEOA3_ENOSUPPORT;
  const int tag = a.taps == 3 ? GEOA3_PROF_CONV5 : GEOA3_PROF_TNETWIDE;
  geoa3_prof_begin(tag, s);
  if (!a.keys_clean &&
      hipMemsetAsync(a.keys, 0, (size_t)a.B * a.Co * sizeof(unsigned long long), s) != hipSuccess)
    return GEOA3_ELAUNCH;
  const size_t lds = (size_t)WM_CI * W2_XP * sizeof(float);
  constexpr int OCC = 3, SLOTS = 32 * OCC;   // per XCD: 768 resident workgroups of 4 waves (occupancy / unit size
                                             // picked on hardware: 4 waves per SIMD and quarter units were slower)
  if (a.taps == 1) hipLaunchKernelGGL((wide_max2_kernel<1, OCC, 8>), dim3(SLOTS * 8), dim3(WM_THREADS), lds, s, a, SLOTS);
  else hipLaunchKernelGGL((wide_max2_kernel<3, OCC, 4>), dim3(SLOTS * 8), dim3(WM_THREADS), lds, s, a, SLOTS);
  launch_wide_finalize(a, s);
  geoa3_prof_end(tag, s);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

template <int TAPS, int NP, int WB_BLOCK, int WB_COLS>
static void launch_wide_bwd_variant(const WideBwdArgs& a, hipStream_t s) {
  dim3 grid((a.N + WB_COLS - 1) / WB_COLS, a.B);
  const size_t lds = ((size_t)WM_CI * (WB_COLS + 1) + 2 * (size_t)NP * WB_BLOCK * TAPS + NP) * sizeof(float);
  auto kern = wide_max_bwd_kernel<TAPS, NP, WB_BLOCK, WB_COLS>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, grid, dim3(128 * NP), lds, s, a);
}

int launch_wide_max_bwd(const WideBwdArgs& a, hipStream_t s) {
  if (a.taps != 1 && a.taps != 3) return GEOA3_ENOSUPPORT;
  // tile width / parts / compaction block picked on hardware (tools/bench_widebwd.py): 64-column tiles leave room for
  // three workgroups per CU, so one workgroup's write-out overlaps its neighbours' hit walks (95 / 150 us)
  if (a.form == 1) {   // the first form (LDS accumulation), kept for tools/bench_widebwd.py and the cross-check test
    if (a.taps == 1) launch_wide_bwd_variant<1, 2, 256, 64>(a, s);
    else launch_wide_bwd_variant<3, 2, 128, 64>(a, s);
  } else {
    if (a.Co * a.taps > 0xffff) return GEOA3_ENOSUPPORT;
    dim3 grid((a.N + 63) / 64, a.B);
    const size_t region = (size_t)WM_CI * 65 > 2 * (size_t)a.Co * a.taps ? (size_t)WM_CI * 65 : 2 * (size_t)a.Co * a.taps;
    const size_t lds = (region + (size_t)a.Co * a.taps + 65 + 2 * BW2_WAVES + 1 + BW2_WAVES * WM_CI + 3) * sizeof(float);
    if (a.taps == 1) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wide_max_bwd2_kernel<1>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(wide_max_bwd2_kernel<1>, grid, dim3(BW2_THREADS), lds, s, a);
    } else {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(wide_max_bwd2_kernel<3>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL(wide_max_bwd2_kernel<3>, grid, dim3(BW2_THREADS), lds, s, a);
    }
  }
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_debug_wide_bwd(const float* g, const int32_t* arg, const float* W, const float* Z, float* dX, int B,
                                    int N, int taps, int form, void* stream) {
  WideBwdArgs a{};
  a.form = form;
  a.g = g; a.arg = arg; a.W = W;
  a.Z = Z; a.sZb = (long)128 * N; a.ldZ = N;
  a.dX = dX; a.sXb = (long)128 * N; a.ldX = N;
  a.Co = 1024; a.N = N; a.B = B; a.taps = taps;
  return launch_wide_max_bwd(a, geoa3_stream(stream));
}
