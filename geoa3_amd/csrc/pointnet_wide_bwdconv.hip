// The sparse backward of a 1024-wide layer AND the 128 -> 64 layer behind it in one kernel (gfx950): what
// wide_max_bwd2_kernel (pointnet_wide.hip) writes out -- the [B,128,N] gradient, 131 MB at B = 250 -- is consumed by a
// gated 128 -> 64 convolution that reads it back.  Here the 64-column tile stays in LDS (column-major), is gated, and
// goes through W2^T on the matrix core; only the [B,64,N] result leaves (a third of the bytes of the two kernels, one
// launch).  One workgroup per (instance, 64-point tile):
//   lists   the (channel, tap) pairs whose arg-max column falls into the tile, by column, in (chunk of 64 channels, tap,
//           channel) order inside a column.  PRE: one contiguous segment of the lists the FORWARD's finalize pass built
//           per instance (wide_finalize_hits_kernel, pointnet_wide.hip), read in place; else built here, phases (1)-(2)
//           of wide_max_bwd2_kernel (N > 4096, or the caller opted out: GEOA3_PN_NO_PRE_LISTS) -- same lists either way;
//   walk    (3) the list cut into eight equal shares, one per wave; a column's sum in registers, list entries through
//           SGPRs (scalar row / gradient addresses), eight weight rows in flight; shares that start inside a column go
//           through side rows and are added in wave order;
//   product (4) tile gated by relu bits, scaled by a power of two from its own maximum and split ONCE into fp16 hi / lo
//           images written over it; W2^T's fragments arrive split from the host (WideBwdArgs::W2th); four waves take one
//           32 x 32 quadrant each on the f16 matrix core; the result leaves gated (or, first-layer form, is contracted
//           with w1 into dx in the same kernel).
// Same sums in the same order as wide_max_bwd2_kernel + the 128 -> 64 convolution: bit-identical to the two-kernel path.
// Reference: the autograd of Model/PointNet.py:80-82,146-147 (conv3 / conv5 + max, conv2 / conv4 + relu).
#include "pointnet_kernels.h"
#include <type_traits>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
constexpr int WM_CI = 128;

__device__ __forceinline__ unsigned bc_exp(float m) {
  const unsigned E = (__float_as_uint(m) >> 23) & 0xffu;
  return E < 14u ? 14u : (E > 254u ? 254u : E);
}
__device__ __forceinline__ float bc_scale(unsigned E) { return __uint_as_float((267u - E) << 23); }
__device__ __forceinline__ float bc_unscale(unsigned E) { return __uint_as_float((E - 13u) << 23); }

constexpr int BW2_WAVES = 8, BW2_THREADS = 64 * BW2_WAVES;
#ifdef GEOA3_BC_STAMPS   // probe build: s_memtime at the phase boundaries of a few workgroups (tools/bc_stamps.py)
__device__ unsigned long long g_bc_stamps[3 * 8 * 16];
#define BC_STAMP(i)                                                                                          \
  do {                                                                                                       \
    if (tid == 0 && blockIdx.x == 5 && (blockIdx.y & 31) == 7 && blockIdx.y < 256)                            \
      g_bc_stamps[((TAPS == 3 ? 2 : (GF ? 1 : 0)) * 8 + (blockIdx.y >> 5)) * 16 + (i)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define BC_STAMP(i)
#endif
constexpr int BC_PT = WM_CI + 4;   // floats per column of the tile (column-major here: the B operand reads 8 consecutive ci)
// PRE: the hit lists come from the forward (WideBwdArgs::hits / hoff); GF: the 64-channel activation is the 3-channel first
// layer (recomputed gate) and its backward finishes here: dx3 += w1^T (gated result), no [B,64,N] output
template <int TAPS, bool GF = false, bool PRE = false>
__device__ __forceinline__ void wide_bwd_conv_body(const WideBwdArgs& a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int COLS = 64;
  float* s_acc = smem;                                                  // [64 columns][BC_PT]; before the walk: the flat list
  int* s_flat = reinterpret_cast<int*>(smem);                           // [Co * TAPS] hits in (channel chunk, tap, channel) order
  float* s_fg = smem + a.Co * TAPS;                                     // [Co * TAPS] their upstream gradients
  const int region = max(COLS * BC_PT, 2 * a.Co * TAPS);
  // the hits by column: (co * TAPS + tap) | (column << 16); one tap: co | (column << 10) in 16 bits -- with the side rows
  // doubling as the first layer's table (GF) the workgroup stays under 40 KB of LDS: four per CU instead of three
  using list_t = typename std::conditional<TAPS == 1, uint16_t, int>::type;
  constexpr int LSH = TAPS == 1 ? 10 : 16;
  list_t* s_list = reinterpret_cast<list_t*>(smem + region);            // [Co * TAPS]
  // (a hit's upstream gradient is read again from g in the walk -- an L2 hit beside the weight row it multiplies --
  //  instead of being carried in a second list: 12 KB of LDS less at three taps, one more workgroup per CU)
  // (PRE: no list in LDS -- the walk reads the forward's segment itself: 12 KB less at three taps, four workgroups per CU)
  int* s_off = reinterpret_cast<int*>(smem + region) + (PRE ? 0 : (TAPS == 1 ? a.Co / 2 : a.Co * TAPS));   // [COLS + 1] column counts -> start offsets
  int* s_wcnt = s_off + COLS + 1;                                       // [BW2_WAVES + 1] hits found by each wave -> offsets
  int* s_sidecol = s_wcnt + BW2_WAVES + 1;                              // [BW2_WAVES]
  float* s_side = reinterpret_cast<float*>(s_sidecol + BW2_WAVES);      // [BW2_WAVES][128]
  float* s_mx = s_side + BW2_WAVES * WM_CI;                             // [8] tile maxima of the waves
  const int tid = threadIdx.x, lane = tid & 63, b = blockIdx.y, m0 = blockIdx.x * COLS;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (uniform: the walk's list entries, rows and columns stay in SGPRs)
  // requested now, used at the very end: the gate word of output row `lane`
  unsigned long long gw2 = 0ull;
  if (!GF && wave < 4) gw2 = a.Zmask2[((size_t)b * ((a.N + 63) >> 6) + blockIdx.x) * 64 + lane];
  // ... and the gate words of tile rows 2 lane, 2 lane + 1 of the 128-channel activation
  const int tiles = (a.N + 63) >> 6;
  const ulonglong2 mkw = reinterpret_cast<const ulonglong2*>(a.Zmask + ((size_t)b * tiles + blockIdx.x) * WM_CI)[lane];   // rows 2 lane, 2 lane + 1
  const int qt = wave & 1, qc = (wave >> 1) & 1;     // waves 0-3: output rows 32 qt .., columns 32 qc ..

  // GF: [64] (w1 row, b1), then [2][32] partial d x -- in the side rows, once the walk is done with them
  float4* s_w1 = reinterpret_cast<float4*>((reinterpret_cast<uintptr_t>(s_side) + 15) & ~(uintptr_t)15);
  const float* gb = a.g + (size_t)b * a.Co;
  const int* argb = a.arg + (size_t)b * a.Co;
  const unsigned long long lt = (1ull << lane) - 1ull;
  BC_STAMP(0);
  constexpr int U = 8;            // list entries (weight rows) in flight per wave in the walk
  const int* seg = nullptr;       // PRE: the tile's segment of the instance's sorted hits
  int ev0 = 0;                    // PRE: the wave's first U entries, requested before the tile is cleared
  if constexpr (PRE) {
    // the tile's lists are one contiguous segment of the instance's sorted hits (wide_finalize_hits_kernel)
    const int* ho = a.hoff + (size_t)b * (a.N + 1);
    const int h0 = ho[m0], h1 = ho[min(m0 + COLS, a.N)], hv = ho[min(m0 + min(tid, COLS), a.N)];
    seg = a.hits + (size_t)b * a.Co * TAPS + h0;
    {
      const int tot = h1 - h0, per = (tot + BW2_WAVES - 1) / BW2_WAVES;
      const int lo = min(tot, wave * per), hi = min(tot, lo + per);
      if (lo < hi) ev0 = seg[min(lo + (lane & (U - 1)), hi - 1)];
    }
    for (int e = tid; e < COLS * BC_PT; e += BW2_THREADS) s_acc[e] = 0.f;   // (while the offsets are on their way)
    if (tid <= COLS) s_off[tid] = hv - h0;
    BC_STAMP(1);
    __syncthreads();
    BC_STAMP(2);
  } else {
  if (tid <= COLS) s_off[tid] = 0;
  __syncthreads();
  {
  // (1) ordered compaction.  Wave w looks at channels [w Co/8, (w+1) Co/8) in chunks of 64; a hit's place in the flat
  // list follows (chunk, tap, channel): ballot + popcount, no ordering left to chance.
  const int cpw = (a.Co + BW2_WAVES - 1) / BW2_WAVES, c_lo = wave * cpw, c_hi = min(a.Co, c_lo + cpw);
  // a wave's channels are at most two chunks of 64 for Co = 1024: their (gradient, arg-max) pairs are loaded ONCE, both
  // chunks in flight together, and serve the counting pass and the placement pass (they used to be re-read: two dependent
  // L2 round trips of the ~16 us a workgroup lives)
  constexpr int MAXCH = 2;
  float gch[MAXCH];
  int bch[MAXCH];
  const bool regs = cpw <= 64 * MAXCH;
#pragma unroll
  for (int ch = 0; ch < MAXCH; ++ch) {
    const int co = c_lo + 64 * ch + lane;
    const bool in = regs && co < c_hi;
    gch[ch] = in ? gb[co] : 0.f;
    bch[ch] = (in ? argb[co] : 0) - TAPS / 2 - m0;
  }
  {
    int cnt = 0;
    int chn = 0;
    for (int c0 = c_lo; c0 < c_hi; c0 += 64, ++chn) {
      const int co = c0 + lane;
      const bool in = co < c_hi;
      const float g = regs ? (chn == 0 ? gch[0] : gch[1]) : (in ? gb[co] : 0.f);
      const int base = regs ? (chn == 0 ? bch[0] : bch[1]) : ((in ? argb[co] : 0) - TAPS / 2 - m0);
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
    int chn = 0;
    for (int c0 = c_lo; c0 < c_hi; c0 += 64, ++chn) {
      const int co = c0 + lane;
      const bool in = co < c_hi;
      const float g = regs ? (chn == 0 ? gch[0] : gch[1]) : (in ? gb[co] : 0.f);
      const int base = regs ? (chn == 0 ? bch[0] : bch[1]) : ((in ? argb[co] : 0) - TAPS / 2 - m0);
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
  }
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
          s_list[slot] = TAPS == 1 ? (list_t)((e & 0xffff) | ((e >> 16) << LSH)) : (list_t)e;
        }
        basec[j] += __popcll(mask);
      }
    }
  }
  __syncthreads();
  for (int e = tid; e < COLS * BC_PT; e += BW2_THREADS) s_acc[e] = 0.f;   // the flat list is done with: the tile
  __syncthreads();
  }
  const int total = s_off[COLS];
  // (3) walk: the flat list is cut into equal shares, one per wave -- arg-max columns cluster on a few "critical" points,
  // so a split by columns leaves most waves idle.  A column that continues from the previous wave's share is summed
  // into the wave's side row and added to the tile afterwards, in wave order (fixed order: deterministic).
  {
    const int per = (total + BW2_WAVES - 1) / BW2_WAVES;
    const int lo = min(total, wave * per), hi = min(total, lo + per);
    const int ldW = a.ldW ? a.ldW : WM_CI;   // (co * TAPS + tap) -th row of 128 input channels
    int cur = -1;
    float acc0 = 0.f, acc1 = 0.f;
    // lane u holds entry h0 + u of the batch at h0 (PRE: as the forward wrote it, (co TAPS + tap) | column << 16, read from
    // global one batch ahead; else the tile's own list in LDS, row | tile column << LSH); the entries travel through SGPRs
    auto load_ev = [&](int h0) -> int {
      const int i = min(h0 + (lane & (U - 1)), hi - 1);
      if constexpr (PRE) return seg[i];
      else return (int)s_list[i];
    };
    auto dec = [&](int e) -> int {
      if constexpr (PRE) return (e & 0xffff) | (((e >> 16) - m0) << LSH);
      else return e;
    };
    int evn = 0;
    if (lo < hi) evn = PRE ? ev0 : load_ev(lo);
    // the column of hit lo started in an earlier share <=> lo is not the first entry of that column's list
    int side_col = -1;
    if (lo < hi) {
      const int c0 = dec(__builtin_amdgcn_readlane(evn, 0)) >> LSH;
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
        *reinterpret_cast<float2*>(s_acc + cur * BC_PT + 2 * lane) = make_float2(acc0, acc1);
      }
    };
    for (int h0 = lo; h0 < hi; h0 += U) {
      float2 w[U];
      float gg[U];
      int cc[U];
      // (v_readlane: the row address, the gradient's address and the column are scalar -- the walk was a quarter of the
      //  kernel's VALU instructions)
      const int ev = evn;
      if (h0 + U < hi) evn = load_ev(h0 + U);
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const bool ok = h0 + u < hi;
        const int e = dec(__builtin_amdgcn_readlane(ev, u));
        const int er = e & ((1 << LSH) - 1);
        w[u] = *reinterpret_cast<const float2*>(a.W + (size_t)er * ldW + 2 * lane);
        gg[u] = ok ? gb[er / TAPS] : 0.f;
        cc[u] = ok ? e >> LSH : -2;
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
    BC_STAMP(3);
  }
  __syncthreads();
  BC_STAMP(4);
  if (tid < WM_CI) {
    for (int w = 1; w < BW2_WAVES; ++w) {
      const int c = s_sidecol[w];
      if (c >= 0) s_acc[c * BC_PT + tid] += s_side[w * WM_CI + tid];
    }
  }
  __syncthreads();
  BC_STAMP(5);
  if (GF && tid < 64) s_w1[tid] = make_float4(a.w1[3 * tid], a.w1[3 * tid + 1], a.w1[3 * tid + 2], a.b1[tid]);    // (read behind the next barriers)
  // (4) the 128 -> 64 layer behind it, on the tile while it is in LDS: gate by the relu bits of the 128-channel
  // activation, tile maximum -> power-of-two scale, and the tile is split ONCE, by all eight waves, into fp16 hi / lo
  // images written over it (row c: 128 hi halves, 128 lo halves, in the 528 bytes its 132 floats had); four waves then
  // take one 32 x 32 quadrant of W2^T tile each on the f16 matrix core with ready operands (a*w = a_hi*w_hi + a_hi*w_lo +
  // a_lo*w_hi as in pointnet_conv_split.hip; W2^T's fragments come split from the host -- W2th -- or as fp32, split
  // here); the result leaves gated by the bits of the 64-channel activation.  (The waves used to split their B operand
  // themselves, two waves the same columns, and every workgroup W2^T: 40 % of the kernel's cycles, tools/bc_stamps.py.)
  // thread = (pair of rows 2 lane, 2 lane + 1; the wave's eight columns): a column is touched by ONE wave
  float tv[16];
  {
    float mx = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 8 * wave + j;
      const float2 v = *reinterpret_cast<const float2*>(s_acc + c * BC_PT + 2 * lane);
      const bool in = m0 + c < a.N;
      tv[2 * j] = ((mkw.x >> c) & 1ull) && in ? v.x : 0.f;
      tv[2 * j + 1] = ((mkw.y >> c) & 1ull) && in ? v.y : 0.f;
      mx = fmaxf(mx, fmaxf(__builtin_fabsf(tv[2 * j]), __builtin_fabsf(tv[2 * j + 1])));
    }
    mx = wave_max(mx);
    if (lane == 0) s_mx[wave] = mx;
  }
  __syncthreads();   // every thread holds its sixteen values: the rows may be overwritten
  BC_STAMP(6);
  float tm = s_mx[0];
#pragma unroll
  for (int w = 1; w < BW2_WAVES; ++w) tm = fmaxf(tm, s_mx[w]);
  const unsigned Ex = bc_exp(tm);
  const float sx = bc_scale(Ex);
  constexpr int BC_PH = 2 * BC_PT;   // halves per column of the images
  _Float16* s_img = reinterpret_cast<_Float16*>(s_acc);
  typedef _Float16 half2v __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float x0 = tv[2 * j] * sx, x1 = tv[2 * j + 1] * sx;
    const _Float16 h0 = (_Float16)x0, h1 = (_Float16)x1;
    _Float16* row = s_img + (8 * wave + j) * BC_PH + 2 * lane;
    *reinterpret_cast<half2v*>(row) = half2v{h0, h1};
    *reinterpret_cast<half2v*>(row + WM_CI) = half2v{(_Float16)(x0 - (float)h0), (_Float16)(x1 - (float)h1)};
  }
  __syncthreads();
  if (wave >= 4) return;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const _Float16* brow = s_img + (32 * qc + (lane & 31)) * BC_PH + (lane >> 5) * 8;
  {
    const half8* wf = reinterpret_cast<const half8*>(a.W2th) + (size_t)qt * 8 * 2 * 64 + lane;   // [qt][c][hi / lo][lane]
    constexpr int WQ = 2;   // k-steps of fragments in flight
    half8 wq[WQ][2];
#pragma unroll
    for (int c = 0; c + 1 < WQ; ++c) {
      wq[c][0] = wf[(2 * c) * 64];
      wq[c][1] = wf[(2 * c + 1) * 64];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c + WQ - 1 < 8) {
        wq[(c + WQ - 1) % WQ][0] = wf[(2 * (c + WQ - 1)) * 64];
        wq[(c + WQ - 1) % WQ][1] = wf[(2 * (c + WQ - 1) + 1) * 64];
      }
      const half8 wh = wq[c % WQ][0], wl = wq[c % WQ][1];
      const half8 xh = *reinterpret_cast<const half8*>(brow + c * 16);
      const half8 xl = *reinterpret_cast<const half8*>(brow + WM_CI + c * 16);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc, 0, 0, 0);
    }
  }
  BC_STAMP(7);
  const float un = bc_unscale(Ex) * a.w2th_unscale;
  // acc[r]: row 32 qt + (r&3) + 8 (r>>2) + 4 (lane>>5), column 32 qc + (lane&31); gate word of row l in lane l of gw2
  const int col = 32 * qc + (lane & 31), m = m0 + col;
  if constexpr (GF) {
    // gate = sign of the first layer, recomputed from the point (Model/PointNet.py:79 behind relu); q = w1^T (gated rows)
    // (lane and column are formed again here: kept alive across the matrix product they cost a spilled register pair)
    const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int mq = m0 + 32 * qc + (ln & 31);
    const bool in = mq < a.N;
    const float* xp = a.x3 + (size_t)b * 3 * a.N + (in ? mq : a.N - 1);
    const float x0 = xp[0], x1 = xp[a.N], x2 = xp[2 * (size_t)a.N];
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float4 w = s_w1[32 * qt + (r & 3) + 8 * (r >> 2) + 4 * (ln >> 5)];
      const float o = (w.x * x0 + w.y * x1 + w.z * x2 + w.w) > 0.f ? acc[r] * un : 0.f;
      q0 += w.x * o;
      q1 += w.y * o;
      q2 += w.z * o;
    }
    q0 += __shfl_xor(q0, 32, 64);    // the two row halves of the tile (lanes l, l + 32 hold the same column)
    q1 += __shfl_xor(q1, 32, 64);
    q2 += __shfl_xor(q2, 32, 64);
    float4* s_q = s_w1 + 64;          // [2 column blocks][32]: the partial sums of row tile 1
    __syncthreads();                  // (waves 4-7 have left: the barrier counts the four that remain) s_w1 is read
    if (qt == 1 && ln < 32) s_q[qc * 32 + ln] = make_float4(q0, q1, q2, 0.f);
    __syncthreads();
    if (qt == 0 && ln < 32 && in) {
      const float4 o = s_q[qc * 32 + ln];
      float* dxp = a.dx3 + (size_t)b * 3 * a.N + mq;    // += : the trunk's gradient is there already
      dxp[0] += q0 + o.x;
      dxp[a.N] += q1 + o.y;
      dxp[2 * (size_t)a.N] += q2 + o.z;
    }
    BC_STAMP(8);
    return;
  }
  float* Y = a.dY + (size_t)b * a.sYb + m;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int rr = 32 * qt + (r & 3) + 8 * (r >> 2);   // + 4 (lane >> 5)
    const unsigned g0 = qc ? (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(gw2 >> 32), rr)
                           : (unsigned)__builtin_amdgcn_readlane((int)(unsigned)gw2, rr);
    const unsigned g1 = qc ? (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(gw2 >> 32), rr + 4)
                           : (unsigned)__builtin_amdgcn_readlane((int)(unsigned)gw2, rr + 4);
    const unsigned gsel = lane < 32 ? g0 : g1;
    const int row = rr + 4 * (lane >> 5);
    if (m < a.N) Y[(size_t)row * a.ldY] = (gsel >> (lane & 31)) & 1u ? acc[r] * un : 0.f;
  }
  BC_STAMP(8);
}

template <int TAPS, bool PRE>
__global__ __launch_bounds__(BW2_THREADS) void wide_bwd_conv_kernel(WideBwdArgs a) {
  wide_bwd_conv_body<TAPS, false, PRE>(a);
}
// the first-layer form: its epilogue would take 72-74 registers (seven waves per SIMD = three workgroups per CU); held at 64
template <bool PRE>
__global__ __launch_bounds__(BW2_THREADS) __attribute__((amdgpu_waves_per_eu(8, 8))) void wide_bwd_conv_first_kernel(WideBwdArgs a) {
  wide_bwd_conv_body<1, true, PRE>(a);
}

}  // namespace

#ifdef GEOA3_BC_STAMPS
extern "C" int geoa3_debug_bc_stamps(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bc_stamps), sizeof(g_bc_stamps)) == hipSuccess ? 0 : -1;
}
#endif
int launch_wide_bwd_conv(const WideBwdArgs& a, hipStream_t s) {
  const bool gf = a.dx3 != nullptr;   // first-layer form: x3, w1, b1, dx3 instead of Zmask2 / dY
  if ((a.taps != 1 && a.taps != 3) || !a.Zmask || !a.W2th || a.Co * a.taps > 0xffff || (a.taps == 1 && a.Co > 1024) ||
      (gf ? (!a.x3 || !a.w1 || !a.b1 || a.taps != 1) : (!a.Zmask2 || !a.dY)))
    return GEOA3_ENOSUPPORT;
  dim3 grid((a.N + 63) / 64, a.B);
  const size_t region = (size_t)64 * BC_PT > 2 * (size_t)a.Co * a.taps ? (size_t)64 * BC_PT : 2 * (size_t)a.Co * a.taps;
  const bool pre = a.hits && a.hoff;
  const size_t lds = (region + (pre ? 0 : (a.taps == 1 ? (size_t)a.Co / 2 : (size_t)a.Co * a.taps)) + 65 + 2 * BW2_WAVES + 1 + BW2_WAVES * WM_CI + 16 + 3 + 4) * sizeof(float);
  auto go = [&](auto kern) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, grid, dim3(BW2_THREADS), lds, s, a);
  };
  if (a.taps == 1 && a.dx3) pre ? go(wide_bwd_conv_first_kernel<true>) : go(wide_bwd_conv_first_kernel<false>);
  else if (a.taps == 1) pre ? go(wide_bwd_conv_kernel<1, true>) : go(wide_bwd_conv_kernel<1, false>);
  else pre ? go(wide_bwd_conv_kernel<3, true>) : go(wide_bwd_conv_kernel<3, false>);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
