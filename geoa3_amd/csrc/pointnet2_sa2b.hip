// PointNet++ SSG, second set-abstraction level: the backward of the two hidden layers AND of the grouping in one pass,
// with the rows ordered by DESTINATION point (gfx950, round 6).
// Reference: pointnet2_modules.py:47-74 (group -> shared MLP (128+3) -> 128 -> 128 -> 256 -> max over the ball's 64 samples),
// pointnet2_utils.py:296-333 (QueryAndGroup), _ext-src/src/group_points_gpu.cu:43-64 (group_points_grad: the scatter-add of
// the grouped gradient back to the points), PointNetPP_ssg.py:68-76.
//
// The centre-major form (pointnet2_sa2.hip: sa2_bwd_kernel) wrote d a0 [B][128][128 centres x 64 samples] -- 1.05 GB at
// B = 250 -- for group_points_grad to read straight back and reduce to [B][512 points][128].  Nothing in
//     d a0[:, row] = gate0[row] . W1^T (gate1[row] . sum_{ch : arg[ch] = row} g[ch] W2[ch][:])
// couples two rows (row = (centre, sample)), so the rows may be taken in any order: here in the order of the point they
// gather FROM.  Then the grouping's scatter-add is a sum over CONSECUTIVE rows, finished on chip, and each point's 512-byte
// gradient row is written once.  Only rows some pooled channel points at are formed at all (a ball holds 25-64 distinct
// points, its 256 channels pick 22-40 of them: NOTEBOOK 10): the matrix-core work of the level's backward halves.
//   sa2b_prep_kernel   one workgroup per cloud: the hit rows (centre, sample) counted, ranked by (destination, centre) through
//                      per-destination centre bit sets (deterministic: no arrival order anywhere), cut into P parts at
//                      destination boundaries and into tiles of 64 rows; the pooled gradient's live entries (channel, value)
//                      laid out row by row in that order
//   sa2b_bwd_kernel    one workgroup per (cloud, part), two tiles at a time: phase 1 the sparse W2 product into an fp32 tile
//                      in LDS (lane = k, W2 rows 32 entries ahead), phase 2 W1^T on the matrix core with split-fp16 operands
//                      (the tile's own power-of-two scale), gated result back into the tile, phase 3 two wavefronts walk the
//                      128 columns in order -- a running sum per output channel, stored when the destination changes (strictly
//                      sequential: the sums do not depend on where tiles or parts are cut) -- while the other two form each
//                      row's coordinate term W_x^T d a0 (3 floats per row: all the centre's d shift is needed for)
//   sa2b_centre_kernel d c[centre] -= sum over its hit rows of that term, sample ascending
// Deterministic and batch-independent: every sum has one fixed order that depends on the cloud alone.
#include "pointnet2_sa2_common.h"
#include "profile.h"

namespace {

constexpr int SB_M = 128;        // centres of the level
constexpr int SB_S = 64;         // samples per ball
constexpr int SB_N1 = 512;       // points the balls gather from (level 1's centroids)
constexpr int SB_ROWS = SB_M * SB_S;      // 8192: most rows a cloud can have
constexpr int SB_ENT = SB_M * S2_C;       // 32768: most entries
#ifndef GEOA3_SB_P
#define GEOA3_SB_P 4
#endif
constexpr int SB_P = GEOA3_SB_P; // parts per cloud (fixed: the tiles, and with them the operand scales, depend on the cloud alone)
constexpr int SB_TILES = SB_ROWS / 64 + SB_P;   // most tiles
#ifndef GEOA3_SB_STOP
#define GEOA3_SB_STOP 0   // (tools: 1-3 end sa2b_prep_kernel early, 4 / 5 drop phase 3 / phase 1 of sa2b_bwd_kernel: timing only)
#endif

// per-cloud scratch (bytes), in this order
constexpr size_t SB_OFF_KEY = 0;                                   // int32 [8192]  dest | sample << 9 | centre << 15 | last-of-dest << 31
constexpr size_t SB_OFF_EC = SB_OFF_KEY + (size_t)SB_ROWS * 4;     // int32 [32768] channel | column << 16 | last-of-row << 31
constexpr size_t SB_OFF_EG = SB_OFF_EC + (size_t)SB_ENT * 4;       // float [32768]
constexpr size_t SB_OFF_TD = SB_OFF_EG + (size_t)SB_ENT * 4;       // int4 [SB_TILES] row0, rows, entry0, entries | (entries of rows 0-31) << 16
constexpr size_t SB_OFF_PT = SB_OFF_TD + (size_t)SB_TILES * 16;    // int32 [8]: tile prefix of the parts
constexpr size_t SB_OFF_HIT = SB_OFF_PT + 32;                      // u64 [128]: bit s = (centre, s) is a hit row
constexpr size_t SB_OFF_Y = SB_OFF_HIT + (size_t)SB_M * 8;         // float [128][64][3]: W_x^T d a0 per row
constexpr size_t SB_BYTES = (SB_OFF_Y + (size_t)SB_ROWS * 12 + 255) / 256 * 256;

struct SbPrepArgs {
  const float* dout;               // [B][256][128] d out2
  const float* outp;               // [B][256][128] out2 (relu gate of the pooled layer)
  const int32_t* arg;              // [B][256][128] arg-max sample
  const int32_t* gidx;             // [B][128][64] ball query
  unsigned char* scratch;          // [B][SB_BYTES]
  float* dr;                       // [B][512][128]: rows of destinations no hit row gathers from are zeroed here
};

constexpr int SBP_T = 1024, SBP_W = SBP_T / 64, SBP_ROUND = 32;   // 16 wavefronts, 32 centres staged per round
struct SbPrepLds {
  unsigned short cnt[SB_M][SB_S];      // live entries per (centre, sample)
  unsigned short dst[SB_M][SB_S];      // destination point
  unsigned short rpos[SB_M][SB_S];     // position of the row in destination order
  unsigned mask[SB_N1][4];             // centres with a hit row at this destination
  int dstart[SB_N1 + 1];               // rows in front of a destination's
  int rent[SB_ROWS];                   // entries of the rows in order -> (scan) first entry of each row
  float g[SBP_ROUND][S2_C + 1];
  unsigned char a[SBP_ROUND][S2_C + 4];
  int hist[SBP_W][64], run[SBP_W][64];
  int wsum[SBP_W];
  int pb[SB_P + 1], ptile[SB_P + 1];
  int total_ent;
};

__device__ __forceinline__ int sb_part_of(const int* pb, int pos) {
  int p = 0;
#pragma unroll
  for (int q = 1; q < SB_P; ++q) p += pos >= pb[q] ? 1 : 0;
  return p;
}

// exclusive scan of n ints in LDS in place by the whole workgroup (n <= 8 * SBP_T); returns the total to every thread
__device__ __forceinline__ int sb_block_scan(int* v, int n, int* wsum, int tid) {
  constexpr int PER = SB_ROWS / SBP_T;   // 8 consecutive values per thread
  const int lane = tid & 63, wave = tid >> 6;
  int x[PER], local = 0;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = tid * PER + i;
    x[i] = e < n ? v[e] : 0;
    local += x[i];
  }
  int incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o, 64);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[wave] = incl;
  __syncthreads();
  int base = 0, total = 0;
#pragma unroll
  for (int w = 0; w < SBP_W; ++w) {
    base += w < wave ? wsum[w] : 0;
    total += wsum[w];
  }
  int runv = base + incl - local;
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int e = tid * PER + i;
    if (e < n) v[e] = runv;
    runv += x[i];
  }
  __syncthreads();
  return total;
}

__global__ __launch_bounds__(SBP_T) void sa2b_prep_kernel(SbPrepArgs A) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sb_sm[];
  SbPrepLds& L = *reinterpret_cast<SbPrepLds*>(sb_sm);
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned char* sc = A.scratch + (size_t)b * SB_BYTES;
  int32_t* rkey = reinterpret_cast<int32_t*>(sc + SB_OFF_KEY);
  int32_t* ec = reinterpret_cast<int32_t*>(sc + SB_OFF_EC);
  float* eg = reinterpret_cast<float*>(sc + SB_OFF_EG);
  int4* tdesc = reinterpret_cast<int4*>(sc + SB_OFF_TD);
  int32_t* ptile_out = reinterpret_cast<int32_t*>(sc + SB_OFF_PT);
  unsigned long long* hit = reinterpret_cast<unsigned long long*>(sc + SB_OFF_HIT);
  const size_t base = (size_t)b * S2_C * SB_M;
  const int tx = tid & 31, ty = tid >> 5;   // staging: 32 centres x 32 channel rows per pass

  // the [256][32] tiles of d out2 (gated by out2 > 0) and arg2 cross through LDS; a round's loads are issued while the
  // round before it is processed
  struct Stage {
    float g[S2_C / 32];
    unsigned char a[S2_C / 32];
  };
  auto stage_load = [&](int m0, Stage& r) {
#pragma unroll
    for (int i = 0; i < S2_C / 32; ++i) {
      const size_t e = base + (size_t)(ty + 32 * i) * SB_M + m0 + tx;
      r.g[i] = A.outp[e] > 0.f ? A.dout[e] : 0.f;
      r.a[i] = (unsigned char)(A.arg[e] & 63);
    }
  };
  auto stage_store = [&](const Stage& r) {
#pragma unroll
    for (int i = 0; i < S2_C / 32; ++i) {
      L.g[tx][ty + 32 * i] = r.g[i];
      L.a[tx][ty + 32 * i] = r.a[i];
    }
  };
  Stage st;
  stage_load(0, st);

  for (int e = tid; e < SB_N1 * 4; e += SBP_T) (&L.mask[0][0])[e] = 0u;
  {   // the ball query's destinations: the whole [128][64] table in one round of loads
    int gv[SB_ROWS / SBP_T];
#pragma unroll
    for (int i = 0; i < SB_ROWS / SBP_T; ++i) gv[i] = A.gidx[(size_t)b * SB_ROWS + i * SBP_T + tid];
#pragma unroll
    for (int i = 0; i < SB_ROWS / SBP_T; ++i) (&L.dst[0][0])[i * SBP_T + tid] = (unsigned short)gv[i];
  }
  // ---- pass 1: live entries per (centre, sample); the ball query's destinations
  for (int m0 = 0; m0 < SB_M; m0 += SBP_ROUND) {
    __syncthreads();
    stage_store(st);
    __syncthreads();
    stage_load((m0 + SBP_ROUND) % SB_M, st);   // (the last round loads pass 2's first)
#pragma unroll
    for (int i = 0; i < SBP_ROUND / SBP_W; ++i) {
      const int ml = wave * (SBP_ROUND / SBP_W) + i, m = m0 + ml;
      L.hist[wave][lane] = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (L.g[ml][q * 64 + lane] != 0.f) atomicAdd(&L.hist[wave][L.a[ml][q * 64 + lane]], 1);
      const int c = L.hist[wave][lane];          // (LDS operations of a wave complete in order)
      L.cnt[m][lane] = (unsigned short)c;
      const unsigned long long hm = __ballot(c > 0);
      if (lane == 0) hit[m] = hm;
    }
  }
  __syncthreads();
  if (GEOA3_SB_STOP == 1) return;
  // ---- hit rows: per destination the set of centres that gather from it (a ball's live samples are distinct points, and
  // a padded sample never wins an arg-max tie against the sample it repeats: at most one hit row per (destination, centre))
  for (int e = tid; e < SB_ROWS; e += SBP_T) {
    const int m = e >> 6, s = e & 63;
    if (L.cnt[m][s] > 0) atomicOr(&L.mask[L.dst[m][s]][m >> 5], 1u << (m & 31));
  }
  __syncthreads();
  {
    int c = 0;
    if (tid < SB_N1) c = __builtin_popcount(L.mask[tid][0]) + __builtin_popcount(L.mask[tid][1]) +
                         __builtin_popcount(L.mask[tid][2]) + __builtin_popcount(L.mask[tid][3]);
    int incl = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o, 64);
      if (lane >= o) incl += t;
    }
    if (lane == 63) L.wsum[wave] = incl;
    __syncthreads();
    int bs = 0;
#pragma unroll
    for (int w = 0; w < SB_N1 / 64; ++w) bs += w < wave ? L.wsum[w] : 0;
    if (tid < SB_N1) L.dstart[tid] = bs + incl - c;
    if (tid == SB_N1 - 1) L.dstart[SB_N1] = bs + incl;
    // destinations nothing gathers from: their gradient row is zero (the walk below never reaches them)
    __syncthreads();
  }
  const int R = L.dstart[SB_N1];
  for (int e = tid; e < SB_N1 * (S2_K / 4); e += SBP_T) {
    const int j = e >> 5;
    if (L.dstart[j + 1] == L.dstart[j])
      reinterpret_cast<float4*>(A.dr + ((size_t)b * SB_N1 + j) * S2_K)[e & 31] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  // ---- the rows in (destination, centre) order
  for (int e = tid; e < SB_ROWS; e += SBP_T) {
    const int m = e >> 6, s = e & 63;
    const int c = L.cnt[m][s];
    if (c > 0) {
      const int j = L.dst[m][s];
      int rank = 0;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned word = L.mask[j][w];
        rank += w < (m >> 5) ? __builtin_popcount(word) : (w == (m >> 5) ? __builtin_popcount(word & ((1u << (m & 31)) - 1u)) : 0);
      }
      const int pos = L.dstart[j] + rank;
      L.rpos[m][s] = (unsigned short)pos;
      L.rent[pos] = c;
      rkey[pos] = j | (s << 9) | (m << 15) | (pos == L.dstart[j + 1] - 1 ? (int)0x80000000 : 0);
    }
  }
  __syncthreads();
  const int total_ent = sb_block_scan(L.rent, R, L.wsum, tid);
  // ---- parts (cut at destination boundaries, balanced by rows) and tiles of 64 rows
  if (tid <= SB_P) {
    int cut = tid == SB_P ? R : 0;
    if (tid > 0 && tid < SB_P) {
      const int target = (int)(((long)R * tid) / SB_P);
      int lo = 0, hi = SB_N1;                          // first destination starting at or behind the target
      while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (L.dstart[mid] < target) lo = mid + 1;
        else hi = mid;
      }
      cut = L.dstart[lo];
    }
    L.pb[tid] = cut;
  }
  __syncthreads();
  if (tid == 0) {
    int t = 0;
    for (int p = 0; p < SB_P; ++p) {
      L.ptile[p] = t;
      t += (L.pb[p + 1] - L.pb[p] + 63) / 64;
    }
    L.ptile[SB_P] = t;
    for (int p = 0; p <= SB_P; ++p) ptile_out[p] = L.ptile[p];
  }
  __syncthreads();
  for (int t = tid; t < L.ptile[SB_P]; t += SBP_T) {
    int p = 0;
#pragma unroll
    for (int q = 1; q < SB_P; ++q) p += t >= L.ptile[q] ? 1 : 0;
    const int row0 = L.pb[p] + 64 * (t - L.ptile[p]);
    const int rows = min(64, L.pb[p + 1] - row0);
    const int e0 = L.rent[row0];
    const int e1 = row0 + rows < R ? L.rent[row0 + rows] : total_ent;
    const int em = rows > 32 ? L.rent[row0 + 32] : e1;        // first entry of row 32: the tile's two waves take 32 rows each
    tdesc[t] = make_int4(row0, rows, e0, (e1 - e0) | ((em - e0) << 16));
  }
  if (GEOA3_SB_STOP == 2) return;
  // ---- pass 2: the entries row by row (ascending channel inside a row: a stable counting sort by sample)
  for (int m0 = 0; m0 < SB_M; m0 += SBP_ROUND) {
    __syncthreads();
    stage_store(st);
    __syncthreads();
    if (m0 + SBP_ROUND < SB_M) stage_load(m0 + SBP_ROUND, st);
#pragma unroll 1
    for (int i = 0; i < SBP_ROUND / SBP_W; ++i) {
      const int ml = wave * (SBP_ROUND / SBP_W) + i, m = m0 + ml;
      int* run = L.run[wave];
      run[lane] = 0;                               // entries of sample `lane` placed so far
      const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int s = L.a[ml][q * 64 + lane];
        const float g = L.g[ml][q * 64 + lane];
        const bool live = g != 0.f;
        unsigned long long grp = __ballot(live);   // live lanes with the same sample as this one
#pragma unroll
        for (int bit = 0; bit < 6; ++bit) {
          const unsigned long long bal = __ballot((s >> bit) & 1);
          grp &= ((s >> bit) & 1) ? bal : ~bal;
        }
        const int r = (int)__builtin_popcountll(grp & below), n = (int)__builtin_popcountll(grp);
        const int start = run[s];
        if (live) {
          const int pos = L.rpos[m][s];
          const int rank = start + r;
          const int p = sb_part_of(L.pb, pos);
          const int col = (pos - L.pb[p]) & 63;
          const int idx = L.rent[pos] + rank;
          ec[idx] = (q * 64 + lane) | (col << 16) | (rank == (int)L.cnt[m][s] - 1 ? (int)0x80000000 : 0);
          eg[idx] = g;
        }
        if (live && r == 0) run[s] = start + n;    // one lane per sample: the next round's entries follow
      }
      // gate words of the centre's hit rows: bit k of row (m, s) = bit s of word k
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
struct SbBwdArgs {
  const unsigned char* scratch;    // [B][SB_BYTES] (sa2b_prep_kernel)
  const uint4* m1;                 // [B * 128][64]: a1 > 0 of the row, bit k (the forward's gate words)
  const uint4* m0;                 // a0 > 0, bit i
  const float* W2;                 // [256][128]
  const _Float16* w1img;           // W1^T as a fragment image (frag_image_kernel)
  const float* w1un;               // [1]: 1 / the image's power-of-two scale
  const float* Wx;                 // [128][3]
  float* dr;                       // [B][512][128] point-major
  unsigned char* scratch_w;        // (the same scratch: the rows' coordinate terms are written into it)
};

constexpr int sa2b_bwd_lds() { return 2 * 64 * S2_PT * 4 + 16 * 4 + 2 * 64 * 4 + 2 * 2 * 64 * 4 * 4 + SB_TILES * 16; }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void sa2b_bwd_kernel(SbBwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sb_sm[];
  float* s_tile = reinterpret_cast<float*>(sb_sm);            // [2 tiles][64 rows][S2_PT]: d a1, then d a0
  float* s_red = s_tile + 2 * 64 * S2_PT;                     // [4] tile maxima of the waves (+ pad to 16)
  int* s_key = reinterpret_cast<int*>(s_red + 16);            // [2][64] the rows' keys
  float4* s_y = reinterpret_cast<float4*>(s_key + 128);       // [2 tiles][2 halves of i][64 rows]: partial coordinate terms
  int4* s_td = reinterpret_cast<int4*>(s_y + 2 * 2 * 64);     // the part's tile descriptors
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / SB_P, part = blockIdx.x % SB_P;
  const unsigned char* sc = a.scratch + (size_t)b * SB_BYTES;
  const int32_t* rkey = reinterpret_cast<const int32_t*>(sc + SB_OFF_KEY);
  const uint4* rg1 = a.m1 + (size_t)b * SB_ROWS;   // indexed by centre * 64 + sample
  const uint4* rg0 = a.m0 + (size_t)b * SB_ROWS;
  const int32_t* ec = reinterpret_cast<const int32_t*>(sc + SB_OFF_EC);
  const float* eg = reinterpret_cast<const float*>(sc + SB_OFF_EG);
  const int4* tdesc = reinterpret_cast<const int4*>(sc + SB_OFF_TD);
  const int32_t* ptile = reinterpret_cast<const int32_t*>(sc + SB_OFF_PT);
  float* yrow = reinterpret_cast<float*>(a.scratch_w + (size_t)b * SB_BYTES + SB_OFF_Y);
  const int t0 = ptile[part], t1 = ptile[part + 1];
  if (t0 >= t1) return;
  const int nt = t1 - t0;
  for (int t = tid; t < nt; t += 256) s_td[t] = tdesc[t0 + t];
  __syncthreads();
  const float unW = a.w1un[0];
  const int ci = wave >> 1, hf = wave & 1;
  float* tile = s_tile + ci * 64 * S2_PT;

  // phase 1: a wave takes 32 ROWS of its tile with all 128 k (lane = k / 2: one 512-byte row of W2 per entry and wave) --
  // the per-entry bookkeeping (two lane reads, the address, the branch) is paid once per entry, not once per half of k
  auto w2row = [&](int cw) {
    return GEOA3_SB_STOP == 6 ? make_float2((float)(cw & 255) * 1e-3f, 0.f)
                              : reinterpret_cast<const float2*>(a.W2 + (size_t)(cw & 0xffff) * S2_K)[lane];
  };
  const half8* img = reinterpret_cast<const half8*>(a.w1img) + (size_t)(2 * hf) * 8 * 2 * 64 + lane;   // row tiles 2 hf, 2 hf + 1
  auto load_a = [&](int c, half8 (&f)[4]) {   // fragments (t, piece) of k-step c
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 2; ++p) f[2 * t + p] = img[(size_t)((t * 8 + c) * 2 + p) * 64];
  };
  // tile t of the part (relative index): row0, rows, entry0, entries | entries of rows 0-31 << 16 -- zeros behind the part's last tile
  auto tile_desc = [&](int t) {
    int4 d = t < nt ? s_td[t] : make_int4(0, 0, 0, 0);
    d.x = __builtin_amdgcn_readfirstlane(d.x); d.y = __builtin_amdgcn_readfirstlane(d.y);
    d.z = __builtin_amdgcn_readfirstlane(d.z); d.w = __builtin_amdgcn_readfirstlane(d.w);
    return d;
  };
  // The entries this wave walks -- the rows 32 hf .. 32 hf + 31 of its tile of every pair, in chunks of 64 as lane vectors --
  // are a sequence a cursor steps through two chunks AHEAD of the one being consumed.  Entries past the end of the rows'
  // range: channel 0, value 0, no flag.
  struct Cursor {
    int t, at, e0, n;    // tile (relative), offset of the chunk inside the range, the range's first entry and entry count
  };
  auto cur_first = [&](int t) {
    const int4 d = tile_desc(t);
    const int n = d.w & 0xffff, mid = (int)((unsigned)d.w >> 16);
    return hf == 0 ? Cursor{t, 0, d.z, mid} : Cursor{t, 0, d.z + mid, n - mid};
  };
  auto cur_next = [&](Cursor c) {   // (uniform)
    if (c.at + 64 < c.n) return Cursor{c.t, c.at + 64, c.e0, c.n};
    return cur_first(c.t + 2);
  };
  auto load_chunk = [&](const Cursor& c, int& vc, float& vg) {
    const bool ok = c.at + lane < c.n;
    vc = ok ? ec[c.e0 + c.at + lane] : 0;
    vg = ok ? eg[c.e0 + c.at + lane] : 0.f;
  };
  // the rows of a tile, lane = column: key, a1 > 0 over all 128 k (phase 1), a0 > 0 for this wave's half of i (epilogue)
  struct Rows {
    int key;
    uint4 g1;
    uint2 g0;
  };
  auto load_rows = [&](int t, Rows& r) {
    const int4 d = tile_desc(t);
    const bool ok = lane < d.y;
    r.key = ok ? rkey[d.x + lane] : 0;
    const int ms = (r.key >> 9) & (SB_ROWS - 1);             // centre * 64 + sample (bits 9-14 | 15-21 of the key)
    r.g1 = ok ? rg1[ms] : make_uint4(0u, 0u, 0u, 0u);
    r.g0 = ok ? reinterpret_cast<const uint2*>(rg0 + ms)[hf] : make_uint2(0u, 0u);
  };

  constexpr int PF = 16;   // W2 rows in flight per wave (512 bytes each)
  float2 wv[PF];
  int cwr[PF];   // the entries whose W2 rows are in flight (uniform)
  Cursor c0 = cur_first(ci), c1 = cur_next(c0), c2 = cur_next(c1);
  int vc, vc1, vc2;
  float vg, vg1, vg2;
  load_chunk(c0, vc, vg);
  load_chunk(c1, vc1, vg1);
  load_chunk(c2, vc2, vg2);
  Rows rw, rwn;
  load_rows(ci, rw);
#pragma unroll
  for (int u = 0; u < PF; ++u) {
    cwr[u] = s2_rl(vc, u);
    wv[u] = w2row(cwr[u]);
  }
  float run_acc = 0.f;   // phase 3, waves 0-1: the open destination's sum for output channel `tid`
  const int gsh = 2 * (lane & 15);   // the lane's k = 2 lane, 2 lane + 1: bits gsh, gsh + 1 of word lane >> 4 of a row's gate

  for (int tp = 0; tp < nt; tp += 2) {
    const int rows = tile_desc(tp + ci).y;
    load_rows(tp + 2 + ci, rwn);   // this wave's tile of the next pair
    if (hf == 0) s_key[ci * 64 + lane] = rw.key;
    half8 af[2][4];
    load_a(0, af[0]);
    // ---- phase 1: this wave's 32 rows of the tile
    if (GEOA3_SB_STOP != 5) {
      {   // rows of the wave's range that do not exist (a part's last tile): zero columns
        const int lo = 32 * hf > rows ? 32 * hf : rows;
#pragma unroll 8
        for (int r = 0; r < 32; ++r)
          if (32 * hf + r >= lo) *reinterpret_cast<float2*>(tile + (32 * hf + r) * S2_PT + 2 * lane) = make_float2(0.f, 0.f);   // (uniform test)
      }
      float acc0 = 0.f, acc1 = 0.f, mx = 0.f;
      for (;;) {
        const bool more = c1.t == c0.t;                 // (uniform) the range has another chunk
        Cursor c3 = cur_next(c2);
        int vc3;
        float vg3;
        load_chunk(c3, vc3, vg3);
#pragma unroll
        for (int h = 0; h < 64 / PF; ++h) {
#pragma unroll
          for (int u = 0; u < PF; ++u) {
            const int cw = cwr[u];
            const float g = s2_rlf(vg, PF * h + u);
            acc0 = __builtin_fmaf(g, wv[u].x, acc0);
            acc1 = __builtin_fmaf(g, wv[u].y, acc1);
            cwr[u] = h + 1 < 64 / PF ? s2_rl(vc, PF * (h + 1) + u) : s2_rl(vc1, u);   // (the next chunk may be the next pair's
            wv[u] = w2row(cwr[u]);                                                    // first: in flight across phases 2, 3)
            if (__builtin_expect(cw < 0, 0)) {   // wave-uniform: the row's last entry
              const int col = (cw >> 16) & 63;
              // a1 > 0 of (k, row) for the lane's two k: the row's 128-bit word, uniform
              const unsigned w0 = (unsigned)s2_rl((int)rw.g1.x, col), w1 = (unsigned)s2_rl((int)rw.g1.y, col);
              const unsigned w2 = (unsigned)s2_rl((int)rw.g1.z, col), w3 = (unsigned)s2_rl((int)rw.g1.w, col);
              const unsigned wsel = lane < 32 ? (lane < 16 ? w0 : w1) : (lane < 48 ? w2 : w3);
              const int b0 = __builtin_amdgcn_sbfe((int)wsel, (unsigned)gsh, 1u), b1 = __builtin_amdgcn_sbfe((int)wsel, (unsigned)gsh + 1u, 1u);
              const float v0 = __int_as_float(__float_as_int(acc0) & b0), v1 = __int_as_float(__float_as_int(acc1) & b1);
              *reinterpret_cast<float2*>(tile + col * S2_PT + 2 * lane) = make_float2(v0, v1);
              mx = fmaxf(mx, fmaxf(__builtin_fabsf(v0), __builtin_fabsf(v1)));
              acc0 = 0.f;
              acc1 = 0.f;
            }
          }
        }
        c0 = c1; c1 = c2; c2 = c3;
        vc = vc1; vg = vg1;
        vc1 = vc2; vg1 = vg2;
        vc2 = vc3; vg2 = vg3;
        if (!more) break;
      }
      mx = wave_max(mx);
      if (lane == 0) s_red[wave] = mx;
    }
    __syncthreads();
    // ---- phase 2: rows 64 hf .. 64 hf + 63 of d a0 for the tile's 64 columns
    f32x16 acc2[2][2];
    const unsigned Ex = s2_exp(fmaxf(s_red[2 * ci], s_red[2 * ci + 1]));
    {
      const float sx = s2_scale(Ex);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc2[cb][t][r] = 0.f;
      const float* brow = tile + (lane & 31) * S2_PT + (lane >> 5) * 8;
#pragma unroll
      for (int c = 0; c < (GEOA3_SB_STOP == 7 ? 1 : 8); ++c) {
        if (c + 1 < 8) load_a(c + 1, af[(c + 1) & 1]);
        half8 xh[2], xl[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const float4 b0 = *reinterpret_cast<const float4*>(brow + cb * 32 * S2_PT + c * 16);
          const float4 b1 = *reinterpret_cast<const float4*>(brow + cb * 32 * S2_PT + c * 16 + 4);
          const float x[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
          s2_split8(x, sx, xh[cb], xl[cb]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const half8 wh = af[c & 1][2 * t], wl = af[c & 1][2 * t + 1];
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            acc2[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[cb], acc2[cb][t], 0, 0, 0);
            acc2[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[cb], acc2[cb][t], 0, 0, 0);
            acc2[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[cb], acc2[cb][t], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();   // both waves of the tile have read d a1: the tile takes d a0 (one 528-byte row per column)
    {
      // gate, un-scale, and the row's coordinate term W_x^T d a0 on the way (this wave's 64 of the 128 i, ascending; W_x rows
      // are uniform: scalar loads); lane = column
      const float unscale = s2_unscale(Ex) * unW;
      const float* wx = a.Wx + 3 * 64 * hf;
      float y0 = 0.f, y1 = 0.f, y2 = 0.f;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const unsigned gw = t == 0 ? rw.g0.x : rw.g0.y;      // a0 > 0 for i = 64 hf + 32 t .. + 31 of the lane's row
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = acc2[0][t][4 * g + i];
            v[4 + i] = acc2[1][t][4 * g + i];
            s2_swap32(v[i], v[4 + i]);    // v[i]: row base + i, v[4 + i]: row base + 4 + i, lane = column
          }
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int on = __builtin_amdgcn_sbfe((int)gw, (unsigned)(8 * g + i), 1u);   // 0 / -1
            v[i] = __int_as_float(__float_as_int(v[i] * unscale) & on);
            const float* w = wx + 3 * (32 * t + 8 * g + i);
            y0 = __builtin_fmaf(w[0], v[i], y0);
            y1 = __builtin_fmaf(w[1], v[i], y1);
            y2 = __builtin_fmaf(w[2], v[i], y2);
          }
          float* dstp = tile + lane * S2_PT + 64 * hf + 32 * t + 8 * g;
          *reinterpret_cast<float4*>(dstp) = make_float4(v[0], v[1], v[2], v[3]);
          *reinterpret_cast<float4*>(dstp + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
      }
      s_y[(ci * 2 + hf) * 64 + lane] = make_float4(y0, y1, y2, 0.f);
    }
    __syncthreads();
    // ---- phase 3
    if (GEOA3_SB_STOP != 4) {
      const int rowsA = tile_desc(tp).y, rowsB = tile_desc(tp + 1).y;
      if (wave < 2) {
        // output channel i = tid: the pair's columns in order; a destination's sum leaves when its last row has been added.
        // The keys as lane vectors (lane = column): a column's key is a v_readlane away, the last-row flags one ballot --
        // no LDS round trip per column; the column values arrive sixteen reads at a time.
        const float* colp = s_tile + tid;
        float* drb = a.dr + (size_t)b * SB_N1 * S2_K + tid;
        const int kA = s_key[lane], kB = s_key[64 + lane];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int n = half == 0 ? rowsA : rowsB;
          const int kv = half == 0 ? kA : kB;
          const unsigned long long last = __ballot(kv < 0);
          const float* cp = colp + half * 64 * S2_PT;
#pragma unroll
          for (int q0 = 0; q0 < 64; q0 += 16) {
            if (q0 < n) {   // (uniform)
              float x[16];
#pragma unroll
              for (int u = 0; u < 16; ++u) x[u] = cp[(q0 + u) * S2_PT];
#pragma unroll
              for (int u = 0; u < 16; ++u) {
                if (q0 + u < n) {
                  run_acc += x[u];
                  if ((last >> (q0 + u)) & 1ull) {
                    drb[(size_t)(s2_rl(kv, q0 + u) & 511) * S2_K] = run_acc;
                    run_acc = 0.f;
                  }
                }
              }
            }
          }
        }
      } else {
        // one row per lane: its coordinate term = the two halves' partial sums (i < 64 first)
        const int c = tid - 128, half = c >> 6, cc = c & 63;
        if (cc < (half == 0 ? rowsA : rowsB)) {
          const float4 ya = s_y[(half * 2 + 0) * 64 + cc], yb = s_y[(half * 2 + 1) * 64 + cc];
          const int key = s_key[half * 64 + cc];
          float* yp = yrow + (size_t)((key >> 9) & (SB_ROWS - 1)) * 3;
          yp[0] = ya.x + yb.x;
          yp[1] = ya.y + yb.y;
          yp[2] = ya.z + yb.z;
        }
      }
    }
    __syncthreads();   // the tiles are rewritten by the next pair
    rw = rwn;
  }
}

// d c[b][m][0..2] -= sum over the centre's hit rows (sample ascending) of the row's coordinate term: shift = b0 - W_x c
__global__ __launch_bounds__(256) void sa2b_centre_kernel(const unsigned char* __restrict__ scratch, float* __restrict__ dnx2,
                                                          long centres) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= centres) return;
  const long b = e / SB_M;
  const int m = (int)(e - b * SB_M);
  const unsigned char* sc = scratch + (size_t)b * SB_BYTES;
  unsigned long long hm = reinterpret_cast<const unsigned long long*>(sc + SB_OFF_HIT)[m];
  const float* y = reinterpret_cast<const float*>(sc + SB_OFF_Y) + (size_t)m * SB_S * 3;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  while (hm) {
    const int s = __builtin_ctzll(hm);
    hm &= hm - 1ull;
    a0 += y[3 * s];
    a1 += y[3 * s + 1];
    a2 += y[3 * s + 2];
  }
  float* q = dnx2 + e * 3;
  q[0] -= a0;
  q[1] -= a1;
  q[2] -= a2;
}

// dp[b][n][0..2] = sum_co Wx[co][0..2] dY[b][n][co] for point-major dY (one wavefront per point, co ascending within a lane,
// the lanes' partial sums in the fixed order of wave_sum)
__global__ __launch_bounds__(256) void affine3_grad_pm_kernel(const float* __restrict__ dY, const float* __restrict__ Wx,
                                                              float* __restrict__ dp, long points) {
  const long pt = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (pt >= points) return;
  const float2 v = reinterpret_cast<const float2*>(dY + pt * S2_K)[lane];
  const float* w = Wx + 6 * lane;
  float a0 = w[0] * v.x + w[3] * v.y, a1 = w[1] * v.x + w[4] * v.y, a2 = w[2] * v.x + w[5] * v.y;
  a0 = wave_sum(a0);
  a1 = wave_sum(a1);
  a2 = wave_sum(a2);
  if (lane == 0) {
    dp[pt * 3] = a0;
    dp[pt * 3 + 1] = a1;
    dp[pt * 3 + 2] = a2;
  }
}

}  // namespace

size_t sa2b_scratch_bytes(int B) { return (size_t)B * SB_BYTES; }

// dout / outp / arg [B][256][128] channel-major, gidx [B][128][64], m1 / m0 [B * 128][64][4] the forward's gate words per row;
// dr [B][512][128] (point-major), dnx1 [B][512][3] = W_x^T dr, dnx2 [B][128][3] -= the centres' share.  M1 = 512 points,
// M2 = 128 centres, 64 samples (PointNetPP_ssg.py:68-76).
int launch_sa2b_prep(const float* dout, const float* outp, const int32_t* arg, const int32_t* gidx, void* scratch, float* dr, int B,
                     hipStream_t s) {
  SbPrepArgs a{dout, outp, arg, gidx, static_cast<unsigned char*>(scratch), dr};
  const int lds = (int)sizeof(SbPrepLds);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa2b_prep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(sa2b_prep_kernel, dim3(B), dim3(SBP_T), lds, s, a);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

int launch_sa2b_bwd(const void* scratch, const unsigned* m1, const unsigned* m0, const float* W2, const void* w1t_img,
                    const float* w1t_un, const float* Wx, float* dr, int B, hipStream_t s) {
  SbBwdArgs a{static_cast<const unsigned char*>(scratch), reinterpret_cast<const uint4*>(m1), reinterpret_cast<const uint4*>(m0), W2,
              static_cast<const _Float16*>(w1t_img), w1t_un, Wx, dr,
              const_cast<unsigned char*>(static_cast<const unsigned char*>(scratch))};
  const int lds = sa2b_bwd_lds();
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa2b_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipLaunchKernelGGL(sa2b_bwd_kernel, dim3(B * SB_P), dim3(256), lds, s, a);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

int launch_sa2b_centre(const void* scratch, float* dnx2, int B, hipStream_t s) {
  const long centres = (long)B * SB_M;
  hipLaunchKernelGGL(sa2b_centre_kernel, dim3((unsigned)((centres + 255) / 256)), dim3(256), 0, s,
                     static_cast<const unsigned char*>(scratch), dnx2, centres);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

int launch_affine3_grad_pm(const float* dY, const float* Wx, float* dp, long points, hipStream_t s) {
  hipLaunchKernelGGL(affine3_grad_pm_kernel, dim3((unsigned)((points + 3) / 4)), dim3(256), 0, s, dY, Wx, dp, points);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
