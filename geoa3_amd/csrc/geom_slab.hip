// Self K-NN with slab pruning (gfx950): the all-pairs kernel of geom_nn.hip, fed with a fraction of the pairs.
//
// A candidate p can only enter the K nearest of q if its distance is within q's radius tau (the K-th distance to last
// iteration's neighbours), and d(q,p) >= (q_a - p_a)^2 along ANY axis a.  So: per instance, bin the cloud along its
// longest bounding-box axis (counting sort into 512 bins, one workgroup, ~3 us), hand the queries out in that order
// (256 consecutive sorted positions per workgroup: a narrow slab), and let the workgroup scan only the CONTIGUOUS run
// of sorted points whose bins intersect [q_a - sqrt(tau), q_a + sqrt(tau)] for any of its queries.  The inner loop is
// the regular LDS-broadcast distance loop of the all-pairs kernel (no per-lane cell walks, no divergence), the lists,
// the compaction and the lexicographic (distance, index) selection are the same, and the bin function is monotone in
// the coordinate, so the result is BIT-IDENTICAL to the all-pairs search for any input; a query without a usable
// radius scans everything.
#include <utility>
#include "geom_internal.h"
#include "profile.h"

namespace {

constexpr int SB_T = 1024;            // binning kernel: threads
constexpr int SB_NB = 512;            // bins
constexpr int SB_PPT = 8;             // up to 8192 points
constexpr int SK_BLOCK = 256;
constexpr int SK_CHUNK = 1024;
constexpr float S_INF = __builtin_inff();

// (per-instance records and rows written by one workgroup each and read by the next kernel: a 128-byte line of their own,
//  no two workgroups -- XCDs -- write into one cache line; NOTEBOOK 5a)
struct alignas(128) SlabGeo {
  float lo, inv_w;
  int axis, pad;
};
constexpr int SB_PITCH = 544;         // ints per row of bin starts (SB_NB + 1 = 513, rounded up to whole lines)

__device__ __forceinline__ int slab_bin(float v, float lo, float inv_w) {
  const int c = (int)floorf((v - lo) * inv_w);
  return c < 0 ? 0 : (c > SB_NB - 1 ? SB_NB - 1 : c);
}

// sorted: [B][3][N] coordinates in bin order, sidx [B][N] original indices, bstart [B][SB_NB+1], geo [B]
__global__ __launch_bounds__(SB_T) void slab_bin_kernel(const float* __restrict__ pc, int N, float* __restrict__ sorted,
                                                        int32_t* __restrict__ sidx, int32_t* __restrict__ bstart,
                                                        SlabGeo* __restrict__ geo) {
  __shared__ float s_red[16][6];
  __shared__ int s_cnt[SB_NB], s_start[SB_NB + 1], s_wsum[16];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* P = pc + (size_t)b * 3 * N;
  float px[SB_PPT], py[SB_PPT], pz[SB_PPT];
  float lo[3] = {S_INF, S_INF, S_INF}, hi[3] = {-S_INF, -S_INF, -S_INF};
#pragma unroll
  for (int p = 0; p < SB_PPT; ++p) {
    const int i = tid + p * SB_T;
    if (i < N) {
      px[p] = P[i];
      py[p] = P[N + i];
      pz[p] = P[2 * N + i];
      lo[0] = fminf(lo[0], px[p]); hi[0] = fmaxf(hi[0], px[p]);
      lo[1] = fminf(lo[1], py[p]); hi[1] = fmaxf(hi[1], py[p]);
      lo[2] = fminf(lo[2], pz[p]); hi[2] = fmaxf(hi[2], pz[p]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    lo[c] = -wave_max(-lo[c]);
    hi[c] = wave_max(hi[c]);
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      s_red[wave][c] = lo[c];
      s_red[wave][3 + c] = hi[c];
    }
  }
  if (tid < SB_NB) s_cnt[tid] = 0;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 16; ++w)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      lo[c] = fminf(lo[c], s_red[w][c]);
      hi[c] = fmaxf(hi[c], s_red[w][3 + c]);
    }
  const float ex = hi[0] - lo[0], ey = hi[1] - lo[1], ez = hi[2] - lo[2];
  const int axis = (ex >= ey && ex >= ez) ? 0 : (ey >= ez ? 1 : 2);
  const float alo = axis == 0 ? lo[0] : (axis == 1 ? lo[1] : lo[2]);
  const float ext = axis == 0 ? ex : (axis == 1 ? ey : ez);
  const float inv_w = ext > 1e-30f ? (float)SB_NB / (ext * 1.00001f) : 0.f;
  int bin[SB_PPT];
#pragma unroll
  for (int p = 0; p < SB_PPT; ++p) {
    const int i = tid + p * SB_T;
    if (i < N) {
      bin[p] = slab_bin(axis == 0 ? px[p] : (axis == 1 ? py[p] : pz[p]), alo, inv_w);
      atomicAdd(&s_cnt[bin[p]], 1);
    }
  }
  __syncthreads();
  // exclusive scan of the 512 counters by the first 512 threads (one each)
  int c = tid < SB_NB ? s_cnt[tid] : 0, incl = c;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) base += (w < wave) ? s_wsum[w] : 0;
  if (tid < SB_NB) {
    s_start[tid] = base + incl - c;
    s_cnt[tid] = 0;
  }
  if (tid == SB_NB - 1) s_start[SB_NB] = base + incl;
  __syncthreads();
  float* S = sorted + (size_t)b * 3 * N;
#pragma unroll
  for (int p = 0; p < SB_PPT; ++p) {
    const int i = tid + p * SB_T;
    if (i < N) {
      const int pos = s_start[bin[p]] + atomicAdd(&s_cnt[bin[p]], 1);
      S[pos] = px[p];
      S[N + pos] = py[p];
      S[2 * N + pos] = pz[p];
      sidx[(size_t)b * N + pos] = i;
    }
  }
  for (int e = tid; e <= SB_NB; e += SB_T) bstart[(size_t)b * SB_PITCH + e] = s_start[e];
  if (tid == 0) geo[b] = SlabGeo{alo, inv_w, axis, 0};
}

typedef float slab_f2 __attribute__((ext_vector_type(2)));
// Two candidates' squared distances to the query per packed instruction.  The query arrives NEGATED (slab_negq, once per
// query): the differences are x + (-q) -- plain v_pk_add_f32 without a source modifier.  (x - q)^2 and (q - x)^2 are the same
// bits.  (Written while `neg` modifiers were suspected of the wrong values of NOTEBOOK 5a; the form that matters turned out to be
// op_sel, which these hand-built pairs never carried.  Kept: it costs nothing and the build refuses op_sel either way.)
__device__ __forceinline__ void slab_sqdist2(float nqx, float nqy, float nqz, float x0, float x1, float y0, float y1,
                                             float z0, float z1, float& d0, float& d1) {
#pragma clang fp contract(off)
  const slab_f2 dx = slab_f2{x0, x1} + slab_f2{nqx, nqx};
  const slab_f2 dy = slab_f2{y0, y1} + slab_f2{nqy, nqy};
  const slab_f2 dz = slab_f2{z0, z1} + slab_f2{nqz, nqz};
  const slab_f2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
  const slab_f2 s = xx + yy;
  const slab_f2 d = s + zz;
  d0 = d.x;
  d1 = d.y;
}
__device__ __forceinline__ void slab_negq(float qx, float qy, float qz, float& nqx, float& nqy, float& nqz) {
  nqx = -qx;
  nqy = -qy;
  nqz = -qz;
  asm volatile("" : "+v"(nqx), "+v"(nqy), "+v"(nqz));   // (opaque: not folded back into a modifier of the packed add)
}

__device__ __forceinline__ bool slab_lex_less(float da, int ia, float db, int ib) {
  return (da < db) || (da == db && ia < ib);
}

// The K smallest of a thread's cnt <= CAP candidates by (distance, index), in order, into slots 0 .. K-1; returns the
// K-th distance.  Small CAP: the whole list goes to registers as 64-bit keys (distance bits : index -- distances are
// >= +0, so the bit patterns order like the values and one integer compare is the lexicographic one), every entry's
// rank is counted over all pairs and the entry is written to the slot of its rank.  No dependent LDS round trips: the
// selection loop below waits for two of them per step (K * cnt / 2 steps: ~30 us per call for a wavefront at K = 17).
template <int CAP>
__device__ __forceinline__ float slab_compact(uint16_t* cand, float* candd, int cnt, int K, int tid) {
  float kth = S_INF;
  if constexpr (CAP <= 40) {
    unsigned long long key[CAP];
    int rank[CAP];
#pragma unroll
    for (int s = 0; s < CAP; ++s) {
      const unsigned d = __float_as_uint(candd[s * SK_BLOCK + tid]);
      const unsigned i = cand[s * SK_BLOCK + tid];
      key[s] = s < cnt ? ((unsigned long long)d << 32) | i : ~0ull;
      rank[s] = CAP - 1 - s;      // as if every later entry were smaller; corrected pair by pair
    }
#pragma unroll
    for (int i = 0; i < CAP; ++i)
#pragma unroll
      for (int j = i + 1; j < CAP; ++j) {
        // lt = key[i] < key[j]; rank[i] -= lt; rank[j] += lt -- compare and both carries back to back (left to the
        // compiler the compares are batched and their lane masks spilled: 4000 v_readlane / v_writelane)
        unsigned long long m;
        asm volatile("v_cmp_lt_u64 %2, %3, %4\n\tv_subbrev_co_u32 %0, vcc, 0, %0, %2\n\t"
                     "v_addc_co_u32 %1, vcc, 0, %1, %2"
                     : "+v"(rank[i]), "+v"(rank[j]), "=&s"(m)
                     : "v"(key[i]), "v"(key[j])
                     : "vcc");
      }
#pragma unroll
    for (int s = 0; s < CAP; ++s) {
      if (rank[s] < K) {
        const float d = __uint_as_float((unsigned)(key[s] >> 32));
        candd[rank[s] * SK_BLOCK + tid] = d;
        cand[rank[s] * SK_BLOCK + tid] = (uint16_t)key[s];
        if (rank[s] == K - 1) kth = d;
      }
    }
    return kth;
  }
  for (int p = 0; p < K; ++p) {
    float bd = candd[p * SK_BLOCK + tid];
    int bidx = cand[p * SK_BLOCK + tid];
    int bs = p;
    for (int s = p + 1; s < cnt; ++s) {
      const float d = candd[s * SK_BLOCK + tid];
      const int i = cand[s * SK_BLOCK + tid];
      if (slab_lex_less(d, i, bd, bidx)) {
        bd = d;
        bidx = i;
        bs = s;
      }
    }
    if (bs != p) {
      candd[bs * SK_BLOCK + tid] = candd[p * SK_BLOCK + tid];
      cand[bs * SK_BLOCK + tid] = cand[p * SK_BLOCK + tid];
      candd[p * SK_BLOCK + tid] = bd;
      cand[p * SK_BLOCK + tid] = (uint16_t)bidx;
    }
    kth = bd;
  }
  return kth;
}

template <int CAP>
__global__ __launch_bounds__(SK_BLOCK) void knn_slab_kernel(const float* __restrict__ R, int N, int K,
                                                            const int32_t* __restrict__ prior,
                                                            const float* __restrict__ sorted,
                                                            const int32_t* __restrict__ sidx,
                                                            const int32_t* __restrict__ bstart,
                                                            const SlabGeo* __restrict__ geo, float* __restrict__ dists,
                                                            int32_t* __restrict__ idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_ref = reinterpret_cast<float*>(smem);                                    // 3*SK_CHUNK floats
  float* s_cd = s_ref + 3 * SK_CHUNK;                                               // [CAP][SK_BLOCK]
  uint16_t* s_ci = reinterpret_cast<uint16_t*>(s_cd + (size_t)CAP * SK_BLOCK);      // [CAP][SK_BLOCK]
  uint16_t* s_id = s_ci + (size_t)CAP * SK_BLOCK;                                   // [SK_CHUNK] original indices
  __shared__ int s_lo, s_hi;
  const int b = blockIdx.y, tid = threadIdx.x;
  const int pos = blockIdx.x * SK_BLOCK + tid;
  const bool live = pos < N;
  const float* Rb = R + (size_t)b * 3 * N;
  const float* Sb = sorted + (size_t)b * 3 * N;
  const int32_t* Ib = sidx + (size_t)b * N;
  const int32_t* bs = bstart + (size_t)b * SB_PITCH;
  const SlabGeo g = geo[b];
  const int pc = live ? pos : N - 1;
  const float qx = Sb[pc], qy = Sb[N + pc], qz = Sb[2 * N + pc];
  float nqx, nqy, nqz;
  slab_negq(qx, qy, qz, nqx, nqy, nqz);
  const int qo = Ib[pc];                                     // the query's original index
  if (tid == 0) {
    s_lo = SB_NB;
    s_hi = -1;
  }
  float tau = S_INF;
  if (prior != nullptr && live) {
    // eight neighbours at a time: their indices, then their coordinates, all in flight together (one index / one point
    // per trip of a loop with an exit is 2 K dependent round trips: 30 us of the kernel at K = 17)
    const int32_t* pr = prior + ((size_t)b * N + qo) * K;
    float t = 0.f;
    bool ok = true;
    for (int m0 = 0; m0 < K; m0 += 8) {
      int j[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) j[u] = m0 + u < K ? pr[m0 + u] : 0;
      float px[8], py[8], pz[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        ok = ok && j[u] >= 0 && j[u] < N;
        const int jc = j[u] < 0 ? 0 : (j[u] >= N ? N - 1 : j[u]);
        px[u] = Rb[jc];
        py[u] = Rb[N + jc];
        pz[u] = Rb[2 * N + jc];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (m0 + u < K) t = fmaxf(t, geoa3_sqdist(qx, qy, qz, px[u], py[u], pz[u]));
    }
    if (ok) tau = t;
  }
  if (!live) tau = -1.f;  // padding lanes never collect candidates
  __syncthreads();
  if (live) {
    int blo = 0, bhi = SB_NB - 1;
    if (tau < S_INF) {
      const float qa = g.axis == 0 ? qx : (g.axis == 1 ? qy : qz);
      const float r = sqrtf(tau) * 1.00001f + 1e-30f;
      blo = slab_bin(qa - r, g.lo, g.inv_w);
      bhi = slab_bin(qa + r, g.lo, g.inv_w);
    }
    atomicMin(&s_lo, blo);
    atomicMax(&s_hi, bhi);
  }
  __syncthreads();
  int c_lo = bs[s_lo], c_hi = bs[s_hi + 1];

  // the list's fill state as the BYTE offset of its next distance slot, cnt * (4 SK_BLOCK) + 4 tid: the slot of the index
  // list is half of it, an accepted candidate advances it by one row -- three address instructions per candidate less
  // than from a count
  static_assert(SK_BLOCK == 256, "aoff >> 10 is the count");
  unsigned char* const cdb = reinterpret_cast<unsigned char*>(s_cd);
  unsigned char* const cib = reinterpret_cast<unsigned char*>(s_ci);
  unsigned aoff = 4u * tid;
  int cnt = 0;
  for (int pass = 0; pass < 2; ++pass) {
    // pass 1 only runs if a (bad) prior left some lane with fewer than K candidates: everything, without pruning
    for (int c0 = c_lo; c0 < c_hi; c0 += SK_CHUNK) {
      const int cn = min(SK_CHUNK, c_hi - c0);
      const int cn4 = (cn + 3) & ~3;
      __syncthreads();
      for (int j = tid; j < cn4; j += SK_BLOCK) {
        const bool ok = j < cn;
        // pad with NaN coordinates: the distance is NaN and `d <= tau` is false
        s_ref[j] = ok ? Sb[c0 + j] : __builtin_nanf("");
        s_ref[SK_CHUNK + j] = ok ? Sb[N + c0 + j] : __builtin_nanf("");
        s_ref[2 * SK_CHUNK + j] = ok ? Sb[2 * N + c0 + j] : __builtin_nanf("");
        s_id[j] = ok ? (uint16_t)Ib[c0 + j] : (uint16_t)0;
      }
      __syncthreads();
      // the next four candidates are requested before this step's appends (LDS stores the loads may not be moved across):
      // two waves per SIMD do not hide an LDS round trip per step (the last step's extra read stays inside the arrays)
      float4 nx = *reinterpret_cast<const float4*>(&s_ref[0]);
      float4 ny = *reinterpret_cast<const float4*>(&s_ref[SK_CHUNK]);
      float4 nz = *reinterpret_cast<const float4*>(&s_ref[2 * SK_CHUNK]);
      for (int j = 0; j < cn4; j += 4) {
        const float4 rx = nx, ry = ny, rz = nz;
        const int jn = j + 4 < SK_CHUNK ? j + 4 : j;
        nx = *reinterpret_cast<const float4*>(&s_ref[jn]);
        ny = *reinterpret_cast<const float4*>(&s_ref[SK_CHUNK + jn]);
        nz = *reinterpret_cast<const float4*>(&s_ref[2 * SK_CHUNK + jn]);
        // two candidates per instruction: v_pk_add_f32 / v_pk_mul_f32 are IEEE per component, so with contraction off
        // the distances are the bits of geoa3_sqdist (3 sub + 3 mul + 2 add = 4 packed instructions per point)
        float da[4];
        slab_sqdist2(nqx, nqy, nqz, rx.x, rx.y, ry.x, ry.y, rz.x, rz.y, da[0], da[1]);
        slab_sqdist2(nqx, nqy, nqz, rx.z, rx.w, ry.z, ry.w, rz.z, rz.w, da[2], da[3]);
        // branch-free append: the candidate is written to the list's next slot whether it qualifies or not and the
        // slot is kept only if it does (a passed-over `if` per candidate is a taken branch per candidate: with two
        // waves per SIMD the scan ran at ~140 cycles per candidate).  cnt <= CAP - 4 here (the check below)
        const uint2 ids = *reinterpret_cast<const uint2*>(&s_id[j]);
        const unsigned id4[4] = {ids.x & 0xffffu, ids.x >> 16, ids.y & 0xffffu, ids.y >> 16};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          *reinterpret_cast<float*>(cdb + aoff) = da[u];
          *reinterpret_cast<uint16_t*>(cib + (aoff >> 1)) = (uint16_t)id4[u];
          aoff += da[u] <= tau ? 4u * SK_BLOCK : 0u;
        }
        if (__builtin_expect(__any(aoff > (unsigned)(CAP - 4) * 4u * SK_BLOCK + 4u * SK_BLOCK - 1u), 0)) {   // cnt > CAP - 4
          cnt = (int)(aoff >> 10);
          if (cnt >= K) {
            tau = slab_compact<CAP>(s_ci, s_cd, cnt, K, tid);
            aoff = (unsigned)K * 4u * SK_BLOCK + 4u * tid;
          }
        }
      }
    }
    cnt = (int)(aoff >> 10);
    const bool short_list = live && cnt < K && K <= N;
    if (!__syncthreads_or(short_list)) break;
    tau = live ? S_INF : -1.f;
    cnt = 0;
    aoff = 4u * tid;
    c_lo = 0;
    c_hi = N;
  }

  if (live) {
    const int keep = cnt < K ? cnt : K;
    slab_compact<CAP>(s_ci, s_cd, cnt, keep, tid);
    float* od = dists + ((size_t)b * N + qo) * K;
    int32_t* oi = idx + ((size_t)b * N + qo) * K;
    for (int m = 0; m < K; ++m) {
      od[m] = m < keep ? s_cd[m * SK_BLOCK + tid] : S_INF;
      oi[m] = m < keep ? (int32_t)s_ci[m * SK_BLOCK + tid] : -1;
    }
  }
}

// ------------------------------------------------------------------------------------------
// The slab kernel for a cloud that fits one staging chunk (N <= SK_CHUNK) and short lists (K <= 20): a candidate is
// remembered as its POSITION in the staged run (2 bytes) instead of (distance, index) (6 bytes); the distance is formed
// again, from the same staged coordinates with the same un-fused operations (the same bits), when the list is ranked.
// 34 KB of LDS per workgroup instead of 76: four workgroups per CU instead of two -- the scan is a dependent chain per
// wavefront (compare -> list cursor -> store address) and runs at the rate of the wavefronts that interleave on a SIMD.
// The ranking (64-bit keys in registers, every pair compared once) writes the result straight to memory.
// ------------------------------------------------------------------------------------------
// compile-time loops (indices into register arrays must be constants; a 56 x 56 nest is beyond the unroller's budget)
template <typename F, int... S>
__device__ __forceinline__ void sp_static_for_impl(F&& f, std::integer_sequence<int, S...>) {
  (f(std::integral_constant<int, S>{}), ...);
}
template <int N_, typename F>
__device__ __forceinline__ void sp_static_for(F&& f) {
  sp_static_for_impl(f, std::make_integer_sequence<int, N_>{});
}
// r += (kj < ki): compare + carry back to back (left to the compiler the compares are batched and their masks spilled)
__device__ __forceinline__ void sp_count_less(int& r, unsigned long long kj, unsigned long long ki) {
  asm volatile("v_cmp_lt_u64 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, 0, %0, vcc" : "+v"(r) : "v"(kj), "v"(ki) : "vcc");
}

// Lists of 32 slots at four waves per SIMD (round 4).  With 40 slots the final ranking's keys (80 registers) spilled 35
// registers at the 128 of four waves -- 140 bytes of scratch per lane, 36 MB each way per launch at 250 instances; at three
// waves (168 VGPRs) nothing spilled but the kernel alone took 116 instead of 101 us.  32 slots: 8 bytes of scratch, the list
// is compacted when a lane passes 28 candidates instead of 36 (seeded lists hold K = 17 + the few points inside the old
// radius: p99 21), and the kernel takes 127 instead of 142 us in the bench's one-stream figure, the iteration 1.740 instead
// of 1.755 ms (three interleaved runs, profiles/round4_ab_slab_waves.txt).
// MULTI: a cloud of more than one staging chunk (N <= 65535) and / or longer lists (K <= SP_CAP - 16): the positions are
// those of the whole sorted cloud and the ranking gathers the coordinates from memory (L2) instead of the staged chunk.
template <int SP_CAP, bool MULTI>
__global__ __launch_bounds__(SK_BLOCK) __attribute__((amdgpu_waves_per_eu(MULTI ? 3 : 4, MULTI ? 3 : 4))) void knn_slabp_kernel(const float* __restrict__ R, int N, int K,
                                                             const int32_t* __restrict__ prior,
                                                             const float* __restrict__ sorted,
                                                             const int32_t* __restrict__ sidx,
                                                             const int32_t* __restrict__ bstart,
                                                             const SlabGeo* __restrict__ geo, float* __restrict__ dists,
                                                             int32_t* __restrict__ idx) {
  __shared__ __attribute__((aligned(16))) float s_ref[3 * SK_CHUNK];
  __shared__ __attribute__((aligned(16))) uint16_t s_id[SK_CHUNK];
  // !MULTI: the lists' LDS also holds, before the scan, the cloud in ORIGINAL order and last iteration's lists (the seed
  // radius is formed from LDS, not by 4 K scattered loads per thread) and, after it, the sorted rows on their way out
  constexpr int SP_KMAX = 20;
  constexpr int SP_SEED = 3 * SK_CHUNK * 4 + SK_BLOCK * SP_KMAX * 2;
  constexpr int SP_UNION = !MULTI && SP_SEED > SP_CAP * SK_BLOCK * 2 ? SP_SEED : SP_CAP * SK_BLOCK * 2;
  __shared__ __attribute__((aligned(16))) unsigned char s_un[SP_UNION];
  uint16_t* const s_pos = reinterpret_cast<uint16_t*>(s_un);
  __shared__ int s_lo, s_hi;
  const int b = blockIdx.y, tid = threadIdx.x;
  const int pos = blockIdx.x * SK_BLOCK + tid;
  const bool live = pos < N;
  const float* Rb = R + (size_t)b * 3 * N;
  const float* Sb = sorted + (size_t)b * 3 * N;
  const int32_t* Ib = sidx + (size_t)b * N;
  const int32_t* bs = bstart + (size_t)b * SB_PITCH;
  const SlabGeo g = geo[b];
  const int pc = live ? pos : N - 1;
  const float qx = Sb[pc], qy = Sb[N + pc], qz = Sb[2 * N + pc];
  float nqx, nqy, nqz;
  slab_negq(qx, qy, qz, nqx, nqy, nqz);
  const int qo = Ib[pc];                                     // the query's original index
  if (tid == 0) {
    s_lo = SB_NB;
    s_hi = -1;
  }
  float tau = S_INF;
  if (!MULTI && prior != nullptr) {
    // The seed radius: the largest distance to last iteration's K neighbours.  The cloud (original order) and the
    // workgroup's 256 rows of the old table are staged through LDS with coalesced loads (a wavefront fetches ITS 64 rows:
    // 64 K consecutive elements of ~4 rows per load instruction instead of 64 rows), then every thread walks its row.
    float* const s_org = reinterpret_cast<float*>(s_un);
    uint16_t* const s_pri = reinterpret_cast<uint16_t*>(s_un + 3 * SK_CHUNK * 4);
    for (int j = tid; j < 3 * N; j += SK_BLOCK) s_org[j] = Rb[j];
    s_id[tid] = (uint16_t)qo;
    __syncthreads();
    const int w0 = tid & ~63, lane = tid & 63;
    const float inv_k = 1.0f / (float)K;
    for (int e = lane; e < 64 * K; e += 64) {
      const int row = (int)(((float)e + 0.5f) * inv_k), m = e - row * K;
      const int32_t v = prior[((size_t)b * N + s_id[w0 + row]) * K + m];
      s_pri[w0 * K + e] = (uint16_t)(v < 0 || v >= N ? 0xffff : v);
    }
    __syncthreads();
    if (live) {
      float t = 0.f;
      bool ok = true;
      for (int m = 0; m < K; ++m) {
        const unsigned j = s_pri[tid * K + m];
        ok = ok && j != 0xffffu;
        const int jc = j >= (unsigned)N ? N - 1 : (int)j;
        t = fmaxf(t, geoa3_sqdist(qx, qy, qz, s_org[jc], s_org[N + jc], s_org[2 * N + jc]));
      }
      if (ok) tau = t;
    }
  }
  if (MULTI && prior != nullptr && live) {
    const int32_t* pr = prior + ((size_t)b * N + qo) * K;
    float t = 0.f;
    bool ok = true;
    for (int m0 = 0; m0 < K; m0 += 8) {
      int j[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) j[u] = m0 + u < K ? pr[m0 + u] : 0;
      float px[8], py[8], pz[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        ok = ok && j[u] >= 0 && j[u] < N;
        const int jc = j[u] < 0 ? 0 : (j[u] >= N ? N - 1 : j[u]);
        px[u] = Rb[jc];
        py[u] = Rb[N + jc];
        pz[u] = Rb[2 * N + jc];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (m0 + u < K) t = fmaxf(t, geoa3_sqdist(qx, qy, qz, px[u], py[u], pz[u]));
    }
    if (ok) tau = t;
  }
  if (!live) tau = -1.f;  // padding lanes never collect candidates
  __syncthreads();
  int blo = SB_NB, bhi = -1;
  if (live) {
    blo = 0;
    bhi = SB_NB - 1;
    if (tau < S_INF) {
      const float qa = g.axis == 0 ? qx : (g.axis == 1 ? qy : qz);
      const float r = sqrtf(tau) * 1.00001f + 1e-30f;
      blo = slab_bin(qa - r, g.lo, g.inv_w);
      bhi = slab_bin(qa + r, g.lo, g.inv_w);
    }
    atomicMin(&s_lo, blo);
    atomicMax(&s_hi, bhi);
  }
  // the workgroup stages the union of its 256 windows; a WAVEFRONT scans only the union of its own 64 (the queries are
  // consecutive along the slab axis: ~2/3 of the workgroup's run)
  const int w_blo = -(int)wave_max(-(float)blo), w_bhi = (int)wave_max((float)bhi);
  int wv_lo = w_blo <= w_bhi ? bs[w_blo] : 0, wv_hi = w_blo <= w_bhi ? bs[w_bhi + 1] : 0;
  __syncthreads();
  int c_lo = bs[s_lo], c_hi = bs[s_hi + 1];

  // a thread's list: positions s_pos[slot][tid]; fill state = the BYTE offset of the next slot, cnt * 2 SK_BLOCK + 2 tid
  static_assert(SK_BLOCK == 256, "poff >> 9 is the count");
  unsigned char* const pb = reinterpret_cast<unsigned char*>(s_pos);
  unsigned poff = 2u * tid;
  // positions -> keys in registers: distance bits : original index : position (indices are distinct, so the position
  // never decides an order; it rides along for the write-back); ~0 beyond the list
  unsigned long long key[SP_CAP];
  auto load_keys = [&](auto C_, int cnt) {
    sp_static_for<decltype(C_)::value>([&](auto S_) {
      constexpr int s = decltype(S_)::value;
      unsigned p = s_pos[s * SK_BLOCK + tid];
      float d;
      unsigned id;
      if (MULTI) {
        p = p < (unsigned)N ? p : 0u;     // (slots beyond the list hold anything)
        d = geoa3_sqdist(qx, qy, qz, Sb[p], Sb[N + p], Sb[2 * N + p]);
        id = (unsigned)Ib[p];
      } else {
        p &= SK_CHUNK - 1;
        d = geoa3_sqdist(qx, qy, qz, s_ref[p], s_ref[SK_CHUNK + p], s_ref[2 * SK_CHUNK + p]);
        id = s_id[p];
      }
      key[s] = s < cnt ? ((unsigned long long)__float_as_uint(d) << 32) | (id << 16) | p : ~0ull;
      if ((s & 7) == 7) __builtin_amdgcn_sched_barrier(0);   // eight gathers in flight, not forty (registers)
    });
  };
  // rank of entry s = the number of smaller keys; no rank array -- the caller uses the rank at once (registers)
  auto rank_of = [&](auto C_, auto S_) {
    constexpr int s = decltype(S_)::value;
    int r = 0;
    sp_static_for<decltype(C_)::value>([&](auto J_) {
      constexpr int j = decltype(J_)::value;
      if constexpr (j != s) sp_count_less(r, key[j], key[s]);
    });
    return r;
  };
  for (int pass = 0; pass < 2; ++pass) {
    // pass 1 only runs if a (bad) prior left some lane with fewer than K candidates: everything, without pruning
   for (int c0 = c_lo; c0 < c_hi; c0 += SK_CHUNK) {      // (!MULTI: N <= SK_CHUNK, the whole run at once)
    const int cn = min(SK_CHUNK, c_hi - c0);
    const int cn4 = (cn + 3) & ~3;
    __syncthreads();
    for (int j = tid; j < cn4; j += SK_BLOCK) {
      const bool ok = j < cn;
      // pad with NaN coordinates: the distance is NaN and `d <= tau` is false
      s_ref[j] = ok ? Sb[c0 + j] : __builtin_nanf("");
      s_ref[SK_CHUNK + j] = ok ? Sb[N + c0 + j] : __builtin_nanf("");
      s_ref[2 * SK_CHUNK + j] = ok ? Sb[2 * N + c0 + j] : __builtin_nanf("");
      s_id[j] = ok ? (uint16_t)Ib[c0 + j] : (uint16_t)0;
    }
    __syncthreads();
    const int j_lo = min(max(wv_lo - c0, 0), SK_CHUNK - 4) & ~3, j_hi = (min(wv_hi - c0, cn) + 3) & ~3;
    float4 nx = *reinterpret_cast<const float4*>(&s_ref[j_lo]);
    float4 ny = *reinterpret_cast<const float4*>(&s_ref[SK_CHUNK + j_lo]);
    float4 nz = *reinterpret_cast<const float4*>(&s_ref[2 * SK_CHUNK + j_lo]);
    for (int j = j_lo; j < j_hi; j += 4) {
      const float4 rx = nx, ry = ny, rz = nz;
      const int jn = j + 4 < SK_CHUNK ? j + 4 : j;
      nx = *reinterpret_cast<const float4*>(&s_ref[jn]);
      ny = *reinterpret_cast<const float4*>(&s_ref[SK_CHUNK + jn]);
      nz = *reinterpret_cast<const float4*>(&s_ref[2 * SK_CHUNK + jn]);
      float da[4];
      slab_sqdist2(nqx, nqy, nqz, rx.x, rx.y, ry.x, ry.y, rz.x, rz.y, da[0], da[1]);
      slab_sqdist2(nqx, nqy, nqz, rx.z, rx.w, ry.z, ry.w, rz.z, rz.w, da[2], da[3]);
      // branch-free append: the position goes to the list's next slot whether the candidate qualifies or not, and the
      // slot is kept only if it does.  cnt <= SP_CAP - 4 here (the check below)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        *reinterpret_cast<uint16_t*>(pb + poff) = (uint16_t)((MULTI ? c0 : 0) + j + u);
        poff += da[u] <= tau ? 2u * SK_BLOCK : 0u;
      }
      if (__builtin_expect(__any(poff > (unsigned)(SP_CAP - 4) * 2u * SK_BLOCK + 2u * SK_BLOCK - 1u), 0)) {   // cnt > CAP - 4
        const int cnt = (int)(poff >> 9);
        if (cnt >= K) {   // keep the K best (in order), tighten the radius; every old slot is in a register by now
          load_keys(std::integral_constant<int, SP_CAP>{}, cnt);
          sp_static_for<SP_CAP>([&](auto S_) {
            constexpr int s = decltype(S_)::value;
            const int r = rank_of(std::integral_constant<int, SP_CAP>{}, S_);
            if (r < K) s_pos[r * SK_BLOCK + tid] = (uint16_t)key[s];
            if (r == K - 1) tau = __uint_as_float((unsigned)(key[s] >> 32));
          });
          poff = (unsigned)K * 2u * SK_BLOCK + 2u * tid;
        }
      }
    }
   }
    const int cnt = (int)(poff >> 9);
    const bool short_list = live && cnt < K && K <= N;
    if (!__syncthreads_or(short_list)) break;
    tau = live ? S_INF : -1.f;
    poff = 2u * tid;
    c_lo = 0;
    c_hi = N;
    wv_lo = 0;
    wv_hi = N;
  }

  if constexpr (MULTI) {
    if (live) {
      const int cnt = (int)(poff >> 9);
      const int keep = cnt < K ? cnt : K;
      constexpr std::integral_constant<int, SP_CAP> C_{};
      load_keys(C_, cnt);
      float* od = dists + ((size_t)b * N + qo) * K;
      int32_t* oi = idx + ((size_t)b * N + qo) * K;
      sp_static_for<SP_CAP>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
        const int r = rank_of(C_, S_);
        if (r < keep) {
          od[r] = __uint_as_float((unsigned)(key[s] >> 32));
          oi[r] = (int32_t)((key[s] >> 16) & 0xffffu);
        }
      });
      for (int m = keep; m < K; ++m) {   // fewer than K points in the cloud
        od[m] = S_INF;
        oi[m] = -1;
      }
    }
  } else {
    // The final ranking at the length the wavefront's lists actually have -- seeded, a list holds the K old neighbours
    // plus the few points that came inside their radius: 20 or 28 slots instead of SP_CAP (every pair of slots is
    // compared: 380 / 756 pairs instead of 1560); the same keys in the same order either way.  The sorted rows leave
    // through LDS: a row is K consecutive floats, so 64 consecutive elements of the workgroup's 256 K are ~4 rows per
    // store instruction instead of 64 (measured: the scattered stores were a quarter of the kernel).
    const int cnt = live ? (int)(poff >> 9) : 0;
    const int keep = cnt < K ? cnt : K;
    const int cls = !__any(cnt > 20) ? 0 : (!__any(cnt > 28) ? 1 : 2);       // uniform over the wavefront
    constexpr std::integral_constant<int, 20> C20{};
    constexpr std::integral_constant<int, 28> C28{};
    constexpr std::integral_constant<int, SP_CAP> CXX{};
    if (cls == 0) load_keys(C20, cnt);
    else if (cls == 1) load_keys(C28, cnt);
    else load_keys(CXX, cnt);
    __syncthreads();                       // every list, the staged run and its ids are in registers: the LDS is free
    float* const s_od = reinterpret_cast<float*>(s_un);          // [256][K]
    uint16_t* const s_oi = reinterpret_cast<uint16_t*>(s_ref);   // [256][K]
    s_id[tid] = (uint16_t)qo;
    for (int m = keep; m < K; ++m) {       // fewer than K points in the cloud (and the padding lanes)
      s_od[tid * K + m] = S_INF;
      s_oi[tid * K + m] = 0xffff;
    }
    auto rank_out = [&](auto C_) {
      sp_static_for<decltype(C_)::value>([&](auto S_) {
        constexpr int s = decltype(S_)::value;
        const int r = rank_of(C_, S_);
        if (r < keep) {
          s_od[tid * K + r] = __uint_as_float((unsigned)(key[s] >> 32));
          s_oi[tid * K + r] = (uint16_t)(key[s] >> 16);
        }
      });
    };
    if (cls == 0) rank_out(C20);
    else if (cls == 1) rank_out(C28);
    else rank_out(CXX);
    __syncthreads();
    const float inv_k = 1.0f / (float)K;
    const int rows = min(SK_BLOCK, N - (int)blockIdx.x * SK_BLOCK);
    for (int e = tid; e < rows * K; e += SK_BLOCK) {
      const int row = (int)(((float)e + 0.5f) * inv_k), m = e - row * K;
      const size_t o = ((size_t)b * N + s_id[row]) * K + m;
      const unsigned v = s_oi[e];
      dists[o] = s_od[e];
      idx[o] = v == 0xffffu ? -1 : (int32_t)v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Cell-grid search, one WAVEFRONT per query (the K = 33 / N = 4096 regime, where the per-thread lists of the slab kernel
// leave room for one wave per SIMD and a 1-D slab still holds 10-15 % of the cloud).
// The cloud is counting-sorted into a 16^3 grid over its bounding box (cell index x fastest, so the cells x0..x1 of a
// (y, z) row are ONE contiguous run of the sorted array).  A query's radius tau is the largest distance to last
// iteration's K neighbours (K distinct points: an upper bound of the true K-th distance), so every true neighbour lies in
// the box of cells that [q - sqrt(tau), q + sqrt(tau)] touches (the cell function is monotone in the coordinate).
// The wave gives every (y, z) row of that box to a lane; the lanes walk their runs in lockstep, candidates with d <= tau
// are appended to the wave's list by ballot + popcount (64 at a time), and the K smallest by (distance, index) are
// picked by rank counting over the list -- every lane ranks its own candidates against all of them.  Distances are the
// un-fused bits of geoa3_sqdist and the order is the lexicographic one of the other kernels: BIT-IDENTICAL results.
// No per-thread lists, ~2.5 KB of LDS per wave: eight waves per SIMD hide the dependent loads of the walk.
// ------------------------------------------------------------------------------------------
constexpr int KG_G = 16, KG_CELLS = KG_G * KG_G * KG_G;
constexpr int KG_T = 1024, KG_PPT = 8;       // cell sort: up to 8192 points
constexpr int KG_CAP = 256;                  // entries of a wave's candidate list
constexpr int KG_QPW = 8;                    // queries per wave (sequential)

constexpr int KG_PITCH = ((16 * 16 * 16 + 1 + 31) / 32) * 32;   // ints per row of cell starts: whole lines (NOTEBOOK 5a)
// Cells that follow the density: per axis 15 interior boundaries near the 1/16 .. 15/16 quantiles of the cloud's
// coordinates (edges of a 256-bin histogram), cell coordinate = the number of boundaries <= the coordinate.  On a cloud
// with a dense part (75 % of the points on 1/64 of the surface) a query's box of uniform cells held 656 candidates at
// N = 4096, K = 32, of quantile cells 123 (108 / 103 on an ellipsoid: tools' CPU count); any monotone cell function keeps
// the search exact.  bnd[16 axis + m]: boundary m = 1..15 (non-decreasing), entry 0 = +inf (never counted).
struct alignas(128) GridGeo {
  float bnd[48];
};

// the largest m in 0..15 with b[m] <= v (b[1..15] non-decreasing; 0 if none): the cell coordinate
__device__ __forceinline__ int kg_cell_tab(float v, const float* b) {
  int c = v >= b[8] ? 8 : 0;
  c += v >= b[c + 4] ? 4 : 0;
  c += v >= b[c + 2] ? 2 : 0;
  c += v >= b[c + 1] ? 1 : 0;
  return c;
}

// sorted [B][3][N] coordinates in cell order, sidx [B][N] original indices, cstart [B][KG_CELLS + 1], geo [B]
__global__ __launch_bounds__(KG_T) void knn_cellsort_kernel(const float* __restrict__ pc, int N, float* __restrict__ sorted,
                                                            int32_t* __restrict__ sidx, int32_t* __restrict__ cstart,
                                                            GridGeo* __restrict__ geo) {
  __shared__ float s_red[16][6];
  __shared__ int s_cnt[KG_CELLS], s_wsum[16];
  __shared__ int s_hist[3][256];
  __shared__ float s_bnd[48];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* P = pc + (size_t)b * 3 * N;
  float px[KG_PPT], py[KG_PPT], pz[KG_PPT];
  float lo[3] = {S_INF, S_INF, S_INF}, hi[3] = {-S_INF, -S_INF, -S_INF};
#pragma unroll
  for (int p = 0; p < KG_PPT; ++p) {
    const int i = tid + p * KG_T;
    if (i < N) {
      px[p] = P[i];
      py[p] = P[N + i];
      pz[p] = P[2 * N + i];
      lo[0] = fminf(lo[0], px[p]); hi[0] = fmaxf(hi[0], px[p]);
      lo[1] = fminf(lo[1], py[p]); hi[1] = fmaxf(hi[1], py[p]);
      lo[2] = fminf(lo[2], pz[p]); hi[2] = fmaxf(hi[2], pz[p]);
    }
  }
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    lo[c] = -wave_max(-lo[c]);
    hi[c] = wave_max(hi[c]);
  }
  if (lane == 0) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      s_red[wave][c] = lo[c];
      s_red[wave][3 + c] = hi[c];
    }
  }
  for (int e = tid; e < KG_CELLS; e += KG_T) s_cnt[e] = 0;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < 16; ++w)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      lo[c] = fminf(lo[c], s_red[w][c]);
      hi[c] = fmaxf(hi[c], s_red[w][3 + c]);
    }
  // ---- per-axis histograms (256 bins over the axis' own extent) -> boundaries near the 16-quantiles
  if (tid < 768) (&s_hist[0][0])[tid] = 0;
  if (tid < 48) s_bnd[tid] = S_INF;           // (a boundary no count reaches -- or entry 0 -- stays +inf)
  float wbin[3], ibin[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float e = hi[c] - lo[c];
    wbin[c] = e * (1.00001f / 256.f);
    ibin[c] = e > 1e-30f ? 256.f / (e * 1.00001f) : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < KG_PPT; ++p) {
    const int i = tid + p * KG_T;
    if (i < N) {
      const float v[3] = {px[p], py[p], pz[p]};
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        int k = (int)floorf((v[c] - lo[c]) * ibin[c]);
        k = k < 0 ? 0 : (k > 255 ? 255 : k);       // (NaN: bin 0)
        atomicAdd(&s_hist[c][k], 1);
      }
    }
  }
  __syncthreads();
  if (wave < 3) {     // one wave per axis: four fine bins per lane
    const int c = wave;
    int h4[4], sum4 = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      h4[u] = s_hist[c][4 * lane + u];
      sum4 += h4[u];
    }
    int incl4 = sum4;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl4, o, 64);
      if (lane >= o) incl4 += v;
    }
    const int total = __builtin_amdgcn_readlane(incl4, 63);     // the points with a finite place (all of them)
    int run4 = incl4 - sum4;
    const float l0 = c == 0 ? lo[0] : (c == 1 ? lo[1] : lo[2]), w0 = c == 0 ? wbin[0] : (c == 1 ? wbin[1] : wbin[2]);
    for (int u = 0; u < 4; ++u) {
      const int before = run4, after = run4 + h4[u];
      run4 = after;
      if (after > before && total > 0) {
        // boundaries m with before < m total / 16 <= after: the upper edge of this bin
        const int m0 = (int)(((long long)before * 16) / total) + 1, m1 = (int)(((long long)after * 16) / total);
        const float edge = l0 + (float)(4 * lane + u + 1) * w0;
        for (int m = m0; m <= m1 && m <= 15; ++m) s_bnd[16 * c + m] = edge;
      }
    }
  }
  __syncthreads();
  int cell[KG_PPT];
#pragma unroll
  for (int p = 0; p < KG_PPT; ++p) {
    const int i = tid + p * KG_T;
    if (i < N) {
      cell[p] = (kg_cell_tab(pz[p], s_bnd + 32) * KG_G + kg_cell_tab(py[p], s_bnd + 16)) * KG_G + kg_cell_tab(px[p], s_bnd);
      atomicAdd(&s_cnt[cell[p]], 1);
    }
  }
  __syncthreads();
  // exclusive scan of the 4096 counters: four consecutive entries per thread
  int c4[4], sum = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    c4[u] = s_cnt[4 * tid + u];
    sum += c4[u];
  }
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  int run = incl - sum;
#pragma unroll
  for (int w = 0; w < 16; ++w) run += (w < wave) ? s_wsum[w] : 0;
  int32_t* cs = cstart + (size_t)b * KG_PITCH;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    s_cnt[4 * tid + u] = run;     // from here on: the fill cursor of the cell
    cs[4 * tid + u] = run;
    run += c4[u];
  }
  if (tid == KG_T - 1) cs[KG_CELLS] = run;
  __syncthreads();
  float* S = sorted + (size_t)b * 3 * N;
#pragma unroll
  for (int p = 0; p < KG_PPT; ++p) {
    const int i = tid + p * KG_T;
    if (i < N) {
      const int pos = atomicAdd(&s_cnt[cell[p]], 1);   // the order inside a cell is free: the selection does not depend on it
      S[pos] = px[p];
      S[N + pos] = py[p];
      S[2 * N + pos] = pz[p];
      sidx[(size_t)b * N + pos] = i;
    }
  }
  if (tid < 48) geo[b].bnd[tid] = s_bnd[tid];
}

__device__ __forceinline__ unsigned long long kg_key(float d, int i) {   // d >= 0: its bits order like the value
  return ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)i;
}

// element idx of a UNIFORM array through a 32-bit byte offset (idx < 2^30): `global_load v, voffset, s[base]` -- no 64-bit
// address arithmetic per lane
template <class T>
__device__ __forceinline__ T kg_ld(const T* base, unsigned idx) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + idx * (unsigned)sizeof(T));
}
template <class T>
__device__ __forceinline__ void kg_st(T* base, unsigned idx, T v) {
  *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + idx * (unsigned)sizeof(T)) = v;
}

__global__ __launch_bounds__(256) void knn_grid_kernel(const float* __restrict__ R, int N, int K,
                                                       const int32_t* __restrict__ prior,
                                                       const float* __restrict__ sorted, const int32_t* __restrict__ sidx,
                                                       const int32_t* __restrict__ cstart, const GridGeo* __restrict__ geo,
                                                       float* __restrict__ dists, int32_t* __restrict__ idx) {
  __shared__ unsigned long long s_key[4][KG_CAP];
  __shared__ int s_rowp[4][64], s_rows[4][64];   // per wave: exclusive prefix of the rows' lengths, their first positions
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // (uniform row bases + UNSIGNED 32-bit indices everywhere below: scalar-base addressing, no 64-bit address arithmetic per
  //  lane -- the kernel is bound by VALU issue)
  const float* Rb = R + (size_t)b * 3 * N;
  const float* Sb = sorted + (size_t)b * 3 * N;
  const float *Sy = Sb + N, *Sz = Sb + 2 * (size_t)N, *Ry = Rb + N, *Rz = Rb + 2 * (size_t)N;
  const int32_t* Ib = sidx + (size_t)b * N;
  const int32_t* cs = cstart + (size_t)b * KG_PITCH;
  // lane 16 axis + m holds boundary m of the axis: a coordinate's cell = the number of its axis' boundaries <= it
  const float bl = lane < 48 ? geo[b].bnd[lane] : S_INF;
  auto cellq = [&](float v, int axis) {      // v wave-uniform
    const unsigned long long mk = __ballot(v >= bl) & (0xfffeull << (16 * axis));
    return (int)__builtin_popcountll(mk);
  };
  unsigned long long* L = s_key[wave];
  int* P = s_rowp[wave];
  int* Sr = s_rows[wave];
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int q0 = (blockIdx.x * 4 + wave) * KG_QPW;
  // A query costs five dependent round trips (its coordinates + index, its prior neighbours, their coordinates, the cell
  // table, the candidates).  The first three are taken off the critical path: lanes 0..KG_QPW-1 load the wave's queries
  // at once, and while query qi is searched the prior indices of query qi + 2 and the neighbour coordinates of query
  // qi + 1 are already in flight (K <= 64: one neighbour per lane).
  float vqx = 0.f, vqy = 0.f, vqz = 0.f;
  int vqo = 0;
  if (lane < KG_QPW && q0 + lane < N) {
    const unsigned ql = (unsigned)(q0 + lane);
    vqx = kg_ld(Sb, ql);
    vqy = kg_ld(Sy, ql);
    vqz = kg_ld(Sz, ql);
    vqo = kg_ld(Ib, ql);
  }
  const bool pipe = K <= 64;
  auto load_prior = [&](int qi) {     // neighbour `lane` of query qi (or -1)
    if (qi >= KG_QPW || q0 + qi >= N || lane >= K) return -1;
    const int32_t* pr = prior + ((size_t)b * N + __builtin_amdgcn_readlane(vqo, qi)) * K;   // uniform
    return (int)kg_ld(pr, (unsigned)lane);
  };
  int jn = pipe ? load_prior(0) : -1;           // indices whose coordinates are requested next
  float cx = 0.f, cy = 0.f, cz = 0.f;            // coordinates of the CURRENT query's neighbour
  bool cok = true;
  auto load_coords = [&](int j) {
    cok = lane >= K || (j >= 0 && j < N);
    const unsigned jj = j >= 0 && j < N ? (unsigned)j : 0u;
    cx = kg_ld(Rb, jj);
    cy = kg_ld(Ry, jj);
    cz = kg_ld(Rz, jj);
  };
  if (pipe) {
    load_coords(jn);
    jn = load_prior(1);
  }
  for (int qi = 0; qi < KG_QPW; ++qi) {
    const int pos = q0 + qi;
    if (pos >= N) break;                       // wave-uniform
    const float qx = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vqx), qi));
    const float qy = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vqy), qi));
    const float qz = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(vqz), qi));
    const int qo = __builtin_amdgcn_readlane(vqo, qi);
    // radius: the largest distance to last iteration's K neighbours
    float t = 0.f;
    bool ok = true;
    if (pipe) {
      ok = cok;
      if (lane < K && cok) t = geoa3_sqdist(qx, qy, qz, cx, cy, cz);
      load_coords(jn);                 // query qi + 1
      jn = load_prior(qi + 2);
    } else {
      for (int m = lane; m < K; m += 64) {
        const int j = prior[((size_t)b * N + qo) * K + m];
        if (j < 0 || j >= N) ok = false;
        else t = fmaxf(t, geoa3_sqdist(qx, qy, qz, Rb[j], Rb[N + j], Rb[2 * N + j]));
      }
    }
    float tau = wave_max(t);
    if (__any(!ok)) tau = S_INF;               // no usable radius: the whole grid
    int cnt = 0;                                // wave-uniform
    // second pass only when a degenerate prior (repeated indices) left fewer than K candidates within its radius
    for (int pass = 0; pass < 2; ++pass) {
    int x0 = 0, x1 = KG_G - 1, y0 = 0, y1 = KG_G - 1, z0 = 0, z1 = KG_G - 1;
    if (tau < S_INF) {
      const float r = sqrtf(tau) * 1.00001f + 1e-30f;
      x0 = cellq(qx - r, 0); x1 = cellq(qx + r, 0);
      y0 = cellq(qy - r, 1); y1 = cellq(qy + r, 1);
      z0 = cellq(qz - r, 2); z1 = cellq(qz + r, 2);
    }
    const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
    const float inv_ny = 1.f / (float)ny;     // row / ny below, exact for row < 256, ny <= 16 (no integer division: ~40 instructions)
    cnt = 0;
    for (int rb = 0; rb < nrows; rb += 64) {
      const int row = rb + lane;
      int s = 0, len = 0;
      if (row < nrows) {
        const int rz = (int)(((float)row + 0.5f) * inv_ny);
        const int zz = z0 + rz, yy = y0 + row - rz * ny;
        const int c0 = (zz * KG_G + yy) * KG_G;
        s = kg_ld(cs, (unsigned)(c0 + x0));
        len = kg_ld(cs, (unsigned)(c0 + x1 + 1)) - s;
      }
      // The rows' point ranges, concatenated, are walked 64 candidates at a time (lane = candidate): a lane per ROW left a
      // quarter of the lanes busy and made the loop as long as the longest row -- one L2 round trip per iteration.
      int incl = len;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      const int T = __builtin_amdgcn_readlane(incl, 63);
      P[lane] = incl - len;      // this wave's own slots: LDS operations of a wave execute in order
      Sr[lane] = s;
#ifndef GEOA3_KG_U
#define GEOA3_KG_U 3
#endif
      constexpr int KG_U = GEOA3_KG_U;   // batches of 64 candidates whose loads are in flight together (a query's ~150 candidates
                                         // used to cost one dependent L2 round trip per 64)
      for (int tb = 0; tb < T; tb += 64 * KG_U) {
      float fx[KG_U], fy[KG_U], fz[KG_U];
      int fo[KG_U];
#pragma unroll
      for (int u = 0; u < KG_U; ++u) {
        const int c = tb + 64 * u + lane;
        fx[u] = fy[u] = fz[u] = 0.f;
        fo[u] = 0;
        if (c < T) {
          int r = 0;             // the last row whose range starts at or before candidate c
#pragma unroll
          for (int st = 32; st > 0; st >>= 1)
            if (P[r + st] <= c) r += st;
          const unsigned j = (unsigned)(Sr[r] + (c - P[r]));
          fx[u] = kg_ld(Sb, j);
          fy[u] = kg_ld(Sy, j);
          fz[u] = kg_ld(Sz, j);
          fo[u] = kg_ld(Ib, j);
        }
      }
#pragma unroll
      for (int u = 0; u < KG_U; ++u) {
        const int t0 = tb + 64 * u;
        if (t0 >= T) break;      // wave-uniform
        const int c = t0 + lane;
        const float d = c < T ? geoa3_sqdist(qx, qy, qz, fx[u], fy[u], fz[u]) : S_INF;
        const int oi = fo[u];
        const bool pass = c < T && d <= tau;
        const unsigned long long mask = __ballot(pass);
        if (pass) L[cnt + __popcll(mask & lt)] = kg_key(d, oi);
        cnt += __popcll(mask);
        if (cnt > KG_CAP - 64) {                // wave-uniform, rare: keep the K best, tighten the radius
          // rank counting over the list; the K smallest move to the front (through registers: 4 entries per lane)
          unsigned long long mine[KG_CAP / 64];
          int rk[KG_CAP / 64];
#pragma unroll
          for (int u = 0; u < KG_CAP / 64; ++u) {
            const int c = lane + 64 * u;
            mine[u] = c < cnt ? L[c] : ~0ull;
            rk[u] = 0;
          }
          for (int j2 = 0; j2 < cnt; ++j2) {
            const unsigned long long kj = L[j2];
#pragma unroll
            for (int u = 0; u < KG_CAP / 64; ++u) rk[u] += kj < mine[u] ? 1 : 0;
          }
          const int keep = cnt < K ? cnt : K;
#pragma unroll
          for (int u = 0; u < KG_CAP / 64; ++u)
            if (lane + 64 * u < cnt && rk[u] < keep) L[rk[u]] = mine[u];
          cnt = keep;
          if (keep == K) tau = __uint_as_float((unsigned)(L[K - 1] >> 32));
        }
      }
      }
    }
    if (cnt >= K || K > N || !(tau < S_INF)) break;
    tau = S_INF;
    }
    // the K smallest of the list by (distance, index): every lane ranks its own entries -- ONE per lane when the list holds
    // at most 64 (the usual case: the K old neighbours + the few points that moved inside their radius), so the rank loop
    // is a broadcast read, a compare and an add per entry instead of four compares (the kernel is bound by VALU issue:
    // ~700 instructions per query, 250 of them here)
    {
      float* od = dists + ((size_t)b * N + qo) * K;
      int32_t* oi = idx + ((size_t)b * N + qo) * K;
      auto rank_emit = [&](auto ul) {
        constexpr int UL = decltype(ul)::value;
        unsigned long long mine[UL];
        int rk[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
          const int c = lane + 64 * u;
          mine[u] = c < cnt ? L[c] : ~0ull;
          rk[u] = 0;
        }
#pragma unroll 4
        for (int j2 = 0; j2 < cnt; ++j2) {
          const unsigned long long kj = L[j2];
#pragma unroll
          for (int u = 0; u < UL; ++u) rk[u] += kj < mine[u] ? 1 : 0;
        }
#pragma unroll
        for (int u = 0; u < UL; ++u)
          if (lane + 64 * u < cnt && rk[u] < K) {
            kg_st(od, (unsigned)rk[u], __uint_as_float((unsigned)(mine[u] >> 32)));
            kg_st(oi, (unsigned)rk[u], (int32_t)(unsigned)(mine[u] & 0xffffffffull));
          }
      };
      if (cnt <= 64) rank_emit(std::integral_constant<int, 1>{});
      else if (cnt <= 128) rank_emit(std::integral_constant<int, 2>{});
      else rank_emit(std::integral_constant<int, KG_CAP / 64>{});
      for (int m = cnt + lane; m < K; m += 64) {   // fewer than K candidates (K > N): the all-pairs kernel's padding
        od[m] = S_INF;
        oi[m] = -1;
      }
    }
  }
}

struct GridScratch {
  float* sorted;
  int32_t* sidx;
  int32_t* cstart;
  GridGeo* geo;
  size_t total;
};
GridScratch grid_carve(void* base, int B, int N) {
  GridScratch s{};
  size_t off = 0;
  char* p = static_cast<char*>(base);
  auto take = [&](size_t bytes) {
    void* r = p ? p + off : nullptr;
    off += (bytes + 255) / 256 * 256;
    return r;
  };
  s.sorted = (float*)take((size_t)B * 3 * N * 4);
  s.sidx = (int32_t*)take((size_t)B * N * 4);
  s.cstart = (int32_t*)take((size_t)B * KG_PITCH * 4);
  s.geo = (GridGeo*)take((size_t)B * sizeof(GridGeo));
  s.total = off;
  return s;
}

struct SlabScratch {
  float* sorted;
  int32_t* sidx;
  int32_t* bstart;
  SlabGeo* geo;
  size_t total;
};
SlabScratch slab_carve(void* base, int B, int N) {
  SlabScratch s{};
  size_t off = 0;
  char* p = static_cast<char*>(base);
  auto take = [&](size_t bytes) {
    void* r = p ? p + off : nullptr;
    off += (bytes + 255) / 256 * 256;
    return r;
  };
  s.sorted = (float*)take((size_t)B * 3 * N * 4);
  s.sidx = (int32_t*)take((size_t)B * N * 4);
  s.bstart = (int32_t*)take((size_t)B * SB_PITCH * 4);
  s.geo = (SlabGeo*)take((size_t)B * sizeof(SlabGeo));
  s.total = off;
  return s;
}

}  // namespace

extern "C" int64_t geoa3_knn_self_scratch_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return -1;
  const size_t a = slab_carve(nullptr, B, N).total, g = grid_carve(nullptr, B, N).total;
  return (int64_t)(a > g ? a : g);
}

extern "C" int geoa3_knn_self(const float* pc, int B, int N, int K, const int32_t* prior, float* dists, int32_t* idx,
                              void* scratch, int method, void* stream) {
  if (!pc || !dists || !idx || B <= 0 || N <= 0 || K <= 0 || K > GEOA3_KNN_MAX_K) return GEOA3_EINVAL;
  hipStream_t s = geoa3_stream(stream);
  geoa3_prof_begin(GEOA3_PROF_KNN, s);
  int rc = GEOA3_OK;
  // method 0: the cell grid (one wave per query) for large lists / large clouds, the slab kernel otherwise
  const bool grid = method == 2 || (method == 0 && (K > 20 || N >= 2048));   // (1, 3, 4: the slab kernels)
  if (!prior || !scratch || N > SB_T * SB_PPT || K > N || ((uintptr_t)scratch & 255) != 0) {
    rc = geoa3_launch_knn(pc, pc, B, N, N, K, prior, dists, idx, nullptr, s);   // nothing to prune with
  } else if (grid && K <= KG_CAP - 64) {
    const GridScratch gs = grid_carve(scratch, B, N);
    hipLaunchKernelGGL(knn_cellsort_kernel, dim3(B), dim3(KG_T), 0, s, pc, N, gs.sorted, gs.sidx, gs.cstart, gs.geo);
    dim3 ggrid((N + 4 * KG_QPW - 1) / (4 * KG_QPW), B);
    hipLaunchKernelGGL(knn_grid_kernel, ggrid, dim3(256), 0, s, pc, N, K, prior, gs.sorted, gs.sidx, gs.cstart, gs.geo,
                       dists, idx);
    if (hipGetLastError() != hipSuccess) rc = GEOA3_ELAUNCH;
  } else {
    const SlabScratch sc = slab_carve(scratch, B, N);
    hipLaunchKernelGGL(slab_bin_kernel, dim3(B), dim3(SB_T), 0, s, pc, N, sc.sorted, sc.sidx, sc.bstart, sc.geo);
    dim3 grid((N + SK_BLOCK - 1) / SK_BLOCK, B);
#define SLAB_LAUNCH(CAPV)                                                                                       \
  do {                                                                                                          \
    constexpr int CAP = CAPV;                                                                                   \
    const size_t lds = 3 * SK_CHUNK * 4 + (size_t)CAP * SK_BLOCK * 6 + SK_CHUNK * 2;                            \
    if (lds > 64 * 1024)                                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(knn_slab_kernel<CAP>),                            \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                          \
    hipLaunchKernelGGL(knn_slab_kernel<CAP>, grid, dim3(SK_BLOCK), lds, s, pc, N, K, prior, sc.sorted, sc.sidx, \
                       sc.bstart, sc.geo, dists, idx);                                                          \
  } while (0)
    // method 3 / 4 (tests, A/B runs): the (distance, index)-list kernel / the position-list kernel whatever the launch size
    const bool slabp = method != 3, slabp_always = method == 4;
    // positions instead of (distance, index) lists: twice the occupancy -- for launches the (distance, index) kernel cannot
    // hold at once (two workgroups per CU); a small shard's launch runs beside the victim's kernels, where the denser
    // kernel cost more than it saved (32 instances: 0.500 -> 0.506 ms per iteration)
    if (K <= 20 && N <= SK_CHUNK && slabp && ((size_t)grid.x * grid.y > 512 || slabp_always))
      hipLaunchKernelGGL((knn_slabp_kernel<32, false>), grid, dim3(SK_BLOCK), 0, s, pc, N, K, prior, sc.sorted, sc.sidx,
                         sc.bstart, sc.geo, dists, idx);
    else if (K <= 40 && N <= 65535 && slabp && ((size_t)grid.x * grid.y > 512 || slabp_always))
      hipLaunchKernelGGL((knn_slabp_kernel<56, true>), grid, dim3(SK_BLOCK), 0, s, pc, N, K, prior, sc.sorted, sc.sidx,
                         sc.bstart, sc.geo, dists, idx);
    else if (K <= 20) SLAB_LAUNCH(40);
    else if (K <= 40) SLAB_LAUNCH(72);
    else SLAB_LAUNCH(96);
#undef SLAB_LAUNCH
    if (hipGetLastError() != hipSuccess) rc = GEOA3_ELAUNCH;
  }
  geoa3_prof_end(GEOA3_PROF_KNN, s);
  return rc;
}
