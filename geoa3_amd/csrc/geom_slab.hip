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
#include "geom_internal.h"
#include "profile.h"

namespace {

constexpr int SB_T = 1024;            // binning kernel: threads
constexpr int SB_NB = 512;            // bins
constexpr int SB_PPT = 8;             // up to 8192 points
constexpr int SK_BLOCK = 256;
constexpr int SK_CHUNK = 1024;
constexpr float S_INF = __builtin_inff();

struct SlabGeo {
  float lo, inv_w;
  int axis, pad;
};

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
  for (int e = tid; e <= SB_NB; e += SB_T) bstart[(size_t)b * (SB_NB + 1) + e] = s_start[e];
  if (tid == 0) geo[b] = SlabGeo{alo, inv_w, axis, 0};
}

typedef float slab_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void slab_sqdist2(float qx, float qy, float qz, float x0, float x1, float y0, float y1,
                                             float z0, float z1, float& d0, float& d1) {
#pragma clang fp contract(off)
  const slab_f2 dx = slab_f2{qx, qx} - slab_f2{x0, x1};
  const slab_f2 dy = slab_f2{qy, qy} - slab_f2{y0, y1};
  const slab_f2 dz = slab_f2{qz, qz} - slab_f2{z0, z1};
  const slab_f2 xx = dx * dx, yy = dy * dy, zz = dz * dz;
  const slab_f2 s = xx + yy;
  const slab_f2 d = s + zz;
  d0 = d.x;
  d1 = d.y;
}

__device__ __forceinline__ bool slab_lex_less(float da, int ia, float db, int ib) {
  return (da < db) || (da == db && ia < ib);
}

template <int CAP>
__device__ __forceinline__ float slab_compact(uint16_t* cand, float* candd, int cnt, int K, int tid) {
  float kth = S_INF;
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
  const int32_t* bs = bstart + (size_t)b * (SB_NB + 1);
  const SlabGeo g = geo[b];
  const int pc = live ? pos : N - 1;
  const float qx = Sb[pc], qy = Sb[N + pc], qz = Sb[2 * N + pc];
  const int qo = Ib[pc];                                     // the query's original index
  if (tid == 0) {
    s_lo = SB_NB;
    s_hi = -1;
  }
  float tau = S_INF;
  if (prior != nullptr && live) {
    const int32_t* pr = prior + ((size_t)b * N + qo) * K;
    float t = 0.f;
    bool ok = true;
    for (int m = 0; m < K; ++m) {
      const int j = pr[m];
      if (j < 0 || j >= N) {
        ok = false;
        break;
      }
      t = fmaxf(t, geoa3_sqdist(qx, qy, qz, Rb[j], Rb[N + j], Rb[2 * N + j]));
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
      for (int j = 0; j < cn4; j += 4) {
        const float4 rx = *reinterpret_cast<const float4*>(&s_ref[j]);
        const float4 ry = *reinterpret_cast<const float4*>(&s_ref[SK_CHUNK + j]);
        const float4 rz = *reinterpret_cast<const float4*>(&s_ref[2 * SK_CHUNK + j]);
        // two candidates per instruction: v_pk_add_f32 / v_pk_mul_f32 are IEEE per component, so with contraction off
        // the distances are the bits of geoa3_sqdist (3 sub + 3 mul + 2 add = 4 packed instructions per point)
        float da[4];
        slab_sqdist2(qx, qy, qz, rx.x, rx.y, ry.x, ry.y, rz.x, rz.y, da[0], da[1]);
        slab_sqdist2(qx, qy, qz, rx.z, rx.w, ry.z, ry.w, rz.z, rz.w, da[2], da[3]);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float d = da[u];
          if (d <= tau) {
            s_cd[cnt * SK_BLOCK + tid] = d;
            s_ci[cnt * SK_BLOCK + tid] = s_id[j + u];
            ++cnt;
          }
        }
        if (__builtin_expect(__any(cnt > CAP - 4), 0)) {
          if (cnt >= K) {
            tau = slab_compact<CAP>(s_ci, s_cd, cnt, K, tid);
            cnt = K;
          }
        }
      }
    }
    const bool short_list = live && cnt < K && K <= N;
    if (!__syncthreads_or(short_list)) break;
    tau = live ? S_INF : -1.f;
    cnt = 0;
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
  s.bstart = (int32_t*)take((size_t)B * (SB_NB + 1) * 4);
  s.geo = (SlabGeo*)take((size_t)B * sizeof(SlabGeo));
  s.total = off;
  return s;
}

}  // namespace

extern "C" int64_t geoa3_knn_self_scratch_bytes(int B, int N) {
  if (B <= 0 || N <= 0) return -1;
  return (int64_t)slab_carve(nullptr, B, N).total;
}

extern "C" int geoa3_knn_self(const float* pc, int B, int N, int K, const int32_t* prior, float* dists, int32_t* idx,
                              void* scratch, void* stream) {
  if (!pc || !dists || !idx || B <= 0 || N <= 0 || K <= 0 || K > GEOA3_KNN_MAX_K) return GEOA3_EINVAL;
  hipStream_t s = geoa3_stream(stream);
  geoa3_prof_begin(GEOA3_PROF_KNN, s);
  int rc = GEOA3_OK;
  if (!prior || !scratch || N > SB_T * SB_PPT || K > N || ((uintptr_t)scratch & 255) != 0) {
    rc = geoa3_launch_knn(pc, pc, B, N, N, K, prior, dists, idx, nullptr, s);   // nothing to prune with
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
    if (K <= 20) SLAB_LAUNCH(40);
    else if (K <= 40) SLAB_LAUNCH(72);
    else SLAB_LAUNCH(96);
#undef SLAB_LAUNCH
    if (hipGetLastError() != hipSuccess) rc = GEOA3_ELAUNCH;
  }
  geoa3_prof_end(GEOA3_PROF_KNN, s);
  return rc;
}
