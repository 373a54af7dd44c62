// Nearest-neighbour kernels (gfx950): K=1 cross search in both directions ("CD kernel") and the
// general K-NN with a per-query candidate buffer.  Replaces pytorch3d.ops.knn_points as called from
// Lib/loss_utils.py:32,33,41,48,57,70,77,92 of the reference.
//
// Layout: clouds are planar [B,3,N]; a workgroup stages the reference cloud of ONE instance in LDS
// as three planes (x|y|z), every lane reads the same LDS address (broadcast, conflict free) while
// holding its own query point(s) in registers.  The N x N distance matrix is never materialised.
#include "common.h"
#include "profile.h"
#include "geom_internal.h"

namespace {

constexpr int NN_BLOCK = 256;
constexpr int NN_CHUNK = 2048;  // reference points staged per LDS pass (24 KB)

// ------------------------------------------------------------------------------------------
// K = 1.  blockIdx.z selects the direction: 0: queries = a, refs = r;  1: queries = r, refs = a.
// Each thread owns QPT queries (q, q+256, ...) so that one LDS broadcast feeds QPT distance chains.
// ------------------------------------------------------------------------------------------
template <int QPT>
__global__ __launch_bounds__(NN_BLOCK) void nn1_pair_kernel(const float* __restrict__ A, const float* __restrict__ R,
                                                            int Na, int Nr, float* __restrict__ d_ar,
                                                            int32_t* __restrict__ i_ar, float* __restrict__ d_ra,
                                                            int32_t* __restrict__ i_ra,
                                                            const uint8_t* __restrict__ only) {
  // `only` (optional, [2][B][max(Na,Nr)] bytes, direction-major): restrict the search to the flagged queries --
  // the exact fallback of a pruned search; a block without a flagged query exits at once.
  __shared__ __attribute__((aligned(16))) float s_ref[3 * NN_CHUNK];
  const int b = blockIdx.y;
  const bool swap = blockIdx.z != 0;
  const float* Q = swap ? R : A;
  const float* P = swap ? A : R;
  const int Nq = swap ? Nr : Na;
  const int Np = swap ? Na : Nr;
  float* dout = swap ? d_ra : d_ar;
  int32_t* iout = swap ? i_ra : i_ar;
  const int q0 = blockIdx.x * (NN_BLOCK * QPT) + threadIdx.x;
  if (blockIdx.x * (NN_BLOCK * QPT) >= Nq) return;  // whole block out of range (uniform)
  const int nmax = Na > Nr ? Na : Nr;
  const uint8_t* sel = only ? only + ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * nmax : nullptr;
  if (sel) {
    bool any = false;
#pragma unroll
    for (int t = 0; t < QPT; ++t) {
      const int q = q0 + t * NN_BLOCK;
      any = any || (q < Nq && sel[q] != 0);
    }
    if (!__syncthreads_or(any)) return;
  }

  const float* Qb = Q + (size_t)b * 3 * Nq;
  const float* Pb = P + (size_t)b * 3 * Np;
  float qx[QPT], qy[QPT], qz[QPT], best[QPT];
  int32_t bi[QPT];
#pragma unroll
  for (int t = 0; t < QPT; ++t) {
    int q = q0 + t * NN_BLOCK;
    int qc = q < Nq ? q : Nq - 1;
    qx[t] = Qb[qc];
    qy[t] = Qb[Nq + qc];
    qz[t] = Qb[2 * Nq + qc];
    best[t] = __builtin_inff();
    bi[t] = 0;
  }

  for (int c0 = 0; c0 < Np; c0 += NN_CHUNK) {
    const int cn = min(NN_CHUNK, Np - c0);
    const int cn4 = (cn + 3) & ~3;
    __syncthreads();
    for (int j = threadIdx.x; j < cn4; j += NN_BLOCK) {
      const bool ok = j < cn;
      // pad with +inf coordinates: distance becomes +inf (or NaN), never < best
      s_ref[j] = ok ? Pb[c0 + j] : __builtin_inff();
      s_ref[NN_CHUNK + j] = ok ? Pb[Np + c0 + j] : __builtin_inff();
      s_ref[2 * NN_CHUNK + j] = ok ? Pb[2 * Np + c0 + j] : __builtin_inff();
    }
    __syncthreads();
    for (int j = 0; j < cn4; j += 4) {
      const float4 rx = *reinterpret_cast<const float4*>(&s_ref[j]);
      const float4 ry = *reinterpret_cast<const float4*>(&s_ref[NN_CHUNK + j]);
      const float4 rz = *reinterpret_cast<const float4*>(&s_ref[2 * NN_CHUNK + j]);
      const float rxa[4] = {rx.x, rx.y, rx.z, rx.w};
      const float rya[4] = {ry.x, ry.y, ry.z, ry.w};
      const float rza[4] = {rz.x, rz.y, rz.z, rz.w};
#pragma unroll
      for (int u = 0; u < 4; ++u) {
#pragma unroll
        for (int t = 0; t < QPT; ++t) {
          const float d = geoa3_sqdist(qx[t], qy[t], qz[t], rxa[u], rya[u], rza[u]);
          const bool lt = d < best[t];  // strict: the first (lowest) index wins a tie
          best[t] = lt ? d : best[t];
          bi[t] = lt ? (c0 + j + u) : bi[t];
        }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < QPT; ++t) {
    int q = q0 + t * NN_BLOCK;
    if (q < Nq && (!sel || sel[q] != 0)) {
      dout[(size_t)b * Nq + q] = best[t];
      iout[(size_t)b * Nq + q] = bi[t];
    }
  }
}

// ------------------------------------------------------------------------------------------
// General K.  One query per thread.  Every reference point whose distance is <= tau (the
// query's current pruning radius) is appended to the thread's candidate list in LDS; when a
// list is full the wavefront compacts: each lane keeps its K best and tightens tau to its K-th
// distance.  With a good prior (last iteration's neighbours) tau is tight from the start and
// no compaction happens.  The final pass orders the survivors by (distance, index).
// ------------------------------------------------------------------------------------------
constexpr int KNN_BLOCK = 256;
constexpr int KNN_CHUNK = 1024;

__device__ __forceinline__ bool lex_less(float da, int ia, float db, int ib) {
  return (da < db) || (da == db && ia < ib);
}

// Select the K lexicographically smallest (d, idx) among this lane's cnt candidates, write them
// sorted to (outd, outi) registers-free via LDS columns; returns the K-th distance.
template <int CAP>
__device__ __forceinline__ float knn_compact(uint16_t* cand /*[CAP][KNN_BLOCK]*/, float* candd /*[CAP][KNN_BLOCK]*/,
                                             int cnt, int K, int tid) {
  // selection by repeated lexicographic-minimum extraction into the front of the list
  // (in-place selection sort over the lane's own column; columns never alias across lanes)
  float kth = __builtin_inff();
  for (int p = 0; p < K; ++p) {
    float bd = candd[p * KNN_BLOCK + tid];
    int bidx = cand[p * KNN_BLOCK + tid];
    int bs = p;
    for (int s = p + 1; s < cnt; ++s) {
      float d = candd[s * KNN_BLOCK + tid];
      int i = cand[s * KNN_BLOCK + tid];
      if (lex_less(d, i, bd, bidx)) {
        bd = d;
        bidx = i;
        bs = s;
      }
    }
    if (bs != p) {
      candd[bs * KNN_BLOCK + tid] = candd[p * KNN_BLOCK + tid];
      cand[bs * KNN_BLOCK + tid] = cand[p * KNN_BLOCK + tid];
      candd[p * KNN_BLOCK + tid] = bd;
      cand[p * KNN_BLOCK + tid] = (uint16_t)bidx;
    }
    kth = bd;
  }
  return kth;
}

template <int CAP>
__global__ __launch_bounds__(KNN_BLOCK) void knn_kernel(const float* __restrict__ Q, const float* __restrict__ R,
                                                        int Nq, int Nr, int K, const int32_t* __restrict__ prior,
                                                        float* __restrict__ dists, int32_t* __restrict__ idx,
                                                        const uint8_t* __restrict__ only) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_ref = reinterpret_cast<float*>(smem);                                    // 3*KNN_CHUNK floats
  float* s_cd = s_ref + 3 * KNN_CHUNK;                                              // [CAP][KNN_BLOCK]
  uint16_t* s_ci = reinterpret_cast<uint16_t*>(s_cd + (size_t)CAP * KNN_BLOCK);     // [CAP][KNN_BLOCK]

  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int q = blockIdx.x * KNN_BLOCK + tid;
  bool live = q < Nq;
  if (only) {  // exact fallback of a pruned search: only the flagged queries, whole block skipped if none
    live = live && only[(size_t)b * Nq + q] != 0;
    if (!__syncthreads_or(live)) return;
  }
  const int qc = q < Nq ? q : Nq - 1;
  const float* Qb = Q + (size_t)b * 3 * Nq;
  const float* Rb = R + (size_t)b * 3 * Nr;
  const float qx = Qb[qc], qy = Qb[Nq + qc], qz = Qb[2 * Nq + qc];

  float tau = __builtin_inff();
  if (prior != nullptr && live) {
    const int32_t* pr = prior + ((size_t)b * Nq + q) * K;
    float t = 0.f;
    bool ok = true;
    for (int m = 0; m < K; ++m) {
      int j = pr[m];
      if (j < 0 || j >= Nr) {
        ok = false;
        break;
      }
      t = fmaxf(t, geoa3_sqdist(qx, qy, qz, Rb[j], Rb[Nr + j], Rb[2 * Nr + j]));
    }
    if (ok) tau = t;
  }
  if (!live) tau = -1.f;  // padding lanes never collect candidates

  int cnt = 0;
  for (int pass = 0; pass < 2; ++pass) {
    // pass 1 only runs if a (bad) prior left some lane with fewer than K candidates
    for (int c0 = 0; c0 < Nr; c0 += KNN_CHUNK) {
      const int cn = min(KNN_CHUNK, Nr - c0);
      const int cn4 = (cn + 3) & ~3;
      __syncthreads();
      for (int j = tid; j < cn4; j += KNN_BLOCK) {
        const bool ok = j < cn;
        // pad with NaN coordinates: the distance is NaN and `d <= tau` is false
        s_ref[j] = ok ? Rb[c0 + j] : __builtin_nanf("");
        s_ref[KNN_CHUNK + j] = ok ? Rb[Nr + c0 + j] : __builtin_nanf("");
        s_ref[2 * KNN_CHUNK + j] = ok ? Rb[2 * Nr + c0 + j] : __builtin_nanf("");
      }
      __syncthreads();
      for (int j = 0; j < cn4; j += 4) {
        const float4 rx = *reinterpret_cast<const float4*>(&s_ref[j]);
        const float4 ry = *reinterpret_cast<const float4*>(&s_ref[KNN_CHUNK + j]);
        const float4 rz = *reinterpret_cast<const float4*>(&s_ref[2 * KNN_CHUNK + j]);
        const float rxa[4] = {rx.x, rx.y, rx.z, rx.w};
        const float rya[4] = {ry.x, ry.y, ry.z, ry.w};
        const float rza[4] = {rz.x, rz.y, rz.z, rz.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float d = geoa3_sqdist(qx, qy, qz, rxa[u], rya[u], rza[u]);
          if (d <= tau) {
            s_cd[cnt * KNN_BLOCK + tid] = d;
            s_ci[cnt * KNN_BLOCK + tid] = (uint16_t)(c0 + j + u);
            ++cnt;
          }
        }
        // a list may grow by 4 per group: compact as soon as any lane could overflow next time
        if (__builtin_expect(__any(cnt > CAP - 4), 0)) {
          if (cnt >= K) {
            tau = knn_compact<CAP>(s_ci, s_cd, cnt, K, tid);
            cnt = K;
          }
        }
      }
    }
    const bool short_list = live && cnt < K && K <= Nr;
    if (!__syncthreads_or(short_list)) break;
    // some lane was starved by a non-distinct / stale prior: redo exactly, without pruning
    tau = live ? __builtin_inff() : -1.f;
    cnt = 0;
  }

  if (live) {
    const int keep = cnt < K ? cnt : K;
    knn_compact<CAP>(s_ci, s_cd, cnt, keep, tid);
    float* od = dists + ((size_t)b * Nq + q) * K;
    int32_t* oi = idx + ((size_t)b * Nq + q) * K;
    for (int m = 0; m < K; ++m) {
      od[m] = m < keep ? s_cd[m * KNN_BLOCK + tid] : __builtin_inff();
      oi[m] = m < keep ? (int32_t)s_ci[m * KNN_BLOCK + tid] : -1;
    }
  }
}

}  // namespace

int geoa3_launch_nn1(const float* a, const float* r, int B, int Na, int Nr, float* d_ar, int32_t* i_ar, float* d_ra,
                     int32_t* i_ra, const uint8_t* only, hipStream_t s) {
  const int ndir = d_ra ? 2 : 1;
  const int nmax = ndir == 2 ? (Na > Nr ? Na : Nr) : Na;
  constexpr int QPT = 2;
  dim3 grid((nmax + NN_BLOCK * QPT - 1) / (NN_BLOCK * QPT), B, ndir);
  hipLaunchKernelGGL(nn1_pair_kernel<QPT>, grid, dim3(NN_BLOCK), 0, s, a, r, Na, Nr, d_ar, i_ar, d_ra, i_ra, only);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

int geoa3_launch_knn(const float* q, const float* r, int B, int Nq, int Nr, int K, const int32_t* prior, float* dists,
                     int32_t* idx, const uint8_t* only, hipStream_t s) {
  dim3 grid((Nq + KNN_BLOCK - 1) / KNN_BLOCK, B);
#define KNN_LAUNCH(CAPV)                                                                                         \
  do {                                                                                                           \
    constexpr int CAP = CAPV;                                                                                    \
    const size_t lds = 3 * KNN_CHUNK * 4 + (size_t)CAP * KNN_BLOCK * 6;                                          \
    if (lds > 64 * 1024)                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(knn_kernel<CAP>),                                  \
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
    hipLaunchKernelGGL(knn_kernel<CAP>, grid, dim3(KNN_BLOCK), lds, s, q, r, Nq, Nr, K, prior, dists, idx, only); \
  } while (0)
  if (K <= 20) KNN_LAUNCH(40);
  else if (K <= 40) KNN_LAUNCH(72);
  else KNN_LAUNCH(96);
#undef KNN_LAUNCH
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_nn1_pair(const float* a, const float* r, int B, int Na, int Nr, float* d_ar, int32_t* i_ar,
                              float* d_ra, int32_t* i_ra, void* stream) {
  if (!a || !r || !d_ar || !i_ar || B <= 0 || Na <= 0 || Nr <= 0) return GEOA3_EINVAL;
  if ((d_ra == nullptr) != (i_ra == nullptr)) return GEOA3_EINVAL;
  geoa3_prof_begin(GEOA3_PROF_NN1, geoa3_stream(stream));
  const int rc = geoa3_launch_nn1(a, r, B, Na, Nr, d_ar, i_ar, d_ra, i_ra, nullptr, geoa3_stream(stream));
  geoa3_prof_end(GEOA3_PROF_NN1, geoa3_stream(stream));
  return rc;
}

extern "C" int geoa3_knn(const float* q, const float* r, int B, int Nq, int Nr, int K, const int32_t* prior,
                         float* dists, int32_t* idx, void* stream) {
  if (!q || !r || !dists || !idx || B <= 0 || Nq <= 0 || Nr <= 0 || K <= 0 || K > GEOA3_KNN_MAX_K)
    return GEOA3_EINVAL;
  if (Nr > 65535) return GEOA3_ENOSUPPORT;  // candidate indices are stored as uint16 in LDS
  geoa3_prof_begin(GEOA3_PROF_KNN, geoa3_stream(stream));
  const int rc = geoa3_launch_knn(q, r, B, Nq, Nr, K, prior, dists, idx, nullptr, geoa3_stream(stream));
  geoa3_prof_end(GEOA3_PROF_KNN, geoa3_stream(stream));
  return rc;
}
