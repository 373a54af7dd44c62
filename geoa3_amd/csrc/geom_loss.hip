// Curvature (kappa) and the fused geometric objective + gradient (gfx950).
// Reference: Lib/loss_utils.py:25-97 (norm_l2_loss, chamfer_loss, pseudo_chamfer_loss, hausdorff_loss,
// _get_kappa_ori, _get_kappa_adv, curvature_loss) and Lib/utility.py:30-31 (_normalize), as combined by
// Attacker/geoA3_attack.py:131-166.  The reference builds [b,3,n,k] neighbour tensors with knn_gather and
// lets autograd scatter the gradient back; here one workgroup owns one instance, keeps the cloud and the
// gradient accumulators in LDS, and never materialises the neighbour tensor.
#include <type_traits>
#include "common.h"
#include "profile.h"

namespace {

constexpr float NORM_EPS = 1e-12f;  // Lib/utility.py:30 (_normalize eps)

// | < normalize(q - p), n > | summed over the k neighbours listed in nb[1..k] (nb[0] is dropped).
template <typename Fetch>
__device__ __forceinline__ float kappa_point(float px, float py, float pz, float nx, float ny, float nz,
                                             const int32_t* __restrict__ nb, int k, Fetch fetch) {
  float acc = 0.f;
  for (int m = 1; m <= k; ++m) {
    float qx, qy, qz;
    fetch(nb[m], qx, qy, qz);
    const float vx = qx - px, vy = qy - py, vz = qz - pz;
    const float r = sqrtf(vx * vx + vy * vy + vz * vz);
    const float inv = 1.0f / fmaxf(r, NORM_EPS);
    const float t = (vx * inv) * nx + (vy * inv) * ny + (vz * inv) * nz;
    acc += fabsf(t);
  }
  return acc / (float)k;
}

// LDS = true (3 N floats fit): the instance's cloud is staged in LDS once per workgroup of 1024 points and the k
// neighbours of a point are gathered from there -- at N = 4096, k = 32 the gather from global memory fetched 6.3 GB for
// 0.16 GB of algorithmic traffic (profiles/round4_v5_c5_summary.md; once per batch).
template <bool LDS>
__global__ __launch_bounds__(LDS ? 1024 : 256) void kappa_kernel(const float* __restrict__ pc, const float* __restrict__ normal,
                                                                 const int32_t* __restrict__ knn_idx,
                                                                 const int32_t* __restrict__ nn_idx, int N, int Nn, int k,
                                                                 float* __restrict__ kappa) {
  extern __shared__ __attribute__((aligned(16))) float s_kp[];
  const int b = blockIdx.y;
  const int i = blockIdx.x * (LDS ? 1024 : 256) + threadIdx.x;
  const float* P = pc + (size_t)b * 3 * N;
  if (LDS) {
    for (int e = threadIdx.x; e < 3 * N; e += 1024) s_kp[e] = P[e];
    __syncthreads();
  }
  if (i >= N) return;
  const float* Nm = normal + (size_t)b * 3 * Nn;
  const int ni = nn_idx ? nn_idx[(size_t)b * N + i] : i;
  const float nx = Nm[ni], ny = Nm[Nn + ni], nz = Nm[2 * Nn + ni];
  const int32_t* nb = knn_idx + ((size_t)b * N + i) * (k + 1);
  const float* Q = LDS ? s_kp : P;
  auto fetch = [&](int j, float& x, float& y, float& z) {
    x = Q[j];
    y = Q[N + j];
    z = Q[2 * N + j];
  };
  kappa[(size_t)b * N + i] = kappa_point(Q[i], Q[N + i], Q[2 * N + i], nx, ny, nz, nb, k, fetch);
}

// ------------------------------------------------------------------------------------------
// Fused objective: one workgroup per instance.
// ------------------------------------------------------------------------------------------
constexpr int GEO_BLOCK = 1024;
constexpr int GEO_WAVES = GEO_BLOCK / GEOA3_WAVE;

struct MaxIdx {
  float v;
  int i;
};
__device__ __forceinline__ MaxIdx better(MaxIdx a, MaxIdx b) {  // larger value, then lower index
  return (b.v > a.v || (b.v == a.v && b.i < a.i)) ? b : a;
}

// DET: the gradient is accumulated by the OWNER of every point in a fixed order (own terms, then the contributions it
// receives from other points sorted by their source) instead of with LDS float atomics, whose order is free: bit-for-
// bit reproducible, independent of timing and of which other instances share the batch.  The reverse lists (who pulls
// on point q: its occurrences in other points' neighbour lists, and the clean points whose nearest adversarial point it
// is) are built per launch in LDS, for as many sources at a time as fit (rcap entries): histogram (integer atomics:
// exact), scan, unordered fill, then every owner sorts its own short list by source key.  A contribution is recomputed
// by the owner with the expression its source uses for its own share, so the two agree bit for bit.
template <bool DET>
__global__ __launch_bounds__(GEO_BLOCK) void geo_loss_grad_kernel(geoa3_geo_args A, int rcap, int nrm_in_lds) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = A.N, k = A.k, b = blockIdx.x, tid = threadIdx.x;
  float* s_ax = sm;            // adv planes
  float* s_ay = sm + N;
  float* s_az = sm + 2 * N;
  float* s_gx = sm + 3 * N;    // gradient accumulators
  float* s_gy = sm + 4 * N;
  float* s_gz = sm + 5 * N;
  float* s_e = sm + 6 * N;     // kappa_adv - kappa_ori[nn]
  float* s_red = sm + 7 * N;   // [GEO_WAVES * 5] reduction scratch + results
  const int Nr = A.Nr > 0 ? A.Nr : N;   // points of the clean cloud (dense-cloud path: N = npoint < Nr)
  const float* adv = A.adv + (size_t)b * 3 * N;
  const float* ori = A.ori + (size_t)b * 3 * Nr;
  const size_t bN = (size_t)b * N, bNr = (size_t)b * Nr;
  const bool do_curv = (A.w_curv != 0.f || A.dkappa != nullptr) && A.knn_adv != nullptr;
  const bool do_cd = A.dis_type == 1;
  const bool do_l2 = A.dis_type == 2;
  const bool two_side = do_cd && !A.single_side && A.d_oa != nullptr;
  const bool do_hd = A.w_hd != 0.f && A.d_ao != nullptr;

  for (int i = tid; i < N; i += GEO_BLOCK) {
    s_ax[i] = adv[i];
    s_ay[i] = adv[N + i];
    s_az[i] = adv[2 * N + i];
    s_gx[i] = 0.f;
    s_gy[i] = 0.f;
    s_gz[i] = 0.f;
  }
  __syncthreads();

  // ---- phase A: per-point terms and block reductions
  float sum_ao = 0.f, sum_oa = 0.f, sum_e2 = 0.f;
  MaxIdx hd{-__builtin_inff(), 0x7fffffff};
  for (int i = tid; i < N; i += GEO_BLOCK) {
    if (do_cd) {
      const float d = A.d_ao[bN + i];
      sum_ao += d;
    } else if (do_l2) {
      const float dx = s_ax[i] - ori[i], dy = s_ay[i] - ori[Nr + i], dz = s_az[i] - ori[2 * Nr + i];
      sum_ao += dx * dx + dy * dy + dz * dz;
    }
    if (do_hd) hd = better(hd, MaxIdx{A.d_ao[bN + i], i});
    if (do_curv) {
      const int ni = A.i_ao[bN + i];
      const float* Nm = A.normal_ori + (size_t)b * 3 * Nr;
      const int32_t* nb = A.knn_adv + (bN + i) * (size_t)(k + 1);
      auto fetch = [&](int j, float& x, float& y, float& z) {
        x = s_ax[j];
        y = s_ay[j];
        z = s_az[j];
      };
      const float kap = kappa_point(s_ax[i], s_ay[i], s_az[i], Nm[ni], Nm[Nr + ni], Nm[2 * Nr + ni], nb, k, fetch);
      const float e = kap - (A.kappa_ori ? A.kappa_ori[bNr + ni] : 0.f);
      s_e[i] = e;
      sum_e2 += e * e;
      if (A.kappa_adv) A.kappa_adv[bN + i] = kap;
    }
  }
  if (two_side)
    for (int i = tid; i < Nr; i += GEO_BLOCK) sum_oa += A.d_oa[bNr + i];
  sum_ao = wave_sum(sum_ao);
  sum_oa = wave_sum(sum_oa);
  sum_e2 = wave_sum(sum_e2);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MaxIdx other{__shfl_xor(hd.v, o, 64), __shfl_xor(hd.i, o, 64)};
    hd = better(hd, other);
  }
  const int wave = tid >> 6, lane = tid & 63;
  if (lane == 0) {
    s_red[wave * 5 + 0] = sum_ao;
    s_red[wave * 5 + 1] = sum_oa;
    s_red[wave * 5 + 2] = sum_e2;
    s_red[wave * 5 + 3] = hd.v;
    s_red[wave * 5 + 4] = __int_as_float(hd.i);
  }
  __syncthreads();
  if (tid == 0) {
    float a = 0.f, o = 0.f, e2 = 0.f;
    MaxIdx h{-__builtin_inff(), 0x7fffffff};
    for (int w = 0; w < GEO_WAVES; ++w) {
      a += s_red[w * 5 + 0];
      o += s_red[w * 5 + 1];
      e2 += s_red[w * 5 + 2];
      h = better(h, MaxIdx{s_red[w * 5 + 3], __float_as_int(s_red[w * 5 + 4])});
    }
    const float invN = 1.0f / (float)N;
    float dis = 0.f;
    if (do_cd) dis = a * invN + (two_side ? o * (1.0f / (float)Nr) : 0.f);
    if (do_l2) dis = a;
    const float hdv = do_hd ? h.v : 0.f;
    const float curv = do_curv ? e2 * invN : 0.f;
    float con = 0.f;
    if (A.dis_type != 0) con = A.w_dis * dis;
    if (do_hd) con = con + A.w_hd * hdv;
    if (do_curv) con = con + A.w_curv * curv;
    if (A.dis_loss) A.dis_loss[b] = dis;
    if (A.hd_loss) A.hd_loss[b] = hdv;
    if (A.curv_loss) A.curv_loss[b] = curv;
    if (A.constrain) A.constrain[b] = con;
    s_red[GEO_WAVES * 5] = __int_as_float(h.i);
  }
  __syncthreads();
  if (A.grad == nullptr) return;
  const int hd_arg = __float_as_int(s_red[GEO_WAVES * 5]);

  // ---- phase B (deterministic): own terms, then the pulls every point receives, source by source in ascending order
  if constexpr (DET) {
    const float invN = 1.0f / (float)N;
    const float c_cd = A.w_dis * invN * 2.0f;
    const float c_cd_r = A.w_dis * (1.0f / (float)Nr) * 2.0f;
    const int k1 = k + 1;
    int* s_cnt = reinterpret_cast<int*>(s_red + GEO_WAVES * 5 + 4);   // [N + 1] counts -> offsets
    float* s_nrm = reinterpret_cast<float*>(s_cnt + N + 1);            // [3 N] normal of every adversarial point (optional)
    int* rlist = reinterpret_cast<int*>(s_nrm + (nrm_in_lds ? 3 * N : 0));   // [rcap] reverse lists of the current chunk
    const float* Nm = A.normal_ori ? A.normal_ori + (size_t)b * 3 * Nr : nullptr;
    if (do_curv && nrm_in_lds)
      for (int i = tid; i < N; i += GEO_BLOCK) {
        const int ni = A.i_ao[bN + i];
        s_nrm[i] = Nm[ni];
        s_nrm[N + i] = Nm[Nr + ni];
        s_nrm[2 * N + i] = Nm[2 * Nr + ni];
      }
    __syncthreads();
    auto normal_of = [&](int i, float& nx, float& ny, float& nz) {
      if (nrm_in_lds) {
        nx = s_nrm[i];
        ny = s_nrm[N + i];
        nz = s_nrm[2 * N + i];
      } else {
        const int ni = A.i_ao[bN + i];
        nx = Nm[ni];
        ny = Nm[Nr + ni];
        nz = Nm[2 * Nr + ni];
      }
    };
    // d kappa-term / d q for the pair (centre with normal n and coefficient dk, neighbour q): what the centre subtracts
    // from its own gradient and q adds to its
    auto pair_grad = [&](float px, float py, float pz, float nx, float ny, float nz, float dk, int q, float& dvx,
                         float& dvy, float& dvz) {
      const float vx = s_ax[q] - px, vy = s_ay[q] - py, vz = s_az[q] - pz;
      const float r = sqrtf(vx * vx + vy * vy + vz * vz);
      const float inv = 1.0f / fmaxf(r, NORM_EPS);
      const float ux = vx * inv, uy = vy * inv, uz = vz * inv;
      const float t = ux * nx + uy * ny + uz * nz;
      const float sg = t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f);
      const float c = dk * sg * inv;
      if (r >= NORM_EPS) {  // d(v/|v|)/dv = (I - u u^T)/|v|
        dvx = c * (nx - t * ux);
        dvy = c * (ny - t * uy);
        dvz = c * (nz - t * uz);
      } else {  // clamp active: v/eps, the norm path carries no gradient
        dvx = c * nx;
        dvy = c * ny;
        dvz = c * nz;
      }
    };
    auto coeff = [&](int i) {
      return (A.dkappa ? A.dkappa[bN + i] : A.w_curv * invN * 2.0f * s_e[i]) / (float)k;
    };
    // own terms
    for (int i = tid; i < N; i += GEO_BLOCK) {
      float gx = 0.f, gy = 0.f, gz = 0.f;
      const float px = s_ax[i], py = s_ay[i], pz = s_az[i];
      if (do_cd || do_hd) {
        const int j = A.i_ao[bN + i];
        const float dx = px - ori[j], dy = py - ori[Nr + j], dz = pz - ori[2 * Nr + j];
        float c = do_cd ? c_cd : 0.f;
        if (do_hd && i == hd_arg) c += A.w_hd * 2.0f;
        gx += c * dx;
        gy += c * dy;
        gz += c * dz;
      }
      if (do_l2) {
        const float c = A.w_dis * 2.0f;
        gx += c * (px - ori[i]);
        gy += c * (py - ori[Nr + i]);
        gz += c * (pz - ori[2 * Nr + i]);
      }
      if (do_curv) {
        float nx, ny, nz;
        normal_of(i, nx, ny, nz);
        const int32_t* nb = A.knn_adv + (bN + i) * (size_t)k1;
        const float dk = coeff(i);
        for (int m = 1; m <= k; ++m) {
          float dvx, dvy, dvz;
          pair_grad(px, py, pz, nx, ny, nz, dk, nb[m], dvx, dvy, dvz);
          gx -= dvx;
          gy -= dvy;
          gz -= dvz;
        }
      }
      s_gx[i] = gx;
      s_gy[i] = gy;
      s_gz[i] = gz;
    }
    // received terms.  Sources are taken in CHUNKS whose reverse lists fit LDS (ascending: neighbour pairs (i, m) keyed
    // i * (k+1) + m, then the clean points); per chunk: histogram over destinations (integer atomics: exact), scan, fill
    // (free order), then every owner sorts its short list by key and adds the contributions in that order -- the result
    // does not depend on the chunking.
    const float cr = (Nr != N) ? c_cd_r : c_cd;
    auto pass = [&](bool knn, int lo, int hi) {     // kNN pairs of the centres lo..hi-1, or the clean points lo..hi-1
      const int nent = knn ? (hi - lo) * k : hi - lo;
      for (int i = tid; i <= N; i += GEO_BLOCK) s_cnt[i] = 0;
      __syncthreads();
      for (int e = tid; e < nent; e += GEO_BLOCK) {
        int q;
        if (knn) {
          const int i = lo + e / k, m = e - (e / k) * k + 1;
          q = A.knn_adv[(bN + i) * (size_t)k1 + m];
        } else {
          q = A.i_oa[bNr + lo + e];
        }
        atomicAdd(&s_cnt[q], 1);
      }
      __syncthreads();
      if (tid < GEOA3_WAVE) {    // exclusive scan of s_cnt[0..N]: 64 lanes x consecutive runs
        const int per = (N + 1 + GEOA3_WAVE - 1) / GEOA3_WAVE, a0 = tid * per, a1 = min(a0 + per, N + 1);
        int sum = 0;
        for (int i = a0; i < a1; ++i) sum += s_cnt[i];
        int incl = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int v = __shfl_up(incl, o, 64);
          if (tid >= o) incl += v;
        }
        int run = incl - sum;
        for (int i = a0; i < a1; ++i) {
          const int c = s_cnt[i];
          s_cnt[i] = run;
          run += c;
        }
      }
      __syncthreads();
      for (int e = tid; e < nent; e += GEO_BLOCK) {   // fill: s_cnt[q] advances from the start to the end of q's list
        int q, key;
        if (knn) {
          const int i = lo + e / k, m = e - (e / k) * k + 1;
          q = A.knn_adv[(bN + i) * (size_t)k1 + m];
          key = i * k1 + m;
        } else {
          q = A.i_oa[bNr + lo + e];
          key = lo + e;
        }
        rlist[atomicAdd(&s_cnt[q], 1)] = key;
      }
      __syncthreads();
      for (int i = tid; i < N; i += GEO_BLOCK) {
        const int st = i ? s_cnt[i - 1] : 0, n = s_cnt[i] - st;
        if (n == 0) continue;
        int* L = rlist + st;
        if (n <= 24) {           // ascending by source key: insertion sort; heap sort for the rare long list
          for (int a = 1; a < n; ++a) {
            const int v = L[a];
            int c = a - 1;
            while (c >= 0 && L[c] > v) {
              L[c + 1] = L[c];
              --c;
            }
            L[c + 1] = v;
          }
        } else {
          auto sift = [&](int start, int end) {
            int root = start;
            for (;;) {
              int child = 2 * root + 1;
              if (child > end) break;
              if (child + 1 <= end && L[child] < L[child + 1]) ++child;
              if (L[root] >= L[child]) break;
              const int tmp = L[root];
              L[root] = L[child];
              L[child] = tmp;
              root = child;
            }
          };
          for (int h0 = (n - 2) / 2; h0 >= 0; --h0) sift(h0, n - 1);
          for (int end = n - 1; end > 0; --end) {
            const int tmp = L[0];
            L[0] = L[end];
            L[end] = tmp;
            sift(0, end - 1);
          }
        }
        float gx = s_gx[i], gy = s_gy[i], gz = s_gz[i];
        const float px = s_ax[i], py = s_ay[i], pz = s_az[i];
        for (int e = 0; e < n; ++e) {
          const int key = L[e];
          if (knn) {               // point `src` lists this point as one of its neighbours
            const int src = key / k1;
            float nx, ny, nz, dvx, dvy, dvz;
            normal_of(src, nx, ny, nz);
            pair_grad(s_ax[src], s_ay[src], s_az[src], nx, ny, nz, coeff(src), i, dvx, dvy, dvz);
            gx += dvx;
            gy += dvy;
            gz += dvz;
          } else {                 // clean point `key` has this point as its nearest adversarial point
            gx += cr * (px - ori[key]);
            gy += cr * (py - ori[Nr + key]);
            gz += cr * (pz - ori[2 * Nr + key]);
          }
        }
        s_gx[i] = gx;
        s_gy[i] = gy;
        s_gz[i] = gz;
      }
      __syncthreads();
    };
    if (do_curv) {
      const int cs = max(1, rcap / k);
      for (int lo = 0; lo < N; lo += cs) pass(true, lo, min(N, lo + cs));
    }
    if (two_side)
      for (int lo = 0; lo < Nr; lo += rcap) pass(false, lo, min(Nr, lo + rcap));
  } else {
  // ---- phase B: d constrain / d adv.  Own-point terms go through registers, neighbour terms
    //      are scattered with LDS float atomics (ds_add_f32).
    const float invN = 1.0f / (float)N;
    const float c_cd = A.w_dis * invN * 2.0f;
    const float c_cd_r = A.w_dis * (1.0f / (float)Nr) * 2.0f;
    if (two_side && Nr != N) {  // ori point i pulls its nearest adversarial point (dense-cloud path: Nr > N)
      for (int i = tid; i < Nr; i += GEO_BLOCK) {
        const int a = A.i_oa[bNr + i];
        const float dx = s_ax[a] - ori[i], dy = s_ay[a] - ori[Nr + i], dz = s_az[a] - ori[2 * Nr + i];
        atomicAdd(&s_gx[a], c_cd_r * dx);
        atomicAdd(&s_gy[a], c_cd_r * dy);
        atomicAdd(&s_gz[a], c_cd_r * dz);
      }
    }
    for (int i = tid; i < N; i += GEO_BLOCK) {
      float gx = 0.f, gy = 0.f, gz = 0.f;
      const float px = s_ax[i], py = s_ay[i], pz = s_az[i];
      if (do_cd || do_hd) {
        const int j = A.i_ao[bN + i];
        const float dx = px - ori[j], dy = py - ori[Nr + j], dz = pz - ori[2 * Nr + j];
        float c = do_cd ? c_cd : 0.f;
        if (do_hd && i == hd_arg) c += A.w_hd * 2.0f;
        gx += c * dx;
        gy += c * dy;
        gz += c * dz;
      }
      if (two_side && Nr == N) {  // ori point i pulls its nearest adversarial point
        const int a = A.i_oa[bN + i];
        const float dx = s_ax[a] - ori[i], dy = s_ay[a] - ori[Nr + i], dz = s_az[a] - ori[2 * Nr + i];
        atomicAdd(&s_gx[a], c_cd * dx);
        atomicAdd(&s_gy[a], c_cd * dy);
        atomicAdd(&s_gz[a], c_cd * dz);
      }
      if (do_l2) {
        const float c = A.w_dis * 2.0f;
        gx += c * (px - ori[i]);
        gy += c * (py - ori[Nr + i]);
        gz += c * (pz - ori[2 * Nr + i]);
      }
      if (do_curv) {
        const int ni = A.i_ao[bN + i];
        const float* Nm = A.normal_ori + (size_t)b * 3 * Nr;
        const float nx = Nm[ni], ny = Nm[Nr + ni], nz = Nm[2 * Nr + ni];
        const int32_t* nb = A.knn_adv + (bN + i) * (size_t)(k + 1);
        const float dk = (A.dkappa ? A.dkappa[bN + i] : A.w_curv * invN * 2.0f * s_e[i]) / (float)k;
        for (int m = 1; m <= k; ++m) {
          const int q = nb[m];
          const float vx = s_ax[q] - px, vy = s_ay[q] - py, vz = s_az[q] - pz;
          const float r = sqrtf(vx * vx + vy * vy + vz * vz);
          const float inv = 1.0f / fmaxf(r, NORM_EPS);
          const float ux = vx * inv, uy = vy * inv, uz = vz * inv;
          const float t = ux * nx + uy * ny + uz * nz;
          const float s = t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f);
          const float c = dk * s * inv;
          float dvx, dvy, dvz;
          if (r >= NORM_EPS) {  // d(v/|v|)/dv = (I - u u^T)/|v|
            dvx = c * (nx - t * ux);
            dvy = c * (ny - t * uy);
            dvz = c * (nz - t * uz);
          } else {  // clamp active: v/eps, the norm path carries no gradient
            dvx = c * nx;
            dvy = c * ny;
            dvz = c * nz;
          }
          atomicAdd(&s_gx[q], dvx);
          atomicAdd(&s_gy[q], dvy);
          atomicAdd(&s_gz[q], dvz);
          gx -= dvx;
          gy -= dvy;
          gz -= dvz;
        }
      }
      atomicAdd(&s_gx[i], gx);
      atomicAdd(&s_gy[i], gy);
      atomicAdd(&s_gz[i], gz);
    }
  }
  __syncthreads();
  float* G = A.grad + (size_t)b * 3 * N;
  for (int i = tid; i < N; i += GEO_BLOCK) {
    G[i] = s_gx[i];
    G[N + i] = s_gy[i];
    G[2 * N + i] = s_gz[i];
  }
}



// ------------------------------------------------------------------------------------------
// Order-free accumulation of gradient terms: 64-bit FIXED POINT at two scales, chosen per instance.
//
// Integer addition is associative, so sums of converted terms do not depend on who adds them in which order (LDS integer
// atomics): deterministic and batch-independent without reverse lists.  Round 4 used ONE fixed scale (2^-44 per unit, every
// term clamped silently at 2^18): right for the attack loop's magnitudes, wrong for a caller-supplied dkappa, for loss
// weights far from 1 and for near-coincident pairs (a pair term is ~ 2 dk / r).  Now, with X = the instance's largest
// coefficient (2 w_curv / (N k) or max |dkappa| / k; the Chamfer coefficients) and Ex = ceil(log2 X):
//   fine    unit 2^(Ex - 40), terms up to 2^(Ex + 10):  4096 of them still fit 63 bits; X-relative precision 2^-40;
//   coarse  unit 2^(Ex - 16), terms up to 2^(Ex + 34):  pairs down to r ~ 1e-10 at the largest coefficient -- they are
//           rare, so their sums live in a small hash pool in LDS keyed by the destination point;
//   beyond that (or NaN, or a full pool): the destination's gradient is written as NaN -- loud, never a silent clamp.
// The result is fine * 2^(Ex - 40) + coarse * 2^(Ex - 16) in fp32.
// ------------------------------------------------------------------------------------------
struct GeoFix {
  float to_f, to_c, from_f, from_c, lim_f, lim_c;
};
__device__ __forceinline__ float geo_pow2(int k) { return __uint_as_float((unsigned)(k + 127) << 23); }   // -126 <= k <= 127
__device__ __forceinline__ GeoFix geo_fix_make(float X) {
  int Ex = (int)((__float_as_uint(X) >> 23) & 0xffu) - 126;        // X < 2^Ex (X = m 2^Ex, 0.5 <= m < 1)
  Ex = Ex < -80 ? -80 : (Ex > 60 ? 60 : Ex);                        // (X = 0, denormal or absurd: any scale will do)
  GeoFix f;
  f.to_f = geo_pow2(40 - Ex);
  f.to_c = geo_pow2(16 - Ex);
  f.from_f = geo_pow2(Ex - 40);
  f.from_c = geo_pow2(Ex - 16);
  f.lim_f = geo_pow2(Ex + 10);
  f.lim_c = geo_pow2(Ex + 34);
  return f;
}
__device__ __forceinline__ unsigned long long geo_fix_conv(float v, float mul) {
  return (unsigned long long)__float2ll_rn(v * mul);
}
// hash pool of destinations with coarse sums: key [cap] (-1 = empty), acc [cap][W] 64-bit words
struct GeoPool {
  int* key;
  unsigned long long* acc;
  int cap;   // a power of two
};
__device__ __forceinline__ int geo_pool_find(const GeoPool& P, int q, bool insert) {
  unsigned h = ((unsigned)q * 0x9E3779B1u) >> 8;
  for (int step = 0; step < P.cap; ++step) {
    const int s = (int)((h + (unsigned)step) & (unsigned)(P.cap - 1));
    int kk = P.key[s];
    if (kk == q) return s;
    if (kk == -1) {
      if (!insert) return -1;
      kk = atomicCAS(&P.key[s], -1, q);
      if (kk == -1 || kk == q) return s;
    }
  }
  return -1;
}
__device__ __forceinline__ void geo_mark_bad(unsigned* s_bad, int q) { atomicOr(&s_bad[q >> 5], 1u << (q & 31)); }
// one gradient term of destination q: into the per-point fine sums (planes of N, or null: everything through the pool,
// words 0-2 fine / 3-5 coarse), the pool's coarse sums, or the sticky NaN flags
template <int W>
__device__ __forceinline__ void geo_fix_add(const GeoFix& F, unsigned long long* fine, int N, const GeoPool& P, unsigned* s_bad,
                                            int q, float x, float y, float z) {
  const float m = fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z));
  if (m <= F.lim_f) {
    if (W == 3) {
      atomicAdd(&fine[q], geo_fix_conv(x, F.to_f));
      atomicAdd(&fine[N + q], geo_fix_conv(y, F.to_f));
      atomicAdd(&fine[2 * N + q], geo_fix_conv(z, F.to_f));
    } else {
      const int s = geo_pool_find(P, q, true);
      if (s < 0) return geo_mark_bad(s_bad, q);
      atomicAdd(&P.acc[s * W + 0], geo_fix_conv(x, F.to_f));
      atomicAdd(&P.acc[s * W + 1], geo_fix_conv(y, F.to_f));
      atomicAdd(&P.acc[s * W + 2], geo_fix_conv(z, F.to_f));
    }
  } else if (m <= F.lim_c) {
    const int s = geo_pool_find(P, q, true);
    if (s < 0) return geo_mark_bad(s_bad, q);
    atomicAdd(&P.acc[s * W + W - 3], geo_fix_conv(x, F.to_c));
    atomicAdd(&P.acc[s * W + W - 2], geo_fix_conv(y, F.to_c));
    atomicAdd(&P.acc[s * W + W - 1], geo_fix_conv(z, F.to_c));
  } else {
    geo_mark_bad(s_bad, q);      // out of range or NaN
  }
}
constexpr int GEO_POOL_CAP = 128;                                     // geo_fused_kernel: overflowed rows (fine + coarse sums)
constexpr size_t GEO_POOL_BYTES = GEO_POOL_CAP * (4 + 6 * 8) + 8;     // keys + sums + alignment
constexpr int GB_POOL_CAP = 256;                                      // geo_big_kernel: destinations with coarse terms
// the instance's largest coefficient: max |dkappa| / k over the block (dkappa mode) or the loss's analytic bound (kappa is a
// mean of |cosines|: |kappa_adv - kappa_ori| <= 1), and the Chamfer coefficients
__device__ __forceinline__ float geo_coef_bound(const geoa3_geo_args& A, int N, int Nr, float block_max_dkappa) {
  const float k = (float)(A.k > 0 ? A.k : 1);
  float X = A.dkappa ? block_max_dkappa / k : fabsf(A.w_curv) * 2.0f / ((float)N * k);
  X = fmaxf(X, fabsf(A.w_dis) * 2.0f / (float)(N < Nr ? N : Nr));
  return X;
}

// ------------------------------------------------------------------------------------------
// Deterministic objective for clouds of at most 1024 points (the default there): one 1024-thread workgroup per instance
// as above, but every (centre, neighbour) PAIR a lane and no dependent chain longer than one round trip.
//
// The kernel above walks a point's k neighbours in serial loops with a dependent load of the neighbour index inside, four
// times over (kappa, own terms, histogram, fill), sorts every reverse list by insertion through LDS and recomputes the
// pulls one after the other: ~100 us per launch whatever the batch size.  Here:
//   phase 1  lane = pair (i, m), G = 2^ceil(log2 k) lanes per centre, N G / 1024 passes with the next pass's table entries
//            already in flight: kappa_adv[i] and e[i] by a G-lane butterfly (DPP), the pair's gradient term, its negative
//            summed over m (the centre's own curvature term), and the pair appends its SOURCE index to the row of its
//            destination in LDS (integer atomic: exact; free order) -- the reverse lists exist after one pass over the
//            table, no histogram / scan / fill.  The clean points append themselves (N + j) to their nearest
//            adversarial point's row.
//   phase 2  thread = owner: its row sorted in REGISTERS (bitonic network of 16 / 32 / 64 keys, no LDS traffic), the pulls
//            recomputed from the LDS-resident cloud, normals and coefficients with the expression phase 1 uses, added
//            in ascending source order.
// Measured steps on the way (250 / 32 instances): a separate pair kernel + range workgroups that gather 12-byte pull
// records from memory: 37 + 72 / 7 + 47 us (the random record reads); the same with a CSR built by global atomics:
// 113 + 148 + 103 us (global atomics run at ~50 G/s here).  Bit-for-bit reproducible for any batch composition.
// Rows that overflow their capacity (coincident points) are summed by their owner from a scan of the table, in the same
// order.
// ------------------------------------------------------------------------------------------
constexpr int GEO_T = 1024;

template <int G>
__device__ __forceinline__ float group_sum(float v) {   // sum over aligned groups of G lanes, in every lane of the group
  if (G >= 2) v += dpp_f32<0xB1, 0xF>(v, 0.f);           // lane ^ 1
  if (G >= 4) v += dpp_f32<0x4E, 0xF>(v, 0.f);           // lane ^ 2
  if (G >= 8) v += dpp_f32<0x141, 0xF>(v, 0.f);          // row_half_mirror
  if (G >= 16) v += dpp_f32<0x140, 0xF>(v, 0.f);         // row_mirror
  if (G >= 32) v += __shfl_xor(v, 16, 64);
  if (G >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// ascending bitonic sort of W keys held in registers (fully unrolled: indices are compile-time constants)
template <int W>
__device__ __forceinline__ void geo_sort_regs(int (&v)[W]) {
#pragma unroll
  for (int k2 = 2; k2 <= W; k2 <<= 1) {
#pragma unroll
    for (int j = k2 >> 1; j > 0; j >>= 1) {
#pragma unroll
      for (int i = 0; i < W; ++i) {
        const int l = i ^ j;
        if (l > i) {
          const bool up = (i & k2) == 0;
          const int lo = min(v[i], v[l]), hi = max(v[i], v[l]);
          v[i] = up ? lo : hi;
          v[l] = up ? hi : lo;
        }
      }
    }
  }
}

// d |<normalize(q - p), n>| * dk / d q for the pair (centre p with normal n and coefficient dk, neighbour q): what the
// centre subtracts from its own gradient and q adds to its
// The pair kernel is bound by VALU issue on the one CU that holds an instance (~100 instructions per pair term, a third
// of them the IEEE sqrt / division sequences): v_sqrt_f32 and v_rcp_f32 (1 ulp each) take 8 us off 51 (250 instances).
// Two ulp per pair term is far inside the parity bars (values rtol 2e-5, gradients 1e-4; the summation order already
// differs from torch's); -DGEOA3_GEO_IEEE restores the correctly rounded forms.
#ifdef GEOA3_GEO_IEEE
#define GEO_SQRT(x) sqrtf(x)
#define GEO_RCP(x) (1.0f / (x))
#else
#define GEO_SQRT(x) __builtin_amdgcn_sqrtf(x)
#define GEO_RCP(x) __builtin_amdgcn_rcpf(x)
#endif
// (The multiply-adds are spelled out and contraction is OFF inside: the overflowed-row path forms a row's terms at two
// call sites -- the pair lane in phase 1, the owner in phase 2 -- whose sums must agree bit for bit whichever of them
// converts a given term; left to -ffp-contract=fast each inlined copy is fused as its surroundings suggest.)
__device__ __forceinline__ void geo_pair_grad(float px, float py, float pz, float nx, float ny, float nz, float dk, float qx,
                                              float qy, float qz, float& dvx, float& dvy, float& dvz, float& t_out) {
#pragma clang fp contract(off)
  const float vx = qx - px, vy = qy - py, vz = qz - pz;
  const float r = GEO_SQRT(__builtin_fmaf(vz, vz, __builtin_fmaf(vy, vy, vx * vx)));
  const float inv = GEO_RCP(fmaxf(r, NORM_EPS));
  const float ux = vx * inv, uy = vy * inv, uz = vz * inv;
  const float t = __builtin_fmaf(uz, nz, __builtin_fmaf(uy, ny, ux * nx));
  const float sg = t > 0.f ? 1.f : (t < 0.f ? -1.f : 0.f);
  const float c = dk * sg * inv;
  if (r >= NORM_EPS) {  // d(v/|v|)/dv = (I - u u^T)/|v|
    dvx = c * __builtin_fmaf(-t, ux, nx);
    dvy = c * __builtin_fmaf(-t, uy, ny);
    dvz = c * __builtin_fmaf(-t, uz, nz);
  } else {              // clamp active: v/eps, the norm path carries no gradient
    dvx = c * nx;
    dvy = c * ny;
    dvz = c * nz;
  }
  t_out = t;
}

template <int G>
__global__ __launch_bounds__(GEO_T) void geo_fused_kernel(geoa3_geo_args A, int C, int R) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = A.N, k = A.k, k1 = k + 1, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Nr = A.Nr > 0 ? A.Nr : N;
  // a random access costs one 16-byte LDS read per record instead of three or four 4-byte ones (bank conflicts are paid
  // per instruction): (x, y, z, coefficient) and (normal, -) of every adversarial point
  float4* s_p = reinterpret_cast<float4*>(sm);           // [N] point + curvature coefficient dk (phase 1: kappa_ori[nn])
  float4* s_n = s_p + N;                                  // [N] normal of the nearest clean point
  float* s_gx = sm + 8 * N;                               // own curvature term of every centre
  float* s_gy = sm + 9 * N;
  float* s_gz = sm + 10 * N;
  int* s_cnt = reinterpret_cast<int*>(sm + 11 * N);       // row lengths
  float* s_red = sm + 12 * N;                             // [16 * 5 + 4]
  uint16_t* s_rows = reinterpret_cast<uint16_t*>(s_red + 16 * 5 + 4);  // [N][C + 1] source ids, fill order
  const int stride = C + 1;
  // Rows longer than C (a dense cluster's clean points all nearest to ONE adversarial point; hubs of the K-NN graph): every
  // term of such a destination goes through fixed-point sums in a small hash pool -- the appends beyond slot C - 1 add theirs
  // in phase 1, the owner adds the C sources its row does hold: order-free, so still bit-reproducible.  (Round 4: the owner
  // walked the whole table, N k entries by one thread: 27x the kernel's time on clustered clouds.)
  GeoPool pool;
  pool.cap = GEO_POOL_CAP;
  pool.acc = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(s_rows + (size_t)R * stride) + 7) & ~(uintptr_t)7);
  pool.key = reinterpret_cast<int*>(pool.acc + GEO_POOL_CAP * 6);
  // Small batches: an instance is split over gridDim.y workgroups by OWNER range [r0, r0 + R) -- every workgroup stages
  // the cloud and forms every centre's coefficient (the pulls need the sources'), but keeps rows, sorts and receives
  // only for its own points: the LDS-bound parts (appends, receives) shrink with the range.  Same bits for any split.
  const int r0 = blockIdx.y * R;
  const float* adv = A.adv + (size_t)b * 3 * N;
  const float* ori = A.ori + (size_t)b * 3 * Nr;
  const size_t bN = (size_t)b * N, bNr = (size_t)b * Nr;
  const bool do_curv = (A.w_curv != 0.f || A.dkappa != nullptr) && A.knn_adv != nullptr;
  const bool do_cd = A.dis_type == 1, do_l2 = A.dis_type == 2;
  const bool two_side = do_cd && !A.single_side && A.d_oa != nullptr;
  const bool do_hd = A.w_hd != 0.f && A.d_ao != nullptr;
  const bool want_grad = A.grad != nullptr;
  const bool first = blockIdx.y == 0;       // writes the loss values and kappa_adv
  constexpr int CPP = GEO_T / G;    // centres per pass
  const int32_t* tab = do_curv ? A.knn_adv + bN * (size_t)k1 : nullptr;

  // ---- stage.  Every load that does not depend on another is issued here (one round trip), the gathers through the
  // nearest-point index right behind them (a second one).
  const int m = tid % G, cl = tid / G;
  // this lane's column of the neighbour table for ALL passes (N / CPP <= G of them): every load in flight before the
  // first pass (one pass ahead left most of each load's latency exposed: 16 passes x ~1 us)
  int q_all[G];
#pragma unroll
  for (int p = 0; p < G; ++p) {
    const int c = p * CPP + cl;
    q_all[p] = (do_curv && c < N && m < k) ? tab[(size_t)c * k1 + 1 + m] : 0;
  }
  const bool me_valid = tid < N;
  const int me = me_valid ? tid : N - 1;
  const float px = adv[me], py = adv[N + me], pz = adv[2 * N + me];
  const int nn = A.i_ao ? A.i_ao[bN + me] : 0;
  const float d_me = A.d_ao ? A.d_ao[bN + me] : 0.f;
  float sum_oa = 0.f;
  int oa_first = 0;
  if (two_side) {
    for (int i = tid; i < Nr; i += GEO_T) sum_oa += A.d_oa[bNr + i];
    if (tid < Nr) oa_first = A.i_oa[bNr + tid];
  }
  const float dkap_me = (do_curv && A.dkappa) ? A.dkappa[bN + me] : 0.f;
  float ox = 0.f, oy = 0.f, oz = 0.f;            // the nearest clean point (own Chamfer / Hausdorff term)
  if (do_cd || do_hd) {
    ox = ori[nn];
    oy = ori[Nr + nn];
    oz = ori[2 * Nr + nn];
  }
  float lx = 0.f, ly = 0.f, lz = 0.f;            // the clean point of the same index (L2 term)
  if (do_l2) {
    lx = ori[me];
    ly = ori[Nr + me];
    lz = ori[2 * Nr + me];
  }
  for (int i = tid; i < GEO_POOL_CAP * 6; i += GEO_T) pool.acc[i] = 0ull;
  for (int i = tid; i < GEO_POOL_CAP; i += GEO_T) pool.key[i] = -1;
  float xmax = fabsf(dkap_me);
  if (A.dkappa) {                                  // (dkappa mode only: the block's largest |dkappa|)
    xmax = wave_max(me_valid ? xmax : 0.f);
    if (lane == 0) s_red[wave] = xmax;
  }
  if (me_valid) {
    s_cnt[me] = 0;
    float kori = 0.f;
    float4 nv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (do_curv) {
      const float* Nm = A.normal_ori + (size_t)b * 3 * Nr;
      nv = make_float4(Nm[nn], Nm[Nr + nn], Nm[2 * Nr + nn], 0.f);
      kori = A.kappa_ori ? A.kappa_ori[bNr + nn] : 0.f;
    }
    s_p[me] = make_float4(px, py, pz, kori);     // (phase 1 turns .w into the coefficient)
    s_n[me] = nv;
  }
  __syncthreads();
  if (A.dkappa) {
    xmax = 0.f;
#pragma unroll
    for (int w = 0; w < GEO_T / 64; ++w) xmax = fmaxf(xmax, s_red[w]);
    __syncthreads();                               // (s_red is reused by the loss reduction below)
  }
  const GeoFix FX = geo_fix_make(geo_coef_bound(A, N, Nr, xmax));
  auto ovf_add = [&](int q, float x, float y, float z) {   // a term of an overflowed row
    const float mm = fmaxf(fmaxf(fabsf(x), fabsf(y)), fabsf(z));
    if (!(mm <= FX.lim_c)) {                               // out of the fixed-point range, or NaN: the owner writes NaN
      atomicOr(&s_cnt[q], (int)0x40000000);
      return;
    }
    const int sl = geo_pool_find(pool, q, true);
    if (sl < 0) return;   // pool FULL: the workgroup has more overflowed rows than slots -- a count every owner sees below,
                          // and then every overflowed row is summed by its owner's walk over the tables instead
    const bool fine = mm <= FX.lim_f;
    const float mul = fine ? FX.to_f : FX.to_c;
    const int o = sl * 6 + (fine ? 0 : 3);
    atomicAdd(&pool.acc[o + 0], geo_fix_conv(x, mul));
    atomicAdd(&pool.acc[o + 1], geo_fix_conv(y, mul));
    atomicAdd(&pool.acc[o + 2], geo_fix_conv(z, mul));
  };

  // ---- phase 1: pairs
  float sum_e2 = 0.f;
  if (do_curv) {
    const float invN = 1.0f / (float)N;
#pragma unroll
    for (int p = 0; p < G; ++p) {
      const int c0 = p * CPP;
      if (c0 >= N) break;
      const int c = c0 + cl;
      const bool cvalid = c < N, active = cvalid && m < k;
      const int cc = cvalid ? c : N - 1;
      const int q = active ? q_all[p] : cc;
      const float4 cp = s_p[cc], nv = s_n[cc], qp = s_p[q];
      int slot = -1;
      if (active && want_grad && (unsigned)(q - r0) < (unsigned)R) {   // the reverse list of q: source c
        slot = atomicAdd(&s_cnt[q], 1) & 0x3fffffff;
        if (slot < C) s_rows[(q - r0) * stride + slot] = (uint16_t)c;
      }
      // kappa term, as kappa_point()
      const float vx = qp.x - cp.x, vy = qp.y - cp.y, vz = qp.z - cp.z;
      const float r = GEO_SQRT(vx * vx + vy * vy + vz * vz);
      const float inv = GEO_RCP(fmaxf(r, NORM_EPS));
      const float t = (vx * inv) * nv.x + (vy * inv) * nv.y + (vz * inv) * nv.z;
      const float kap = group_sum<G>(active ? fabsf(t) : 0.f) / (float)k;
      const float e = kap - cp.w;          // (.w still holds kappa_ori[nn]: the group reads it before its lane 0 writes)
      const float dkap = A.dkappa ? A.dkappa[bN + cc] : 0.f;
      const float dk = (A.dkappa ? dkap : A.w_curv * invN * 2.0f * e) / (float)k;
      float dvx, dvy, dvz, t2;
      geo_pair_grad(cp.x, cp.y, cp.z, nv.x, nv.y, nv.z, dk, qp.x, qp.y, qp.z, dvx, dvy, dvz, t2);
      const float sx = group_sum<G>(active ? dvx : 0.f), sy = group_sum<G>(active ? dvy : 0.f),
                  sz = group_sum<G>(active ? dvz : 0.f);
      if (slot >= C) ovf_add(q, dvx, dvy, dvz);     // the row of q is full: this pair's pull, in fixed point
      if (m == 0 && cvalid) {
        sum_e2 += e * e;
        if (A.kappa_adv && first) A.kappa_adv[bN + c] = kap;
        s_p[c].w = dk;
        s_gx[c] = -sx;
        s_gy[c] = -sy;
        s_gz[c] = -sz;
      }
    }
  }
  (void)dkap_me;
  if (two_side && want_grad)       // the clean points: source id N + j in the row of their nearest adversarial point
    for (int j = tid; j < Nr; j += GEO_T) {
      const int q = j == tid ? oa_first : A.i_oa[bNr + j];
      if ((unsigned)(q - r0) >= (unsigned)R) continue;
      const int slot = atomicAdd(&s_cnt[q], 1) & 0x3fffffff;
      if (slot < C) s_rows[(q - r0) * stride + slot] = (uint16_t)(N + j);
      else {
        const float crj = (Nr != N) ? A.w_dis * (1.0f / (float)Nr) * 2.0f : A.w_dis * (1.0f / (float)N) * 2.0f;
        const float4 qp = s_p[q];
        ovf_add(q, crj * (qp.x - ori[j]), crj * (qp.y - ori[Nr + j]), crj * (qp.z - ori[2 * Nr + j]));
      }
    }

  // ---- loss values and the Hausdorff arg-max
  float sum_ao = 0.f;
  MaxIdx hd{-__builtin_inff(), 0x7fffffff};
  if (me_valid) {
    if (do_cd) sum_ao = d_me;
    if (do_l2) {
      const float dx = px - lx, dy = py - ly, dz = pz - lz;
      sum_ao = dx * dx + dy * dy + dz * dz;
    }
    if (do_hd) hd = MaxIdx{d_me, me};
  }
  sum_ao = wave_sum(sum_ao);
  sum_oa = wave_sum(sum_oa);
  sum_e2 = wave_sum(sum_e2);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MaxIdx other{__shfl_xor(hd.v, o, 64), __shfl_xor(hd.i, o, 64)};
    hd = better(hd, other);
  }
  if (lane == 0) {
    s_red[wave * 5 + 0] = sum_ao;
    s_red[wave * 5 + 1] = sum_oa;
    s_red[wave * 5 + 2] = sum_e2;
    s_red[wave * 5 + 3] = hd.v;
    s_red[wave * 5 + 4] = __int_as_float(hd.i);
  }
  __syncthreads();
  {   // every thread forms the arg-max itself (sixteen broadcast reads); thread 0 writes the values
    float a = 0.f, o = 0.f, e2 = 0.f;
    MaxIdx h{-__builtin_inff(), 0x7fffffff};
#pragma unroll
    for (int w = 0; w < GEO_T / 64; ++w) {
      a += s_red[w * 5 + 0];
      o += s_red[w * 5 + 1];
      e2 += s_red[w * 5 + 2];
      h = better(h, MaxIdx{s_red[w * 5 + 3], __float_as_int(s_red[w * 5 + 4])});
    }
    hd = h;
    if (tid == 0 && first) {
      const float invN = 1.0f / (float)N;
      float dis = 0.f;
      if (do_cd) dis = a * invN + (two_side ? o * (1.0f / (float)Nr) : 0.f);
      if (do_l2) dis = a;
      const float hdv = do_hd ? h.v : 0.f;
      const float curv = do_curv ? e2 * invN : 0.f;
      float con = 0.f;
      if (A.dis_type != 0) con = A.w_dis * dis;
      if (do_hd) con = con + A.w_hd * hdv;
      if (do_curv) con = con + A.w_curv * curv;
      if (A.dis_loss) A.dis_loss[b] = dis;
      if (A.hd_loss) A.hd_loss[b] = hdv;
      if (A.curv_loss) A.curv_loss[b] = curv;
      if (A.constrain) A.constrain[b] = con;
    }
  }
  if (!want_grad) return;
  const int hd_arg = hd.i;
  const bool mine = me_valid && (unsigned)(me - r0) < (unsigned)R;     // this thread owns a point of the range

  // ---- phase 2: every owner sums its own terms, then what it receives in ascending source order
  const float invN = 1.0f / (float)N;
  const float c_cd = A.w_dis * invN * 2.0f;
  const float cr = (Nr != N) ? A.w_dis * (1.0f / (float)Nr) * 2.0f : c_cd;
  float gx = 0.f, gy = 0.f, gz = 0.f;
  if (mine) {
    if (do_cd || do_hd) {
      const float dx = px - ox, dy = py - oy, dz = pz - oz;
      float c = do_cd ? c_cd : 0.f;
      if (do_hd && me == hd_arg) c += A.w_hd * 2.0f;
      gx += c * dx;
      gy += c * dy;
      gz += c * dz;
    }
    if (do_l2) {
      const float c = A.w_dis * 2.0f;
      gx += c * (px - lx);
      gy += c * (py - ly);
      gz += c * (pz - lz);
    }
    if (do_curv) {
      gx += s_gx[me];
      gy += s_gy[me];
      gz += s_gz[me];
    }
  }
  auto receive_rec = [&](int src, float4 sp, float4 sn) {      // the same with the records already read
    if (src < N) {
      float dvx, dvy, dvz, t2;
      geo_pair_grad(sp.x, sp.y, sp.z, sn.x, sn.y, sn.z, sp.w, px, py, pz, dvx, dvy, dvz, t2);
      gx += dvx;
      gy += dvy;
      gz += dvz;
    } else {
      const int jj = src - N;
      gx += cr * (px - ori[jj]);
      gy += cr * (py - ori[Nr + jj]);
      gz += cr * (pz - ori[2 * Nr + jj]);
    }
  };
  auto receive = [&](int src) {
    if (src < N) {              // point `src` lists this point as one of its neighbours
      const float4 sp = s_p[src], sn = s_n[src];
      float dvx, dvy, dvz, t2;
      geo_pair_grad(sp.x, sp.y, sp.z, sn.x, sn.y, sn.z, sp.w, px, py, pz, dvx, dvy, dvz, t2);
      gx += dvx;
      gy += dvy;
      gz += dvz;
    } else {                    // clean point `src - N` has this point as its nearest adversarial point
      const int jj = src - N;
      gx += cr * (px - ori[jj]);
      gy += cr * (py - ori[Nr + jj]);
      gz += cr * (pz - ori[2 * Nr + jj]);
    }
  };
  const int ncnt_raw = mine ? s_cnt[me] : 0;
  const int ncnt = ncnt_raw & 0x3fffffff;
  const int n = ncnt <= C ? ncnt : 0;
  // how many rows of this workgroup overflowed: a function of the tables alone (not of arrival order).  More than the pool
  // has slots (hubs of the K-NN graph, many clean points sharing few adversarial neighbours): the pool's contents are
  // incomplete for SOME rows -- which ones is arrival order -- so NO row uses it: every overflowed row's owner walks the
  // neighbour table and the clean cloud's nearest-point table for its sources, ascending -- the sums the ordered path
  // forms, slow (N k + Nr reads per row) and exact.  (Round 5 wrote NaN here.)
  const int n_ovf = __syncthreads_count(mine && ncnt > C);
  const bool pool_short = n_ovf > GEO_POOL_CAP;
  uint16_t* L = s_rows + (mine ? me - r0 : 0) * stride;
  int nmax = n;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o, 64));
  // Rows are sorted in registers as runs of at most 32 keys, written back in place (the row is this thread's); a row of
  // 33..64 keys is two runs merged while it is consumed (a 64-key network needs more than the 128 registers a lane has
  // at 1024 threads, and in-degree + clean points exceeds 32 for some point of most instances).
  auto sort_run = [&](auto wc, int base) __attribute__((always_inline)) {
    constexpr int W = decltype(wc)::value;
    int key[W];
#pragma unroll
    for (int j = 0; j < W; ++j) key[j] = base + j < n ? (int)L[base + j] : 0x7fffffff;
    geo_sort_regs<W>(key);
#pragma unroll
    for (int j = 0; j < W; ++j)
      if (base + j < n) L[base + j] = (uint16_t)key[j];
  };
  if (nmax > 1 && nmax <= 16) {
    sort_run(std::integral_constant<int, 16>{}, 0);
  } else if (nmax > 16 && nmax <= 64) {
#pragma unroll 1
    for (int base = 0; base < nmax; base += 32) sort_run(std::integral_constant<int, 32>{}, base);   // (one copy of the network)
  } else if (nmax > 64) {      // selection sort through LDS (rows this long need k > 24)
    for (int e = 0; e + 1 < n; ++e) {
      int bj = e, bv = L[e];
      for (int j = e + 1; j < n; ++j) {
        const int kk = L[j];
        if (kk < bv) {
          bv = kk;
          bj = j;
        }
      }
      L[bj] = L[e];
      L[e] = (uint16_t)bv;
    }
  }
  // The row is one sorted run [0, n), or -- 33..64 keys -- two: [0, 32) and [32, n).  Each run holds its neighbour
  // sources first, then its clean points (ids >= N).  The clean points' pulls need coordinates from memory: they are taken
  // after the neighbour pulls, four loads in flight, so that no wavefront waits for one lane's global load inside the
  // main loop (with them inside it, the loop ran at ~1 us per iteration).
  const bool two_runs = nmax > 32 && nmax <= 64;
  const int e0 = two_runs ? (n < 32 ? n : 32) : n;       // end of run 0
  int ka = e0, kb = n;                                    // ends of the runs' neighbour parts
  while (ka > 0 && (int)L[ka - 1] >= N) --ka;
  while (kb > e0 && (int)L[kb - 1] >= N) --kb;
  const int nk = ka + (kb - e0);
  int nkmax = nk;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) nkmax = max(nkmax, __shfl_xor(nkmax, o, 64));
  if (!two_runs) {
    // (rolled: unrolled, the kernel is 19 K instructions); the next source's records are read while this one's term
    // is formed
    int src = nk > 0 ? (int)L[0] : 0;
    float4 sp = s_p[src], sn = s_n[src];
    for (int j = 0; j < nkmax; ++j) {
      const int src1 = j + 1 < nk ? (int)L[j + 1] : 0;
      const float4 sp1 = s_p[src1], sn1 = s_n[src1];
      if (j < nk) receive_rec(src, sp, sn);
      src = src1;
      sp = sp1;
      sn = sn1;
    }
  } else {                             // merged
    int i0 = 0, i1 = e0;
    for (int j = 0; j < nkmax; ++j)
      if (j < nk) {
        const int a0 = i0 < ka ? (int)L[i0] : 0x7fffffff, a1 = i1 < kb ? (int)L[i1] : 0x7fffffff;
        const bool first_run = a0 <= a1;
        receive(first_run ? a0 : a1);
        i0 += first_run ? 1 : 0;
        i1 += first_run ? 0 : 1;
      }
  }
  if (__builtin_amdgcn_ballot_w64(n > nk) != 0) {     // clean points, ascending (merged the same way)
    int i0 = ka, i1 = kb;
    for (int left = n - nk; __builtin_amdgcn_ballot_w64(left > 0) != 0; left -= 4) {
      int id[4];
      float v[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        id[u] = -1;
        if (u < left) {
          const int a0 = i0 < e0 ? (int)L[i0] : 0x7fffffff, a1 = i1 < n ? (int)L[i1] : 0x7fffffff;
          const bool first_run = a0 <= a1;
          id[u] = (first_run ? a0 : a1) - N;
          i0 += first_run ? 1 : 0;
          i1 += first_run ? 0 : 1;
          v[u][0] = ori[id[u]];
          v[u][1] = ori[Nr + id[u]];
          v[u][2] = ori[2 * Nr + id[u]];
        }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (id[u] >= 0) {
          gx += cr * (px - v[u][0]);
          gy += cr * (py - v[u][1]);
          gz += cr * (pz - v[u][2]);
        }
    }
  }
  if (ncnt > C && pool_short) {
    if (do_curv)
      for (int c = 0; c < N; ++c) {                 // sources ascending: point c lists this point among its k neighbours
        const int32_t* row = tab + (size_t)c * k1 + 1;
        bool hit = false;
        for (int mm = 0; mm < k; ++mm) hit = hit || row[mm] == me;
        if (hit) receive(c);
      }
    if (two_side)
      for (int j = 0; j < Nr; ++j)                  // clean points whose nearest adversarial point this is, ascending
        if (A.i_oa[bNr + j] == me) receive(N + j);
  } else if (ncnt > C) {
    // overflowed row: the C sources the row holds (WHICH ones is the appends' arrival order) are converted by the owner and
    // added to what the later arrivals put into the pool in phase 1 -- integer sums, so the split does not matter
    long long fxs = 0, fys = 0, fzs = 0, cxs = 0, cys = 0, czs = 0;
    bool bad = (ncnt_raw & 0x40000000) != 0;
    for (int e = 0; e < C; ++e) {
      const int src = (int)s_rows[(me - r0) * stride + e];
      float dvx, dvy, dvz;
      if (src < N) {
        const float4 sp = s_p[src], sn = s_n[src];
        float t2;
        geo_pair_grad(sp.x, sp.y, sp.z, sn.x, sn.y, sn.z, sp.w, px, py, pz, dvx, dvy, dvz, t2);
      } else {
        const int jj = src - N;
        dvx = cr * (px - ori[jj]);
        dvy = cr * (py - ori[Nr + jj]);
        dvz = cr * (pz - ori[2 * Nr + jj]);
      }
      const float mm = fmaxf(fmaxf(fabsf(dvx), fabsf(dvy)), fabsf(dvz));
      if (mm <= FX.lim_f) {
        fxs += (long long)geo_fix_conv(dvx, FX.to_f);
        fys += (long long)geo_fix_conv(dvy, FX.to_f);
        fzs += (long long)geo_fix_conv(dvz, FX.to_f);
      } else if (mm <= FX.lim_c) {
        cxs += (long long)geo_fix_conv(dvx, FX.to_c);
        cys += (long long)geo_fix_conv(dvy, FX.to_c);
        czs += (long long)geo_fix_conv(dvz, FX.to_c);
      } else {
        bad = true;
      }
    }
    const int sl = geo_pool_find(pool, me, false);
    if (sl >= 0) {
      fxs += (long long)pool.acc[sl * 6 + 0];
      fys += (long long)pool.acc[sl * 6 + 1];
      fzs += (long long)pool.acc[sl * 6 + 2];
      cxs += (long long)pool.acc[sl * 6 + 3];
      cys += (long long)pool.acc[sl * 6 + 4];
      czs += (long long)pool.acc[sl * 6 + 5];
    } else {
      bad = true;       // (cannot happen: a row overflows only through appends that reached the pool, and it was not full)
    }
    gx += __ll2float_rn(fxs) * FX.from_f + __ll2float_rn(cxs) * FX.from_c;
    gy += __ll2float_rn(fys) * FX.from_f + __ll2float_rn(cys) * FX.from_c;
    gz += __ll2float_rn(fzs) * FX.from_f + __ll2float_rn(czs) * FX.from_c;
    if (bad) gx = gy = gz = __builtin_nanf("");
  }
  if (mine) {
    float* Gd = A.grad + (size_t)b * 3 * N;
    Gd[me] = gx;
    Gd[N + me] = gy;
    Gd[2 * N + me] = gz;
  }
}

// ------------------------------------------------------------------------------------------
// 1024 < N <= 4096 (BASELINE configs[4]: N = 4096, k = 32): the pair-parallel objective with the gradient accumulated in
// FIXED POINT -- the cloud, its normals, coefficients and reverse-list rows no longer fit one workgroup's LDS together
// (geo_fused_kernel), and the one-workgroup kernel with chunked reverse lists re-read table and normals ~14 times
// (1.94 GB and 0.83 ms per launch at 250 instances).  geo_big_gather_kernel writes one 16-byte record per point
// (normal of the nearest clean point, its kappa), geo_big_kernel<G> does the rest: one workgroup per instance, positions
// (48 KB) and three planes of 64-bit sums (96 KB) in LDS, the table and the records streamed once.
// HBM per launch: the table once (135 MB) + ~25 MB.  Measured: 15 + 120 us (round 4).
// (Measured on the way: owner-range workgroups with reverse-list rows in LDS, four threads per owner -- 0.15 + 1.0 ms: the
//  instance's sixteen workgroups each stage 112 KB of records and scan the whole table, and ranking rows of up to 91
//  sources is VALU-bound on the one CU; rows split four ways overflow on the k-NN graph's hubs, 9 ms.)
// ------------------------------------------------------------------------------------------
constexpr int GB_T = 1024;

// per adversarial point: (normal of its nearest clean point, kappa_ori of it) as one 16-byte record (coalesced for the
// objective kernel, which would otherwise chain two gathers per centre and pass)
__global__ __launch_bounds__(256) void geo_big_gather_kernel(geoa3_geo_args A, float4* __restrict__ ctr) {
  const int N = A.N, Nr = A.Nr > 0 ? A.Nr : N, b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const size_t bN = (size_t)b * N, bNr = (size_t)b * Nr;
  const int nn = A.i_ao[bN + i];
  const float* Nm = A.normal_ori + bNr * 3;
  ctr[bN + i] = make_float4(Nm[nn], Nm[Nr + nn], Nm[2 * Nr + nn], A.kappa_ori ? A.kappa_ori[bNr + nn] : 0.f);
}

template <int G>
__global__ __launch_bounds__(GB_T) void geo_big_kernel(geoa3_geo_args A, const float4* __restrict__ ctr) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int N = A.N, k = A.k, k1 = k + 1, b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Nr = A.Nr > 0 ? A.Nr : N;
  unsigned long long* s_acc = reinterpret_cast<unsigned long long*>(sm);   // [3][N] fixed-point sums: own curvature term + pulls
  float* s_pos = reinterpret_cast<float*>(s_acc + 3 * N);                  // [N][3]
  float* s_red = s_pos + 3 * N;                                            // [16 * 5 + 4]
  GeoPool pool;                                                            // destinations with coarse-scale terms
  pool.cap = GB_POOL_CAP;
  pool.acc = reinterpret_cast<unsigned long long*>((reinterpret_cast<uintptr_t>(s_red + 16 * 5 + 4) + 7) & ~(uintptr_t)7);
  pool.key = reinterpret_cast<int*>(pool.acc + GB_POOL_CAP * 3);
  unsigned* s_bad = reinterpret_cast<unsigned*>(pool.key + GB_POOL_CAP);   // [ceil(N / 32)] sticky "not representable"
  const size_t bN = (size_t)b * N, bNr = (size_t)b * Nr;
  const float* adv = A.adv + bN * 3;
  const float* ori = A.ori + bNr * 3;
  const int32_t* tab = A.knn_adv + bN * (size_t)k1;
  const bool do_cd = A.dis_type == 1, do_l2 = A.dis_type == 2;
  const bool two_side = do_cd && !A.single_side && A.d_oa != nullptr;
  const bool do_hd = A.w_hd != 0.f && A.d_ao != nullptr;
  const bool want_grad = A.grad != nullptr;
  for (int i = tid; i < N; i += GB_T) {
    s_pos[3 * i] = adv[i];
    s_pos[3 * i + 1] = adv[N + i];
    s_pos[3 * i + 2] = adv[2 * N + i];
    s_acc[i] = 0ull;
    s_acc[N + i] = 0ull;
    s_acc[2 * N + i] = 0ull;
  }
  for (int i = tid; i < GB_POOL_CAP * 3; i += GB_T) pool.acc[i] = 0ull;
  for (int i = tid; i < GB_POOL_CAP; i += GB_T) pool.key[i] = -1;
  for (int i = tid; i < (N + 31) / 32; i += GB_T) s_bad[i] = 0u;
  float xmax = 0.f;
  if (A.dkappa) {                                  // (dkappa mode only: the block's largest |dkappa|)
    for (int i = tid; i < N; i += GB_T) xmax = fmaxf(xmax, fabsf(A.dkappa[bN + i]));
    xmax = wave_max(xmax);
    if (lane == 0) s_red[wave] = xmax;
  }
  __syncthreads();
  if (A.dkappa) {
    xmax = 0.f;
#pragma unroll
    for (int w = 0; w < GB_T / 64; ++w) xmax = fmaxf(xmax, s_red[w]);
    __syncthreads();                               // (s_red is reused by the loss reduction below)
  }
  const GeoFix FX = geo_fix_make(geo_coef_bound(A, N, Nr, xmax));
  const float invN = 1.0f / (float)N;
  // ---- pairs: lane = (centre c, neighbour m); kappa_adv[c] and the centre's coefficient by a G-lane butterfly, then the
  // pair's term goes to q's sums and its negative, summed over m, to c's (geo_fused_kernel's phase 1, same expressions)
  float sum_e2 = 0.f;
  {
    constexpr int CPP = GB_T / G, U = 4;
    const int m = tid % G, cl = tid / G;
    const bool lane_on = m < k;
    for (int c0 = 0; c0 < N; c0 += CPP * U) {
      int q[U];
      float4 cv[U];
      float dkp[U];
      float csx[U], csy[U], csz[U];   // the group sums of the U centres' pair terms (the same in every lane of a group)
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * CPP + cl;
        const int cc = c < N ? c : N - 1;
        q[u] = (c < N && lane_on) ? tab[(size_t)c * k1 + 1 + m] : cc;
        cv[u] = ctr[bN + cc];
        dkp[u] = A.dkappa ? A.dkappa[bN + cc] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int c = c0 + u * CPP + cl;
        const bool cvalid = c < N, active = cvalid && lane_on;
        const int cc = cvalid ? c : N - 1;
        const float cx = s_pos[3 * cc], cy = s_pos[3 * cc + 1], cz = s_pos[3 * cc + 2];
        const float qx = s_pos[3 * q[u]], qy = s_pos[3 * q[u] + 1], qz = s_pos[3 * q[u] + 2];
        const float4 nv = cv[u];
        const float vx = qx - cx, vy = qy - cy, vz = qz - cz;
        const float r = GEO_SQRT(vx * vx + vy * vy + vz * vz);
        const float inv = GEO_RCP(fmaxf(r, NORM_EPS));
        const float t = (vx * inv) * nv.x + (vy * inv) * nv.y + (vz * inv) * nv.z;
        const float kap = group_sum<G>(active ? fabsf(t) : 0.f) / (float)k;
        const float e = kap - nv.w;
        const float dk = (A.dkappa ? dkp[u] : A.w_curv * invN * 2.0f * e) / (float)k;
        float dvx, dvy, dvz, t2;
        geo_pair_grad(cx, cy, cz, nv.x, nv.y, nv.z, dk, qx, qy, qz, dvx, dvy, dvz, t2);
        const float sx = group_sum<G>(active ? dvx : 0.f), sy = group_sum<G>(active ? dvy : 0.f),
                    sz = group_sum<G>(active ? dvz : 0.f);
        if (active && want_grad) geo_fix_add<3>(FX, s_acc, N, pool, s_bad, q[u], dvx, dvy, dvz);
        if (m == 0 && cvalid) {
          sum_e2 += e * e;
          if (A.kappa_adv) A.kappa_adv[bN + c] = kap;
        }
        csx[u] = sx;
        csy[u] = sy;
        csz[u] = sz;
      }
      // the centres' own terms: lane m < U of a group takes centre m of the U just done -- one conversion + atomic pass
      // with U lanes per group instead of U passes with one (the kernel is bound by VALU issue; the sums are integers:
      // who adds them does not matter)
      if (want_grad && m < U) {
        const int c = c0 + m * CPP + cl;
        if (c < N) {
          float sx = csx[0], sy = csy[0], sz = csz[0];
#pragma unroll
          for (int u = 1; u < U; ++u) {
            sx = m == u ? csx[u] : sx;
            sy = m == u ? csy[u] : sy;
            sz = m == u ? csz[u] : sz;
          }
          geo_fix_add<3>(FX, s_acc, N, pool, s_bad, c, -sx, -sy, -sz);
        }
      }
    }
  }
  const float c_cd = A.w_dis * invN * 2.0f;
  const float cr = (Nr != N) ? A.w_dis * (1.0f / (float)Nr) * 2.0f : c_cd;
  if (two_side && want_grad)          // clean point j pulls on its nearest adversarial point
    for (int j = tid; j < Nr; j += GB_T) {
      const int q = A.i_oa[bNr + j];
      geo_fix_add<3>(FX, s_acc, N, pool, s_bad, q, cr * (s_pos[3 * q] - ori[j]), cr * (s_pos[3 * q + 1] - ori[Nr + j]),
                     cr * (s_pos[3 * q + 2] - ori[2 * Nr + j]));
    }
  // ---- loss values and the Hausdorff arg-max, fixed order
  float sum_ao = 0.f, sum_oa = 0.f;
  MaxIdx hd{-__builtin_inff(), 0x7fffffff};
  for (int i = tid; i < N; i += GB_T) {
    if (do_cd || do_hd) {
      const float d = A.d_ao[bN + i];
      if (do_cd) sum_ao += d;
      if (do_hd) hd = better(hd, MaxIdx{d, i});
    }
    if (do_l2) {
      const float dx = s_pos[3 * i] - ori[i], dy = s_pos[3 * i + 1] - ori[Nr + i], dz = s_pos[3 * i + 2] - ori[2 * Nr + i];
      sum_ao += dx * dx + dy * dy + dz * dz;
    }
  }
  if (two_side)
    for (int i = tid; i < Nr; i += GB_T) sum_oa += A.d_oa[bNr + i];
  sum_ao = wave_sum(sum_ao);
  sum_oa = wave_sum(sum_oa);
  sum_e2 = wave_sum(sum_e2);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    MaxIdx other{__shfl_xor(hd.v, o, 64), __shfl_xor(hd.i, o, 64)};
    hd = better(hd, other);
  }
  if (lane == 0) {
    s_red[wave * 5 + 0] = sum_ao;
    s_red[wave * 5 + 1] = sum_oa;
    s_red[wave * 5 + 2] = sum_e2;
    s_red[wave * 5 + 3] = hd.v;
    s_red[wave * 5 + 4] = __int_as_float(hd.i);
  }
  __syncthreads();       // (also: every pull is in the sums)
  {
    float a = 0.f, o = 0.f, e2 = 0.f;
    MaxIdx h{-__builtin_inff(), 0x7fffffff};
#pragma unroll
    for (int w = 0; w < GB_T / 64; ++w) {
      a += s_red[w * 5 + 0];
      o += s_red[w * 5 + 1];
      e2 += s_red[w * 5 + 2];
      h = better(h, MaxIdx{s_red[w * 5 + 3], __float_as_int(s_red[w * 5 + 4])});
    }
    hd = h;
    if (tid == 0) {
      float dis = 0.f;
      if (do_cd) dis = a * invN + (two_side ? o * (1.0f / (float)Nr) : 0.f);
      if (do_l2) dis = a;
      const float hdv = do_hd ? h.v : 0.f;
      const float curv = e2 * invN;
      float con = 0.f;
      if (A.dis_type != 0) con = A.w_dis * dis;
      if (do_hd) con = con + A.w_hd * hdv;
      con = con + A.w_curv * curv;
      if (A.dis_loss) A.dis_loss[b] = dis;
      if (A.hd_loss) A.hd_loss[b] = hdv;
      if (A.curv_loss) A.curv_loss[b] = curv;
      if (A.constrain) A.constrain[b] = con;
    }
  }
  const int hd_arg = hd.i;
  if (!want_grad) return;
  // ---- every point: its Chamfer / Hausdorff / L2 terms, then its fixed-point sums (own curvature term + pulls)
  float* Gd = A.grad + bN * 3;
  for (int i = tid; i < N; i += GB_T) {
    const float px = s_pos[3 * i], py = s_pos[3 * i + 1], pz = s_pos[3 * i + 2];
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (do_cd || do_hd) {
      const int nn = A.i_ao[bN + i];
      const float dx = px - ori[nn], dy = py - ori[Nr + nn], dz = pz - ori[2 * Nr + nn];
      float c = do_cd ? c_cd : 0.f;
      if (do_hd && i == hd_arg) c += A.w_hd * 2.0f;
      gx += c * dx;
      gy += c * dy;
      gz += c * dz;
    }
    if (do_l2) {
      const float c = A.w_dis * 2.0f;
      gx += c * (px - ori[i]);
      gy += c * (py - ori[Nr + i]);
      gz += c * (pz - ori[2 * Nr + i]);
    }
    float ax = __ll2float_rn((long long)s_acc[i]) * FX.from_f, ay = __ll2float_rn((long long)s_acc[N + i]) * FX.from_f,
          az = __ll2float_rn((long long)s_acc[2 * N + i]) * FX.from_f;
    const int sl = geo_pool_find(pool, i, false);
    if (sl >= 0) {
      ax += __ll2float_rn((long long)pool.acc[sl * 3 + 0]) * FX.from_c;
      ay += __ll2float_rn((long long)pool.acc[sl * 3 + 1]) * FX.from_c;
      az += __ll2float_rn((long long)pool.acc[sl * 3 + 2]) * FX.from_c;
    }
    gx += ax;
    gy += ay;
    gz += az;
    if ((s_bad[i >> 5] >> (i & 31)) & 1u) gx = gy = gz = __builtin_nanf("");   // a term beyond 2^34 X, a NaN, or a full pool
    Gd[i] = gx;
    Gd[N + i] = gy;
    Gd[2 * N + i] = gz;
  }
}

}  // namespace

extern "C" int geoa3_kappa(const float* pc, const float* normal, const int32_t* knn_idx, const int32_t* nn_idx,
                           int B, int N, int Nn, int k, float* kappa, void* stream) {
  if (!pc || !normal || !knn_idx || !kappa || B <= 0 || N <= 0 || k <= 0 || Nn < 0) return GEOA3_EINVAL;
  if (Nn == 0) Nn = N;
  if (!nn_idx && Nn != N) return GEOA3_EINVAL;
  const size_t lds = (size_t)3 * N * sizeof(float);
  if (N >= 2048 && lds <= 150 * 1024) {   // (below 2048 points the cloud is L2-resident anyway and four waves fill faster)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kappa_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kappa_kernel<true>, dim3((N + 1023) / 1024, B), dim3(1024), lds, geoa3_stream(stream), pc, normal, knn_idx,
                       nn_idx, N, Nn, k, kappa);
  } else {
    hipLaunchKernelGGL(kappa_kernel<false>, dim3((N + 255) / 256, B), dim3(256), 0, geoa3_stream(stream), pc, normal, knn_idx,
                       nn_idx, N, Nn, k, kappa);
  }
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_geo_loss_grad(const geoa3_geo_args* a, void* stream) {
  if (!a || !a->adv || !a->ori || a->B <= 0 || a->N <= 0 || a->Nr < 0) return GEOA3_EINVAL;
  if (a->dis_type == 2 && a->Nr > 0 && a->Nr != a->N) return GEOA3_EINVAL;  // norm_l2_loss needs equal sizes
  if (a->dis_type == 1 && (!a->d_ao || !a->i_ao)) return GEOA3_EINVAL;
  if (a->w_hd != 0.f && (!a->d_ao || !a->i_ao)) return GEOA3_EINVAL;
  if ((a->w_curv != 0.f || a->dkappa) && (!a->knn_adv || !a->normal_ori || !a->i_ao || a->k <= 0))
    return GEOA3_EINVAL;
  if (a->w_curv != 0.f && !a->dkappa && !a->kappa_ori) return GEOA3_EINVAL;
  const bool do_curv = (a->w_curv != 0.f || a->dkappa) && a->knn_adv;
  // (k > 32 with a scratch buffer: the fixed-point kernel below -- geo_fused_kernel<64> carries 272 bytes of scratch per lane)
  if ((a->deterministic || !a->grad) && a->N <= GEO_T && a->k <= 64 && !(a->k > 32 && a->scratch && do_curv)) {
    // the pair-parallel kernel: the cloud, its normals, coefficients and own terms (10 N floats), row lengths, and one row
    // of C source ids per point; rows of 2 (k + clean points per point) + 16 ids (in-degrees of a k-NN graph concentrate
    // around k), at least 32, at most what fits
    const int N = a->N, Nr = a->Nr > 0 ? a->Nr : a->N;
    const bool two_side = a->dis_type == 1 && !a->single_side && a->d_oa != nullptr;
    if (two_side && !a->i_oa) return GEOA3_EINVAL;
    const int per = (do_curv ? a->k : 0) + (two_side ? (Nr + N - 1) / N : 0);
    const size_t fixed = ((size_t)12 * N + 16 * 5 + 4) * sizeof(float);
    const size_t room = 160 * 1024 - 256 - fixed - GEO_POOL_BYTES;
    // owner ranges per instance: enough workgroups for the chip when the batch is small (a function of the batch size only
    // -- and the results do not depend on it)
    int S = 1;
    while (S < 4 && a->B * S * 2 <= 256 && (N + 2 * S - 1) / (2 * S) >= 64 && a->grad) S *= 2;
    const int R = (((N + S - 1) / S + 63) / 64) * 64;
    int C = 2 * per + 16 < 32 ? 32 : 2 * per + 16;
    const int Cfit = (int)(room / ((size_t)R * sizeof(uint16_t))) - 1;
    if (C > Cfit) C = Cfit;
    if (N + Nr < 65535 && C >= 32 && 2 * C >= 3 * per + 16) {
      const size_t lds = fixed + (size_t)R * (C + 1) * sizeof(uint16_t) + GEO_POOL_BYTES;
      int G = 1;
      while (G < a->k && do_curv) G *= 2;
      hipStream_t s = geoa3_stream(stream);
      geoa3_prof_begin(GEOA3_PROF_GEO, s);
#define GEOA3_FUSED_CASE(GG)                                                                                           \
  if (G == GG) {                                                                                                       \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(geo_fused_kernel<GG>),                                     \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                   \
    hipLaunchKernelGGL(geo_fused_kernel<GG>, dim3(a->B, S), dim3(GEO_T), lds, s, *a, C, R);                                  \
  }
      GEOA3_FUSED_CASE(1)
      GEOA3_FUSED_CASE(2)
      GEOA3_FUSED_CASE(4)
      GEOA3_FUSED_CASE(8)
      GEOA3_FUSED_CASE(16)
      GEOA3_FUSED_CASE(32)
      GEOA3_FUSED_CASE(64)
#undef GEOA3_FUSED_CASE
      geoa3_prof_end(GEOA3_PROF_GEO, s);
      GEOA3_CHECK_LAUNCH();
      return GEOA3_OK;
    }
  }
  if ((a->deterministic || !a->grad) && a->scratch && do_curv && (a->N > GEO_T || a->k > 32) && a->N <= 4096 && a->k <= 64) {
    // the pair-parallel kernel with fixed-point sums (see geo_big_kernel)
    const int N = a->N;
    const bool two_side = a->dis_type == 1 && !a->single_side && a->d_oa != nullptr;
    if (two_side && !a->i_oa) return GEOA3_EINVAL;
    {
      int G = 1;
      while (G < a->k) G *= 2;
      if (G < 16) G = 16;
      float4* ctr = reinterpret_cast<float4*>(a->scratch);
      hipStream_t s = geoa3_stream(stream);
      const size_t lds2 = (size_t)36 * N + (16 * 5 + 4) * sizeof(float) + 8 + GB_POOL_CAP * (3 * 8 + 4) + ((size_t)(N + 31) / 32) * 4;
      geoa3_prof_begin(GEOA3_PROF_GEO, s);
      hipLaunchKernelGGL(geo_big_gather_kernel, dim3((N + 255) / 256, a->B), dim3(256), 0, s, *a, ctr);
#define GEOA3_BIG_CASE(GG)                                                                                             \
  if (G == GG) {                                                                                                       \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(geo_big_kernel<GG>),                                       \
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2);                                  \
    hipLaunchKernelGGL(geo_big_kernel<GG>, dim3(a->B), dim3(GB_T), lds2, s, *a, ctr);                                  \
  }
      GEOA3_BIG_CASE(16)
      GEOA3_BIG_CASE(32)
      GEOA3_BIG_CASE(64)
#undef GEOA3_BIG_CASE
      geoa3_prof_end(GEOA3_PROF_GEO, s);
      GEOA3_CHECK_LAUNCH();
      return GEOA3_OK;
    }
  }
  const size_t base = ((size_t)7 * a->N + GEO_WAVES * 5 + 4) * sizeof(float);
  if (base > 160 * 1024) return GEOA3_ENOSUPPORT;  // N <= ~5800 points per instance
  const size_t det_idx = ((size_t)a->N + 1) * sizeof(int);
  // (clouds whose reverse lists do not fit beside the cloud -- N > ~4800 -- take the atomic kernel: same values, free
  // summation order, documented in geoa3_hip.h)
  if (a->deterministic && a->grad && base + det_idx + 2048 * sizeof(int) <= 160 * 1024 - 512) {
    // counts / offsets, then (if they fit) the per-point normals, then the reverse lists of one chunk of sources: all of
    // them in one chunk when they fit (N = 1024, k = 16: 70 KB), at least 2048 entries otherwise
    const size_t Nr = a->Nr > 0 ? a->Nr : a->N;
    const size_t idx = ((size_t)a->N + 1) * sizeof(int), nrm = (size_t)3 * a->N * sizeof(float);
    const size_t cap = 160 * 1024 - 512, all = (size_t)a->N * a->k > Nr ? (size_t)a->N * a->k : Nr;
    const int nrm_in = a->normal_ori && base + idx + nrm + (all < 8192 ? all : 8192) * sizeof(int) <= cap;
    size_t rcap = (cap - base - idx - (nrm_in ? nrm : 0)) / sizeof(int);
    if (rcap > all) rcap = all;
    if (rcap < 1) rcap = 1;
    const size_t lds = base + idx + (nrm_in ? nrm : 0) + rcap * sizeof(int);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(geo_loss_grad_kernel<true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    geoa3_prof_begin(GEOA3_PROF_GEO, geoa3_stream(stream));
    hipLaunchKernelGGL(geo_loss_grad_kernel<true>, dim3(a->B), dim3(GEO_BLOCK), lds, geoa3_stream(stream), *a, (int)rcap,
                       nrm_in);
    geoa3_prof_end(GEOA3_PROF_GEO, geoa3_stream(stream));
    GEOA3_CHECK_LAUNCH();
    return GEOA3_OK;
  }
  const size_t lds = base;
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(geo_loss_grad_kernel<false>),
                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  geoa3_prof_begin(GEOA3_PROF_GEO, geoa3_stream(stream));
  hipLaunchKernelGGL(geo_loss_grad_kernel<false>, dim3(a->B), dim3(GEO_BLOCK), lds, geoa3_stream(stream), *a, 0, 0);
  geoa3_prof_end(GEOA3_PROF_GEO, geoa3_stream(stream));
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
