// Geometry helpers of the dense-cloud attack path, the point-removal defence and the smoothness measurement
// (gfx950).  Reference: Lib/utility.py:91-108 (estimate_normal_via_ori_normal), :116-149 (estimate_perpendicular),
// :175-187 (farthest_points_sample); defense.py:18-45; Measurement/compute_data_smoothness.py:37-67.
// All of them sit on top of the K-NN kernels of geom_nn.hip; the reference runs them as chains of dense torch ops
// (an [n,n] distance matrix per cloud, per-point numpy eig loops, python loops over instances).
#include "common.h"

namespace {

constexpr float INF = __builtin_inff();

// ------------------------------------------------------------------------------------------
// farthest_points_sample (Lib/utility.py:175-187): m-1 rounds of
//     dists = min(dists, |p - p_last|);  next = argmax(dists)        (first maximal index)
// One workgroup of 1024 threads per cloud; every thread keeps PPT points and their running distances in
// registers, so a round is PPT distance updates, a shuffle arg-max per wave and ONE barrier (the per-wave winners
// and their coordinates go through a double-buffered LDS slot; every thread re-reduces the 16 slots itself).
// The distance is sqrt(fl(fl(dx^2+dy^2)+dz^2)) as torch.norm(dim=1) evaluates it; sqrt is correctly rounded.
// ------------------------------------------------------------------------------------------
constexpr int FPS_T = 1024;
constexpr int FPS_W = FPS_T / GEOA3_WAVE;

template <int PPT>
__global__ __launch_bounds__(FPS_T) void fps_sample_kernel(const float* __restrict__ pc, int N, int m,
                                                           const int32_t* __restrict__ start,
                                                           int32_t* __restrict__ idx_out,
                                                           float* __restrict__ pts_out) {
  __shared__ float s_val[2][FPS_W], s_x[2][FPS_W], s_y[2][FPS_W], s_z[2][FPS_W];
  __shared__ int s_idx[2][FPS_W];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6;
  const float* P = pc + (size_t)b * 3 * N;
  float px[PPT], py[PPT], pz[PPT], dist[PPT];
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int i = tid + p * FPS_T;
    const bool ok = i < N;
    px[p] = ok ? P[i] : 0.f;
    py[p] = ok ? P[N + i] : 0.f;
    pz[p] = ok ? P[2 * N + i] : 0.f;
    dist[p] = ok ? INF : -1.f;   // padding never wins: real distances are >= 0
  }
  int cur = start[b];
  cur = cur < 0 ? 0 : (cur >= N ? N - 1 : cur);
  float cx = P[cur], cy = P[N + cur], cz = P[2 * N + cur];
  for (int r = 0; r < m; ++r) {
    if (tid == 0) {
      idx_out[(size_t)b * m + r] = cur;
      if (pts_out) {
        pts_out[((size_t)b * 3 + 0) * m + r] = cx;
        pts_out[((size_t)b * 3 + 1) * m + r] = cy;
        pts_out[((size_t)b * 3 + 2) * m + r] = cz;
      }
    }
    if (r == m - 1) break;
    float bv = -2.f, bx = 0.f, by = 0.f, bz = 0.f;
    int bi = 0x7fffffff;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const float d = sqrtf(geoa3_sqdist(px[p], py[p], pz[p], cx, cy, cz));
      dist[p] = fminf(dist[p], d);
      const bool take = dist[p] > bv;   // strict: the lowest index of equal maxima stays (ascending p = ascending i)
      bv = take ? dist[p] : bv;
      bi = take ? tid + p * FPS_T : bi;
      bx = take ? px[p] : bx;
      by = take ? py[p] : by;
      bz = take ? pz[p] : bz;
    }
    const int li = bi;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      const bool take = ov > bv || (ov == bv && oi < bi);
      bv = take ? ov : bv;
      bi = take ? oi : bi;
    }
    const int buf = r & 1;
    if (li == bi) {   // the lane that owns the wave's winner publishes it
      s_val[buf][wave] = bv;
      s_idx[buf][wave] = bi;
      s_x[buf][wave] = bx;
      s_y[buf][wave] = by;
      s_z[buf][wave] = bz;
    }
    __syncthreads();
    float gv = s_val[buf][0];
    int gi = s_idx[buf][0], gs = 0;
#pragma unroll
    for (int w = 1; w < FPS_W; ++w) {
      const float v = s_val[buf][w];
      const int i = s_idx[buf][w];
      const bool take = v > gv || (v == gv && i < gi);
      gv = take ? v : gv;
      gi = take ? i : gi;
      gs = take ? w : gs;
    }
    cur = gi;
    cx = s_x[buf][gs];
    cy = s_y[buf][gs];
    cz = s_z[buf][gs];
  }
}

// Faster form for clouds that fit LDS (N <= 8192): the arg-max is a max over 64-bit keys (distance bits, ~index) --
// distances are >= 0, so their bit patterns order like unsigned integers, and the complemented index makes the LOWEST
// index win a tie -- reduced inside a wavefront with DPP row operations (quad_perm, row_half_mirror, row_mirror,
// row_bcast15/31: common.h), across wavefronts through one LDS slot per wave, and the winner's coordinates are an
// LDS read of the staged cloud.
template <int FT, int PPT>   // FT threads, PPT points per thread: fewer, fatter waves make the per-round barrier cheaper
__global__ __launch_bounds__(FT) void fps_sample_lds_kernel(const float* __restrict__ pc, int N, int m,
                                                               const int32_t* __restrict__ start,
                                                               int32_t* __restrict__ idx_out,
                                                               float* __restrict__ pts_out) {
  extern __shared__ __attribute__((aligned(16))) float s_cloud[];   // [3][N]
  __shared__ unsigned long long s_key[2][FT / 64];
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* P = pc + (size_t)b * 3 * N;
  float px[PPT], py[PPT], pz[PPT], dist[PPT];
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int i = tid + p * FT;
    const bool ok = i < N;
    px[p] = ok ? P[i] : 0.f;
    py[p] = ok ? P[N + i] : 0.f;
    pz[p] = ok ? P[2 * N + i] : 0.f;
    dist[p] = ok ? INF : -1.f;
    if (ok) {
      s_cloud[i] = px[p];
      s_cloud[N + i] = py[p];
      s_cloud[2 * N + i] = pz[p];
    }
  }
  int cur = start[b];
  cur = cur < 0 ? 0 : (cur >= N ? N - 1 : cur);
  __syncthreads();
  for (int r = 0; r < m; ++r) {
    const float cx = s_cloud[cur], cy = s_cloud[N + cur], cz = s_cloud[2 * N + cur];
    if (tid == 0) {
      idx_out[(size_t)b * m + r] = cur;
      if (pts_out) {
        pts_out[((size_t)b * 3 + 0) * m + r] = cx;
        pts_out[((size_t)b * 3 + 1) * m + r] = cy;
        pts_out[((size_t)b * 3 + 2) * m + r] = cz;
      }
    }
    if (r == m - 1) break;
    unsigned long long key = 0ull;   // padding lanes: below every real key (real distances are >= 0 -> bits >= 0)
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const int i = tid + p * FT;
      const float d = sqrtf(geoa3_sqdist(px[p], py[p], pz[p], cx, cy, cz));
      dist[p] = fminf(dist[p], d);
      if (i < N) {
        const unsigned long long k = ((unsigned long long)__float_as_uint(dist[p]) << 32) | (0xFFFFFFFFu - (unsigned)i);
        key = k > key ? k : key;
      }
    }
    key = wave_max_u64(key);
    const int buf = r & 1;
    if (lane == 0) s_key[buf][wave] = key;
    __syncthreads();
    unsigned long long k2 = lane < FT / 64 ? s_key[buf][lane] : 0ull;
    k2 = row16_max_u64(k2);           // at most 16 waves: one DPP row
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)k2);
    cur = (int)(0xFFFFFFFFu - lo);
  }
}

// ------------------------------------------------------------------------------------------
// estimate_normal_via_ori_normal (Lib/utility.py:91-108) from the cross K-NN table of adv against ori.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_normal_kernel(const float* __restrict__ knn_d,
                                                         const int32_t* __restrict__ knn_i,
                                                         const float* __restrict__ normal_ori, int Nq, int Nr, int K,
                                                         float* __restrict__ out) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= Nq) return;
  const float* Nm = normal_ori + (size_t)b * 3 * Nr;
  const int32_t* nb = knn_i + ((size_t)b * Nq + i) * K;
  float sx = 0.f, sy = 0.f, sz = 0.f;
  for (int m = 0; m < K; ++m) {
    const int j = nb[m];
    sx += Nm[j];
    sy += Nm[Nr + j];
    sz += Nm[2 * Nr + j];
  }
  const float inv_k = 1.0f / (float)K;
  sx *= inv_k;
  sy *= inv_k;
  sz *= inv_k;
  const float nrm = sqrtf(sx * sx + sy * sy + sz * sz) + 1e-12f;
  float ox = sx / nrm, oy = sy / nrm, oz = sz / nrm;
  if (knn_d[((size_t)b * Nq + i) * K] < 1e-6f) {   // the point did not move: its own original normal
    const int j = nb[0];
    ox = Nm[j];
    oy = Nm[Nr + j];
    oz = Nm[2 * Nr + j];
  }
  float* O = out + (size_t)b * 3 * Nq;
  O[i] = ox;
  O[Nq + i] = oy;
  O[2 * Nq + i] = oz;
}

// ------------------------------------------------------------------------------------------
// Local frames: eigen-decomposition of the covariance (factor 1/(k-1)) of the k nearest neighbours of every
// point (Lib/utility.py:122-133; Measurement/compute_data_smoothness.py:52-61).  The covariance is accumulated in
// fp32 like the reference's bmm; the 3x3 symmetric eigenproblem is solved by cyclic Jacobi in fp64 (6 sweeps).
// Output: eigenvalues ascending [B,3,N]; eigenvectors [B,3(e),3(xyz),N], each with its largest-magnitude
// component positive (the reference's LAPACK sign is implementation-defined).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void jacobi_rot(double& app, double& aqq, double& apq, double& apr, double& aqr,
                                           double& vp0, double& vp1, double& vp2, double& vq0, double& vq1,
                                           double& vq2) {
  if (fabs(apq) < 1e-300) return;
  const double theta = (aqq - app) / (2.0 * apq);
  const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
  const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
  app -= t * apq;
  aqq += t * apq;
  apq = 0.0;
  const double npr = c * apr - s * aqr, nqr = s * apr + c * aqr;
  apr = npr;
  aqr = nqr;
  double a, bb;
  a = c * vp0 - s * vq0; bb = s * vp0 + c * vq0; vp0 = a; vq0 = bb;
  a = c * vp1 - s * vq1; bb = s * vp1 + c * vq1; vp1 = a; vq1 = bb;
  a = c * vp2 - s * vq2; bb = s * vp2 + c * vq2; vp2 = a; vq2 = bb;
}

__global__ __launch_bounds__(256) void local_frame_kernel(const float* __restrict__ pc,
                                                          const int32_t* __restrict__ knn, int N, int K1,
                                                          float* __restrict__ evals, float* __restrict__ evecs) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float* P = pc + (size_t)b * 3 * N;
  const int32_t* nb = knn + ((size_t)b * N + i) * K1;
  const int k = K1 - 1;
  float mx = 0.f, my = 0.f, mz = 0.f;
  for (int m = 1; m <= k; ++m) {
    const int j = nb[m];
    mx += P[j];
    my += P[N + j];
    mz += P[2 * N + j];
  }
  const float inv_k = 1.0f / (float)k;
  mx *= inv_k;
  my *= inv_k;
  mz *= inv_k;
  float cxx = 0.f, cxy = 0.f, cxz = 0.f, cyy = 0.f, cyz = 0.f, czz = 0.f;
  for (int m = 1; m <= k; ++m) {
    const int j = nb[m];
    const float x = P[j] - mx, y = P[N + j] - my, z = P[2 * N + j] - mz;
    cxx += x * x;
    cxy += x * y;
    cxz += x * z;
    cyy += y * y;
    cyz += y * z;
    czz += z * z;
  }
  const float fact = 1.0f / (float)(k - 1);
  double a00 = cxx * fact, a01 = cxy * fact, a02 = cxz * fact, a11 = cyy * fact, a12 = cyz * fact, a22 = czz * fact;
  double v00 = 1, v01 = 0, v02 = 0, v10 = 0, v11 = 1, v12 = 0, v20 = 0, v21 = 0, v22 = 1;  // v{e}{c}: column e
  for (int sweep = 0; sweep < 6; ++sweep) {
    jacobi_rot(a00, a11, a01, a02, a12, v00, v01, v02, v10, v11, v12);
    jacobi_rot(a00, a22, a02, a01, a12, v00, v01, v02, v20, v21, v22);
    jacobi_rot(a11, a22, a12, a01, a02, v10, v11, v12, v20, v21, v22);
  }
  // sort ascending (3-element network), eigenvectors follow
#define GEOA3_SWAP_E(wa, wb, a0, a1, a2, b0, b1, b2) \
  if (wb < wa) {                                     \
    double t_;                                       \
    t_ = wa; wa = wb; wb = t_;                       \
    t_ = a0; a0 = b0; b0 = t_;                       \
    t_ = a1; a1 = b1; b1 = t_;                       \
    t_ = a2; a2 = b2; b2 = t_;                       \
  }
  GEOA3_SWAP_E(a00, a11, v00, v01, v02, v10, v11, v12)
  GEOA3_SWAP_E(a11, a22, v10, v11, v12, v20, v21, v22)
  GEOA3_SWAP_E(a00, a11, v00, v01, v02, v10, v11, v12)
#undef GEOA3_SWAP_E
  const size_t bN = (size_t)b * 3 * N;
  evals[bN + i] = (float)a00;
  evals[bN + N + i] = (float)a11;
  evals[bN + 2 * N + i] = (float)a22;
  float* E = evecs + (size_t)b * 9 * N;
  auto put = [&](int e, double x, double y, double z) {
    const double ax = fabs(x), ay = fabs(y), az = fabs(z);
    const double lead = (ax >= ay && ax >= az) ? x : (ay >= az ? y : z);
    const double sgn = lead < 0.0 ? -1.0 : 1.0;
    E[(size_t)(e * 3 + 0) * N + i] = (float)(sgn * x);
    E[(size_t)(e * 3 + 1) * N + i] = (float)(sgn * y);
    E[(size_t)(e * 3 + 2) * N + i] = (float)(sgn * z);
  };
  put(0, v00, v01, v02);
  put(1, v10, v11, v12);
  put(2, v20, v21, v22);
}

// estimate_perpendicular's tail (Lib/utility.py:146-149): clamp(v1*aux1) + clamp(v2*aux2), v1 = eigenvector of the
// largest eigenvalue, v2 = of the middle one; aux = sigma * randn drawn by the caller.
__global__ __launch_bounds__(256) void perp_jitter_kernel(const float* __restrict__ evecs,
                                                          const float* __restrict__ aux1,
                                                          const float* __restrict__ aux2, int N, float clip,
                                                          float* __restrict__ out) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float* E = evecs + (size_t)b * 9 * N;
  const float a1 = aux1[(size_t)b * N + i], a2 = aux2[(size_t)b * N + i];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float t1 = fminf(fmaxf(E[(size_t)(6 + c) * N + i] * a1, -clip), clip);
    const float t2 = fminf(fmaxf(E[(size_t)(3 + c) * N + i] * a2, -clip), clip);
    out[((size_t)b * 3 + c) * N + i] = t1 + t2;
  }
}

// compute_data_smoothness.py:63-67: s_i = mean_{m=1..k} |<q_m - p_i, n_i>|, n_i = eigenvector of the SMALLEST
// eigenvalue; out[b] = max_i s_i (as float bits through an unsigned atomicMax: s_i >= 0).
__global__ __launch_bounds__(256) void smoothness_kernel(const float* __restrict__ pc, const int32_t* __restrict__ knn,
                                                         const float* __restrict__ evecs, int N, int K1,
                                                         float* __restrict__ per_point, unsigned* __restrict__ out) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  float s = 0.f;
  if (i < N) {
    const float* P = pc + (size_t)b * 3 * N;
    const float* E = evecs + (size_t)b * 9 * N;
    const int32_t* nb = knn + ((size_t)b * N + i) * K1;
    const float nx = E[i], ny = E[(size_t)N + i], nz = E[(size_t)2 * N + i];
    const float px = P[i], py = P[N + i], pz = P[2 * N + i];
    const int k = K1 - 1;
    for (int m = 1; m <= k; ++m) {
      const int j = nb[m];
      const float t = (P[j] - px) * nx + (P[N + j] - py) * ny + (P[2 * N + j] - pz) * nz;
      s += fabsf(t);
    }
    s /= (float)k;
    if (per_point) per_point[(size_t)b * N + i] = s;
  }
  s = wave_max(s);
  if ((threadIdx.x & 63) == 0) atomicMax(out + b, __float_as_uint(s));
}

// ------------------------------------------------------------------------------------------
// Statistical outlier removal (defense.py:26-45).
// ------------------------------------------------------------------------------------------
// dis[b,i] = mean of the K smallest non-self values of sqrt(sum_c (p_j - p_i + 1e-10)^2) (defense.py:27-28; the
// smallest of the K+1 values, normally the point itself at sqrt(3e-20), is dropped like the reference's [:, :, 1:]).
constexpr int SOR_T = 256;
__global__ __launch_bounds__(SOR_T) void sor_stat_kernel(const float* __restrict__ pc, int N, int K,
                                                         float* __restrict__ dis) {
  extern __shared__ __attribute__((aligned(16))) float sor_sm[];
  float* s_best = sor_sm;                      // [K+1][SOR_T] ascending per thread column
  float* s_tile = sor_sm + (size_t)(K + 1) * SOR_T;   // [3][SOR_T]
  const int b = blockIdx.y, tid = threadIdx.x, i = blockIdx.x * SOR_T + tid;
  const float* P = pc + (size_t)b * 3 * N;
  const bool live = i < N;
  const float qx = live ? P[i] : 0.f, qy = live ? P[N + i] : 0.f, qz = live ? P[2 * N + i] : 0.f;
  const int K1 = K + 1;
  for (int m = 0; m < K1; ++m) s_best[m * SOR_T + tid] = INF;
  float worst = INF;
  for (int base = 0; base < N; base += SOR_T) {
    __syncthreads();
    const int j = base + tid;
    s_tile[tid] = j < N ? P[j] : 0.f;
    s_tile[SOR_T + tid] = j < N ? P[N + j] : 0.f;
    s_tile[2 * SOR_T + tid] = j < N ? P[2 * N + j] : 0.f;
    __syncthreads();
    const int cnt = min(SOR_T, N - base);
    for (int t = 0; t < cnt; ++t) {
      float d;
      {
#pragma clang fp contract(off)
        const float dx = (s_tile[t] - qx) + 1e-10f, dy = (s_tile[SOR_T + t] - qy) + 1e-10f,
                    dz = (s_tile[2 * SOR_T + t] - qz) + 1e-10f;
        const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
        const float s = xx + yy;
        d = s + zz;
      }
      if (d < worst) {   // insert into the ascending column
        int m = K1 - 1;
        while (m > 0 && s_best[(m - 1) * SOR_T + tid] > d) {
          s_best[m * SOR_T + tid] = s_best[(m - 1) * SOR_T + tid];
          --m;
        }
        s_best[m * SOR_T + tid] = d;
        worst = s_best[(K1 - 1) * SOR_T + tid];
      }
    }
  }
  if (!live) return;
  float acc = 0.f;
  for (int m = 1; m < K1; ++m) acc += sqrtf(s_best[m * SOR_T + tid]);
  dis[(size_t)b * N + i] = acc / (float)K;
}

// Keep-set selection, one workgroup per cloud, indices compacted in ascending order (defense.py:31-45).
//   mode 0 outliers_fixNum:   keep the N - drop_num smallest dis (equal values: lower index first)
//   mode 1 outliers_variance: keep dis < mean + alpha * std (unbiased std, float64 accumulation)
constexpr int SEL_T = 1024;
__global__ __launch_bounds__(SEL_T) void sor_select_kernel(const float* __restrict__ dis, int N, int mode, int drop_num,
                                                           float alpha, int32_t* __restrict__ idx_out,
                                                           int32_t* __restrict__ count_out,
                                                           float* __restrict__ stats_out) {
  extern __shared__ __attribute__((aligned(16))) float sel_sm[];
  float* s_d = sel_sm;   // [N]
  __shared__ double s_red[SEL_T / GEOA3_WAVE];
  __shared__ double s_mean, s_thr;
  __shared__ int s_wcnt[SEL_T / GEOA3_WAVE];
  __shared__ int s_base;
  const int b = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const float* D = dis + (size_t)b * N;
  for (int i = tid; i < N; i += SEL_T) s_d[i] = D[i];
  if (tid == 0) {
    s_base = 0;
    s_thr = 0.0;
  }
  __syncthreads();
  if (mode == 1) {
    double acc = 0.0;
    for (int i = tid; i < N; i += SEL_T) acc += (double)s_d[i];
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) s_red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < SEL_T / GEOA3_WAVE; ++w) t += s_red[w];
      s_mean = t / (double)N;
    }
    __syncthreads();
    const double mean = s_mean;
    acc = 0.0;
    for (int i = tid; i < N; i += SEL_T) {
      const double e = (double)s_d[i] - mean;
      acc += e * e;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    __syncthreads();
    if (lane == 0) s_red[wave] = acc;
    __syncthreads();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < SEL_T / GEOA3_WAVE; ++w) t += s_red[w];
      const double sd = sqrt(t / (double)(N - 1));
      // the reference forms mean + alpha*std in fp32 (defense.py:33-35)
      float thr;
      {
#pragma clang fp contract(off)
        const float scaled = alpha * (float)sd;
        thr = (float)mean + scaled;
      }
      s_thr = (double)thr;
      if (stats_out) {
        stats_out[(size_t)b * 2] = (float)mean;
        stats_out[(size_t)b * 2 + 1] = (float)sd;
      }
    }
    __syncthreads();
  }
  const float thr = (float)s_thr;
  const int keep_n = N - drop_num;
  for (int base = 0; base < N; base += SEL_T) {
    const int i = base + tid;
    bool keep = false;
    if (i < N) {
      const float di = s_d[i];
      if (mode == 1) {
        keep = di < thr;
      } else {
        int rank = 0;
        for (int j = 0; j < N; ++j) {
          const float dj = s_d[j];
          rank += (dj < di || (dj == di && j < i)) ? 1 : 0;
        }
        keep = rank < keep_n;
      }
    }
    const unsigned long long mask = __ballot(keep);
    if (lane == 0) s_wcnt[wave] = __popcll(mask);
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_wcnt[w];
    if (keep) idx_out[(size_t)b * N + off + __popcll(mask & ((1ull << lane) - 1ull))] = i;
    __syncthreads();
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < SEL_T / GEOA3_WAVE; ++w) t += s_wcnt[w];
      s_base += t;
    }
    __syncthreads();
  }
  const int total = s_base;
  for (int i = total + tid; i < N; i += SEL_T) idx_out[(size_t)b * N + i] = -1;
  if (tid == 0) count_out[b] = total;
}

}  // namespace

extern "C" int geoa3_fps_sample(const float* pc, int B, int N, int m, const int32_t* start, int32_t* idx, float* pts,
                                void* stream) {
  if (!pc || !start || !idx || B <= 0 || N <= 0 || m <= 0) return GEOA3_EINVAL;
  hipStream_t s = geoa3_stream(stream);
#define GEOA3_FPS_LDS_CASE(FT, PPT)                                                                              \
  if (N <= PPT * FT) {                                                                                           \
    const size_t lds = (size_t)3 * N * sizeof(float);                                                            \
    auto kern = fps_sample_lds_kernel<FT, PPT>;                                                                  \
    if (lds > 48 * 1024)                                                                                         \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                       \
    hipLaunchKernelGGL(kern, dim3(B), dim3(FT), lds, s, pc, N, m, start, idx, pts);                              \
    GEOA3_CHECK_LAUNCH();                                                                                        \
    return GEOA3_OK;                                                                                             \
  }
  // 512 threads measured best (1024: 8 % slower, 256: 8-19 %): 0.33 ms for 512 of 1024 points, 1.30 ms for 1024 of 4096
  GEOA3_FPS_LDS_CASE(512, 2) GEOA3_FPS_LDS_CASE(512, 4) GEOA3_FPS_LDS_CASE(512, 8) GEOA3_FPS_LDS_CASE(512, 16)
#undef GEOA3_FPS_LDS_CASE
#define GEOA3_FPS_CASE(PPT)                                                                                      \
  if (N <= PPT * FPS_T) {                                                                                        \
    hipLaunchKernelGGL(fps_sample_kernel<PPT>, dim3(B), dim3(FPS_T), 0, s, pc, N, m, start, idx, pts);           \
    GEOA3_CHECK_LAUNCH();                                                                                        \
    return GEOA3_OK;                                                                                             \
  }
  GEOA3_FPS_CASE(1)
  GEOA3_FPS_CASE(2)
  GEOA3_FPS_CASE(4)
  GEOA3_FPS_CASE(8)
  GEOA3_FPS_CASE(16)
#undef GEOA3_FPS_CASE
  return GEOA3_ENOSUPPORT;   // N <= 16384 points per cloud
}

extern "C" int geoa3_knn_normal(const float* knn_d, const int32_t* knn_i, const float* normal_ori, int B, int Nq, int Nr,
                                int K, float* out, void* stream) {
  if (!knn_d || !knn_i || !normal_ori || !out || B <= 0 || Nq <= 0 || Nr <= 0 || K <= 0) return GEOA3_EINVAL;
  hipLaunchKernelGGL(knn_normal_kernel, dim3((Nq + 255) / 256, B), dim3(256), 0, geoa3_stream(stream), knn_d, knn_i,
                     normal_ori, Nq, Nr, K, out);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_local_frames(const float* pc, const int32_t* knn_idx, int B, int N, int K1, float* evals,
                                  float* evecs, void* stream) {
  if (!pc || !knn_idx || !evals || !evecs || B <= 0 || N <= 0 || K1 < 3) return GEOA3_EINVAL;
  hipLaunchKernelGGL(local_frame_kernel, dim3((N + 255) / 256, B), dim3(256), 0, geoa3_stream(stream), pc, knn_idx, N,
                     K1, evals, evecs);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_perp_jitter(const float* evecs, const float* aux1, const float* aux2, int B, int N, float clip,
                                 float* out, void* stream) {
  if (!evecs || !aux1 || !aux2 || !out || B <= 0 || N <= 0 || !(clip > 0.f)) return GEOA3_EINVAL;
  hipLaunchKernelGGL(perp_jitter_kernel, dim3((N + 255) / 256, B), dim3(256), 0, geoa3_stream(stream), evecs, aux1,
                     aux2, N, clip, out);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_smoothness(const float* pc, const int32_t* knn_idx, const float* evecs, int B, int N, int K1,
                                float* per_point, float* out, void* stream) {
  if (!pc || !knn_idx || !evecs || !out || B <= 0 || N <= 0 || K1 < 2) return GEOA3_EINVAL;
  hipStream_t s = geoa3_stream(stream);
  if (hipMemsetAsync(out, 0, (size_t)B * sizeof(float), s) != hipSuccess) return GEOA3_ELAUNCH;
  hipLaunchKernelGGL(smoothness_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, pc, knn_idx, evecs, N, K1,
                     per_point, reinterpret_cast<unsigned*>(out));
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_sor_statistic(const float* pc, int B, int N, int K, float* dis, void* stream) {
  if (!pc || !dis || B <= 0 || N <= 0 || K <= 0 || K >= N || K > GEOA3_KNN_MAX_K) return GEOA3_EINVAL;
  const size_t lds = ((size_t)(K + 1) * SOR_T + 3 * SOR_T) * sizeof(float);
  if (lds > 64 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sor_stat_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds);
  hipLaunchKernelGGL(sor_stat_kernel, dim3((N + SOR_T - 1) / SOR_T, B), dim3(SOR_T), lds, geoa3_stream(stream), pc, N,
                     K, dis);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_sor_select(const float* dis, int B, int N, int mode, int drop_num, float alpha, int32_t* idx,
                                int32_t* count, float* stats, void* stream) {
  if (!dis || !idx || !count || B <= 0 || N <= 1 || (mode != 0 && mode != 1)) return GEOA3_EINVAL;
  if (mode == 0 && (drop_num < 0 || drop_num > N)) return GEOA3_EINVAL;
  const size_t lds = (size_t)N * sizeof(float);
  if (lds > 150 * 1024) return GEOA3_ENOSUPPORT;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sor_select_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(sor_select_kernel, dim3(B), dim3(SEL_T), lds, geoa3_stream(stream), dis, N, mode, drop_num, alpha,
                     idx, count, stats);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
