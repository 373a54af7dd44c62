// Curvature (kappa) and the fused geometric objective + gradient (gfx950).
// Reference: Lib/loss_utils.py:25-97 (norm_l2_loss, chamfer_loss, pseudo_chamfer_loss, hausdorff_loss,
// _get_kappa_ori, _get_kappa_adv, curvature_loss) and Lib/utility.py:30-31 (_normalize), as combined by
// Attacker/geoA3_attack.py:131-166.  The reference builds [b,3,n,k] neighbour tensors with knn_gather and
// lets autograd scatter the gradient back; here one workgroup owns one instance, keeps the cloud and the
// gradient accumulators in LDS, and never materialises the neighbour tensor.
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

__global__ __launch_bounds__(256) void kappa_kernel(const float* __restrict__ pc, const float* __restrict__ normal,
                                                    const int32_t* __restrict__ knn_idx,
                                                    const int32_t* __restrict__ nn_idx, int N, int Nn, int k,
                                                    float* __restrict__ kappa) {
  const int b = blockIdx.y;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const float* P = pc + (size_t)b * 3 * N;
  const float* Nm = normal + (size_t)b * 3 * Nn;
  const int ni = nn_idx ? nn_idx[(size_t)b * N + i] : i;
  const float nx = Nm[ni], ny = Nm[Nn + ni], nz = Nm[2 * Nn + ni];
  const int32_t* nb = knn_idx + ((size_t)b * N + i) * (k + 1);
  auto fetch = [&](int j, float& x, float& y, float& z) {
    x = P[j];
    y = P[N + j];
    z = P[2 * N + j];
  };
  kappa[(size_t)b * N + i] = kappa_point(P[i], P[N + i], P[2 * N + i], nx, ny, nz, nb, k, fetch);
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

// The non-deterministic form (geoa3_geo_args.deterministic == 0): neighbour terms are scattered with LDS float atomics,
// whose order is free.  The default is the pair-parallel, owner-ordered pair of kernels below.
__global__ __launch_bounds__(GEO_BLOCK) void geo_loss_grad_atomic_kernel(geoa3_geo_args A) {
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

  {
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

}  // namespace

extern "C" int geoa3_kappa(const float* pc, const float* normal, const int32_t* knn_idx, const int32_t* nn_idx,
                           int B, int N, int Nn, int k, float* kappa, void* stream) {
  if (!pc || !normal || !knn_idx || !kappa || B <= 0 || N <= 0 || k <= 0 || Nn < 0) return GEOA3_EINVAL;
  if (Nn == 0) Nn = N;
  if (!nn_idx && Nn != N) return GEOA3_EINVAL;
  dim3 grid((N + 255) / 256, B);
  hipLaunchKernelGGL(kappa_kernel, grid, dim3(256), 0, geoa3_stream(stream), pc, normal, knn_idx, nn_idx, N, Nn,
                     k, kappa);
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
  const size_t base = ((size_t)7 * a->N + GEO_WAVES * 5 + 4) * sizeof(float);
  if (base > 160 * 1024) return GEOA3_ENOSUPPORT;  // N <= ~5800 points per instance
  if (a->deterministic && a->grad) {
    // counts / offsets, then (if they fit) the per-point normals, then the reverse lists of one chunk of sources: all of
    // them in one chunk when they fit (N = 1024, k = 16: 70 KB), at least 2048 entries otherwise
    const size_t Nr = a->Nr > 0 ? a->Nr : a->N;
    const size_t idx = ((size_t)a->N + 1) * sizeof(int), nrm = (size_t)3 * a->N * sizeof(float);
    const size_t cap = 160 * 1024 - 512, all = (size_t)a->N * a->k > Nr ? (size_t)a->N * a->k : Nr;
    if (base + idx + 2048 * sizeof(int) > cap) return GEOA3_ENOSUPPORT;
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
