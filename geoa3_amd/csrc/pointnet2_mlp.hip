// Epilogues of the PointNet++ shared MLPs (pointnet2_modules.py:9-19,62-70: Conv2d 1x1 + BatchNorm2d + ReLU, then
// max over the nsample axis), as single-pass HIP kernels around the GEMMs: with the eval-mode BatchNorm scale folded
// into the GEMM weights on the host, a layer's tail is y = relu(z + shift[c]); the last layer's tail additionally
// takes the max over the 64 (or k*64) samples of a ball and never writes y.  The [B,C,npoint,nsample] tensors are
// 2-4 GB at B=250, so every avoided pass over them is ~1 ms.
#include "common.h"

namespace {

// y = relu(z + shift[c]) in place; z [B*C rows][L]; 16-byte vectorised; block = (row, 1024-element segment)
__global__ __launch_bounds__(256) void bias_relu_kernel(float* __restrict__ z, const float* __restrict__ shift, int C,
                                                        long L, unsigned segs) {
  const long row = blockIdx.x / segs;
  const unsigned seg = blockIdx.x % segs;
  const float s = shift[row % C];
  float* p = z + row * L;
  const long i = ((long)seg * 256 + threadIdx.x) * 4;
  if (i + 3 < L && (L & 3) == 0) {
    float4 v = *reinterpret_cast<float4*>(p + i);
    v.x = fmaxf(v.x + s, 0.f);
    v.y = fmaxf(v.y + s, 0.f);
    v.z = fmaxf(v.z + s, 0.f);
    v.w = fmaxf(v.w + s, 0.f);
    *reinterpret_cast<float4*>(p + i) = v;
  } else {
    for (long j = i; j < min(i + 4, L); ++j) p[j] = fmaxf(p[j] + s, 0.f);
  }
}

// dz = (y > 0) ? g : 0   (may run in place on g)
__global__ __launch_bounds__(256) void relu_grad_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                        float* __restrict__ dz, long total) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < total) {
    const float4 yy = *reinterpret_cast<const float4*>(y + i);
    float4 gg = *reinterpret_cast<const float4*>(g + i);
    gg.x = yy.x > 0.f ? gg.x : 0.f;
    gg.y = yy.y > 0.f ? gg.y : 0.f;
    gg.z = yy.z > 0.f ? gg.z : 0.f;
    gg.w = yy.w > 0.f ? gg.w : 0.f;
    *reinterpret_cast<float4*>(dz + i) = gg;
  } else {
    for (long j = i; j < total; ++j) dz[j] = y[j] > 0.f ? g[j] : 0.f;
  }
}

// out[row] = relu(max_s z[row][s] + shift[c]), arg[row] = first maximising s.
// S == 64 (every SSG ball): 16 lanes x 16-byte loads cover one row, a wavefront streams 4 rows per load and 16 rows
// in total (4 independent 1-KiB loads in flight); the max / arg-max of a row is a 4-step shuffle inside its 16 lanes.
__global__ __launch_bounds__(256) void bias_relu_max64_kernel(const float* __restrict__ z,
                                                              const float* __restrict__ shift, float* __restrict__ out,
                                                              int32_t* __restrict__ arg, int C, long M, long rows) {
  const int lane = threadIdx.x & 63, sub = lane & 15, rsel = lane >> 4;
  const long row0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16;
  float4 v[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long row = row0 + i * 4 + rsel;
    v[i] = row < rows ? *reinterpret_cast<const float4*>(z + row * 64 + sub * 4)
                      : make_float4(-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff());
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const long row = row0 + i * 4 + rsel;
    float m = v[i].x;
    int a = sub * 4;
    if (v[i].y > m) { m = v[i].y; a = sub * 4 + 1; }
    if (v[i].z > m) { m = v[i].z; a = sub * 4 + 2; }
    if (v[i].w > m) { m = v[i].w; a = sub * 4 + 3; }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const float m2 = __shfl_xor(m, o, 64);
      const int a2 = __shfl_xor(a, o, 64);
      const bool take = m2 > m || (m2 == m && a2 < a);
      m = take ? m2 : m;
      a = take ? a2 : a;
    }
    if (sub == 0 && row < rows) {
      const int c = (int)((row / M) % C);
      out[row] = fmaxf(m + shift[c], 0.f);
      arg[row] = a;
    }
  }
}

// general S: one wavefront per row
__global__ __launch_bounds__(256) void bias_relu_max_kernel(const float* __restrict__ z,
                                                            const float* __restrict__ shift, float* __restrict__ out,
                                                            int32_t* __restrict__ arg, int C, long M, int S, long rows) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  const float* p = z + row * S;
  float v = -__builtin_inff();
  int a = 0x7fffffff;
  for (int s = lane; s < S; s += 64) {
    const float x = p[s];
    if (x > v) { v = x; a = s; }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const float v2 = __shfl_xor(v, o, 64);
    const int a2 = __shfl_xor(a, o, 64);
    const bool take = v2 > v || (v2 == v && a2 < a);
    v = take ? v2 : v;
    a = take ? a2 : a;
  }
  if (lane == 0) {
    const int c = (int)((row / M) % C);
    out[row] = fmaxf(v + shift[c], 0.f);
    arg[row] = a;
  }
}

// dz[row][s] = (s == arg[row] && out[row] > 0) ? g[row] : 0 ; 16-byte stores, 4 elements per thread
__global__ __launch_bounds__(256) void bias_relu_max_grad_kernel(const float* __restrict__ g,
                                                                 const float* __restrict__ out,
                                                                 const int32_t* __restrict__ arg, float* __restrict__ dz,
                                                                 int S, long total) {
  const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
  if (e >= total) return;
  const long row = e / S;
  const int s0 = (int)(e - row * S);
  const float gv = out[row] > 0.f ? g[row] : 0.f;
  const int a = arg[row];
  float4 v;
  v.x = s0 == a ? gv : 0.f;
  v.y = s0 + 1 == a ? gv : 0.f;
  v.z = s0 + 2 == a ? gv : 0.f;
  v.w = s0 + 3 == a ? gv : 0.f;
  *reinterpret_cast<float4*>(dz + e) = v;
}

// y[row][s] = relu(z[row][s] + shift[row]) in place; one wavefront per row
__global__ __launch_bounds__(256) void shift_relu_kernel(float* __restrict__ z, const float* __restrict__ shift, long rows,
                                                         int S) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float sh = shift[row];
  float* zr = z + row * S;
  for (int s = threadIdx.x & 63; s < S; s += 64) zr[s] = fmaxf(zr[s] + sh, 0.f);
}

// dz[row][s] = y[row][s] > 0 ? g[row][s] : 0;  dshift[row] = sum_s dz[row][s] (lanes in a fixed order)
__global__ __launch_bounds__(256) void shift_relu_grad_kernel(const float* __restrict__ y, const float* __restrict__ g,
                                                              float* __restrict__ dz, float* __restrict__ dshift,
                                                              long rows, int S) {
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  if (y == nullptr) {   // g is already gated: only the row sums
    for (int s = lane; s < S; s += 64) acc += g[row * S + s];
  } else {
    for (int s = lane; s < S; s += 64) {
      const float v = y[row * S + s] > 0.f ? g[row * S + s] : 0.f;
      dz[row * S + s] = v;
      acc += v;
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) dshift[row] = acc;
}

}  // namespace

extern "C" int geoa3_pn2_shift_relu(float* z, const float* shift, long rows, int S, void* stream) {
  if (!z || !shift || rows <= 0 || S <= 0 || (rows + 3) / 4 > 2147483647L) return GEOA3_EINVAL;
  hipLaunchKernelGGL(shift_relu_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, geoa3_stream(stream), z, shift,
                     rows, S);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_pn2_shift_relu_grad(const float* y, const float* g, float* dz, float* dshift, long rows, int S,
                                         void* stream) {
  if (!g || (y && !dz) || !dshift || rows <= 0 || S <= 0 || (rows + 3) / 4 > 2147483647L) return GEOA3_EINVAL;
  hipLaunchKernelGGL(shift_relu_grad_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, geoa3_stream(stream), y, g,
                     dz, dshift, rows, S);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_pn2_bias_relu(float* z, const float* shift, int B, int C, long L, void* stream) {
  if (!z || !shift || B <= 0 || C <= 0 || L <= 0) return GEOA3_EINVAL;
  const unsigned segs = (unsigned)((L + 1023) / 1024);
  const long blocks = (long)B * C * segs;
  if (blocks > 2147483647L) return GEOA3_ENOSUPPORT;
  hipLaunchKernelGGL(bias_relu_kernel, dim3((unsigned)blocks), dim3(256), 0, geoa3_stream(stream), z, shift, C, L, segs);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_pn2_relu_grad(const float* y, const float* g, float* dz, long total, void* stream) {
  if (!y || !g || !dz || total <= 0) return GEOA3_EINVAL;
  hipLaunchKernelGGL(relu_grad_kernel, dim3((unsigned)((total + 1023) / 1024)), dim3(256), 0, geoa3_stream(stream), y, g,
                     dz, total);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_pn2_bias_relu_max(const float* z, const float* shift, int B, int C, long M, int S, float* out,
                                       int32_t* arg, void* stream) {
  if (!z || !shift || !out || !arg || B <= 0 || C <= 0 || M <= 0 || S <= 0) return GEOA3_EINVAL;
  const long rows = (long)B * C * M;
  if (S == 64)
    hipLaunchKernelGGL(bias_relu_max64_kernel, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, geoa3_stream(stream), z,
                       shift, out, arg, C, M, rows);
  else
    hipLaunchKernelGGL(bias_relu_max_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, geoa3_stream(stream), z,
                       shift, out, arg, C, M, S, rows);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_pn2_bias_relu_max_grad(const float* g, const float* out, const int32_t* arg, int B, int C, long M,
                                            int S, float* dz, void* stream) {
  if (!g || !out || !arg || !dz || B <= 0 || C <= 0 || M <= 0 || S <= 0) return GEOA3_EINVAL;
  if (S % 4 != 0) return GEOA3_ENOSUPPORT;
  const long total = (long)B * C * M * S;
  hipLaunchKernelGGL(bias_relu_max_grad_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0,
                     geoa3_stream(stream), g, out, arg, dz, S, total);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
