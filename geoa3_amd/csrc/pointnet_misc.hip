// Small PointNet pieces that are not MFMA-shaped (gfx950): the backward of the 3-channel input layers with the 3x3
// input transform (Model/PointNet.py:79,137-139).  Their forward is folded into the 64-input convolution that follows
// (pointnet_gemm.hip, conv_cm64_kernel<.., FIRST>).
#include "pointnet_kernels.h"

namespace {

// One workgroup per instance (deterministic dT reduction).
__global__ __launch_bounds__(256) void conv_in3_bwd_kernel(const float* __restrict__ g, const float* __restrict__ W,
                                                           const float* __restrict__ T, const float* __restrict__ x,
                                                           float* __restrict__ dx, float* __restrict__ dT,
                                                           int accumulate, int N) {
  __shared__ float s_w[64 * 3], s_t[9], s_red[4][9];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 192) s_w[tid] = W[tid];
  if (tid < 9) s_t[tid] = T ? T[(size_t)b * 9 + tid] : ((tid % 4 == 0) ? 1.f : 0.f);
  __syncthreads();
  const float* gb = g + (size_t)b * 64 * N;
  const float* xb = x + (size_t)b * 3 * N;
  float* dxb = dx + (size_t)b * 3 * N;
  float t[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) t[i] = 0.f;
  for (int n = tid; n < N; n += 256) {
    float q0 = 0.f, q1 = 0.f, q2 = 0.f;  // d/d x' (the transformed input)
#pragma unroll 8
    for (int co = 0; co < 64; ++co) {
      const float gv = gb[(size_t)co * N + n];
      q0 += s_w[co * 3] * gv;
      q1 += s_w[co * 3 + 1] * gv;
      q2 += s_w[co * 3 + 2] * gv;
    }
    // x' = T^T x  =>  dx[d] = sum_c T[d][c] q[c]
    float d0 = s_t[0] * q0 + s_t[1] * q1 + s_t[2] * q2;
    float d1 = s_t[3] * q0 + s_t[4] * q1 + s_t[5] * q2;
    float d2 = s_t[6] * q0 + s_t[7] * q1 + s_t[8] * q2;
    if (accumulate) {
      d0 += dxb[n];
      d1 += dxb[N + n];
      d2 += dxb[2 * N + n];
    }
    dxb[n] = d0;
    dxb[N + n] = d1;
    dxb[2 * N + n] = d2;
    if (dT) {
      const float x0 = xb[n], x1 = xb[N + n], x2 = xb[2 * N + n];
      t[0] += x0 * q0; t[1] += x0 * q1; t[2] += x0 * q2;
      t[3] += x1 * q0; t[4] += x1 * q1; t[5] += x1 * q2;
      t[6] += x2 * q0; t[7] += x2 * q1; t[8] += x2 * q2;
    }
  }
  if (dT) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const float v = wave_sum(t[i]);
      if (lane == 0) s_red[wave][i] = v;
    }
    __syncthreads();
    if (tid < 9) dT[(size_t)b * 9 + tid] = s_red[0][tid] + s_red[1][tid] + s_red[2][tid] + s_red[3][tid];
  }
}

}  // namespace

int launch_conv_in3_bwd(const float* g, const float* W, const float* T, const float* x, float* dx, float* dT,
                        int accumulate, int B, int N, hipStream_t s) {
  hipLaunchKernelGGL(conv_in3_bwd_kernel, dim3(B), dim3(256), 0, s, g, W, T, x, dx, dT, accumulate, N);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
