// micro-benchmark of grid_nn1_kernel: hipcc --offload-arch=gfx950 -O3 -o nn1_ub nn1_ub.hip
#include "../../geoa3_amd/csrc/geom_grid.hip"
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
bool geoa3_prof_on() { return false; }
void geoa3_prof_begin(int, hipStream_t) {}
void geoa3_prof_end(int, hipStream_t) {}
template <int MODE>
float run(const float* A, const float* R, int B, int N, int* prior_ar, int* prior_ra, float* d_ar, int* i_ar, float* d_ra, int* i_ra, int iters) {
  const size_t lds = grid_nn1_lds(N);
  auto k = grid_nn1_kernel<1, MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(B, 2), dim3(GT), lds, 0, A, R, N, N, prior_ar, prior_ra, d_ar, i_ar, d_ra, i_ra, NN1_WIDE);
  hipEventRecord(e0, 0);
  for (int w = 0; w < iters; ++w) hipLaunchKernelGGL(k, dim3(B, 2), dim3(GT), lds, 0, A, R, N, N, prior_ar, prior_ra, d_ar, i_ar, d_ra, i_ra, NN1_WIDE);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / iters;
}
int main(int argc, char** argv) {
  const int B = 250, N = 1024;
  const float noise = argc > 1 ? atof(argv[1]) : 0.03f;
  std::vector<float> ori((size_t)B * 3 * N), adv((size_t)B * 3 * N);
  srand(5);
  auto rnd = [] { return (rand() % 20001 - 10000) * 1e-4f; };
  for (int b = 0; b < B; ++b)
    for (int i = 0; i < N; ++i) {
      float x = rnd(), y = rnd(), z = rnd();
      const float n = sqrtf(x * x + y * y + z * z) + 1e-6f;
      x /= n; y /= n; z /= n;
      const size_t o = (size_t)b * 3 * N + i;
      ori[o] = x; ori[o + N] = y; ori[o + 2 * N] = z;
      adv[o] = x + noise * rnd(); adv[o + N] = y + noise * rnd(); adv[o + 2 * N] = z + noise * rnd();
    }
  float *dA, *dR, *d_ar, *d_ra; int *i_ar, *i_ra;
  hipMalloc(&dA, adv.size() * 4); hipMalloc(&dR, ori.size() * 4);
  hipMalloc(&d_ar, (size_t)B * N * 4); hipMalloc(&d_ra, (size_t)B * N * 4); hipMalloc(&i_ar, (size_t)B * N * 4); hipMalloc(&i_ra, (size_t)B * N * 4);
  hipMemcpy(dA, adv.data(), adv.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dR, ori.data(), ori.size() * 4, hipMemcpyHostToDevice);
  // first run without priors fills i_*, then they serve as priors (the loop's steady state)
  run<0>(dA, dR, B, N, nullptr, nullptr, d_ar, i_ar, d_ra, i_ra, 1);
  printf("noise %.2f\n", noise);
  printf("full, own-index seed  %.1f us\n", run<0>(dA, dR, B, N, nullptr, nullptr, d_ar, i_ar, d_ra, i_ra, 10));
  printf("full, prior seed      %.1f us\n", run<0>(dA, dR, B, N, i_ar, i_ra, d_ar, i_ar, d_ra, i_ra, 10));
  printf("balanced, prior seed  %.1f us\n", run<4>(dA, dR, B, N, i_ar, i_ra, d_ar, i_ar, d_ra, i_ra, 10));
  printf("build only            %.1f us\n", run<1>(dA, dR, B, N, i_ar, i_ra, d_ar, i_ar, d_ra, i_ra, 10));
  printf("seeds only            %.1f us\n", run<2>(dA, dR, B, N, i_ar, i_ra, d_ar, i_ar, d_ra, i_ra, 10));
  {
    unsigned long long* cnt; hipMalloc(&cnt, 64); hipMemset(cnt, 0, 64);
    const size_t lds = grid_nn1_lds(N);
    auto k = grid_nn1_kernel<1, 3>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(k, dim3(B, 2), dim3(GT), lds, 0, dA, dR, N, N, i_ar, i_ra, (float*)cnt, i_ar, (float*)cnt, i_ra, NN1_WIDE);
    unsigned long long h[8]; hipMemcpy(h, cnt, 64, hipMemcpyDeviceToHost);
    printf("columns iterated per query: <=4: %llu  <=16: %llu  <=64: %llu  >64: %llu\n", h[2], h[3], h[4], h[5]);
    printf("per query: %.1f columns, %.1f candidates\n", h[0] / (2.0 * B * N), h[1] / (2.0 * B * N));
  }
  return 0;
}
