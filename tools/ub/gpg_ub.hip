// micro-benchmark of group_points_grad64_kernel variants (what bounds it?): hipcc --offload-arch=gfx950 -O3 -o gpg_ub gpg_ub.hip
#include "../../geoa3_amd/csrc/pointnet2_ops.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
template <int MODE, int PF = 8>
float run(const float* G, const int* I, float* out, float* rs, int B, int C, int N, int M, int iters) {
  const size_t lds = (size_t)4 * GPG_CT * N * sizeof(float);
  auto k = group_points_grad64_kernel<true, MODE, PF, GPG_CT>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(C / GPG_CT, B), dim3(256), lds, 0, G, I, out, C, N, M, rs);
  hipEventRecord(e0, 0);
  for (int w = 0; w < iters; ++w) hipLaunchKernelGGL(k, dim3(C / GPG_CT, B), dim3(256), lds, 0, G, I, out, C, N, M, rs);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / iters;
}
int main() {
  const int B = 250, C = 128, N = 512, M = 128;
  std::vector<int> idx((size_t)B * M * 64);
  srand(1);
  for (size_t r = 0; r < (size_t)B * M; ++r) {
    int cnt = 20 + rand() % 45;
    std::vector<int> pool(N);
    for (int i = 0; i < N; ++i) pool[i] = i;
    for (int i = 0; i < cnt; ++i) std::swap(pool[i], pool[i + rand() % (N - i)]);
    std::sort(pool.begin(), pool.begin() + cnt);
    for (int s = 0; s < 64; ++s) idx[r * 64 + s] = s < cnt ? pool[s] : pool[0];
  }
  float *G, *out, *rs; int* I;
  hipMalloc(&G, (size_t)B * C * M * 64 * 4); hipMalloc(&out, (size_t)B * C * N * 4); hipMalloc(&rs, (size_t)B * C * M * 4);
  hipMalloc(&I, idx.size() * 4);
  hipMemcpy(I, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
  hipMemset(G, 0, (size_t)B * C * M * 64 * 4);
  printf("mode0 (atomics)      %.1f us\n", run<0>(G, I, out, rs, B, C, N, M, 10));
  printf("mode1 (plain rmw)    %.1f us\n", run<1>(G, I, out, rs, B, C, N, M, 10));
  printf("mode2 (no lds update)%.1f us\n", run<2>(G, I, out, rs, B, C, N, M, 10));
  printf("mode0 no rowsum      %.1f us\n", run<0>(G, I, out, nullptr, B, C, N, M, 10));
  printf("mode0 PF=2           %.1f us\n", (run<0, 2>(G, I, out, rs, B, C, N, M, 10)));
  printf("mode0 PF=6           %.1f us\n", (run<0, 6>(G, I, out, rs, B, C, N, M, 10)));
  printf("mode0 PF=8           %.1f us\n", (run<0, 8>(G, I, out, rs, B, C, N, M, 10)));
  printf("mode2 PF=8           %.1f us\n", (run<2, 8>(G, I, out, rs, B, C, N, M, 10)));
  return 0;
}
