// micro-benchmark of sa2_bwd_kernel (which phase bounds it?): hipcc --offload-arch=gfx950 -O3 -fno-honor-nans -fno-slp-vectorize -I../../include -o sa2_ub sa2_ub.hip
#include "../../geoa3_amd/csrc/pointnet2_sa2.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
template <int MODE>
float run(const Sa2BwdArgs& a, int iters) {
  const int lds = sa2_bwd_lds();
  auto k = sa2_bwd_kernel<MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(512), dim3(256), lds, 0, a);
  hipEventRecord(e0, 0);
  for (int w = 0; w < iters; ++w) hipLaunchKernelGGL(k, dim3(512), dim3(256), lds, 0, a);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / iters;
}
int main() {
  const int B = 250, M = 128;
  const long centres = (long)B * M;
  std::vector<int> arg(centres * 256);
  std::vector<float> g(centres * 256), w2(256 * 128), w1(128 * 128);
  std::vector<unsigned long long> mk(centres * 128);
  srand(2);
  for (auto& v : arg) v = rand() & 63;
  for (auto& v : g) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : w2) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : w1) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : mk) v = ((unsigned long long)rand() << 33) ^ ((unsigned long long)rand() << 11) ^ rand();
  float *dg, *dw2, *dw1, *eg, *da0; int *darg, *ec; unsigned long long *m0, *m1;
  hipMalloc(&dg, g.size() * 4); hipMalloc(&darg, arg.size() * 4); hipMalloc(&eg, g.size() * 4); hipMalloc(&ec, g.size() * 4);
  hipMalloc(&dw2, w2.size() * 4); hipMalloc(&dw1, w1.size() * 4); hipMalloc(&m0, mk.size() * 8); hipMalloc(&m1, mk.size() * 8);
  hipMalloc(&da0, (size_t)B * 128 * M * 64 * 4);
  hipMemcpy(dg, g.data(), g.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(darg, arg.data(), arg.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(m0, mk.data(), mk.size() * 8, hipMemcpyHostToDevice);
  hipMemcpy(m1, mk.data(), mk.size() * 8, hipMemcpyHostToDevice);
  launch_sa2_sort(dg, darg, eg, ec, centres, 0);
  char* scr; hipMalloc(&scr, 65536 + 256);
  launch_frag_image(dw1, 128, 128, scr, (float*)(scr + 65536), 0);
  Sa2BwdArgs a{eg, ec, dw2, (_Float16*)scr, (float*)(scr + 65536), m1, m0, da0, B, M};
  printf("mode0 (all)        %.1f us\n", run<0>(a, 5));
  printf("mode1 (no phase 1) %.1f us\n", run<1>(a, 5));
  printf("mode2 (no phase 2) %.1f us\n", run<2>(a, 5));
  printf("mode3 (no stores)  %.1f us\n", run<3>(a, 5));
  printf("mode0 (all)        %.1f us\n", run<0>(a, 5));
  return 0;
}
