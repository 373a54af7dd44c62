// micro-benchmark of sa2_bwd_kernel (which phase bounds it?): hipcc --offload-arch=gfx950 -O3 -fno-honor-nans -fno-slp-vectorize -I../../include -o sa2_ub sa2_ub.hip
#include "../../geoa3_amd/csrc/pointnet2_sa2.hip"
#include "sa2_bwdl_experiment.h"
#include <cstdio>
#include <cmath>
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
template <int MODE>
float runl(const Sa2BwdLArgs& a, int iters) {
  const int lds = sa2_bwdl_lds();
  auto k = sa2_bwdl_kernel<MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(1024), lds, 0, a);
  hipEventRecord(e0, 0);
  for (int w = 0; w < iters; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(1024), lds, 0, a);
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
  unsigned short* eo; hipMalloc(&eo, centres * 64 * 2);
  hipLaunchKernelGGL(sa2_offsets_kernel, dim3((unsigned)((centres * 64 + 255) / 256)), dim3(256), 0, 0, ec, eo, centres);
  char* scr; hipMalloc(&scr, 65536 + 256);
  launch_frag_image(dw1, 128, 128, scr, (float*)(scr + 65536), 0);
  Sa2BwdArgs a{eg, ec, dw2, (_Float16*)scr, (float*)(scr + 65536), m1, m0, da0, B, M};
  printf("mode0 (all)        %.1f us\n", run<0>(a, 5));
  printf("mode1 (no phase 1) %.1f us\n", run<1>(a, 5));
  printf("mode2 (no phase 2) %.1f us\n", run<2>(a, 5));
  printf("mode3 (no stores)  %.1f us\n", run<3>(a, 5));
  printf("mode0 (all)        %.1f us\n", run<0>(a, 5));
  std::vector<float> ref((size_t)B * 128 * M * 64), got(ref.size());
  hipMemcpy(ref.data(), da0, ref.size() * 4, hipMemcpyDeviceToHost);
  hipMemset(da0, 0xff, ref.size() * 4);
  float* dw1t; hipMalloc(&dw1t, w1.size() * 4);   // the image above was made from `w1` as W1^T itself
  hipMemcpy(dw1t, w1.data(), w1.size() * 4, hipMemcpyHostToDevice);
  long long* dbg; hipMalloc(&dbg, 16 * 64);
  Sa2BwdLArgs al{eg, ec, eo, dw2, dw1t, m1, m0, da0, B, M, dbg};
  printf("L mode0 (all)        %.1f us\n", runl<0>(al, 5));
  hipMemcpy(got.data(), da0, got.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0, scale = 0;
  size_t bad = 0;
  for (size_t i = 0; i < ref.size(); ++i) {
    const double d = fabs((double)ref[i] - got[i]);
    if (!(d <= 1e30)) ++bad;
    if (d > worst) worst = d;
    if (fabs(ref[i]) > scale) scale = fabs(ref[i]);
  }
  printf("L vs old: max |diff| %.3g of %.3g, non-finite %zu\n", worst, scale, bad);
  printf("L mode1 (no phase 1) %.1f us\n", runl<1>(al, 5));
  printf("L mode2 (no phase 2) %.1f us\n", runl<2>(al, 5));
  printf("L mode3 (no stores)  %.1f us\n", runl<3>(al, 5));
  printf("L mode0 (all)        %.1f us\n", runl<0>(al, 5));
  printf("L mode4 (timed)      %.1f us\n", runl<4>(al, 1));
  long long t[16][8];
  hipMemcpy(t, dbg, sizeof(t), hipMemcpyDeviceToHost);
  const char* nm[8] = {"stage+fetch", "barrier A", "split 0", "barrier B0", "first (M|P)", "second(P|M)", "C+split1+B1", "M1+P1a"};
  for (int i = 0; i < 8; ++i) {
    printf("   %-12s", nm[i]);
    for (int w = 0; w < 16; ++w) printf(" %5.0f", t[w][i] / 125.0);
    printf("\n");
  }
  return 0;
}
