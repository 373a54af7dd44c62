// micro-benchmark of sa2_fwd8_kernel (which phase bounds it?): hipcc --offload-arch=gfx950 -O3 -o sa2f_ub sa2f_ub.hip
#include "../../geoa3_amd/csrc/pointnet2_sa2.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
template <int MODE>
float run8(const Sa2FwdArgs& a, int iters) {
  const int lds = sa2_fwd8_lds();
  auto k = sa2_fwd8_kernel<MODE>;
  hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, a);
  hipEventRecord(e0, 0);
  for (int w = 0; w < iters; ++w) hipLaunchKernelGGL(k, dim3(256), dim3(512), lds, 0, a);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  return ms * 1000.f / iters;
}
int main() {
  const int B = 250, M = 128, N1 = 512;
  const long centres = (long)B * M;
  std::vector<int> idx(centres * 64);
  std::vector<float> rT((size_t)B * N1 * 128), shift((size_t)B * 128 * M), w1(128 * 128), w2(256 * 128), b1(128), b2(256);
  srand(3);
  for (long r = 0; r < centres; ++r) {
    int cnt = 20 + rand() % 45;
    std::vector<int> pool(N1);
    for (int i = 0; i < N1; ++i) pool[i] = i;
    for (int i = 0; i < cnt; ++i) std::swap(pool[i], pool[i + rand() % (N1 - i)]);
    std::sort(pool.begin(), pool.begin() + cnt);
    for (int s = 0; s < 64; ++s) idx[r * 64 + s] = s < cnt ? pool[s] : pool[0];
  }
  auto rnd = [] { return (rand() % 2001 - 1000) * 1e-3f; };
  for (auto& v : rT) v = rnd();
  for (auto& v : shift) v = 0.2f * rnd();
  for (auto& v : w1) v = 0.1f * rnd();
  for (auto& v : w2) v = 0.1f * rnd();
  for (auto& v : b1) v = 0.1f * rnd();
  for (auto& v : b2) v = 0.1f * rnd();
  float *drT, *dsh, *dw1, *dw2, *db1, *db2, *out; int *didx, *arg; unsigned *m0, *m1; char* scr;
  hipMalloc(&drT, rT.size() * 4); hipMalloc(&dsh, shift.size() * 4); hipMalloc(&dw1, w1.size() * 4); hipMalloc(&dw2, w2.size() * 4);
  hipMalloc(&db1, 512); hipMalloc(&db2, 1024); hipMalloc(&out, centres * 256 * 4); hipMalloc(&arg, centres * 256 * 4);
  hipMalloc(&didx, idx.size() * 4); hipMalloc(&m0, centres * 128 * 8); hipMalloc(&m1, centres * 128 * 8); hipMalloc(&scr, 196608 + 512);
  hipMemcpy(drT, rT.data(), rT.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dsh, shift.data(), shift.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw1, w1.data(), w1.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw2, w2.data(), w2.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(db1, b1.data(), 512, hipMemcpyHostToDevice);
  hipMemcpy(db2, b2.data(), 1024, hipMemcpyHostToDevice);
  hipMemcpy(didx, idx.data(), idx.size() * 4, hipMemcpyHostToDevice);
  _Float16* img1 = (_Float16*)scr; _Float16* img2 = (_Float16*)(scr + 65536); float* un = (float*)(scr + 196608);
  launch_frag_image(dw1, 128, 128, img1, un, 0);
  launch_frag_image(dw2, 256, 128, img2, un + 64, 0);
  Sa2FwdArgs a{drT, didx, dsh, db1, db2, img1, un, img2, un + 64, out, arg, m0, m1, B, N1, M};
  printf("fwd8 all               %.1f us\n", run8<0>(a, 5));
  printf("fwd8 no W2 MFMAs       %.1f us\n", run8<1>(a, 5));
  printf("fwd8 no W1 MFMAs       %.1f us\n", run8<2>(a, 5));
  printf("fwd8 no pooled stores  %.1f us\n", run8<3>(a, 5));
  printf("fwd8 no gather         %.1f us\n", run8<4>(a, 5));
  printf("fwd8 all               %.1f us\n", run8<0>(a, 5));
  return 0;
}
