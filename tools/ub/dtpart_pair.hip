// NOTEBOOK 5a, second reproducer: the loop's REAL producer / consumer pair, stand-alone (no torch, no Python):
// conv_bwd_chain_kernel (1000 workgroups at B = 250, N = 1024; its last instructions store 36 bytes of dT3 partial sums
// per workgroup) followed by reduce_dT_kernel (9 workgroups; its first instructions read them), on random inputs, the two
// input sets alternating so that a stale partial row shows as the OTHER set's value.
//
//   hipcc --offload-arch=gfx950 -O3 -o dtpart_pair dtpart_pair.hip -L<dir of a libgeoa3_hip.so build> -lgeoa3_hip -ldl
//   LD_LIBRARY_PATH=<that dir> ./dtpart_pair [pairs] [mode]
// Run against tools/ub/lib_two_wave (`python -m geoa3_amd.build --variant tools/ub/lib_two_wave --no-file-flags`: the chain
// kernels compiled WITH packed-FP32 instructions, the faulty build) and against the product library.  A build of
// tools/ub/conv_bwd_chain_probe.patch (stage checksums inside the kernel) adds the per-stage report.
#include "../../geoa3_amd/csrc/pointnet_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                        \
    }                                                                                 \
  } while (0)

__global__ void compare_kernel(const float* __restrict__ got, const float* __restrict__ want, const float* __restrict__ other,
                               int n, unsigned long long* __restrict__ bad, unsigned it) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n) return;
  const unsigned g = __float_as_uint(got[e]);
  if (g != __float_as_uint(want[e])) {
    atomicAdd(&bad[0], 1ull);
    atomicAdd(&bad[g == __float_as_uint(other[e]) ? 1 : 2], 1ull);
    atomicMax(&bad[3], (unsigned long long)it);
    atomicAdd(&bad[4 + (e / 9 >= n / 9 - 16 ? 1 : 0)], 1ull);   // [5]: in the last 16 instances
  }
}
// the partial rows themselves: got / want / other are [rows][pitch]; a row is compared over its 9 floats
__global__ void compare_rows_kernel(const float* __restrict__ got, const float* __restrict__ want, const float* __restrict__ other,
                                    int rows, int pitch, unsigned long long* __restrict__ bad) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  bool eq = true, st = true;
  for (int j = 0; j < 9; ++j) {
    const unsigned g = __float_as_uint(got[(size_t)r * pitch + j]);
    eq &= g == __float_as_uint(want[(size_t)r * pitch + j]);
    st &= g == __float_as_uint(other[(size_t)r * pitch + j]);
    if (g != __float_as_uint(want[(size_t)r * pitch + j])) {
      atomicAdd(&bad[16 + j], 1ull);                                              // which float of the row
      if (g == __float_as_uint(other[(size_t)r * pitch + j])) atomicAdd(&bad[26 + j], 1ull);   // ... and it holds the old value
    }
  }
  if (!eq) {
    atomicAdd(&bad[8], 1ull);
    atomicAdd(&bad[st ? 9 : 10], 1ull);
    atomicAdd(&bad[12 + (r & 3)], 1ull);   // which of the instance's four workgroups
  }
}
__global__ void compare_dx_kernel(const float* __restrict__ got, const float* __restrict__ want, size_t n,
                                  unsigned long long* __restrict__ bad) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    if (__float_as_uint(got[i]) != __float_as_uint(want[i])) {
      atomicAdd(&bad[40], 1ull);
      const size_t col = i % 1024, comp = (i / 1024) % 3;
      atomicAdd(&bad[64 + (col & 63)], 1ull);
      atomicAdd(&bad[128 + comp], 1ull);
      atomicAdd(&bad[132 + ((col >> 6) & 3)], 1ull);
      const float rel = fabsf(got[i] - want[i]) / fmaxf(fabsf(want[i]), 1e-20f);
      atomicAdd(&bad[rel < 1e-6f ? 41 : rel < 1e-4f ? 42 : rel < 1e-2f ? 43 : 44], 1ull);
    }
}
// stage checksums of the probe build: [5][B * N]; bad[48 + stage]: columns whose checksum differs, bad[56 + stage]: of those in lanes 48-63
__global__ void compare_stages_kernel(const unsigned* __restrict__ got, const unsigned* __restrict__ want, size_t bn,
                                      unsigned long long* __restrict__ bad) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < 5 * bn; i += (size_t)gridDim.x * 256)
    if (got[i] != want[i]) {
      const int st = (int)(i / bn);
      atomicAdd(&bad[48 + st], 1ull);
      if (((i % bn) & 63) >= 48) atomicAdd(&bad[56 + st], 1ull);
    }
}
__global__ void dirty_kernel(float* __restrict__ buf, size_t n, float v) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) buf[i] = v + (float)i;
}

static float* dev_random(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 1;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = ((int)(s >> 8) % 20001 - 10000) * 1e-4f * scale;
  }
  float* d;
  CHECK(hipMalloc(&d, n * 4));
  CHECK(hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice));
  return d;
}

// bitwise comparison of a buffer of 32-bit words; bad[0]: words that differ, bad[1 + lane]: by word index mod 64
__global__ void compare_words_kernel(const unsigned* __restrict__ got, const unsigned* __restrict__ want, size_t n,
                                     unsigned long long* __restrict__ bad) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    if (got[i] != want[i]) {
      atomicAdd(&bad[0], 1ull);
      atomicAdd(&bad[1 + (i & 63)], 1ull);
    }
}

// The forward chain x -> conv1 -> conv2 -> T-Net(64).conv1 -> T-Net(64).conv2 (conv_chain_kernel<3, true, .>, two
// workgroups per CU as well) on random inputs, every output compared bit for bit with a device-synchronised first run.
static int forward_chain_soak(int pairs, hipStream_t s) {
  const int B = 250, N = 1024;
  const size_t n64 = (N + 63) / 64;
  ConvChainArgs a{};
  a.x3 = dev_random((size_t)B * 3 * N, 1.f, 31);
  a.T3 = dev_random((size_t)B * 9, 1.f, 32);
  a.w1 = dev_random(64 * 3, 0.5f, 33);
  a.b1 = dev_random(64, 0.1f, 34);
  a.N = N; a.B = B; a.ns = 3;
  float *h2, *c2, *h2r, *c2r;
  unsigned long long *m[3], *mr[3];
  CHECK(hipMalloc(&h2, (size_t)B * 64 * N * 4));
  CHECK(hipMalloc(&c2, (size_t)B * 128 * N * 4));
  CHECK(hipMalloc(&h2r, (size_t)B * 64 * N * 4));
  CHECK(hipMalloc(&c2r, (size_t)B * 128 * N * 4));
  const size_t msz[3] = {(size_t)B * 64 * n64 * 8, (size_t)B * 64 * n64 * 8, (size_t)B * 128 * n64 * 8};
  for (int i = 0; i < 3; ++i) {
    CHECK(hipMalloc(&m[i], msz[i]));
    CHECK(hipMalloc(&mr[i], msz[i]));
  }
  a.st[0] = ChainStage{dev_random(64 * 64, 0.3f, 35), 0, dev_random(64, 0.1f, 36), h2, (long)64 * N, m[0], 64};
  a.st[1] = ChainStage{dev_random(64 * 64, 0.3f, 37), 0, dev_random(64, 0.1f, 38), nullptr, 0, m[1], 64};
  a.st[2] = ChainStage{dev_random(128 * 64, 0.3f, 39), 0, dev_random(128, 0.1f, 40), c2, (long)128 * N, m[2], 128};
  unsigned long long* bad;
  CHECK(hipMalloc(&bad, 5 * 65 * 8));
  CHECK(hipMemset(bad, 0, 5 * 65 * 8));
  if (launch_conv_chain(a, s) != GEOA3_OK) return 3;
  CHECK(hipDeviceSynchronize());
  CHECK(hipMemcpy(h2r, h2, (size_t)B * 64 * N * 4, hipMemcpyDeviceToDevice));
  CHECK(hipMemcpy(c2r, c2, (size_t)B * 128 * N * 4, hipMemcpyDeviceToDevice));
  for (int i = 0; i < 3; ++i) CHECK(hipMemcpy(mr[i], m[i], msz[i], hipMemcpyDeviceToDevice));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, s));
  for (int it = 0; it < pairs; ++it) {
    launch_conv_chain(a, s);
    hipLaunchKernelGGL(compare_words_kernel, dim3(1024), dim3(256), 0, s, (const unsigned*)h2, (const unsigned*)h2r, (size_t)B * 64 * N, bad);
    hipLaunchKernelGGL(compare_words_kernel, dim3(1024), dim3(256), 0, s, (const unsigned*)c2, (const unsigned*)c2r, (size_t)B * 128 * N, bad + 65);
    for (int i = 0; i < 3; ++i)
      hipLaunchKernelGGL(compare_words_kernel, dim3(256), dim3(256), 0, s, (const unsigned*)m[i], (const unsigned*)mr[i], msz[i] / 4, bad + 65 * (2 + i));
  }
  CHECK(hipEventRecord(e1, s));
  CHECK(hipStreamSynchronize(s));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[5 * 65];
  CHECK(hipMemcpy(h, bad, sizeof(h), hipMemcpyDeviceToHost));
  printf("forward chain (3 stages): %d launches, %.1f us each incl. the checks: wrong words h2 %llu, c2 %llu, gate masks %llu %llu %llu\n",
         pairs, ms * 1000.f / pairs, h[0], h[65], h[130], h[195], h[260]);
  if (h[0] + h[65]) {
    printf("        wrong h2 + c2 words by column mod 64:");
    for (int l = 0; l < 64; ++l) printf("%s%llu", l % 16 == 0 ? " | " : " ", h[1 + l] + h[66 + l]);
    printf("\n");
  }
  return 0;
}

#include <dlfcn.h>

int main(int argc, char** argv) {
  const int pairs = argc > 1 ? atoi(argv[1]) : 20000;
  const int mode = argc > 2 ? atoi(argv[2]) : 0;   // bit 0: a 256 MB streaming write in front of every producer; bit 1: the forward chain kernel as well
  const int pitch = argc > 3 ? atoi(argv[3]) : 32;  // floats per partial row in the library under test (probe builds: 9)
  const int B = 250, N = 1024, nparts = (N + 255) / 256;
  ConvBwdChainArgs a[2] = {};
  float* Xa = dev_random((size_t)B * 64 * N, 1.f, 1);
  float* Wa = dev_random((size_t)B * 64 * 64, 0.2f, 2);
  float* Xb = dev_random((size_t)B * 64 * N, 1.f, 3);
  float* Wb = dev_random(64 * 64, 0.2f, 4);
  float* W2t = dev_random(64 * 64, 0.2f, 5);
  float* T3 = dev_random((size_t)B * 9, 1.f, 6);
  float* w1 = dev_random(64 * 3, 0.5f, 7);
  float* b1 = dev_random(64, 0.1f, 8);
  float* mask = dev_random((size_t)B * (N / 64) * 64 * 2, 1.f, 9);   // random gate bits
  float *dx, *dTpart, *gT3, *ref[2], *big, *refpart[2], *refdx[2];
  unsigned long long* bad;
  CHECK(hipMalloc(&dx, (size_t)B * 3 * N * 4));
  CHECK(hipMalloc(&dTpart, (size_t)B * nparts * 32 * 4 + 4096));
  CHECK(hipMalloc(&gT3, (size_t)B * 9 * 4));
  CHECK(hipMalloc(&ref[0], (size_t)B * 9 * 4));
  CHECK(hipMalloc(&ref[1], (size_t)B * 9 * 4));
  CHECK(hipMalloc(&bad, 2048));
  for (int p = 0; p < 2; ++p) CHECK(hipMalloc(&refdx[p], (size_t)B * 3 * N * 4));
  for (int p = 0; p < 2; ++p) CHECK(hipMalloc(&refpart[p], (size_t)B * nparts * 32 * 4 + 4096));
  const size_t bign = (size_t)64 << 20;
  CHECK(hipMalloc(&big, bign * 4));
  for (int p = 0; p < 2; ++p) {
    a[p].Xa = Xa; a[p].Wa = Wa; a[p].sWa = 4096; a[p].Xb = Xb; a[p].Wb = Wb;
    a[p].Zmask = reinterpret_cast<const unsigned long long*>(mask);
    a[p].W2t = W2t; a[p].x3 = dev_random((size_t)B * 3 * N, 1.f, 20 + p); a[p].T3 = T3; a[p].w1 = w1; a[p].b1 = b1;
    a[p].dx = dx; a[p].dTpart = dTpart; a[p].N = N; a[p].B = B;
  }
  hipStream_t s;
  CHECK(hipStreamCreate(&s));
  typedef int (*setbuf_fn)(unsigned*);
  setbuf_fn set_buf = (setbuf_fn)dlsym(RTLD_DEFAULT, "geoa3_probe_set_buf");
  unsigned *stg = nullptr, *stg_ref[2] = {nullptr, nullptr};
  const size_t bn = (size_t)B * N;
  if (set_buf) {
    CHECK(hipMalloc(&stg, 5 * bn * 4));
    for (int p = 0; p < 2; ++p) CHECK(hipMalloc(&stg_ref[p], 5 * bn * 4));
    set_buf(stg);
  }
  // references: every launch fenced by a device synchronisation, twice (they must agree)
  for (int rep = 0; rep < 2; ++rep)
    for (int p = 0; p < 2; ++p) {
      CHECK(hipMemsetAsync(bad, 0, 2048, s));
      if (launch_conv_bwd_chain(a[p], s) != GEOA3_OK) return 3;
      CHECK(hipDeviceSynchronize());
      if (rep == 0 && stg) CHECK(hipMemcpy(stg_ref[p], stg, 5 * bn * 4, hipMemcpyDeviceToDevice));
      if (rep == 0) CHECK(hipMemcpy(refdx[p], dx, (size_t)B * 3 * N * 4, hipMemcpyDeviceToDevice));
      if (rep == 0) CHECK(hipMemcpy(refpart[p], dTpart, (size_t)B * nparts * 32 * 4, hipMemcpyDeviceToDevice));
      if (launch_reduce_dT(dTpart, nparts, rep == 0 ? ref[p] : gT3, B, s) != GEOA3_OK) return 3;
      CHECK(hipDeviceSynchronize());
      if (rep == 1) {
        hipLaunchKernelGGL(compare_kernel, dim3((B * 9 + 255) / 256), dim3(256), 0, s, gT3, ref[p], ref[1 - p], B * 9, bad, 0u);
        unsigned long long h[8];
        CHECK(hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost));
        if (h[0]) printf("synchronised reference runs disagree (%llu values)\n", h[0]);
      }
    }
  CHECK(hipMemsetAsync(bad, 0, 2048, s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, s));
  for (int it = 1; it <= pairs; ++it) {
    const int p = it & 1;
    if (mode & 1) hipLaunchKernelGGL(dirty_kernel, dim3(2048), dim3(256), 0, s, big, bign, (float)it);
    launch_conv_bwd_chain(a[p], s);
    launch_reduce_dT(dTpart, nparts, gT3, B, s);
    if (stg) hipLaunchKernelGGL(compare_stages_kernel, dim3(512), dim3(256), 0, s, stg, stg_ref[p], bn, bad);
    hipLaunchKernelGGL(compare_dx_kernel, dim3(512), dim3(256), 0, s, dx, refdx[p], (size_t)B * 3 * N, bad);
    hipLaunchKernelGGL(compare_rows_kernel, dim3((B * nparts + 255) / 256), dim3(256), 0, s, dTpart, refpart[p], refpart[1 - p],
                       B * nparts, pitch, bad);
    hipLaunchKernelGGL(compare_kernel, dim3((B * 9 + 255) / 256), dim3(256), 0, s, gT3, ref[p], ref[1 - p], B * 9, bad, (unsigned)it);
  }
  CHECK(hipEventRecord(e1, s));
  CHECK(hipStreamSynchronize(s));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[256];
  CHECK(hipMemcpy(h, bad, 2048, hipMemcpyDeviceToHost));
  printf("mode %d: %d pairs, %.1f us per pair: wrong dT3 values %llu (equal to the other input set's: %llu, other: %llu; %llu in the "
         "last 16 instances), last at pair %llu\n", mode, pairs, ms * 1000.f / pairs, h[0], h[1], h[2], h[5], h[3]);
  printf("        partial rows as the checker kernel reads them afterwards: %llu wrong (%llu equal to the previous launch's row = stale, %llu other); "
         "by workgroup of the instance: %llu %llu %llu %llu\n", h[8], h[9], h[10], h[12], h[13], h[14], h[15]);
  printf("        wrong floats by position in the row (of which equal to the previous launch's value):");
  for (int j = 0; j < 9; ++j) printf(" %llu(%llu)", h[16 + j], h[26 + j]);
  printf("\n        wrong dx values: %llu of %zu; relative error < 1e-6: %llu, < 1e-4: %llu, < 1e-2: %llu, larger: %llu\n", h[40],
         (size_t)pairs * B * 3 * N, h[41], h[42], h[43], h[44]);
  printf("        wrong dx by component x/y/z: %llu %llu %llu; by wave of the workgroup: %llu %llu %llu %llu; by lane:", h[128], h[129], h[130],
         h[132], h[133], h[134], h[135]);
  for (int l = 0; l < 64; ++l) printf("%s%llu", l % 16 == 0 ? " | " : " ", h[64 + l]);
  printf("\n");
  if (stg) {
    printf("        probe: columns whose stage checksum differs from the reference run (of which lanes 48-63):");
    const char* nm[5] = {"W2^T dh2 from the accumulators", "after W3eff^T Ga", "first-layer rows from LDS (broadcast ds_read_b128)", "p = T3^T x at its use", "q"};
    for (int st = 0; st < 5; ++st) printf("  %s %llu (%llu)", nm[st], h[48 + st], h[56 + st]);
    printf("\n");
  }
  typedef int (*probe_fn)(unsigned long long*);
  if (probe_fn geoa3_probe_read = (probe_fn)dlsym(RTLD_DEFAULT, "geoa3_probe_read")) {   // builds with -DGEOA3_HAZARD_PROBE=3 only
    unsigned long long pc[4] = {0, 0, 0, 0};
    geoa3_probe_read(pc);
    printf("        probe: gate words parked in SGPRs that differ from a fresh v_readlane at the point of use: %llu (bits: lo %08llx hi %08llx)\n",
           pc[0], pc[1] & 0xffffffffull, pc[1] >> 32);
  }
  if (mode & 2) return forward_chain_soak(pairs, s);
  return 0;
}
