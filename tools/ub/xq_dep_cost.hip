// What one cross-queue event dependency costs the MAIN queue (NOTEBOOK 8, row 37): a chain of dependent kernels on one stream,
// (a) as it is, (b) with a tiny kernel forked to a second stream behind every link and joined in front of the next one
// (record main -> wait side -> side kernel -> record side -> wait main: the pattern of a side queue inside one call),
// (c) with the tiny kernel in stream order instead.  (b) - (a) per link = two event records + two cross-queue waits as the main
// queue sees them; (c) - (a) = what the fork was meant to hide.  Links of 5 / 50 / 400 us (busy loops on all CUs).
//   hipcc --offload-arch=gfx950 -O3 -o xq_dep_cost xq_dep_cost.hip && ./xq_dep_cost
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void spin_kernel(float* out, int iters) {
  float a = threadIdx.x * 1e-3f;
  for (int i = 0; i < iters; ++i) a = __builtin_fmaf(a, 1.0001f, 0.5f);
  if (a == 12345.f) out[blockIdx.x] = a;
}

#define CK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e__)); return 1; } } while (0)

int main() {
  float* d;
  CK(hipMalloc(&d, 1 << 20));
  hipStream_t sm, ss;
  CK(hipStreamCreateWithFlags(&sm, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&ss, hipStreamNonBlocking));
  const int L = 200;
  hipEvent_t t0e, t1e;
  CK(hipEventCreate(&t0e));
  CK(hipEventCreate(&t1e));
  hipEvent_t ef[L], ej[L];
  for (int i = 0; i < L; ++i) {
    CK(hipEventCreateWithFlags(&ef[i], hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ej[i], hipEventDisableTiming));
  }
  // iterations of the busy loop for ~5 / 50 / 400 us links (4 cycles per dependent fma, measured: 14 ns per iteration)
  const int link_iters[3] = {360, 3600, 29000};
  const char* link_name[3] = {"~5 us", "~50 us", "~400 us"};
  for (int li = 0; li < 3; ++li) {
    double ms[3] = {0, 0, 0};
    for (int mode = 0; mode < 3; ++mode) {
      for (int rep = 0; rep < 4; ++rep) {   // first repetition: warm-up
        CK(hipDeviceSynchronize());
        // a ~30 ms blocker in front: the host enqueues the whole chain while it runs, the events time the GPU alone
        hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, sm, d, 2000000);
        CK(hipEventRecord(t0e, sm));
        for (int i = 0; i < L; ++i) {
          hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(256), 0, sm, d, link_iters[li]);
          if (mode == 1) {
            CK(hipEventRecord(ef[i], sm));
            CK(hipStreamWaitEvent(ss, ef[i], 0));
            hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(256), 0, ss, d + 4096, 210);   // ~3 us beside the next link
            CK(hipEventRecord(ej[i], ss));
          } else if (mode == 2) {
            hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(256), 0, sm, d + 4096, 210);
          }
          if (mode == 1 && i > 0) CK(hipStreamWaitEvent(sm, ej[i - 1], 0));   // joined one link later: the side kernel had a whole link to finish
        }
        CK(hipEventRecord(t1e, sm));
        CK(hipDeviceSynchronize());
        float t = 0.f;
        CK(hipEventElapsedTime(&t, t0e, t1e));
        if (rep > 0) ms[mode] += t / 3;
      }
    }
    printf("links of %-7s: chain alone %.1f us per link; + fork / join of a 3-us kernel %.1f (%+.1f); the 3-us kernel in stream order "
           "%.1f (%+.1f)\n", link_name[li], 1e3 * ms[0] / L, 1e3 * ms[1] / L, 1e3 * (ms[1] - ms[0]) / L, 1e3 * ms[2] / L,
           1e3 * (ms[2] - ms[0]) / L);
  }
  return 0;
}
