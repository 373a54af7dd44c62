// Stand-alone reproducer for NOTEBOOK 5a: is a small store made by the LAST instructions of a kernel's workgroups visible
// to the FIRST instructions of the next kernel of the same stream, when several workgroups (of different XCDs) write into
// one cache line?
//
//   hipcc --offload-arch=gfx950 -O3 -o xcd_visibility xcd_visibility.hip
//   ./xcd_visibility [launch pairs, default 100000] [workgroups, default 1000]
//
// producer<<<G workgroups>>>: optional streaming work (so that the XCD's L2 holds dirty lines of its own), then nine
//   threads store value(iteration, workgroup, j) into part[workgroup * pitch + j] (pitch 9 floats = 36 bytes, three and a
//   half writers per 128-byte line -- the layout of the loop's dTpart in round 2 -- or 32 floats = a line per writer).
// consumer<<<G/4 workgroups>>>: its first instructions read the 4 x 9 values of "its" instance and compare them with
//   value(iteration, ...).  A mismatch that equals value(iteration - 1, ...) is a STALE read.
// Variants: release (agent-scope fence behind the store), acquire (buffer_inv sc1 before the read), sc1 (agent-scope
// atomic store / load), a second stream kept busy with an unrelated streaming kernel (the loop has a geometry stream).
// Nothing synchronises with the host inside the loop; counts are read at the end.  Output: one line per variant.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) {                                                           \
      fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                        \
    }                                                                                 \
  } while (0)

enum { F_PAD = 1, F_REL = 2, F_ACQ = 4, F_SC1 = 8, F_WORK = 16, F_STREAM2 = 32, F_WAIT = 64, F_DIRTY = 128, F_SMALLC = 256, F_DELAY = 512 };

__device__ __forceinline__ float value(unsigned it, unsigned wg, unsigned j) {
  unsigned h = it * 2654435761u ^ (wg * 40503u + j * 9973u + 12345u);
  h ^= h >> 15;
  h *= 2246822519u;
  h ^= h >> 13;
  return __uint_as_float(0x3f800000u | (h & 0x007fffffu));   // [1, 2): never NaN, the bit pattern is the check
}

template <int FLAGS>
__global__ __launch_bounds__(256) void producer(float* __restrict__ part, int pitch, unsigned it, const float* __restrict__ src,
                                                float* __restrict__ sink, int work) {
  const unsigned wg = blockIdx.x, t = threadIdx.x;
  if (FLAGS & F_WORK) {   // streaming read-modify-write of this workgroup's slab: dirty lines in this XCD's L2
    float acc = 0.f;
    const size_t base = (size_t)wg * work * 256;
    for (int i = 0; i < work; ++i) acc += src[base + (size_t)i * 256 + t];
    if (FLAGS & F_DIRTY)   // ... and as many dirty lines: 64 KB per workgroup (64 MB per launch at 1000 workgroups)
      for (int i = 0; i < work; ++i) sink[base + (size_t)i * 256 + t] = acc + (float)(it + i);
    else
      sink[base + t] = acc + (float)it;
  }
  if (t < 9) {
    float* p = part + (size_t)wg * pitch + t;
    const float v = value(it, wg, t);
    if (FLAGS & F_SC1)
      __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      *p = v;
    if (FLAGS & F_WAIT) __builtin_amdgcn_s_waitcnt(0);
    if (FLAGS & F_REL) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  }
}

template <int FLAGS>
__global__ __launch_bounds__(64) void consumer(const float* __restrict__ part, int pitch, unsigned it,
                                               unsigned long long* __restrict__ bad, float* __restrict__ sum, unsigned ninst) {
  unsigned inst = blockIdx.x, t = threadIdx.x;
  if (FLAGS & F_SMALLC) {   // the loop's reducer: a few workgroups, a thread per output element
    const unsigned e = blockIdx.x * 64 + t;
    inst = e / 36;
    t = e % 36;
    if (inst >= ninst) return;
  }
  if (FLAGS & F_DELAY)
    for (int w = 0; w < 64; ++w) __builtin_amdgcn_s_sleep(127);
  if (FLAGS & F_ACQ) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (t < 36) {
    const unsigned wg = inst * 4 + t / 9, j = t % 9;
    const float* p = part + (size_t)wg * pitch + j;
    float v;
    if (FLAGS & F_SC1)
      v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else
      v = *p;
    const float want = value(it, wg, j);
    if (__float_as_uint(v) != __float_as_uint(want)) {
      atomicAdd(&bad[0], 1ull);
      if (__float_as_uint(v) == __float_as_uint(value(it - 1, wg, j)))
        atomicAdd(&bad[1], 1ull);
      else
        atomicAdd(&bad[2], 1ull);
      atomicMax(&bad[3], (unsigned long long)it);
      if (bad[4] == 0) atomicCAS(&bad[4], 0ull, ((unsigned long long)it << 32) | (wg << 8) | j);
    }
    if (sum) sum[inst * 64 + t] = v;   // the reducer of the loop writes what it read
  }
}

__global__ __launch_bounds__(256) void bystander(float* __restrict__ buf, size_t n, float a) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) buf[i] = buf[i] * a + 1.f;
}

template <int FLAGS>
static void run(const char* name, int pairs, int G, float* part, const float* src, float* sink, float* sum,
                unsigned long long* bad, hipStream_t s, hipStream_t s2, float* by, size_t byn) {
  const int pitch = (FLAGS & F_PAD) ? 32 : 9;
  const int work = 64;
  CHECK(hipMemsetAsync(bad, 0, 64, s));
  CHECK(hipMemsetAsync(part, 0, (size_t)G * 32 * 4, s));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, s));
  for (int it = 1; it <= pairs; ++it) {
    if ((FLAGS & F_STREAM2) && (it % 8) == 1) hipLaunchKernelGGL(bystander, dim3(512), dim3(256), 0, s2, by, byn, 0.5f);
    hipLaunchKernelGGL(producer<FLAGS>, dim3(G), dim3(256), 0, s, part, pitch, (unsigned)it, src, sink, work);
    hipLaunchKernelGGL(consumer<FLAGS>, dim3((FLAGS & F_SMALLC) ? (G / 4 * 36 + 63) / 64 : G / 4), dim3(64), 0, s, part, pitch,
                       (unsigned)it, bad, sum, (unsigned)(G / 4));
  }
  CHECK(hipEventRecord(e1, s));
  CHECK(hipStreamSynchronize(s));
  if (FLAGS & F_STREAM2) CHECK(hipStreamSynchronize(s2));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[8];
  CHECK(hipMemcpy(h, bad, 64, hipMemcpyDeviceToHost));
  printf("%-58s pairs %d  wrong values %llu (stale by one launch %llu, other %llu)  last at pair %llu  first (pair %llu, wg %llu, j %llu)  %.2f us/pair\n",
         name, pairs, h[0], h[1], h[2], h[3], h[4] >> 32, (h[4] >> 8) & 0xffffff, h[4] & 0xff, ms * 1000.f / pairs);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int pairs = argc > 1 ? atoi(argv[1]) : 100000;
  const int G = argc > 2 ? atoi(argv[2]) : 1000;
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s (%s), %d CUs; %d producer workgroups, %d launch pairs per variant\n", prop.name, prop.gcnArchName,
         prop.multiProcessorCount, G, pairs);
  float *part, *src, *sink, *sum, *by;
  unsigned long long* bad;
  const size_t slab = (size_t)G * 64 * 256;
  const size_t byn = (size_t)64 << 20;
  CHECK(hipMalloc(&part, (size_t)G * 32 * 4));
  CHECK(hipMalloc(&src, slab * 4));
  CHECK(hipMalloc(&sink, slab * 4));
  CHECK(hipMalloc(&sum, (size_t)G / 4 * 64 * 4));
  CHECK(hipMalloc(&by, byn * 4));
  CHECK(hipMalloc(&bad, 64));
  CHECK(hipMemset(src, 0, slab * 4));
  CHECK(hipMemset(by, 0, byn * 4));
  hipStream_t s, s2;
  CHECK(hipStreamCreate(&s));
  CHECK(hipStreamCreate(&s2));
#define RUN(F, NAME) run<F>(NAME, pairs, G, part, src, sink, sum, bad, s, s2, by, byn)
  RUN(0, "36-byte rows, plain stores, bare producer");
  RUN(F_WORK, "36-byte rows, plain stores");
  RUN(F_WORK | F_STREAM2, "36-byte rows, plain stores, second stream busy");
  RUN(F_WORK | F_STREAM2 | F_WAIT, "36-byte rows, s_waitcnt behind the store, 2nd stream");
  RUN(F_WORK | F_STREAM2 | F_ACQ, "36-byte rows, acquire in the consumer, 2nd stream");
  RUN(F_WORK | F_STREAM2 | F_PAD, "128-byte rows, plain stores, 2nd stream");
  RUN(F_WORK | F_STREAM2 | F_PAD | F_SC1, "128-byte rows, sc1 atomic store + load, 2nd stream");
  RUN(F_WORK | F_STREAM2 | F_SC1, "36-byte rows, sc1 atomic store + load, 2nd stream");
  RUN(F_WORK | F_STREAM2 | F_REL, "36-byte rows, release behind the store, 2nd stream");
  RUN(F_WORK | F_STREAM2 | F_PAD | F_REL, "128-byte rows, release behind the store, 2nd stream");
  RUN(F_WORK | F_STREAM2 | F_REL | F_ACQ, "36-byte rows, release + acquire, 2nd stream");
  // the producer leaves 64 MB of dirty lines per launch in the L2s (the loop's backward kernels write 65-400 MB each)
  RUN(F_WORK | F_DIRTY, "36-byte rows, 64 MB dirtied per launch");
  RUN(F_WORK | F_DIRTY | F_SMALLC, "36-byte rows, 64 MB dirtied, small reducer");
  RUN(F_WORK | F_DIRTY | F_SMALLC | F_DELAY, "36-byte rows, 64 MB dirtied, small reducer sleeps 4 us");
  RUN(F_WORK | F_DIRTY | F_SMALLC | F_PAD, "128-byte rows, 64 MB dirtied, small reducer");
  RUN(F_WORK | F_DIRTY | F_SMALLC | F_REL, "36-byte rows, 64 MB dirtied, small reducer, release");
  RUN(F_WORK | F_DIRTY | F_SMALLC | F_ACQ, "36-byte rows, 64 MB dirtied, small reducer, acquire");
  return 0;
}
