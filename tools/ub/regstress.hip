// NOTEBOOK 5a, hypothesis test: do two wavefronts of a 200+ VGPR kernel on one SIMD disturb each other's registers?
//   hipcc --offload-arch=gfx950 -O3 -o regstress regstress.hip ;  ./regstress [launches]
// Every lane keeps R registers alive through a chain of integer updates (optionally with MFMAs and v_permlane32_swap between
// the rounds, the instruction classes conv_bwd_chain_kernel mixes), then runs the inverse updates in place
// and counts the lanes / registers that differ.  Two workgroups per CU (LDS), amdgpu_waves_per_eu(2, 2): two waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// (inline assembly, one register at a time: left to the scheduler the 200 chains are interleaved with temporaries and spill)
__device__ __forceinline__ void step(unsigned& v, unsigned A, unsigned C) {
  asm volatile("v_mul_lo_u32 %0, %0, %1\n\tv_add_u32 %0, %0, %2" : "+v"(v) : "s"(A), "s"(C));
}
__device__ __forceinline__ void unstep(unsigned& v, unsigned Ai, unsigned C) {
  asm volatile("v_sub_u32 %0, %0, %2\n\tv_mul_lo_u32 %0, %0, %1" : "+v"(v) : "s"(Ai), "s"(C));
}

template <int R, int MODE>   // MODE bit 0: MFMAs between the rounds, bit 1: v_permlane32_swap on register pairs (twice = identity),
                             // bit 2: NS values read with v_readlane_b32 into SGPRs before the rounds, kept there, checked after
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void regstress(unsigned long long* bad, int rounds,
                                                                                             float* sink) {
  extern __shared__ float lds[];
  const unsigned tid = blockIdx.x * 256 + threadIdx.x;
  unsigned v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = tid * 2654435761u + r * 40503u;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  constexpr int NS = 76;
  unsigned sg[NS];
  const unsigned wbase = (unsigned)__builtin_amdgcn_readfirstlane((int)((tid & ~63u) * 2246822519u));
  if (MODE & 4) {
    const unsigned lane0 = threadIdx.x & 63;
    const unsigned w0 = (wbase + lane0 * 2654435761u) | 0x80008000u, w1 = (wbase ^ (lane0 * 40503u + 77u)) | 0x80008000u;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      sg[i] = (unsigned)__builtin_amdgcn_readlane((int)(i < 64 ? w0 : w1), i & 63);
      asm volatile("" : "+s"(sg[i]));     // in an SGPR from here on
    }
  }
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int r = 0; r < R; ++r) step(v[r], 1664525u, 1013904223u);
    if (MODE & 2) {
#pragma unroll
      for (int r = 0; r + 1 < R; r += 2) {
        auto p = __builtin_amdgcn_permlane32_swap(v[r], v[r + 1], false, false);
        auto q = __builtin_amdgcn_permlane32_swap(p[0], p[1], false, false);
        v[r] = q[0];
        v[r + 1] = q[1];
      }
    }
    if (MODE & 1) {
      half8 a, b;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a[j] = (_Float16)(float)((v[j] >> 20) & 7);
        b[j] = (_Float16)(float)((v[8 + j] >> 20) & 7);
      }
#pragma unroll
      for (int m = 0; m < 6; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    if ((it & 7) == 0) lds[threadIdx.x] = acc[0];      // (keeps the accumulators alive)
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i];
  if (s == 12345.678f) sink[tid] = s;
  // verification in place: the inverse update `rounds` times brings every register back to its seed
  for (int it = 0; it < rounds; ++it) {
#pragma unroll
    for (int r = 0; r < R; ++r) unstep(v[r], 4276115653u, 1013904223u);
  }
  if (MODE & 4) {
    unsigned wb2 = wbase;
    asm volatile("" : "+s"(wb2));
    unsigned diff_or = 0, nbad = 0;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
      const unsigned l = i & 63;
      const unsigned e = i < 64 ? ((wb2 + l * 2654435761u) | 0x80008000u) : ((wb2 ^ (l * 40503u + 77u)) | 0x80008000u);
      asm volatile("" : "+s"(sg[i]));
      const unsigned d = sg[i] ^ e;
      diff_or |= d;
      nbad += d != 0 ? 1 : 0;
    }
    if (nbad && (threadIdx.x & 63) == 0) {
      atomicAdd(&bad[9], (unsigned long long)nbad);
      atomicOr(&bad[10], (unsigned long long)diff_or);
    }
  }
  const unsigned lane = threadIdx.x & 63;
  unsigned tid2 = tid;
  asm volatile("" : "+v"(tid2));     // (recompute the seeds here: kept from the start they would double the live registers)
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (v[r] != tid2 * 2654435761u + r * 40503u) {
      atomicAdd(&bad[0], 1ull);
      atomicAdd(&bad[1 + (lane >> 4)], 1ull);
      atomicMax(&bad[8], (unsigned long long)r);
    }
}

// The instruction pattern of conv_bwd_chain_kernel's 3x3 transform as the compiler emitted it: an SGPR PAIR assembled with
// s_mov_b32 right in front of a packed-FP32 instruction that reads it, one half overwritten right behind it.
typedef float float2v __attribute__((ext_vector_type(2)));
template <int VARIANT>   // 0: s_mov, s_mov, v_pk_mul, s_mov (as compiled); 1: the same with s_nop 3 in front of the packed op;
                         // 2: no trailing overwrite
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void pkstress(unsigned long long* bad, int rounds,
                                                                                            const float* __restrict__ tab) {
  extern __shared__ float lds[];
  const unsigned lane = threadIdx.x & 63;
  float2v x = {1.0f + 0.001f * (float)(threadIdx.x & 255), 2.0f + 0.003f * (float)(threadIdx.x & 127)};
  float busy[24];
#pragma unroll
  for (int i = 0; i < 24; ++i) busy[i] = 1.0f + (float)i + (float)lane;
  for (int it = 0; it < rounds; ++it) {
    const float a = tab[(blockIdx.x * 7 + it) & 1023], b = tab[(blockIdx.x * 13 + it * 3 + 5) & 1023], c = tab[(it * 11 + 9) & 1023];
    float2v r;
    if (VARIANT == 0)
      asm volatile("s_mov_b32 s40, %2\n\ts_mov_b32 s41, %3\n\tv_pk_mul_f32 %0, %1, s[40:41]\n\ts_mov_b32 s41, %4\n\ts_mov_b32 s40, %4"
                   : "=v"(r) : "v"(x), "s"(a), "s"(b), "s"(c) : "s40", "s41");
    else if (VARIANT == 1)
      asm volatile("s_mov_b32 s40, %2\n\ts_mov_b32 s41, %3\n\ts_nop 3\n\tv_pk_mul_f32 %0, %1, s[40:41]\n\ts_nop 3\n\ts_mov_b32 s41, %4\n\ts_mov_b32 s40, %4"
                   : "=v"(r) : "v"(x), "s"(a), "s"(b), "s"(c) : "s40", "s41");
    else
      asm volatile("s_mov_b32 s40, %2\n\ts_mov_b32 s41, %3\n\tv_pk_mul_f32 %0, %1, s[40:41]"
                   : "=v"(r) : "v"(x), "s"(a), "s"(b), "s"(c) : "s40", "s41");
    const float e0 = x[0] * a, e1 = x[1] * b;
    if (__float_as_uint(r[0]) != __float_as_uint(e0) || __float_as_uint(r[1]) != __float_as_uint(e1)) {
      atomicAdd(&bad[0], 1ull);
      atomicAdd(&bad[1 + (lane >> 4)], 1ull);
      if (__float_as_uint(r[0]) == __float_as_uint(x[0] * c) || __float_as_uint(r[1]) == __float_as_uint(x[1] * c)) atomicAdd(&bad[9], 1ull);
    }
#pragma unroll
    for (int i = 0; i < 24; ++i) busy[i] = busy[i] * 1.0001f + r[i & 1];     // VALU traffic between the probes
    x[0] += 0.25f;
    x[1] -= 0.125f;
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 24; ++i) s += busy[i];
  if (s == 12345.678f) lds[threadIdx.x] = s;
}

template <int VARIANT>
static void run_pk(const char* name, int launches, unsigned long long* bad, const float* tab) {
  CHECK(hipMemset(bad, 0, 128));
  const size_t lds = 56 * 1024;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(pkstress<VARIANT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((pkstress<VARIANT>), dim3(1000), dim3(256), lds, 0, bad, 200, tab);
  CHECK(hipDeviceSynchronize());
  unsigned long long h[16];
  CHECK(hipMemcpy(h, bad, 128, hipMemcpyDeviceToHost));
  printf("%-58s %d launches x 256000 lanes x 200 probes: wrong products %llu (lanes 0-15: %llu, 16-31: %llu, 32-47: %llu, 48-63: %llu; "
         "equal to the product with the OVERWRITING value: %llu)\n", name, launches, h[0], h[1], h[2], h[3], h[4], h[9]);
  fflush(stdout);
}

template <int R, int MODE>
static void run(const char* name, int launches, unsigned long long* bad, float* sink) {
  CHECK(hipMemset(bad, 0, 128));
  const size_t lds = 56 * 1024;    // two workgroups per CU
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(regstress<R, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((regstress<R, MODE>), dim3(1000), dim3(256), lds, 0, bad, 24, sink);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipDeviceSynchronize());
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long h[16];
  CHECK(hipMemcpy(h, bad, 128, hipMemcpyDeviceToHost));
  printf("%-58s %d launches x 1000 workgroups, %.1f us each: wrong registers %llu (lanes 0-15: %llu, 16-31: %llu, 32-47: %llu, 48-63: %llu); "
         "wrong SGPRs %llu (bits that differed: %08llx)\n",
         name, launches, ms * 1000.f / launches, h[0], h[1], h[2], h[3], h[4], h[9], h[10]);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 20000;
  unsigned long long* bad;
  float* sink;
  CHECK(hipMalloc(&bad, 128));
  CHECK(hipMalloc(&sink, 1000 * 256 * 4));
  {
    float htab[1024];
    for (int i = 0; i < 1024; ++i) htab[i] = 0.5f + 0.001f * (float)((i * 7919) % 1000);
    float* tab;
    CHECK(hipMalloc(&tab, sizeof(htab)));
    CHECK(hipMemcpy(tab, htab, sizeof(htab), hipMemcpyHostToDevice));
    const int pl = launches / 10 > 0 ? launches / 10 : 1;
    run_pk<0>("s_mov pair -> v_pk_mul_f32 s[40:41] -> s_mov (as compiled)", pl, bad, tab);
    run_pk<1>("the same with s_nop 3 around the packed instruction", pl, bad, tab);
    run_pk<2>("without the trailing overwrite", pl, bad, tab);
  }
  run<200, 0>("200 live registers, integer updates only", launches, bad, sink);
  run<184, 1>("184 registers + MFMA 32x32x16 f16 between the rounds", launches, bad, sink);
  run<200, 2>("200 registers + v_permlane32_swap pairs", launches, bad, sink);
  run<184, 3>("184 registers + MFMA + v_permlane32_swap", launches, bad, sink);
  run<184, 7>("184 registers + MFMA + swap + 76 SGPRs from v_readlane", launches, bad, sink);
  run<120, 5>("120 registers + MFMA + 76 SGPRs from v_readlane", launches, bad, sink);
  return 0;
}
