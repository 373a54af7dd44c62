// Shared helpers for the gfx950 kernels of libgeoa3_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/geoa3_hip.h"

#define GEOA3_WAVE 64

#define GEOA3_CHECK_LAUNCH()                         \
  do {                                               \
    hipError_t e__ = hipGetLastError();              \
    if (e__ != hipSuccess) return GEOA3_ELAUNCH;     \
  } while (0)

static inline hipStream_t geoa3_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Squared distance, un-fused: fl(fl(fl(dx*dx)+fl(dy*dy))+fl(dz*dz)) -- the evaluation order the
// CPU oracle (and torch's ((a-b)**2).sum(1)) uses, so indices can be compared bit-for-bit.
__device__ __forceinline__ float geoa3_sqdist(float ax, float ay, float az, float bx, float by, float bz) {
#pragma clang fp contract(off)
  float dx = ax - bx, dy = ay - by, dz = az - bz;
  float xx = dx * dx, yy = dy * dy, zz = dz * dz;
  float s = xx + yy;
  return s + zz;
}

// wave-wide (64 lanes) reductions via DPP-lowered shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
