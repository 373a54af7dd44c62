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

// ------------------------------------------------------------------------------------------
// Max over 64-bit keys with DPP row operations (~10 cycles per step instead of a ds_bpermute round trip): the
// arg-max reductions of the farthest-point samplers pack (value bits, complemented index) into one key.
// ------------------------------------------------------------------------------------------
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned long long dpp_max_u64(unsigned long long v) {
  const int lo = (int)(unsigned)v, hi = (int)(unsigned)(v >> 32);
  const unsigned olo = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, CTRL, ROWMASK, 0xF, false);
  const unsigned ohi = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWMASK, 0xF, false);
  const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
  return o > v ? o : v;
}
// max over the 16 lanes of every DPP row (all 16 lanes end with it)
__device__ __forceinline__ unsigned long long row16_max_u64(unsigned long long v) {
  v = dpp_max_u64<0xB1, 0xF>(v);    // quad_perm [1,0,3,2]
  v = dpp_max_u64<0x4E, 0xF>(v);    // quad_perm [2,3,0,1]
  v = dpp_max_u64<0x141, 0xF>(v);   // row_half_mirror
  v = dpp_max_u64<0x140, 0xF>(v);   // row_mirror
  return v;
}
// max over the wavefront, returned uniformly (read from lane 63)
__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v) {
  v = row16_max_u64(v);
  v = dpp_max_u64<0x142, 0xA>(v);   // row_bcast15 into rows 1 and 3
  v = dpp_max_u64<0x143, 0xC>(v);   // row_bcast31 into rows 2 and 3
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), 63);
  return ((unsigned long long)hi << 32) | lo;
}
