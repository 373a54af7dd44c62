// Shared helpers for the gfx950 kernels of libgeoa3_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/geoa3_hip.h"
#include "../../include/geoa3_hip_debug.h"

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

// The bits of a float, opaque to the optimiser: files compiled with -fno-honor-nans fold `(bits & 0x7fffffff) >= 0x7f800000`
// into `|x| == inf` (the NaN half of the class test dropped) -- what must SEE a NaN tests these bits.
__device__ __forceinline__ unsigned geoa3_opaque_bits(float x) {
  unsigned u = __float_as_uint(x);
  asm("" : "+v"(u));
  return u;
}
__device__ __forceinline__ bool geoa3_nonfinite(float x) { return (geoa3_opaque_bits(x) & 0x7fffffffu) >= 0x7f800000u; }

// wave-wide (64 lanes) reductions with DPP row operations (quad_perm, row_half_mirror, row_mirror, row_bcast15/31):
// ~10 cycles per step where __shfl_xor compiles to a ds_bpermute round trip through the LDS crossbar.  The result is
// read from lane 63 and returned uniformly.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ float dpp_f32(float v, float old) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROWMASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum(float v) {
  v += dpp_f32<0xB1, 0xF>(v, 0.f);    // quad_perm [1,0,3,2]
  v += dpp_f32<0x4E, 0xF>(v, 0.f);    // quad_perm [2,3,0,1]
  v += dpp_f32<0x141, 0xF>(v, 0.f);   // row_half_mirror
  v += dpp_f32<0x140, 0xF>(v, 0.f);   // row_mirror: every lane of a row holds the row sum
  v += dpp_f32<0x142, 0xA>(v, 0.f);   // row_bcast15: rows 1 and 3 add the sum of the row before them
  v += dpp_f32<0x143, 0xC>(v, 0.f);   // row_bcast31: rows 2 and 3 add the sum of rows 0-1
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, dpp_f32<0xB1, 0xF>(v, v));
  v = fmaxf(v, dpp_f32<0x4E, 0xF>(v, v));
  v = fmaxf(v, dpp_f32<0x141, 0xF>(v, v));
  v = fmaxf(v, dpp_f32<0x140, 0xF>(v, v));
  v = fmaxf(v, dpp_f32<0x142, 0xA>(v, v));
  v = fmaxf(v, dpp_f32<0x143, 0xC>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
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

// The same maximum as TWO 32-bit reductions -- the high words, then the low words of the lanes that hold the largest high
// word -- each step one `v_max_u32` with a DPP operand where the 64-bit form takes two moves, a 64-bit compare and two
// selects (the sampler's round is this reduction's dependent chain: 42 -> 14 instructions).
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_max_u32(unsigned v) {
  // (old = 0, the maximum's identity: lanes a row mask leaves out, or without a source lane, see 0 -- and the compiler
  // folds the move into `v_max_u32_dpp`)
  const unsigned o = (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROWMASK, 0xF, true);
  return o > v ? o : v;
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = dpp_max_u32<0xB1, 0xF>(v);
  v = dpp_max_u32<0x4E, 0xF>(v);
  v = dpp_max_u32<0x141, 0xF>(v);
  v = dpp_max_u32<0x140, 0xF>(v);
  v = dpp_max_u32<0x142, 0xA>(v);
  v = dpp_max_u32<0x143, 0xC>(v);
  return (unsigned)__builtin_amdgcn_readlane((int)v, 63);
}
__device__ __forceinline__ unsigned long long wave_max_u64_split(unsigned long long v) {
  const unsigned hi = (unsigned)(v >> 32), lo = (unsigned)v;
  const unsigned hmax = wave_max_u32(hi);
  const unsigned lmax = wave_max_u32(hi == hmax ? lo : 0u);
  return ((unsigned long long)hmax << 32) | lmax;
}

// ascending in-place sort of a short int list owned by ONE thread (reverse lists of the deterministic scatter-adds):
// insertion sort, heap sort beyond 24 entries (bounded work for degenerate inputs whose lists are long)
__device__ __forceinline__ void geoa3_sort_ints(int* L, int n) {
  if (n <= 24) {
    for (int a = 1; a < n; ++a) {
      const int v = L[a];
      int c = a - 1;
      while (c >= 0 && L[c] > v) {
        L[c + 1] = L[c];
        --c;
      }
      L[c + 1] = v;
    }
    return;
  }
  auto sift = [&](int start, int end) {
    int root = start;
    for (;;) {
      int child = 2 * root + 1;
      if (child > end) break;
      if (child + 1 <= end && L[child] < L[child + 1]) ++child;
      if (L[root] >= L[child]) break;
      const int tmp = L[root];
      L[root] = L[child];
      L[child] = tmp;
      root = child;
    }
  };
  for (int h0 = (n - 2) / 2; h0 >= 0; --h0) sift(h0, n - 1);
  for (int end = n - 1; end > 0; --end) {
    const int tmp = L[0];
    L[0] = L[end];
    L[end] = tmp;
    sift(0, end - 1);
  }
}
