// The matrix-core filter search as a kernel of its own: clouds of up to 1024 points (one workgroup of 512 threads per
// (instance, direction): no grid at all) and of more than 4096 (slices of 1024 queries).  See geom_filter.h.
#include "geom_filter.h"

namespace {

constexpr int NF_T = 512;

__global__ __launch_bounds__(NF_T, 2) void nn1_filter_kernel(const float* __restrict__ A, const float* __restrict__ R, int Na, int Nr,
                                                             const int32_t* prior_ar, const int32_t* prior_ra, float* d_ar,
                                                             int32_t* i_ar, float* d_ra, int32_t* i_ra) {
  extern __shared__ __attribute__((aligned(16))) unsigned char f_smem[];
  const int b = blockIdx.x;
  const bool swap = blockIdx.y != 0;
  const int q0 = blockIdx.z * nf::Cfg<NF_T>::SLICE;
  const int32_t* prior = swap ? prior_ra : prior_ar;
  const int Nq = swap ? Nr : Na, M = swap ? Na : Nr;
  if (q0 >= Nq) return;
  nf::search<NF_T>((swap ? A : R) + (size_t)b * 3 * M, (swap ? R : A) + (size_t)b * 3 * Nq, M, Nq, q0,
                   prior ? prior + (size_t)b * Nq : nullptr, (swap ? d_ra : d_ar) + (size_t)b * Nq,
                   (swap ? i_ra : i_ar) + (size_t)b * Nq, f_smem);
}

}  // namespace

// Every query of every instance (clouds of any size).
int geoa3_launch_nn1_filter(const float* a, const float* r, int B, int Na, int Nr, const int32_t* prior_ar,
                            const int32_t* prior_ra, float* d_ar, int32_t* i_ar, float* d_ra, int32_t* i_ra, hipStream_t s) {
  const int M = Na > Nr ? Na : Nr;
  constexpr size_t lds = nf::Cfg<NF_T>::LDS;
  constexpr int slice = nf::Cfg<NF_T>::SLICE;
  static_assert(lds <= 80 * 1024, "two workgroups per CU");
  // (once per device: the call costs host time in a loop that is enqueue-bound at small batches; nothing a result depends on)
  static bool attr_set[64] = {};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nn1_filter_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  dim3 grid(B, d_ra ? 2 : 1, (M + slice - 1) / slice);
  hipLaunchKernelGGL(nn1_filter_kernel, grid, dim3(NF_T), lds, s, a, r, Na, Nr, prior_ar, prior_ra, d_ar, i_ar, d_ra, i_ra);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
