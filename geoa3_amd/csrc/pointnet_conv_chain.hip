// Chains of the 64-input shared-MLP layers of the PointNet trunk in ONE kernel (split-fp16 arithmetic of
// pointnet_conv_split.hip), Model/PointNet.py:139-145:
//     [x -> conv1 ->] conv2 -> T-Net(64).conv1 -> T-Net(64).conv2        (h2 written, c1 only as gate bits, c2 written)
//     h2 -> conv3 (feature transform folded into its weights) -> conv4   (h3 only as gate bits, h4 written)
//
// Why: the layers are bandwidth-bound (a 64 -> 64 layer at 250 instances reads and writes 65 MB each in ~36 us) and the
// backward needs the middle activations only as relu gate BITS, so in a chain they never leave the registers: a lane
// owns one point, after a layer's epilogue it holds that point's 64 outputs -- exactly the form the next layer's K loop
// reads its input rows in.  Every stage performs the operations of conv_cm64s_kernel on the same values in the same
// order (weights scaled per 64-row block from the block's maximum, the running per-wave activation scale, three f16
// products per 16 k, fp32 sums), so a chain produces the bits of the layer-by-layer launches.
#include "pointnet_kernels.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void cc_swap32(float& a, float& b) {   // a[32..63] <-> b[0..31]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ float cc_first_layer(const float4 w, float p0, float p1, float p2) {
  return w.x * p0 + w.y * p1 + w.z * p2 + w.w;   // the expression of pointnet_gemm.hip first_layer (same bits)
}
__device__ __forceinline__ unsigned cc_exp(float m) {   // as cs_exp (pointnet_conv_split.hip)
  const unsigned E = (__float_as_uint(m) >> 23) & 0xffu;
  return E < 14u ? 14u : (E > 254u ? 254u : E);
}
__device__ __forceinline__ float cc_scale(unsigned E) { return __uint_as_float((267u - E) << 23); }
__device__ __forceinline__ float cc_unscale(unsigned E) { return __uint_as_float((E - 13u) << 23); }

// d[lane l] = v (wave-uniform value, constant lane): one instruction instead of compare + select
__device__ __forceinline__ void cc_writelane(int& d, int v, int l) {
  asm("v_writelane_b32 %0, %1, %2" : "+v"(d) : "s"(v), "n"(l));
}

constexpr int CC_PITCH = 64 * 2 + 16;            // bytes per weight row and piece (conflict-free ds_read_b128)
constexpr int CC_BLK = 2 * 64 * CC_PITCH;        // one 64 x 64 weight block: hi image, lo image

// NS stages (the last with COL = 64 or 128 output channels, the others 64); FIRST: the input rows are relu(w1 (T^T x) +
// b1) computed from the 3-channel cloud.  One wavefront = 64 points, four per workgroup; grid (column blocks, instances).
template <int NS, bool FIRST, int COL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_chain_kernel(ConvChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cc_smem[];
  constexpr int NBLK = NS - 1 + COL / 64;        // 64-row weight blocks of all stages
  float4* s_w1 = reinterpret_cast<float4*>(cc_smem + NBLK * CC_BLK);   // [64] (w1 row, b1)
  float* s_red = reinterpret_cast<float*>(s_w1 + 64);                  // [NBLK][4 waves]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cblk = blockIdx.x, b = blockIdx.y;
  const int col = cblk * 256 + wave * 64 + lane;
  const bool live = col < a.N, wave_live = cblk * 256 + wave * 64 < a.N;
  const unsigned long long livemask = __builtin_amdgcn_ballot_w64(live);
  const size_t mword = ((size_t)b * ((a.N + 63) >> 6) + (size_t)(cblk * 4 + wave));   // x Co: gate words [B][column block][row]

  float o[64];   // this lane's point: the 64 input rows of the current stage
  float p0 = 0.f, p1 = 0.f, p2 = 0.f;
  if (FIRST) {
    const float* xp = a.x3 + (size_t)b * 3 * a.N + (live ? col : a.N - 1);
    const float x0 = xp[0], x1 = xp[a.N], x2 = xp[2 * (size_t)a.N];
    p0 = x0;
    p1 = x1;
    p2 = x2;
    if (a.T3) {   // bmm(pc^T, T)^T : x'[c] = sum_d x[d] T[d][c]   (Model/PointNet.py:138)
      const float* t = a.T3 + (size_t)b * 9;
      p0 = x0 * t[0] + x1 * t[3] + x2 * t[6];
      p1 = x0 * t[1] + x1 * t[4] + x2 * t[7];
      p2 = x0 * t[2] + x1 * t[5] + x2 * t[8];
    }
    if (tid < 64) s_w1[tid] = make_float4(a.w1[3 * tid], a.w1[3 * tid + 1], a.w1[3 * tid + 2], a.b1[tid]);
  } else {   // all 64 rows in flight at once
    const float* X = a.X + (size_t)b * a.sXb;      // uniform row base + lane offset: scalar-base addressing
    const unsigned xc = (unsigned)(live ? col : a.N - 1);
#pragma unroll
    for (int u = 0; u < 64; ++u) o[u] = (X + (size_t)u * a.ldX)[xc];
  }

  // ---- weights of every stage: 64 x 64 values per block, 16 per thread (element e = tid + 256 i: row e / 64, k = e % 64),
  // the block's maximum, its power-of-two scale, the two fp16 images in LDS; per-row bias, one row per lane
  float biasv[NBLK];
  unsigned Ew[NBLK];
  {
    float wv[NBLK][16];
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      const int s = blk < NS - 1 ? blk : NS - 1, rb = blk - s;
      const float* W = a.st[s].W + (size_t)b * a.st[s].sWb + (size_t)rb * 4096;
      biasv[blk] = a.st[s].bias[rb * 64 + lane];
      float m = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        wv[blk][i] = W[tid + 256 * i];
        m = fmaxf(m, __builtin_fabsf(wv[blk][i]));
      }
      m = wave_max(m);
      if (lane == 0) s_red[blk * 4 + wave] = m;
    }
    __syncthreads();
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      Ew[blk] = cc_exp(fmaxf(fmaxf(s_red[blk * 4], s_red[blk * 4 + 1]), fmaxf(s_red[blk * 4 + 2], s_red[blk * 4 + 3])));
      const float sw = cc_scale(Ew[blk]);
      unsigned char* s_wh = cc_smem + blk * CC_BLK;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int e = tid + 256 * i, co = e >> 6, k = e & 63;
        const float v = wv[blk][i] * sw;
        const _Float16 h = (_Float16)v;
        *reinterpret_cast<_Float16*>(s_wh + co * CC_PITCH + k * 2) = h;
        *reinterpret_cast<_Float16*>(s_wh + 64 * CC_PITCH + co * CC_PITCH + k * 2) = (_Float16)(v - (float)h);
      }
    }
    __syncthreads();
  }
  if (FIRST) {
#pragma unroll
    for (int u = 0; u < 64; ++u) o[u] = fmaxf(cc_first_layer(s_w1[u], p0, p1, p2), 0.f);
  }

#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    const int s = blk < NS - 1 ? blk : NS - 1, rb = blk - s;
    const bool last = s == NS - 1;
    const ChainStage& st = a.st[s];
    const unsigned char* arow = cc_smem + blk * CC_BLK + (lane & 31) * CC_PITCH + (lane >> 5) * 16;   // A: row r, k = 8h + j

    f32x16 acc[2][2];   // [column block][row tile]
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;
    unsigned Ex = 14u;   // running exponent of the wave's activation scale
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float* x = o + 16 * c;
      float m = 0.f;
#pragma unroll
      for (int u = 0; u < 16; ++u) m = fmaxf(m, __builtin_fabsf(x[u]));
      const unsigned E = cc_exp(wave_max(live ? m : 0.f));   // = the maximum over the live lanes' |x| (0 for the others)
      if (E > Ex) {   // wave-uniform: shrink the scale, rescale the sums (exact)
        if (c > 0) {
          const unsigned d = E - Ex;
          const float f = d > 126u ? 0.f : __uint_as_float((127u - d) << 23);
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[cb][t][r] *= f;
        }
        Ex = E;
      }
      const float sx = cc_scale(Ex);
      half8 xh[2], xl[2];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v0 = x[j] * sx, v1 = x[8 + j] * sx;
        cc_swap32(v0, v1);    // v0: column block 0, v1: column block 1; lanes (column, k half)
        const _Float16 h0 = (_Float16)v0, h1 = (_Float16)v1;
        xh[0][j] = h0;
        xl[0][j] = (_Float16)(v0 - (float)h0);
        xh[1][j] = h1;
        xl[1][j] = (_Float16)(v1 - (float)h1);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const half8 wh = *reinterpret_cast<const half8*>(arow + t * 32 * CC_PITCH + c * 32);
        const half8 wl = *reinterpret_cast<const half8*>(arow + t * 32 * CC_PITCH + c * 32 + 64 * CC_PITCH);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[cb], acc[cb][t], 0, 0, 0);
          acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[cb], acc[cb][t], 0, 0, 0);
          acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[cb], acc[cb][t], 0, 0, 0);
        }
      }
    }
    const float unscale = cc_unscale(Ex) * cc_unscale(Ew[blk]);

    // epilogue: every pair of accumulator registers becomes two 64-column rows (row, row + 4); bias + relu; the row's
    // gate bits go to the stage's mask word (lane = row), the row itself into o[] -- the next stage's input, and what is
    // stored below if the stage has an output tensor
    float res[64];
    int mlo = 0, mhi = 0;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          v[i] = acc[0][t][4 * g + i];
          v[4 + i] = acc[1][t][4 * g + i];
          cc_swap32(v[i], v[4 + i]);    // v[i]: row base+i, v[4+i]: row base+4+i, lane = column
        }
        const int row0 = t * 32 + 8 * g;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float r = v[i] * unscale;
          r += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(biasv[blk]), row0 + i));
          r = fmaxf(r, 0.f);
          const unsigned long long mk = __builtin_amdgcn_ballot_w64(r > 0.f) & livemask;
          cc_writelane(mlo, (int)(unsigned)mk, row0 + i);
          cc_writelane(mhi, (int)(unsigned)(mk >> 32), row0 + i);
          res[row0 + i] = r;
        }
      }
    }
    if (wave_live)
      st.Ymask[mword * st.Co + rb * 64 + lane] = ((unsigned long long)(unsigned)mhi << 32) | (unsigned)mlo;
    if (st.Y && live) {   // one divergent region for the 64 row stores; uniform row base + lane offset
      float* Y = st.Y + (size_t)b * st.sYb + (size_t)rb * 64 * a.N;
#pragma unroll
      for (int r = 0; r < 64; ++r) (Y + (size_t)r * a.N)[(unsigned)col] = res[r];
    }
    if (!last) {
#pragma unroll
      for (int r = 0; r < 64; ++r) o[r] = res[r];
    }
  }
}

template <int NS, bool FIRST, int COL>
void launch_chain(const ConvChainArgs& a, hipStream_t s) {
  constexpr int NBLK = NS - 1 + COL / 64;
  const size_t lds = (size_t)NBLK * CC_BLK + 64 * 16 + NBLK * 4 * sizeof(float);
  auto kern = conv_chain_kernel<NS, FIRST, COL>;
  // (up to 74 KB of dynamic LDS)
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(kern, dim3((a.N + 255) / 256, a.B), dim3(256), lds, s, a);
}

}  // namespace

int launch_conv_chain(const ConvChainArgs& a, hipStream_t s) {
  if (a.ns < 2 || a.ns > 3 || a.B <= 0 || a.N <= 0) return GEOA3_EINVAL;
  for (int i = 0; i < a.ns; ++i) {
    if (!a.st[i].W || !a.st[i].bias || !a.st[i].Ymask) return GEOA3_EINVAL;
    if (a.st[i].Co != (i + 1 < a.ns ? 64 : 128)) return GEOA3_ENOSUPPORT;
  }
  const bool first = a.x3 != nullptr;
  if (first ? (!a.w1 || !a.b1) : !a.X) return GEOA3_EINVAL;
  if (a.ns == 3 && first) launch_chain<3, true, 128>(a, s);
  else if (a.ns == 2 && !first) launch_chain<2, false, 128>(a, s);
  else return GEOA3_ENOSUPPORT;
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
