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

// One 64 x 64 layer on the wave's 64 points: x[k] = input row k of this lane's point, arow = this lane's A-operand row of
// the layer's weight block in LDS; acc[column block][row tile] in the scaled domain, Ex = the exponent of the activation
// scale at the end (conv_cm64s_kernel's K loop: chunks of 16 rows, running per-wave power-of-two scale)
__device__ __forceinline__ void cc_mm(const float (&xin)[64], const unsigned char* arow, bool live, f32x16 (&acc)[2][2],
                                      unsigned& Ex) {
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;
  Ex = 14u;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float* x = xin + 16 * c;
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
}

// accumulators -> rows: out[r] = acc value of row r for this lane's point (lane = column), still in the scaled domain
__device__ __forceinline__ void cc_rows(const f32x16 (&acc)[2][2], float (&out)[64]) {
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v0 = acc[0][t][4 * g + i], v1 = acc[1][t][4 * g + i];
        cc_swap32(v0, v1);    // v0: row base + i, v1: row base + 4 + i
        out[t * 32 + 8 * g + i] = v0;
        out[t * 32 + 8 * g + 4 + i] = v1;
      }
}

// the 64 x 64 weight block W (element (co, k) at W[co * sco + k * sk], one of the strides 1) -> registers (16 values per
// thread, element e = tid + 256 i of the contiguous storage) and the wave's maximum |w| into s_red[wave]
__device__ __forceinline__ void cc_wload(const float* W, bool kcontig, int tid, float (&wv)[16], float* s_red) {
  float m = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    wv[i] = W[tid + 256 * i];
    m = fmaxf(m, __builtin_fabsf(wv[i]));
  }
  (void)kcontig;
  m = wave_max(m);
  if ((tid & 63) == 0) s_red[tid >> 6] = m;
}
// ... scaled by the block's power of two and split into the two fp16 images [64][CC_PITCH] at s_wh; returns the exponent
__device__ __forceinline__ unsigned cc_wsplit(const float (&wv)[16], bool kcontig, int tid, const float* s_red,
                                              unsigned char* s_wh) {
  const unsigned Ew = cc_exp(fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3])));
  const float sw = cc_scale(Ew);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = tid + 256 * i;
    const int co = kcontig ? e >> 6 : e & 63, k = kcontig ? e & 63 : e >> 6;   // transposed storage: co contiguous
    const float v = wv[i] * sw;
    const _Float16 h = (_Float16)v;
    *reinterpret_cast<_Float16*>(s_wh + co * CC_PITCH + k * 2) = h;
    *reinterpret_cast<_Float16*>(s_wh + 64 * CC_PITCH + co * CC_PITCH + k * 2) = (_Float16)(v - (float)h);
  }
  return Ew;
}

// NS stages (the last with COL = 64 or 128 output channels, the others 64); FIRST: the input rows are relu(w1 (T^T x) +
// b1) computed from the 3-channel cloud.  One wavefront = 64 points, four per workgroup; grid (column blocks, instances).
template <int NS, bool FIRST, int COL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_chain_kernel(ConvChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cc_smem[];
  constexpr int NBLK = NS - 1 + COL / 64;        // 64-row weight blocks of all stages
  float4* s_w1 = reinterpret_cast<float4*>(cc_smem + NBLK * CC_BLK);   // [64] (w1 row, b1)
  float* s_red = reinterpret_cast<float*>(s_w1 + 64);                  // [NBLK][4 waves]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  const int nblk256 = (a.N + 255) >> 8;
  if (FIRST && tid < 64) s_w1[tid] = make_float4(a.w1[3 * tid], a.w1[3 * tid + 1], a.w1[3 * tid + 2], a.b1[tid]);

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
      cc_wload(W, true, tid, wv[blk], s_red + blk * 4);
    }
    __syncthreads();
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      Ew[blk] = cc_wsplit(wv[blk], true, tid, s_red + blk * 4, cc_smem + blk * CC_BLK);
    }
    __syncthreads();
  }
  // The workgroup's column blocks of 256 points (gridDim.x workgroups per instance share them round robin: the weight
  // images above are built once per workgroup, and a launch of 1000 blocks becomes ONE round of 500 workgroups on the 512
  // slots of the chip instead of two rounds of which the second is 95 % full).  No barrier below: the waves run apart.
  for (int cblk = blockIdx.x; cblk < nblk256; cblk += gridDim.x) {
  const int col = cblk * 256 + wave * 64 + lane;
  const bool live = col < a.N, wave_live = cblk * 256 + wave * 64 < a.N;
  const unsigned long long livemask = __builtin_amdgcn_ballot_w64(live);
  const size_t mword = ((size_t)b * ((a.N + 63) >> 6) + (size_t)(cblk * 4 + wave));   // x Co: gate words [B][column block][row]
  float o[64];   // this lane's point: the 64 input rows of the current stage
  if (FIRST) {
    const float* xp = a.x3 + (size_t)b * 3 * a.N + (live ? col : a.N - 1);
    const float x0 = xp[0], x1 = xp[a.N], x2 = xp[2 * (size_t)a.N];
    float p0 = x0, p1 = x1, p2 = x2;
    if (a.T3) {   // bmm(pc^T, T)^T : x'[c] = sum_d x[d] T[d][c]   (Model/PointNet.py:138)
      const float* t = a.T3 + (size_t)b * 9;
      p0 = x0 * t[0] + x1 * t[3] + x2 * t[6];
      p1 = x0 * t[1] + x1 * t[4] + x2 * t[7];
      p2 = x0 * t[2] + x1 * t[5] + x2 * t[8];
    }
#pragma unroll
    for (int u = 0; u < 64; ++u) o[u] = fmaxf(cc_first_layer(s_w1[u], p0, p1, p2), 0.f);
  } else {   // all 64 rows in flight at once
    const float* X = a.X + (size_t)b * a.sXb;      // uniform row base + lane offset: scalar-base addressing
    const unsigned xc = (unsigned)(live ? col : a.N - 1);
#pragma unroll
    for (int u = 0; u < 64; ++u) o[u] = (X + (size_t)u * a.ldX)[xc];
  }

#pragma unroll
  for (int blk = 0; blk < NBLK; ++blk) {
    const int s = blk < NS - 1 ? blk : NS - 1, rb = blk - s;
    const bool last = s == NS - 1;
    const ChainStage& st = a.st[s];
    const unsigned char* arow = cc_smem + blk * CC_BLK + (lane & 31) * CC_PITCH + (lane >> 5) * 16;   // A: row r, k = 8h + j

    f32x16 acc[2][2];   // [column block][row tile]
    unsigned Ex;
    cc_mm(o, arow, live, acc, Ex);
    const float unscale = cc_unscale(Ex) * cc_unscale(Ew[blk]);

    // epilogue: every pair of accumulator registers becomes two 64-column rows (row, row + 4); bias + relu; the row's
    // gate bits go to the stage's mask word (lane = row), the row itself into o[] -- the next stage's input, and what is
    // stored below if the stage has an output tensor
    float res[64];
    int mlo = 0, mhi = 0;
    cc_rows(acc, res);
#pragma unroll
    for (int row = 0; row < 64; ++row) {
      float r = res[row] * unscale;
      r += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(biasv[blk]), row));
      r = fmaxf(r, 0.f);
      const unsigned long long mk = __builtin_amdgcn_ballot_w64(r > 0.f) & livemask;
      cc_writelane(mlo, (int)(unsigned)mk, row);
      cc_writelane(mhi, (int)(unsigned)(mk >> 32), row);
      res[row] = r;
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
}

// The backward of the trunk's front in one kernel (Model/PointNet.py:137-144 backward), the bits of the three launches
// it replaces (two 64 x 64 input-gradient layers, the second accumulating onto the first, then conv_gate_first's form):
//   dh2 = gate_h2( W3eff^T Ga + Wc1^T Gb )          Ga = d/d(pre-activation of h3), Gb = that of the T-Net's c1
//   g1  = gate_first( W2^T dh2 ),  q = w1^T g1,  dx = T q,  dTpart = per-workgroup sums of x q^T
// dh2 and g1 never leave the registers.
// Two waves per SIMD -- and NO packed-FP32 instructions (NOTEBOOK 5a).  Compiled with SLP vectorisation (604 v_pk_mul_f32 /
// v_pk_fma_f32 / v_pk_add_f32, among them the 3x3 transform with SGPR-pair operands) this kernel computed wrong values in
// lanes 48-63 of ~1e-4 of its workgroups whenever two of its wavefronts shared a SIMD: p = T3^T x came out wrong at the start,
// d = T q and the dx stores at the end (stage checksums of tools/ub/conv_bwd_chain_probe.patch in tools/ub/dtpart_pair.hip);
// with the transform through vector loads the rate falls 1000-fold, without packed instructions altogether
// (geoa3_amd/build.py FILE_FLAGS: -fno-slp-vectorize for this file) it is 0 of 1e8 workgroups at the same 48 us.
// The faulty build for the harness: `python -m geoa3_amd.build --variant tools/ub/lib_two_wave --no-file-flags`.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_bwd_chain_kernel(ConvBwdChainArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char cc_smem[];
  float4* s_w1 = reinterpret_cast<float4*>(cc_smem + 3 * CC_BLK);   // [64] (w1 row, b1)
  float* s_red = reinterpret_cast<float*>(s_w1 + 64);               // [3][4 waves]
  float* s_part = s_red + 12;                                       // [4 waves][9]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cblk = blockIdx.x, b = blockIdx.y;
  const int col = cblk * 256 + wave * 64 + lane;
  const bool live = col < a.N, wave_live = cblk * 256 + wave * 64 < a.N;
  const unsigned xc = (unsigned)(live ? col : a.N - 1);
  const size_t mword = ((size_t)b * ((a.N + 63) >> 6) + (size_t)(cblk * 4 + wave)) * 64;

  float o[64], xb[64];
  {
    const float* X = a.Xa + (size_t)b * 64 * a.N;
#pragma unroll
    for (int u = 0; u < 64; ++u) o[u] = (X + (size_t)u * a.N)[xc];
  }
  const float* xp = a.x3 + (size_t)b * 3 * a.N + xc;
  const float x0 = xp[0], x1 = xp[a.N], x2 = xp[2 * (size_t)a.N];
  float p0 = x0, p1 = x1, p2 = x2;
  if (a.T3) {
    const float* t = a.T3 + (size_t)b * 9;
    p0 = x0 * t[0] + x1 * t[3] + x2 * t[6];
    p1 = x0 * t[1] + x1 * t[4] + x2 * t[7];
    p2 = x0 * t[2] + x1 * t[5] + x2 * t[8];
  }
  if (tid < 64) s_w1[tid] = make_float4(a.w1[3 * tid], a.w1[3 * tid + 1], a.w1[3 * tid + 2], a.b1[tid]);
  const unsigned long long zmv = wave_live ? a.Zmask[mword + lane] : 0ull;   // lane = row of dh2

  unsigned Ew[3];
  {
    float wv[3][16];
    cc_wload(a.Wa + (size_t)b * a.sWa, false, tid, wv[0], s_red);
    cc_wload(a.Wb, false, tid, wv[1], s_red + 4);
    cc_wload(a.W2t, true, tid, wv[2], s_red + 8);
    __syncthreads();
    Ew[0] = cc_wsplit(wv[0], false, tid, s_red, cc_smem);
    Ew[1] = cc_wsplit(wv[1], false, tid, s_red + 4, cc_smem + CC_BLK);
    Ew[2] = cc_wsplit(wv[2], true, tid, s_red + 8, cc_smem + 2 * CC_BLK);
    __syncthreads();
  }
  {
    const float* X = a.Xb + (size_t)b * 64 * a.N;
#pragma unroll
    for (int u = 0; u < 64; ++u) xb[u] = (X + (size_t)u * a.N)[xc];
  }
  const unsigned char* arow = cc_smem + (lane & 31) * CC_PITCH + (lane >> 5) * 16;
  f32x16 acc[2][2];
  unsigned Ex;
  float res[64];
  // W3eff^T Ga
  cc_mm(o, arow, live, acc, Ex);
  {
    const float un = cc_unscale(Ex) * cc_unscale(Ew[0]);
    cc_rows(acc, o);
#pragma unroll
    for (int r = 0; r < 64; ++r) o[r] *= un;
  }
  // + Wc1^T Gb, then the relu gate of h2
  cc_mm(xb, arow + CC_BLK, live, acc, Ex);
  {
    const float un = cc_unscale(Ex) * cc_unscale(Ew[1]);
    cc_rows(acc, res);
    const unsigned sh = lane & 31;
#pragma unroll
    for (int r = 0; r < 64; ++r) {
      float v = res[r] * un;
      v += o[r];
      const unsigned mlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)zmv, r);
      const unsigned mhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(zmv >> 32), r);
      o[r] = (((lane < 32 ? mlo : mhi) >> sh) & 1u) ? v : 0.f;
    }
  }
  // W2^T dh2, the first layer's gate, its backward
  cc_mm(o, arow + 2 * CC_BLK, live, acc, Ex);
  float q0 = 0.f, q1 = 0.f, q2 = 0.f;
  {
    const float un = cc_unscale(Ex) * cc_unscale(Ew[2]);
    cc_rows(acc, res);
    if (live) {
#pragma unroll
      for (int r = 0; r < 64; ++r) {
        float v = res[r] * un;
        const float4 w = s_w1[r];
        v = cc_first_layer(w, p0, p1, p2) > 0.f ? v : 0.f;
        q0 = fmaf(w.x, v, q0);
        q1 = fmaf(w.y, v, q1);
        q2 = fmaf(w.z, v, q2);
      }
    }
  }
  // x' = T^T x  =>  dx[d] = sum_c T[d][c] q[c];  dT[d][c] = sum_n x[d][n] q[c][n]
  float d0 = q0, d1 = q1, d2 = q2;
  if (a.T3) {
    const float* t = a.T3 + (size_t)b * 9;
    d0 = fmaf(t[2], q2, fmaf(t[1], q1, t[0] * q0));   // explicit: the chain kernel must form the same bits
    d1 = fmaf(t[5], q2, fmaf(t[4], q1, t[3] * q0));
    d2 = fmaf(t[8], q2, fmaf(t[7], q1, t[6] * q0));
  }
  if (live) {
    float* dxp = a.dx + (size_t)b * 3 * a.N + col;
    dxp[0] = d0;
    dxp[a.N] = d1;
    dxp[2 * (size_t)a.N] = d2;
  }
  if (a.dTpart) {   // workgroup-uniform
    const float prod[9] = {x0 * q0, x0 * q1, x0 * q2, x1 * q0, x1 * q1, x1 * q2, x2 * q0, x2 * q1, x2 * q2};
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const float v = wave_sum(prod[i]);
      if (lane == 0) s_part[wave * 9 + i] = v;
    }
    __syncthreads();
    if (tid < 9) {
      a.dTpart[((size_t)b * gridDim.x + cblk) * GEOA3_DT_PITCH + tid] = s_part[tid] + s_part[9 + tid] + s_part[18 + tid] + s_part[27 + tid];
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
  // two column blocks per workgroup when that still leaves more than one round of workgroups (never for a small shard)
  const int nblk = (a.N + 255) / 256;
  const int gx = (size_t)nblk * a.B > 768 ? (nblk + 1) / 2 : nblk;
  hipLaunchKernelGGL(kern, dim3(gx, a.B), dim3(256), lds, s, a);
}

}  // namespace

int launch_conv_chain(const ConvChainArgs& a, hipStream_t s) {
  if (a.ns < 1 || a.ns > 3 || a.B <= 0 || a.N <= 0) return GEOA3_EINVAL;
  for (int i = 0; i < a.ns; ++i) {
    if (!a.st[i].W || !a.st[i].bias || !a.st[i].Ymask) return GEOA3_EINVAL;
    if (a.st[i].Co != (i + 1 < a.ns ? 64 : 128)) return GEOA3_ENOSUPPORT;
  }
  const bool first = a.x3 != nullptr;
  if (first ? (!a.w1 || !a.b1) : !a.X) return GEOA3_EINVAL;
  if (a.ns == 3 && first) launch_chain<3, true, 128>(a, s);
  else if (a.ns == 1 && first) launch_chain<1, true, 128>(a, s);
  else if (a.ns == 2 && !first) launch_chain<2, false, 128>(a, s);
  else return GEOA3_ENOSUPPORT;
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

int launch_conv_bwd_chain(const ConvBwdChainArgs& a, hipStream_t s) {
  if (!a.Xa || !a.Wa || !a.Xb || !a.Wb || !a.Zmask || !a.W2t || !a.x3 || !a.w1 || !a.b1 || !a.dx || a.B <= 0 || a.N <= 0)
    return GEOA3_EINVAL;
  const size_t lds = (size_t)3 * CC_BLK + 64 * 16 + (12 + 36) * sizeof(float);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bwd_chain_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)lds);
  hipLaunchKernelGGL(conv_bwd_chain_kernel, dim3((a.N + 255) / 256, a.B), dim3(256), lds, s, a);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
