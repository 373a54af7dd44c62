// PointNet++ SSG, second set-abstraction level (gfx950): the level's FORWARD in one kernel, the pre-transformed first layer
// per point, and the split-fp16 fragment images of the level-2 / level-3 matrices.
// Reference: pointnet2_modules.py:57-70 (shared MLP (128+3) -> 128 -> 128 -> 256, max over the 64 samples of a ball),
// PointNetPP_ssg.py:68-76.  (The level's backward lives in pointnet2_sa2b.hip: the rows in destination order.)
// Both relu gates leave the forward as bit masks per ROW (centre, sample): [B * centres][64 samples][4 x 32 bits over the 128
// channels] -- what the backward reads, 16 bytes per row; neither activation is ever written.
#include "pointnet_kernels.h"
#include "pointnet2_sa2_common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------------
// Forward of the same level in one kernel: gather + shift + relu (a0), W1 (+ b1, relu: a1), W2 max-pooled over the 64
// samples -- pointnet2_modules.py:57-70 behind the pre-transformed first layer (r = W_f f + W_x xyz per POINT, shift =
// b0 - W_x c per centre: pointnet2_net.hip).  The unfused chain wrote and re-read the [B,128,8192] tensors a0 and a1
// (0.36 + 0.58 + 0.81 ms at B = 250); here neither exists: a workgroup keeps two centres' activations in LDS as split
// fp16 images (sample-major, one power-of-two scale per centre from the tile's own maximum) that the matrix-core loops
// read as ready operands:
//   A  (2 waves per centre, lane = sample): the sample's 64 channels of its wave's half are one 256-byte gather from the
//      point-major r; + shift, relu, gate bits (m0), maximum -> scale -> hi / lo images of a0;
//   B  (wave = 64 output rows x one centre): a1 = relu(W1 a0 + b1) with W1 as A fragments streamed from an L2-resident
//      image (sa2_img_kernel), gate bits (m1), maximum -> scale -> the images are OVERWRITTEN with a1;
//   C  the pooled layer TRANSPOSED (activations as A: rows = samples, W2 fragments as B: columns = channels) in two
//      passes of 64 channels per wave: a lane ends with one channel and 32 of its samples, the max over samples is
//      lane-local plus one exchange (first maximal sample on ties, as F.max_pool2d).
// Arithmetic of pointnet_conv_split.hip: a*w = a_hi*w_hi + a_hi*w_lo + a_lo*w_hi, fp32 accumulation.
constexpr int S2_PH = S2_K * 2 + 16;     // bytes per sample row of an fp16 image (conflict-free 16-byte reads)

// W[R][K] (R % 32 == 0, K % 16 == 0) -> power-of-two scale from its maximum, split, in fragment order:
// [tile = row >> 5][cg = k >> 4][piece][lane = 32 (k half) + (row & 31)][8 k]; un[0] = 1 / scale.  One workgroup (runs
// once per set of weights: geoa3_pn2ssg_pack_images).
__global__ __launch_bounds__(256) void frag_image_kernel(const float* __restrict__ W, int R, int K, _Float16* __restrict__ img,
                                                         float* __restrict__ un) {
  __shared__ float s_m[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float m = 0.f;
  for (int e = tid; e < R * K; e += 256) m = fmaxf(m, __builtin_fabsf(W[e]));
  m = wave_max(m);
  if (lane == 0) s_m[wave] = m;
  __syncthreads();
  const unsigned Ew = s2_exp(fmaxf(fmaxf(s_m[0], s_m[1]), fmaxf(s_m[2], s_m[3])));
  const float sw = s2_scale(Ew);
  if (tid == 0) un[0] = s2_unscale(Ew);
  const int kc = K >> 4;
  for (int e = tid; e < R * K; e += 256) {
    const int row = e / K, k = e - row * K;
    const int tile = row >> 5, r = row & 31, cg = k >> 4, h = (k >> 3) & 1, j = k & 7;
    const float v = W[e] * sw;
    const _Float16 hi = (_Float16)v;
    const size_t f = ((size_t)tile * kc + cg) * 2;
    img[((f + 0) * 64 + h * 32 + r) * 8 + j] = hi;
    img[((f + 1) * 64 + h * 32 + r) * 8 + j] = (_Float16)(v - (float)hi);
  }
}

struct Sa2FwdArgs {
  const float* rT;                 // [B][N1][128]
  const int32_t* gidx;             // [B][M][64]
  const float* shift;              // [B][128][M]
  const float* b1;                 // [128]
  const float* b2;                 // [256]
  const _Float16* img1; const float* un1;   // W1 [128][128]
  const _Float16* img2; const float* un2;   // W2 [256][128]
  float* out; int32_t* arg;        // [B][256][M]
  unsigned* m0;                    // [B * M][64 samples][4]: a0 > 0, bit c of word w = channel 32 w + c
  unsigned* m1;                    // the same for a1
  int B, N1, M;
};

// ------------------------------------------------------------------------------------------------------------------
// The forward as ONE 8-wave workgroup per CU.  What bounded the first, 4-wave / two-workgroups-per-CU form (round 4; removed
// in round 6) was weight traffic per wave: both waves of
// a centre streamed all of W1 and half of W2 through a two-step register ring (a deeper ring spilled at 256 VGPRs), so part
// of every L2 round trip was exposed.  Here
//   * W1's fragment image (64 KB) is copied into LDS once per workgroup: layer 1 reads its A fragments from LDS;
//   * W2 is split eight ways: wave w owns channel tile w (32 channels) for BOTH centres, so every fragment of W2 is
//     loaded once per pair and CU (16 x 1 KB per wave), through a ring S8_RING k-steps deep;
//   * layer 1 is split as (centre, 32-row tile) over the eight waves; the gather as (centre, 32-channel quarter).
constexpr int S8_RING = 4;
constexpr int S8_GRP = 4;       // consecutive pairs a workgroup takes in a row: their 8 centres' pooled outputs leave as one 32-byte run per channel
constexpr int sa2_fwd8_lds() { return S2_K * S2_K * 4 + 2 * 2 * 64 * S2_PH + (16 + S2_K + S2_C) * 4 + 2 * S2_C * 2 * S8_GRP * 4; }

template <int MODE>   // 0 = shipped; tools/ub/sa2f_ub.hip: 1 no W2 MFMAs, 2 no W1 MFMAs, 3 no pooled stores, 4 no gather, 5 no syncs C
__global__ __launch_bounds__(512) void sa2_fwd8_kernel(Sa2FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s2_sm[];
  half8* s_w1 = reinterpret_cast<half8*>(s2_sm);                      // fragment image of W1: [(tile * 8 + c) * 2 + piece][lane]
  unsigned char* s_img = s2_sm + S2_K * S2_K * 4;                     // [2 centres][hi / lo][64 samples][S2_PH]
  float* s_red = reinterpret_cast<float*>(s_img + 2 * 2 * 64 * S2_PH);   // [0..7] a0 maxima, [8..15] a1 maxima (per wave)
  float* s_b1 = s_red + 16;
  float* s_b2 = s_b1 + S2_K;
  float* s_out = s_b2 + S2_C;                                          // [256][8]: pooled outputs of a group's 8 centres
  int* s_arg = reinterpret_cast<int*>(s_out + S2_C * 2 * S8_GRP);      // [256][8]
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ci = wave >> 2, qt = wave & 3;       // phases A / B: this wave's centre and its channel quarter / row tile
  unsigned char* thi = s_img + ci * 2 * 64 * S2_PH;
  unsigned char* tlo = thi + 64 * S2_PH;
  {
    const half8* src = reinterpret_cast<const half8*>(a.img1);
    for (int e = tid; e < S2_K * S2_K * 4 / 16; e += 512) s_w1[e] = src[e];
  }
  for (int e = tid; e < S2_K; e += 512) s_b1[e] = a.b1[e];
  for (int e = tid; e < S2_C; e += 512) s_b2[e] = a.b2[e];
  const float un1 = a.un1[0], un2 = a.un2[0];
  const half8* B2 = reinterpret_cast<const half8*>(a.img2) + (size_t)wave * 8 * 2 * 64 + lane;   // channel tile `wave`
  auto load_w2 = [&](int c, half8 (&f)[2]) {
    f[0] = B2[(size_t)(c * 2 + 0) * 64];
    f[1] = B2[(size_t)(c * 2 + 1) * 64];
  };
  const int pairs = a.B * a.M / 2;   // B * M < 2^31 (checked by the launcher); a pair never straddles two instances (M even)
  int gi = 0;
  float shv = 0.f;
  float4 gv[8];
  auto request = [&](int p) {     // the pair's gather for this wave: sample `lane` of centre ci, channels 32 qt .. + 31
    const int centre = 2 * p + ci;
    const int b = (2 * p) / a.M, m = centre - b * a.M;
    gi = a.gidx[(size_t)centre * 64 + lane];
    shv = a.shift[((size_t)b * S2_K + 32 * qt + l31) * a.M + m];
    const float4* src = reinterpret_cast<const float4*>(a.rT + ((size_t)b * a.N1 + gi) * S2_K + 32 * qt);
#pragma unroll
    for (int q = 0; q < 8; ++q) gv[q] = MODE == 4 ? make_float4(0.1f * q, 0.2f, -0.1f, 0.3f) : src[q];
  };
  // A workgroup takes GROUPS of S8_GRP consecutive pairs (group g, g + gridDim.x, ...): the pooled outputs of a group's 8
  // consecutive centres are collected in LDS and leave as one 32-byte run per channel.  (Written pair by pair -- 8 bytes
  // per channel, 512 bytes apart -- every store was a quarter of a 32-byte sector: 328 MB of HBM writes per launch for
  // 131 MB of results, profiles/round6_v1_c4_pmc.csv.)  M % 8 == 0 (the launcher checks): a group never straddles two clouds.
  const int groups = pairs / S8_GRP;
  if ((int)blockIdx.x < groups) request(blockIdx.x * S8_GRP);
  __syncthreads();

  for (int grp = blockIdx.x; grp < groups; grp += gridDim.x)
  for (int pq = 0; pq < S8_GRP; ++pq) {
    const int p = grp * S8_GRP + pq;
    const size_t centre = (size_t)(2 * p + ci);
    half8 ring[S8_RING][2];
#pragma unroll
    for (int i = 0; i + 1 < S8_RING; ++i) load_w2(i, ring[i]);
    // ---- A: a0 = relu(r[sample] + shift): lane = sample, this wave's 32 channels
    float carry;
    {
      float v[32];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        v[4 * q] = gv[q].x;
        v[4 * q + 1] = gv[q].y;
        v[4 * q + 2] = gv[q].z;
        v[4 * q + 3] = gv[q].w;
      }
      unsigned gw = 0u;                 // a0 > 0 over this wave's 32 channels of the lane's sample: one word of the row's gate
      float mx = 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        v[c] = fmaxf(v[c] + s2_rlf(shv, c), 0.f);
        mx = fmaxf(mx, v[c]);
        gw |= v[c] > 0.f ? 1u << c : 0u;
      }
      a.m0[(centre * 64 + lane) * 4 + qt] = gw;
      mx = wave_max(mx);
      if (lane == 0) s_red[wave] = mx;
      __syncthreads();
      const unsigned Ea = s2_exp(fmaxf(fmaxf(s_red[4 * ci], s_red[4 * ci + 1]), fmaxf(s_red[4 * ci + 2], s_red[4 * ci + 3])));
      const float sa = s2_scale(Ea);
#pragma unroll
      for (int j8 = 0; j8 < 4; ++j8) {
        half8 hh, ll;
        {
          const float x8[8] = {v[8 * j8], v[8 * j8 + 1], v[8 * j8 + 2], v[8 * j8 + 3], v[8 * j8 + 4], v[8 * j8 + 5], v[8 * j8 + 6], v[8 * j8 + 7]};
          s2_split8(x8, sa, hh, ll);
        }
        *reinterpret_cast<half8*>(thi + lane * S2_PH + (32 * qt + 8 * j8) * 2) = hh;
        *reinterpret_cast<half8*>(tlo + lane * S2_PH + (32 * qt + 8 * j8) * 2) = ll;
      }
      carry = s2_unscale(Ea);
    }
    __syncthreads();
    // ---- B: a1 rows 32 qt .. 32 qt + 31 of centre ci
    {
      f32x16 acc[2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[cb][r] = 0.f;
      const unsigned char* xrow = thi + l31 * S2_PH + h * 16;
      const half8* wf = s_w1 + (size_t)qt * 8 * 2 * 64 + lane;
      // operands of k-step c + 1 are read from LDS while the matrix core works on step c
      half8 wq[2][2], xq[2][4];
      auto read_b = [&](int c, half8 (&w)[2], half8 (&x)[4]) {
        w[0] = wf[(c * 2 + 0) * 64];
        w[1] = wf[(c * 2 + 1) * 64];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          x[2 * cb] = *reinterpret_cast<const half8*>(xrow + cb * 32 * S2_PH + c * 32);
          x[2 * cb + 1] = *reinterpret_cast<const half8*>(xrow + cb * 32 * S2_PH + c * 32 + 64 * S2_PH);
        }
      };
      read_b(0, wq[0], xq[0]);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c + 1 < 8) read_b(c + 1, wq[(c + 1) & 1], xq[(c + 1) & 1]);
        const half8 wh = wq[c & 1][0], wl = wq[c & 1][1];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const half8 xh = xq[c & 1][2 * cb], xl = xq[c & 1][2 * cb + 1];
          if (MODE == 2) {
            acc[cb][0] += (float)wh[0] + (float)xh[0] + (float)wl[0] + (float)xl[0];
            continue;
          }
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl, acc[cb], 0, 0, 0);
          acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh, acc[cb], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // acc[cb][r]: row 32 qt + (r&3) + 8 (r>>2) + 4 h, sample 32 cb + l31
      const float un = carry * un1;
      float mh = 0.f;
      unsigned g0 = 0u, g1 = 0u;        // a1 > 0 of the lane's 16 rows (channels rr + 4 h of the tile) for samples l31 / 32 + l31
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2);   // + 4 h
        const float bias = s_b1[32 * qt + rr + 4 * h];
        const float o0 = fmaxf(acc[0][r] * un + bias, 0.f), o1 = fmaxf(acc[1][r] * un + bias, 0.f);
        acc[0][r] = o0;
        acc[1][r] = o1;
        mh = fmaxf(mh, fmaxf(o0, o1));
        g0 |= o0 > 0.f ? 1u << rr : 0u;
        g1 |= o1 > 0.f ? 1u << rr : 0u;
      }
      {   // the other 16 channels of the tile sit in the lane 32 away: a row's word = both halves; lane (h, l31) stores sample 32 h + l31
        g0 <<= 4 * h;
        g1 <<= 4 * h;
        const unsigned f0 = g0 | (unsigned)__shfl_xor((int)g0, 32, 64), f1 = g1 | (unsigned)__shfl_xor((int)g1, 32, 64);
        a.m1[(centre * 64 + lane) * 4 + qt] = h ? f1 : f0;
      }
      mh = wave_max(mh);
      if (lane == 0) s_red[8 + wave] = mh;
      __syncthreads();   // every wave is done reading a0; the maxima are visible
      const unsigned Eh =
          s2_exp(fmaxf(fmaxf(s_red[8 + 4 * ci], s_red[9 + 4 * ci]), fmaxf(s_red[10 + 4 * ci], s_red[11 + 4 * ci])));
      const float sh = s2_scale(Eh);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          typedef _Float16 half4 __attribute__((ext_vector_type(4)));
          half4 hh, ll;
          {
            unsigned h0, l0, h1, l1;
            s2_split2(acc[cb][4 * g4], acc[cb][4 * g4 + 1], sh, h0, l0);
            s2_split2(acc[cb][4 * g4 + 2], acc[cb][4 * g4 + 3], sh, h1, l1);
            typedef unsigned uint2v __attribute__((ext_vector_type(2)));
            hh = __builtin_bit_cast(half4, uint2v{h0, h1});
            ll = __builtin_bit_cast(half4, uint2v{l0, l1});
          }
          const int k0 = 32 * qt + 8 * g4 + 4 * h;
          *reinterpret_cast<half4*>(thi + (32 * cb + l31) * S2_PH + k0 * 2) = hh;
          *reinterpret_cast<half4*>(tlo + (32 * cb + l31) * S2_PH + k0 * 2) = ll;
        }
      carry = s2_unscale(Eh);     // of this wave's centre ci; phase C needs both centres' (below)
    }
    __syncthreads();
    // the next pair's gather: in flight across phase C
    {
      const int pn = pq + 1 < S8_GRP ? p + 1 : (grp + (int)gridDim.x) * S8_GRP;
      if (pn < pairs) request(pn);
    }
    // ---- C: pooled channels 32 wave .. + 31 for both centres, transposed (rows = samples)
    {
      f32x16 acc[4];     // [2 centre + column block]
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
      const unsigned char* xrow = s_img + l31 * S2_PH + h * 16;
      half8 xc[2][8];
      auto read_c = [&](int c, half8 (&x)[8]) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const unsigned char* xr = xrow + (q >> 1) * 2 * 64 * S2_PH + (q & 1) * 32 * S2_PH + c * 32;
          x[2 * q] = *reinterpret_cast<const half8*>(xr);
          x[2 * q + 1] = *reinterpret_cast<const half8*>(xr + 64 * S2_PH);
        }
      };
      read_c(0, xc[0]);
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c + S8_RING - 1 < 8) load_w2(c + S8_RING - 1, ring[(c + S8_RING - 1) % S8_RING]);
        if (c + 1 < 8) read_c(c + 1, xc[(c + 1) & 1]);
        const half8 wh = ring[c % S8_RING][0], wl = ring[c % S8_RING][1];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const half8 xh = xc[c & 1][2 * q], xl = xc[c & 1][2 * q + 1];
          if (MODE == 1) {
            acc[q][0] += (float)wh[0] + (float)xh[0] + (float)wl[0] + (float)xl[0];
            continue;
          }
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh, acc[q], 0, 0, 0);
          acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl, acc[q], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // acc[2 cc + cb][r]: channel 32 wave + l31, sample 32 cb + (r&3) + 8 (r>>2) + 4 h of centre cc (ascending)
#pragma unroll
      for (int cc = 0; cc < 2; ++cc) {
        const unsigned Eh =
            s2_exp(fmaxf(fmaxf(s_red[8 + 4 * cc], s_red[9 + 4 * cc]), fmaxf(s_red[10 + 4 * cc], s_red[11 + 4 * cc])));
        const float un = s2_unscale(Eh) * un2;
        float v = -__builtin_inff();
        int smp = 0;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const bool gt = acc[2 * cc + cb][r] > v;
            v = gt ? acc[2 * cc + cb][r] : v;
            smp = gt ? 32 * cb + mfma_row(r, lane) : smp;
          }
        const float ov = __shfl_xor(v, 32, 64);
        const int os = __shfl_xor(smp, 32, 64);
        const bool take = ov > v || (ov == v && os < smp);
        v = take ? ov : v;
        smp = take ? os : smp;
        if (lane < 32 && (MODE != 3 || v == 123.f)) {
          const int ch = 32 * wave + lane;
          s_out[ch * (2 * S8_GRP) + 2 * pq + cc] = fmaxf(v * un + s_b2[ch], 0.f);     // the (positive) scale commutes with the max
          s_arg[ch * (2 * S8_GRP) + 2 * pq + cc] = smp;
        }
      }
    }
    __syncthreads();   // the images and s_red are rewritten by the next pair
    if (pq == S8_GRP - 1 && MODE != 3) {   // the group's 8 centres: thread (channel, half) writes 16 bytes of each table
      const int c0 = 2 * S8_GRP * grp;                     // first centre of the group (global index)
      const int b = c0 / a.M, m0 = c0 - b * a.M;
      const int ch = tid >> 1, hh = tid & 1;
      const size_t e = ((size_t)b * S2_C + ch) * a.M + m0 + 4 * hh;
      *reinterpret_cast<float4*>(a.out + e) = *reinterpret_cast<const float4*>(s_out + ch * (2 * S8_GRP) + 4 * hh);
      *reinterpret_cast<int4*>(a.arg + e) = *reinterpret_cast<const int4*>(s_arg + ch * (2 * S8_GRP) + 4 * hh);
    }
  }
}


// ------------------------------------------------------------------------------------------------------------------
// Level 2's pre-transformed first layer per POINT, in one kernel (round 5):
//   rT[p][co] = sum_k f[p][k] Wf[co][k] + sum_d Wx[co][d] xyz[p][d]        f = level 1's pooled features, point-major
// (pointnet2_modules.py:57-70 behind the rearrangement of pointnet2_net.hip).  It was four launches -- transpose to
// channel-major, the generic split convolution, + W_x xyz, transpose back: 0.13 ms for 0.13 GB that a single pass moves
// in 0.03.  Level 1 writes its features point-major and level 2's gather wants r point-major: with the POINTS as the
// matrix core's rows neither transpose exists.  A wave owns 32 points: their 128 features are 64 registers, the tile's
// own power-of-two scale comes from their maximum, the three coordinates ride in a ninth k-step whose B fragments are
// W_x at the image's scale; four 32-channel column tiles of 27 matrix instructions each.
struct Sa2PreArgs {
  const float* X;        // [P][128]; XT: channel-major [P / Np][128][Np]
  const float* xyz;      // [P][3], or null: no coordinate term
  const _Float16* img;   // fragment image of Wf (launch_frag_image: rows = output channels)
  const float* un;       // [1]: 1 / the image's scale
  const float* Wx;       // [128][3]
  float* Y;              // [P][128]
  long P;                // points (a multiple of 32)
  int Np;                // XT: points per instance (a multiple of 32)
};

constexpr int sa2_pre_lds() { return S2_K * S2_K * 4 + 4 * 2 * 64 * 16; }

// XT: the same product with X channel-major (the backward's d f = W_f^T d r from the grouping gradient's layout, written
// centroid-major for level 1's backward: again no transpose)
template <bool XT, bool XYZ>   // XYZ: with the coordinate term (a.xyz, a.Wx)
__global__ __launch_bounds__(256) void sa2_pre_kernel(Sa2PreArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s2_sm[];
  half8* s_w = reinterpret_cast<half8*>(s2_sm);                     // [(tile * 8 + c) * 2 + piece][lane]
  half8* s_wx = s_w + S2_K * S2_K * 4 / 16;                         // [tile * 2 + piece][lane]: the coordinates' k-step
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {
    const half8* src = reinterpret_cast<const half8*>(a.img);
    for (int e = tid; e < S2_K * S2_K * 4 / 16; e += 256) s_w[e] = src[e];
  }
  const float unW = a.un[0];
  if (XYZ) {   // B fragments of the ninth k-step: lane (column co = 32 t + l31, k = 8 h + j): W_x[co][k] for k < 3, at the image's scale
    const float sw = 1.0f / unW;   // (a power of two)
    const int t = wave;            // four waves, four column tiles
    const int co = 32 * t + l31;
    float x[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (h == 0) {
      x[0] = a.Wx[3 * co];
      x[1] = a.Wx[3 * co + 1];
      x[2] = a.Wx[3 * co + 2];
    }
    half8 wh, wl;
    s2_split8(x, sw, wh, wl);
    s_wx[(t * 2 + 0) * 64 + lane] = wh;
    s_wx[(t * 2 + 1) * 64 + lane] = wl;
  }
  __syncthreads();
  const long tiles = a.P / 32;
  for (long tl = (long)blockIdx.x * 4 + wave; tl < tiles; tl += (long)gridDim.x * 4) {
    const long p0 = tl * 32;
    // ---- the tile's operands: row = point p0 + l31, k = 16 c + 8 h + j
    float raw[8][8];
    float mx = 0.f;
    if (!XT) {
      const float* row = a.X + (p0 + l31) * S2_K + 8 * h;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float4 b0 = *reinterpret_cast<const float4*>(row + 16 * c);
        const float4 b1 = *reinterpret_cast<const float4*>(row + 16 * c + 4);
        raw[c][0] = b0.x; raw[c][1] = b0.y; raw[c][2] = b0.z; raw[c][3] = b0.w;
        raw[c][4] = b1.x; raw[c][5] = b1.y; raw[c][6] = b1.z; raw[c][7] = b1.w;
      }
    } else {   // X[b][k][n]: a lane's eight k are Np floats apart, the 32 points of a half-wave contiguous
      const long b = p0 / a.Np;
      const float* col = a.X + (b * S2_K + 8 * h) * a.Np + (p0 - b * a.Np) + l31;
#pragma unroll
      for (int c = 0; c < 8; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[c][j] = col[(size_t)(16 * c + j) * a.Np];
    }
#pragma unroll
    for (int c = 0; c < 8; ++c)
#pragma unroll
      for (int j = 0; j < 8; ++j) mx = fmaxf(mx, __builtin_fabsf(raw[c][j]));
    float q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (XYZ && h == 0) {
      const float* pz = a.xyz + (p0 + l31) * 3;
      q[0] = pz[0];
      q[1] = pz[1];
      q[2] = pz[2];
    }
    mx = fmaxf(mx, fmaxf(__builtin_fabsf(q[0]), fmaxf(__builtin_fabsf(q[1]), __builtin_fabsf(q[2]))));
    const unsigned Ex = s2_exp(wave_max(mx));
    const float sx = s2_scale(Ex);
    half8 ah[9], al[9];
#pragma unroll
    for (int c = 0; c < 8; ++c) s2_split8(raw[c], sx, ah[c], al[c]);
    s2_split8(q, sx, ah[8], al[8]);
    const float unscale = s2_unscale(Ex) * unW;
    float* Y = a.Y + p0 * S2_K + l31;
#pragma unroll 1
    for (int t = 0; t < 4; ++t) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
      const half8* wf = s_w + (size_t)t * 8 * 2 * 64 + lane;
#pragma unroll
      for (int c = 0; c < (XYZ ? 9 : 8); ++c) {
        const half8 bh = c < 8 ? wf[(c * 2 + 0) * 64] : s_wx[(t * 2 + 0) * 64 + lane];
        const half8 bl = c < 8 ? wf[(c * 2 + 1) * 64] : s_wx[(t * 2 + 1) * 64 + lane];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c], bh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c], bl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[c], bh, acc, 0, 0, 0);
      }
      // acc[r]: point p0 + (r & 3) + 8 (r >> 2) + 4 h, channel 32 t + l31: 128 contiguous bytes per point and half-wave
#pragma unroll
      for (int r = 0; r < 16; ++r) Y[(size_t)mfma_row(r, lane) * S2_K + 32 * t] = acc[r] * unscale;
    }
  }
}

}  // namespace

int launch_sa2_pre(const float* X, bool x_channel_major, int Np, const float* xyz, const void* wf_img, const float* wf_un,
                   const float* Wx, float* Y, long P, hipStream_t s) {
  if (P <= 0 || P % 32 != 0 || (x_channel_major && (Np <= 0 || Np % 32 != 0 || P % Np != 0)) || (xyz && !Wx)) return GEOA3_ENOSUPPORT;
  Sa2PreArgs a{X, xyz, static_cast<const _Float16*>(wf_img), wf_un, Wx, Y, P, Np};
  const int lds = sa2_pre_lds();
  const long tiles = P / 32;
  const unsigned grid = (unsigned)((tiles + 3) / 4 < 512 ? (tiles + 3) / 4 : 512);   // two workgroups per CU (72 KB of LDS each)
#define GEOA3_PRE(XT_, XYZ_)                                                                                             \
  do {                                                                                                                   \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa2_pre_kernel<XT_, XYZ_>),                                   \
                              hipFuncAttributeMaxDynamicSharedMemorySize, lds);                                          \
    hipLaunchKernelGGL((sa2_pre_kernel<XT_, XYZ_>), dim3(grid), dim3(256), lds, s, a);                                    \
  } while (0)
  if (x_channel_major && xyz) GEOA3_PRE(true, true);
  else if (x_channel_major) GEOA3_PRE(true, false);
  else if (xyz) GEOA3_PRE(false, true);
  else GEOA3_PRE(false, false);
#undef GEOA3_PRE
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

int launch_frag_image(const float* W, int R, int K, void* img, float* un, hipStream_t s) {
  if (R % 32 != 0 || K % 16 != 0) return GEOA3_EINVAL;
  hipLaunchKernelGGL(frag_image_kernel, dim3(1), dim3(256), 0, s, W, R, K, static_cast<_Float16*>(img), un);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

int launch_sa2_fwd(const float* rT, const int32_t* gidx, const float* shift, const void* w1_img, const float* w1_un,
                   const float* b1, const void* w2_img, const float* w2_un, const float* b2, float* out, int32_t* arg,
                   unsigned* m0, unsigned* m1, int B, int N1, int M, hipStream_t s) {
  if (M % (2 * S8_GRP) != 0 || (long)B * M > 0x3fffffffL) return GEOA3_ENOSUPPORT;
  Sa2FwdArgs a{rT, gidx, shift, b1, b2, static_cast<const _Float16*>(w1_img), w1_un, static_cast<const _Float16*>(w2_img),
               w2_un, out, arg, m0, m1, B, N1, M};
  const long pairs = (long)B * M / 2;
  const int lds = sa2_fwd8_lds();
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa2_fwd8_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const long groups = pairs / S8_GRP;
  const unsigned grid = (unsigned)(groups < 256 ? groups : 256);   // one 8-wave workgroup per CU, persistent
  hipLaunchKernelGGL(sa2_fwd8_kernel<0>, dim3(grid), dim3(512), lds, s, a);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
