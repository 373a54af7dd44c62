// PointNet++ SSG, second set-abstraction level, backward of the two hidden layers in ONE kernel (gfx950).
// Reference: pointnet2_modules.py:57-70 (shared MLP (128+3) -> 128 -> 128 -> 256, max over the 64 samples of a ball),
// PointNetPP_ssg.py:68-76; the autograd of F.max_pool2d / Conv2d / ReLU behind it.
//
// With z = W2 a1 max-pooled over a centre's samples, the pooled gradient reaches exactly ONE sample per channel:
//   d a1[k][s] = [a1[k][s] > 0] * sum_{ch : arg[ch] == s} g[ch] W2[ch][k]                      (256 x 128 MACs per centre)
//   d a0[i][s] = [a0[i][s] > 0] * sum_k W1[k][i] d a1[k][s]                                     (dense 128 x 128 x 64)
// The unfused form ran the first line as a dense K = 256 convolution over a one-hot operand (1.04 ms at B = 250) and
// wrote / re-read the [B,128,8192] tensor d a1 and both activations as relu gates (0.58 ms for the second layer).
// Here a workgroup takes two centres at a time:
//   phase 1 (VALU, 2 wavefronts per centre, lane = k): the centre's 256 (channel, sample) pairs arrive SORTED by sample
//     (sa2_sort_kernel: stable counting sort, so the channels of a sample stay in ascending order); a lane accumulates
//     g * W2[ch][k] in a register over the channels of one sample -- W2 rows prefetched 32 entries ahead, they are the
//     only memory traffic -- and writes the gated sum once per sample into a sample-major fp32 tile in LDS;
//   phase 2 (matrix core, wave = 64 output rows x one centre): d a0 = W1^T tile with split-fp16 operands (the arithmetic
//     of pointnet_conv_split.hip; W1^T scaled + split once per workgroup into LDS, the tile scaled by a power of two from
//     its own maximum and split as it is read), gated by bits and written as 256-byte rows.
// Both relu gates come as bit masks ([B * centres][128] 64-bit words, bit s = sample s; ConvArgs::Ymask layout): d a1
// never exists in memory and neither activation is read.  Deterministic: fixed summation orders, no atomics.
#include "pointnet_kernels.h"
#include "pointnet2_sa2_common.h"

namespace {

// One wavefront per centre: entries (g, channel) sorted by the channel's arg-max sample, ascending channel inside a
// sample.  ent_c = channel | sample << 16 | (last entry of its sample) << 31.
// A stable counting sort in four rounds of 64 channels (round 5; round 4 swept the 64 sample values with four ballots each:
// 256 ballots per centre, 85 us per launch): a histogram of the samples in LDS, its exclusive prefix (lane = sample), and per
// round the lanes holding the same sample found by six ballots (one per bit of the sample): the rank inside the group is a
// population count, the group's first lane advances the sample's running offset for the next round.
// the sort of ONE centre's 256 (value, sample) pairs by a wavefront: lane holds channels q * 64 + lane
__device__ __forceinline__ void sa2_sort_centre(const int (&a)[4], const float (&g)[4], int* hist, int* run, int lane,
                                                float* __restrict__ eg, int32_t* __restrict__ ec) {
  hist[lane] = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) atomicAdd(&hist[a[q]], 1);
  const int tot = hist[lane];             // entries of sample `lane` (LDS operations of a wave complete in order)
  int incl = tot;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  run[lane] = incl - tot;                 // where the sample's entries start
  const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    unsigned long long m = ~0ull;
#pragma unroll
    for (int bit = 0; bit < 6; ++bit) {
      const unsigned long long bal = __ballot((a[q] >> bit) & 1);
      m &= ((a[q] >> bit) & 1) ? bal : ~bal;
    }
    const int r = (int)__builtin_popcountll(m & below), n = (int)__builtin_popcountll(m);
    const int start = run[a[q]];
    const int pos = start + r;
    if (r == 0) run[a[q]] = start + n;    // one lane per sample: the next round's entries follow
    const int end = __shfl(incl, a[q], 64);       // one past the sample's last entry
    ec[pos] = (q * 64 + lane) | (a[q] << 16) | (pos == end - 1 ? (int)0x80000000 : 0);
    eg[pos] = g[q];
  }
}

__global__ __launch_bounds__(256) void sa2_sort_kernel(const float* __restrict__ gz, const int32_t* __restrict__ argt,
                                                       float* __restrict__ ent_g, int32_t* __restrict__ ent_c, long centres) {
  __shared__ int s_hist[4][64], s_run[4][64];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long c = (long)blockIdx.x * 4 + wave;
  if (c >= centres) return;                       // (whole waves leave: no barrier below)
  int a[4];
  float g[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    a[q] = argt[c * S2_C + q * 64 + lane] & 63;
    g[q] = gz[c * S2_C + q * 64 + lane];
  }
  sa2_sort_centre(a, g, s_hist[wave], s_run[wave], lane, ent_g + c * S2_C, ent_c + c * S2_C);
}

// The same lists straight from the channel-major tensors the level-3 backward leaves (round 5): d out2, out2 (the relu gate)
// and arg2, all [B][256][M] -- the gated transpose of the gradient, the transpose of the arg-max table and the sort were three
// launches (20 + 18 + 30 us) around two [B,128,256] tensors.  A workgroup takes 32 centres of one instance: the [256][32]
// tiles arrive as 128-byte rows, cross through LDS (value rows of 257 floats, samples as bytes), and each of 16 wavefronts
// sorts two centres.
constexpr int S2_SORT_M = 32, S2_SORT_W = 16;   // 16 wavefronts: two centres each
__global__ __launch_bounds__(64 * S2_SORT_W) void sa2_sort_cm_kernel(const float* __restrict__ dout, const float* __restrict__ outp,
                                                                     const int32_t* __restrict__ arg, float* __restrict__ ent_g,
                                                                     int32_t* __restrict__ ent_c, int M) {
  __shared__ float s_g[S2_SORT_M][S2_C + 1];
  __shared__ unsigned char s_a[S2_SORT_M][S2_C + 4];
  __shared__ int s_hist[S2_SORT_W][64], s_run[S2_SORT_W][64];
  const int b = blockIdx.y, m0 = blockIdx.x * S2_SORT_M, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tx = tid & 31, ty = tid >> 5;   // 32 channel rows per pass
  const size_t base = (size_t)b * S2_C * M;
#pragma unroll
  for (int i = 0; i < S2_C / (2 * S2_SORT_W); ++i) {
    const int ch = ty + 2 * S2_SORT_W * i, m = m0 + tx;
    if (m < M) {
      const size_t e = base + (size_t)ch * M + m;
      s_g[tx][ch] = outp[e] > 0.f ? dout[e] : 0.f;
      s_a[tx][ch] = (unsigned char)(arg[e] & 63);
    }
  }
  __syncthreads();
  for (int i = 0; i < S2_SORT_M / S2_SORT_W; ++i) {
    const int ml = wave * (S2_SORT_M / S2_SORT_W) + i;
    if (m0 + ml >= M) break;                       // (wave-uniform)
    int a[4];
    float g[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = s_a[ml][q * 64 + lane];
      g[q] = s_g[ml][q * 64 + lane];
    }
    const size_t c = (size_t)b * M + m0 + ml;
    sa2_sort_centre(a, g, s_hist[wave], s_run[wave], lane, ent_g + c * S2_C, ent_c + c * S2_C);
  }
}

struct Sa2BwdArgs {
  const float* ent_g;              // [centres][256]
  const int32_t* ent_c;            // [centres][256]
  const float* W2;                 // [256][128]
  const _Float16* w1img;           // W1^T as a fragment image (frag_image_kernel)
  const float* w1un;               // [1]: 1 / the image's power-of-two scale
  const unsigned long long* m1;    // [centres][128]: a1 > 0
  const unsigned long long* m0;    // [centres][128]: a0 > 0
  float* da0;                      // [B][128][M * 64]
  int B, M;                        // M centres per instance (even)
};

constexpr int sa2_bwd_lds() { return 2 * 64 * S2_PT * 4 + 16 * 4; }

template <int MODE>   // 0 = shipped; 1 / 2 / 3: without phase 1 / phase 2 / the stores (tools/ub/sa2_ub.hip)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void sa2_bwd_kernel(Sa2BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s2_sm[];
  float* s_tile = reinterpret_cast<float*>(s2_sm);   // [2 centres][64 samples][S2_PT]
  float* s_red = s_tile + 2 * 64 * S2_PT;            // [4] tile maxima of the waves
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float unW = a.w1un[0];
  const int ci = wave >> 1, hf = wave & 1;
  float* tile = s_tile + ci * 64 * S2_PT;
  const long pairs = (long)a.B * a.M / 2;
  const int ldY = a.M * 64;
  const int k = 64 * hf + lane;

  // Two workgroups per CU (68 KB of LDS each): one is on the VALU (phase 1) while the other is on the matrix core.
  // Everything a pair needs from memory is requested ahead: its entries and gate words one iteration early, its first
  // S2_PF rows of W2 while the pair before it is in phase 2, the W1^T fragments one k-step ahead.
  struct PairRegs {
    int vc[4];
    float vg[4];
    unsigned long long mk, gm;   // gates: a1 of row k (phase 1), a0 of row 64 hf + lane (epilogue)
  };
  auto load_pair = [&](long p, PairRegs& r) {
    const long centre = 2 * p + ci;
    r.mk = a.m1[centre * S2_K + k];
    r.gm = a.m0[centre * S2_K + k];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      r.vc[q] = a.ent_c[centre * S2_C + q * 64 + lane];
      r.vg[q] = a.ent_g[centre * S2_C + q * 64 + lane];
    }
  };
  auto w2row = [&](int cw) { return (a.W2 + (size_t)(cw & 0xffff) * S2_K)[k]; };   // uniform row + lane offset
  const half8* img = reinterpret_cast<const half8*>(a.w1img) + (size_t)(2 * hf) * 8 * 2 * 64 + lane;   // row tiles 2 hf, 2 hf + 1
  auto load_a = [&](int c, half8 (&f)[4]) {   // fragments (t, piece) of k-step c
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int p = 0; p < 2; ++p) f[2 * t + p] = img[(size_t)((t * 8 + c) * 2 + p) * 64];
  };

  PairRegs cur, nxt;
  float wv[S2_PF];
  int cwr[S2_PF];   // the entries whose W2 rows are in flight (uniform)
  if ((long)blockIdx.x < pairs) {
    load_pair(blockIdx.x, cur);
#pragma unroll
    for (int u = 0; u < S2_PF; ++u) {
      cwr[u] = s2_rl(cur.vc[0], u);
      wv[u] = w2row(cwr[u]);
    }
  }

  for (long p = blockIdx.x; p < pairs; p += gridDim.x) {
    const long centre = 2 * p + ci;
    load_pair(p + gridDim.x < pairs ? p + gridDim.x : p, nxt);
    half8 af[2][4];
    load_a(0, af[0]);
    // ---- phase 1: this wave's 64 k of the centre's tile
    if (MODE != 1) {
#pragma unroll 8
      for (int s = 0; s < 64; ++s) tile[s * S2_PT + k] = 0.f;     // samples no channel points at
      float acc = 0.f, mx = 0.f;
#pragma unroll
      for (int e = 0; e < S2_C; ++e) {
        const int u = e % S2_PF;
        const int cw = cwr[u];
        const float g = s2_rlf(cur.vg[e >> 6], e & 63);
        acc = __builtin_fmaf(g, wv[u], acc);
        if (e + S2_PF < S2_C) {
          cwr[u] = s2_rl(cur.vc[(e + S2_PF) >> 6], (e + S2_PF) & 63);
          wv[u] = w2row(cwr[u]);
        }
        if (__builtin_expect(cw < 0, 0)) {   // wave-uniform: the sample's last channel
          const int col = (cw >> 16) & 63;
          // a1 > 0 of (row k, sample col): bit col of the lane's gate word -> 0 / -1 by a bit-field extract, AND
          const int gbit = __builtin_amdgcn_sbfe(col < 32 ? (int)(unsigned)cur.mk : (int)(unsigned)(cur.mk >> 32), (unsigned)(col & 31), 1u);
          const float val = __int_as_float(__float_as_int(acc) & gbit);
          tile[col * S2_PT + k] = val;
          mx = fmaxf(mx, __builtin_fabsf(val));
          acc = 0.f;
        }
      }
      mx = wave_max(mx);
      if (lane == 0) s_red[wave] = mx;
    }
    // the next pair's first rows of W2: in flight across phase 2
#pragma unroll
    for (int u = 0; u < S2_PF; ++u) {
      cwr[u] = s2_rl(nxt.vc[0], u);
      wv[u] = w2row(cwr[u]);
    }
    __syncthreads();
    // ---- phase 2: rows 64 hf .. 64 hf + 63 of d a0 for the centre's 64 samples
    if (MODE != 2) {
      const unsigned Ex = s2_exp(fmaxf(s_red[2 * ci], s_red[2 * ci + 1]));
      const float sx = s2_scale(Ex);
      f32x16 acc[2][2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[cb][t][r] = 0.f;
      const float* brow = tile + (lane & 31) * S2_PT + (lane >> 5) * 8;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        if (c + 1 < 8) load_a(c + 1, af[(c + 1) & 1]);
        half8 xh[2], xl[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          const float4 b0 = *reinterpret_cast<const float4*>(brow + cb * 32 * S2_PT + c * 16);
          const float4 b1 = *reinterpret_cast<const float4*>(brow + cb * 32 * S2_PT + c * 16 + 4);
          const float x[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
          s2_split8(x, sx, xh[cb], xl[cb]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const half8 wh = af[c & 1][2 * t], wl = af[c & 1][2 * t + 1];
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[cb], acc[cb][t], 0, 0, 0);
            acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[cb], acc[cb][t], 0, 0, 0);
            acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[cb], acc[cb][t], 0, 0, 0);
          }
        }
      }
      const float unscale = s2_unscale(Ex) * unW;
      const int b = (int)(centre / a.M), m = (int)(centre - (long)b * a.M);
      float* Y = a.da0 + (size_t)b * S2_K * ldY + (size_t)m * 64 + lane;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v[8];
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            v[i] = acc[0][t][4 * g + i];
            v[4 + i] = acc[1][t][4 * g + i];
            s2_swap32(v[i], v[4 + i]);    // v[i]: row base + i, v[4 + i]: row base + 4 + i, lane = sample
          }
          const int row0 = 64 * hf + 32 * t + 8 * g;
#pragma unroll
          for (int i = 0; i < 8; ++i) {   // the row's gate word sits in lane 32 t + 8 g + i
            const unsigned glo = (unsigned)s2_rl((int)(unsigned)cur.gm, 32 * t + 8 * g + i);
            const unsigned ghi = (unsigned)s2_rl((int)(unsigned)(cur.gm >> 32), 32 * t + 8 * g + i);
            const bool on = ((lane < 32 ? glo >> lane : ghi >> (lane - 32)) & 1u) != 0u;
            if (MODE != 3) Y[(size_t)(row0 + i) * ldY] = on ? v[i] * unscale : 0.f;
          }
        }
    }
    __syncthreads();   // the tiles are rewritten by the next pair
    cur = nxt;
  }
}


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
  unsigned long long* m0;          // [B * M][128]
  unsigned long long* m1;
  int B, N1, M;
};

constexpr int sa2_fwd_lds() { return 2 * 2 * 64 * S2_PH + (8 + S2_K + S2_C) * 4; }

template <int MODE, int S2_RING = 2>   // MODE 0 = shipped; tools/ub/sa2f_ub.hip: 1 no gather, 2 no W1 MFMAs, 3 no W2 MFMAs,
                                       // 4 no pooled stores, 5 no gate stores, 6 weight loads hoisted; S2_RING: k-steps in flight
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void sa2_fwd_kernel(Sa2FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s2_sm[];
  float* s_red = reinterpret_cast<float*>(s2_sm + 2 * 2 * 64 * S2_PH);   // [0..3] a0 maxima, [4..7] a1 maxima
  float* s_b1 = s_red + 8;
  float* s_b2 = s_b1 + S2_K;
  const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ci = wave >> 1, hf = wave & 1;
  unsigned char* thi = s2_sm + ci * 2 * 64 * S2_PH;
  unsigned char* tlo = thi + 64 * S2_PH;
  for (int e = tid; e < S2_K; e += 256) s_b1[e] = a.b1[e];
  for (int e = tid; e < S2_C; e += 256) s_b2[e] = a.b2[e];
  const float un1 = a.un1[0], un2 = a.un2[0];
  // fragment (tile, c, piece) of an image sits at ((tile * 8 + c) * 2 + piece) * 64 + lane (in 16-byte elements)
  const half8* A1 = reinterpret_cast<const half8*>(a.img1) + (size_t)(2 * hf) * 8 * 2 * 64 + lane;
  const half8* B2 = reinterpret_cast<const half8*>(a.img2) + (size_t)(4 * hf) * 8 * 2 * 64 + lane;
  // The 24 k-steps of a pair -- 8 of W1 (steps 0-7), 2 x 8 of W2 (channel tiles 2 pass + tt) -- take their four weight
  // fragments from a ring S2_RING steps deep: a step's matrix work (12 MFMAs, ~0.16 us) is far shorter than an L2 round trip.
  auto load_step = [&](int step, half8 (&f)[4]) {
    if (step < 8) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          f[2 * t + p] = MODE == 6 ? A1[(t * 2 + p) * 64] * (_Float16)step : A1[(size_t)((t * 8 + step) * 2 + p) * 64];
    } else {
      const int pass = (step - 8) >> 3, c = (step - 8) & 7;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int p = 0; p < 2; ++p)
          f[2 * tt + p] = MODE == 6 ? B2[(tt * 2 + p) * 64] * (_Float16)step : B2[(size_t)(((2 * pass + tt) * 8 + c) * 2 + p) * 64];
    }
  };
  const long pairs = (long)a.B * a.M / 2;
  // a pair's sample indices and shifts are requested one iteration ahead (the gather itself waits on the indices; the
  // other workgroup of the CU covers it)
  int gi = 0, gi_n = 0;
  float shv = 0.f, shv_n = 0.f;
  auto request = [&](long p, int& i, float& sh) {
    const long centre = 2 * p + ci;
    const int b = (int)(centre / a.M), m = (int)(centre - (long)b * a.M);
    i = a.gidx[centre * 64 + lane];
    sh = a.shift[((size_t)b * S2_K + 64 * hf + lane) * a.M + m];
  };
  if ((long)blockIdx.x < pairs) request(blockIdx.x, gi, shv);
  __syncthreads();

  for (long p = blockIdx.x; p < pairs; p += gridDim.x) {
    const long centre = 2 * p + ci;
    const int b = (int)(centre / a.M), m = (int)(centre - (long)b * a.M);
    float4 gv[16];
    {
      const float4* src = reinterpret_cast<const float4*>(a.rT + ((size_t)b * a.N1 + gi) * S2_K + 64 * hf);
#pragma unroll
      for (int q = 0; q < 16; ++q) gv[q] = MODE == 1 ? make_float4(0.1f * q, 0.2f, -0.1f, 0.3f) : src[q];
    }
    request(p + gridDim.x < pairs ? p + gridDim.x : p, gi_n, shv_n);
    half8 fr[S2_RING][4];
#pragma unroll
    for (int i = 0; i + 1 < S2_RING; ++i) load_step(i, fr[i]);
    // ---- A: a0 = relu(r[sample] + shift), this wave's 64 channels of its centre
    float mx = 0.f;
    {
      float v[64];
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        v[4 * q] = gv[q].x;
        v[4 * q + 1] = gv[q].y;
        v[4 * q + 2] = gv[q].z;
        v[4 * q + 3] = gv[q].w;
      }
      unsigned long long gw = 0ull;
#pragma unroll
      for (int c = 0; c < 64; ++c) {
        v[c] = fmaxf(v[c] + s2_rlf(shv, c), 0.f);
        mx = fmaxf(mx, v[c]);
        const unsigned long long bal = __ballot(v[c] > 0.f);
        gw = lane == c ? bal : gw;
      }
      if (MODE != 5) a.m0[centre * S2_K + 64 * hf + lane] = gw;
      mx = wave_max(mx);
      if (lane == 0) s_red[wave] = mx;
      __syncthreads();
      const unsigned Ea = s2_exp(fmaxf(s_red[2 * ci], s_red[2 * ci + 1]));
      const float sa = s2_scale(Ea);
#pragma unroll
      for (int j8 = 0; j8 < 8; ++j8) {
        half8 hh, ll;
        {
          const float x8[8] = {v[8 * j8], v[8 * j8 + 1], v[8 * j8 + 2], v[8 * j8 + 3], v[8 * j8 + 4], v[8 * j8 + 5], v[8 * j8 + 6], v[8 * j8 + 7]};
          s2_split8(x8, sa, hh, ll);
        }
        *reinterpret_cast<half8*>(thi + lane * S2_PH + (64 * hf + 8 * j8) * 2) = hh;
        *reinterpret_cast<half8*>(tlo + lane * S2_PH + (64 * hf + 8 * j8) * 2) = ll;
      }
      mx = s2_unscale(Ea);   // carried to phase B
    }
    __syncthreads();
    // ---- B: a1 rows 64 hf .. 64 hf + 63 of the centre
    {
      f32x16 acc[2][2];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[cb][t][r] = 0.f;
      const unsigned char* xrow = thi + l31 * S2_PH + h * 16;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        load_step(c + S2_RING - 1, fr[(c + S2_RING - 1) % S2_RING]);
        half8 xh[2], xl[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
          xh[cb] = *reinterpret_cast<const half8*>(xrow + cb * 32 * S2_PH + c * 32);
          xl[cb] = *reinterpret_cast<const half8*>(xrow + cb * 32 * S2_PH + c * 32 + 64 * S2_PH);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const half8 wh = fr[c % S2_RING][2 * t], wl = fr[c % S2_RING][2 * t + 1];
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            if (MODE == 2) {
              acc[cb][t][0] += (float)wh[0] + (float)xh[cb][0] + (float)wl[0] + (float)xl[cb][0];
              continue;
            }
            acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xh[cb], acc[cb][t], 0, 0, 0);
            acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, xl[cb], acc[cb][t], 0, 0, 0);
            acc[cb][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, xh[cb], acc[cb][t], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);   // one k-step of operands in flight, not eight (256 VGPRs)
      }
      // acc[cb][t][r]: row 64 hf + 32 t + (r&3) + 8 (r>>2) + 4 h, sample 32 cb + l31
      const float un = mx * un1;
      float mh = 0.f;
      unsigned long long gw = 0ull;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rr = 32 * t + (r & 3) + 8 * (r >> 2);   // + 4 h
          const float bias = s_b1[64 * hf + rr + 4 * h];
          const float o0 = fmaxf(acc[0][t][r] * un + bias, 0.f), o1 = fmaxf(acc[1][t][r] * un + bias, 0.f);
          acc[0][t][r] = o0;
          acc[1][t][r] = o1;
          mh = fmaxf(mh, fmaxf(o0, o1));
          const unsigned long long k0 = __ballot(o0 > 0.f), k1 = __ballot(o1 > 0.f);
          const unsigned long long w0 = (k0 & 0xffffffffull) | (k1 << 32);            // row rr: samples 0-31 | 32-63
          const unsigned long long w1 = (k0 >> 32) | (k1 & 0xffffffff00000000ull);    // row rr + 4
          gw = lane == rr ? w0 : (lane == rr + 4 ? w1 : gw);
        }
      if (MODE != 5) a.m1[centre * S2_K + 64 * hf + lane] = gw;
      mh = wave_max(mh);
      if (lane == 0) s_red[4 + wave] = mh;
      __syncthreads();   // every wave is done reading a0; the maxima are visible
      const unsigned Eh = s2_exp(fmaxf(s_red[4 + 2 * ci], s_red[5 + 2 * ci]));
      const float sh = s2_scale(Eh);
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            typedef _Float16 half4 __attribute__((ext_vector_type(4)));
            half4 hh, ll;
            {
              unsigned h0, l0, h1, l1;
              s2_split2(acc[cb][t][4 * g4], acc[cb][t][4 * g4 + 1], sh, h0, l0);
              s2_split2(acc[cb][t][4 * g4 + 2], acc[cb][t][4 * g4 + 3], sh, h1, l1);
              typedef unsigned uint2v __attribute__((ext_vector_type(2)));
              hh = __builtin_bit_cast(half4, uint2v{h0, h1});
              ll = __builtin_bit_cast(half4, uint2v{l0, l1});
            }
            const int k0 = 64 * hf + 32 * t + 8 * g4 + 4 * h;
            *reinterpret_cast<half4*>(thi + (32 * cb + l31) * S2_PH + k0 * 2) = hh;
            *reinterpret_cast<half4*>(tlo + (32 * cb + l31) * S2_PH + k0 * 2) = ll;
          }
      mx = s2_unscale(Eh);
    }
    __syncthreads();
    // ---- C: pooled channels 128 hf + 64 pass .. + 63, transposed
    {
      const float un = mx * un2;
      const unsigned char* xrow = thi + l31 * S2_PH + h * 16;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        f32x16 acc[2][2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
          for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[cb][tt][r] = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const int step = 8 + pass * 8 + c;
          if (step + S2_RING - 1 < 24) load_step(step + S2_RING - 1, fr[(step + S2_RING - 1) % S2_RING]);
          half8 xh[2], xl[2];
#pragma unroll
          for (int cb = 0; cb < 2; ++cb) {
            xh[cb] = *reinterpret_cast<const half8*>(xrow + cb * 32 * S2_PH + c * 32);
            xl[cb] = *reinterpret_cast<const half8*>(xrow + cb * 32 * S2_PH + c * 32 + 64 * S2_PH);
          }
#pragma unroll
          for (int tt = 0; tt < 2; ++tt) {
            const half8 wh = fr[step % S2_RING][2 * tt], wl = fr[step % S2_RING][2 * tt + 1];
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
              if (MODE == 3) {
                acc[cb][tt][0] += (float)wh[0] + (float)xh[cb][0] + (float)wl[0] + (float)xl[cb][0];
                continue;
              }
              acc[cb][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[cb], wh, acc[cb][tt], 0, 0, 0);
              acc[cb][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl[cb], wh, acc[cb][tt], 0, 0, 0);
              acc[cb][tt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh[cb], wl, acc[cb][tt], 0, 0, 0);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
        // acc[cb][tt][r]: channel 128 hf + 64 pass + 32 tt + l31, sample 32 cb + (r&3) + 8 (r>>2) + 4 h (ascending)
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          float v = -__builtin_inff();
          int smp = 0;
#pragma unroll
          for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const bool gt = acc[cb][tt][r] > v;
              v = gt ? acc[cb][tt][r] : v;
              smp = gt ? 32 * cb + mfma_row(r, lane) : smp;
            }
          const float ov = __shfl_xor(v, 32, 64);
          const int os = __shfl_xor(smp, 32, 64);
          const bool take = ov > v || (ov == v && os < smp);
          v = take ? ov : v;
          smp = take ? os : smp;
          if (lane < 32 && (MODE != 4 || v == 123.f)) {
            const int ch = 128 * hf + 64 * pass + 32 * tt + lane;
            const size_t e = ((size_t)b * S2_C + ch) * a.M + m;
            a.out[e] = fmaxf(v * un + s_b2[ch], 0.f);     // the (positive) scale commutes with the max
            a.arg[e] = smp;
          }
        }
      }
    }
    __syncthreads();   // the images are rewritten by the next pair
    gi = gi_n;
    shv = shv_n;
  }
}


// ------------------------------------------------------------------------------------------------------------------
// The forward as ONE 8-wave workgroup per CU (the shipped form; sa2_fwd_kernel above is the first version, kept for
// tools/ub/sa2f_ub.hip).  What bounded the 4-wave / two-workgroups-per-CU form was weight traffic per wave: both waves of
// a centre streamed all of W1 and half of W2 through a two-step register ring (a deeper ring spilled at 256 VGPRs), so part
// of every L2 round trip was exposed.  Here
//   * W1's fragment image (64 KB) is copied into LDS once per workgroup: layer 1 reads its A fragments from LDS;
//   * W2 is split eight ways: wave w owns channel tile w (32 channels) for BOTH centres, so every fragment of W2 is
//     loaded once per pair and CU (16 x 1 KB per wave), through a ring S8_RING k-steps deep;
//   * layer 1 is split as (centre, 32-row tile) over the eight waves; the gather as (centre, 32-channel quarter).
// Same arithmetic and the same per-centre scales as the first version: bit-identical results.
constexpr int S8_RING = 4;
constexpr int sa2_fwd8_lds() { return S2_K * S2_K * 4 + 2 * 2 * 64 * S2_PH + (16 + S2_K + S2_C) * 4; }

template <int MODE>   // 0 = shipped; tools/ub/sa2f_ub.hip: 1 no W2 MFMAs, 2 no W1 MFMAs, 3 no pooled stores, 4 no gather, 5 no syncs C
__global__ __launch_bounds__(512) void sa2_fwd8_kernel(Sa2FwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s2_sm[];
  half8* s_w1 = reinterpret_cast<half8*>(s2_sm);                      // fragment image of W1: [(tile * 8 + c) * 2 + piece][lane]
  unsigned char* s_img = s2_sm + S2_K * S2_K * 4;                     // [2 centres][hi / lo][64 samples][S2_PH]
  float* s_red = reinterpret_cast<float*>(s_img + 2 * 2 * 64 * S2_PH);   // [0..7] a0 maxima, [8..15] a1 maxima (per wave)
  float* s_b1 = s_red + 16;
  float* s_b2 = s_b1 + S2_K;
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
  if ((int)blockIdx.x < pairs) request(blockIdx.x);
  __syncthreads();

  for (int p = blockIdx.x; p < pairs; p += gridDim.x) {
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
      unsigned long long gw = 0ull;
      float mx = 0.f;
#pragma unroll
      for (int c = 0; c < 32; ++c) {
        v[c] = fmaxf(v[c] + s2_rlf(shv, c), 0.f);
        mx = fmaxf(mx, v[c]);
        const unsigned long long bal = __ballot(v[c] > 0.f);
        gw = lane == c ? bal : gw;
      }
      if (lane < 32) a.m0[centre * S2_K + 32 * qt + lane] = gw;
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
      unsigned long long gw = 0ull;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rr = (r & 3) + 8 * (r >> 2);   // + 4 h
        const float bias = s_b1[32 * qt + rr + 4 * h];
        const float o0 = fmaxf(acc[0][r] * un + bias, 0.f), o1 = fmaxf(acc[1][r] * un + bias, 0.f);
        acc[0][r] = o0;
        acc[1][r] = o1;
        mh = fmaxf(mh, fmaxf(o0, o1));
        const unsigned long long k0 = __ballot(o0 > 0.f), k1 = __ballot(o1 > 0.f);
        const unsigned long long w0 = (k0 & 0xffffffffull) | (k1 << 32);            // row rr: samples 0-31 | 32-63
        const unsigned long long w1 = (k0 >> 32) | (k1 & 0xffffffff00000000ull);    // row rr + 4
        gw = lane == rr ? w0 : (lane == rr + 4 ? w1 : gw);
      }
      if (lane < 32) a.m1[centre * S2_K + 32 * qt + lane] = gw;
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
    if (p + gridDim.x < pairs) request(p + gridDim.x);
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
          const int b = (2 * p) / a.M, m = 2 * p + cc - b * a.M;
          const int ch = 32 * wave + lane;
          const size_t e = ((size_t)b * S2_C + ch) * a.M + m;
          a.out[e] = fmaxf(v * un + s_b2[ch], 0.f);     // the (positive) scale commutes with the max
          a.arg[e] = smp;
        }
      }
    }
    __syncthreads();   // the images and s_red are rewritten by the next pair
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

int launch_sa2_sort(const float* gz, const int32_t* argt, float* ent_g, int32_t* ent_c, long centres, hipStream_t s) {
  hipLaunchKernelGGL(sa2_sort_kernel, dim3((unsigned)((centres + 3) / 4)), dim3(256), 0, s, gz, argt, ent_g, ent_c, centres);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

int launch_sa2_sort_cm(const float* dout, const float* outp, const int32_t* arg, float* ent_g, int32_t* ent_c, int B, int M,
                       hipStream_t s) {
  hipLaunchKernelGGL(sa2_sort_cm_kernel, dim3((M + S2_SORT_M - 1) / S2_SORT_M, B), dim3(64 * S2_SORT_W), 0, s, dout, outp, arg, ent_g, ent_c, M);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

int launch_frag_image(const float* W, int R, int K, void* img, float* un, hipStream_t s) {
  if (R % 32 != 0 || K % 16 != 0) return GEOA3_EINVAL;
  hipLaunchKernelGGL(frag_image_kernel, dim3(1), dim3(256), 0, s, W, R, K, static_cast<_Float16*>(img), un);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

int launch_sa2_bwd(const float* ent_g, const int32_t* ent_c, const float* W2, const void* w1t_img, const float* w1t_un,
                   const unsigned long long* m1, const unsigned long long* m0, float* da0, int B, int M, hipStream_t s) {
  if (M % 2 != 0) return GEOA3_ENOSUPPORT;
  Sa2BwdArgs a{ent_g, ent_c, W2, static_cast<const _Float16*>(w1t_img), w1t_un, m1, m0, da0, B, M};
  const long pairs = (long)B * M / 2;
  const int lds = sa2_bwd_lds();
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa2_bwd_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const unsigned grid = (unsigned)(pairs < 512 ? pairs : 512);   // two workgroups per CU, persistent
  hipLaunchKernelGGL(sa2_bwd_kernel<0>, dim3(grid), dim3(256), lds, s, a);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

int launch_sa2_fwd(const float* rT, const int32_t* gidx, const float* shift, const void* w1_img, const float* w1_un,
                   const float* b1, const void* w2_img, const float* w2_un, const float* b2, float* out, int32_t* arg,
                   unsigned long long* m0, unsigned long long* m1, int B, int N1, int M, hipStream_t s) {
  if (M % 2 != 0 || (long)B * M > 0x3fffffffL) return GEOA3_ENOSUPPORT;
  Sa2FwdArgs a{rT, gidx, shift, b1, b2, static_cast<const _Float16*>(w1_img), w1_un, static_cast<const _Float16*>(w2_img),
               w2_un, out, arg, m0, m1, B, N1, M};
  const long pairs = (long)B * M / 2;
  const int lds = sa2_fwd8_lds();
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sa2_fwd8_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  const unsigned grid = (unsigned)(pairs < 256 ? pairs : 256);   // one 8-wave workgroup per CU, persistent
  hipLaunchKernelGGL(sa2_fwd8_kernel<0>, dim3(grid), dim3(512), lds, s, a);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
