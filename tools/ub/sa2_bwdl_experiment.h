// EXPERIMENT (round 5), not part of the library: the level-2 backward of PointNet++ SSG with both weight matrices on
// chip.  Built only by tools/ub/sa2_ub.hip, which includes geoa3_amd/csrc/pointnet2_sa2.hip first (the helpers
// s2_split2, s2_exp, ... and the shipped sa2_bwd_kernel it is compared with).  Result: same values,
// 560-640 us against 670-700 us on uniformly random arg-max samples, but 634 us against 600 us inside configs[3], where
// ball queries with few distinct points put most of a centre's 256 entries on a few samples.
#pragma once
#include <type_traits>
namespace {

__global__ __launch_bounds__(256) void sa2_offsets_kernel(const int32_t* __restrict__ ent_c, unsigned short* __restrict__ ent_o,
                                                          long centres) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;   // (centre, sample): entries sorted by sample -> first entry
  if (i >= centres * 64) return;
  const long c = i >> 6;
  const int smp = (int)(i & 63);
  int below = 0;
  for (int e = 0; e < S2_C; ++e) below += ((ent_c[c * S2_C + e] >> 16) & 63) < smp;
  ent_o[i] = (unsigned short)below;
}

// ---- the same backward with BOTH weight matrices on chip (round 5) ---------------------------------------------------
// sa2_bwd_kernel above re-reads, per centre, the 256 rows of W2 its entries name (128 KB) and the W1^T fragment image
// (64 KB) from L2: 6.3 GB per launch at B = 250, and its phases add up instead of overlapping (profiles/round5_*).  Here
// one 16-wave workgroup per CU keeps W2 (fp32, 128 KB) in LDS and W1^T as split-fp16 fragments in registers (a wave's
// 16 rows x 128 k: 32 registers), and walks its centres in HALVES of 32 samples (the operand images are 16.5 KB):
//   phase 1 (wave = 4 samples of the centre, lane = 2 k): a sample's entries are a contiguous run of the sorted list
//     (sa2_sort_kernel's per-sample offsets, `ent_o`), held in registers; eight rows of W2 in flight from LDS; the same
//     ascending-channel fma chain per (sample, k) as above, gated; per half: maximum -> power-of-two scale, each wave
//     splits ITS two rows once into the hi / lo fp16 images (no operand is split twice);
//   phase 2 (wave = 16 rows x 16 samples on v_mfma_f32_16x16x32_f16, 12 per half): ready-made operands from LDS; gated
//     by bits, stored as 64-byte row segments.
// What is left of the memory traffic is the lists (6 KB per centre) and the 1.05 GB of d a0.
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Sa2BwdLArgs {
  const float* ent_g;              // [centres][256]
  const int32_t* ent_c;            // [centres][256]
  const unsigned short* ent_o;     // [centres][64]: first entry of each sample (0 .. 256)
  const float* W2;                 // [256][128]
  const float* W1T;                // [128][128]: W1^T
  const unsigned long long* m1;    // [centres][128]: a1 > 0
  const unsigned long long* m0;    // [centres][128]: a0 > 0
  float* da0;                      // [B][128][M * 64]
  int B, M;
  long long* dbg;                  // MODE 4 (tools/ub/sa2_ub.hip): cycles per section of wave 0 of workgroup 0
};

constexpr int S2L_WAVES = 16;
constexpr int S2L_EO = 68;               // ints per offset list (65 used: [64] = 256)
constexpr int S2L_EN = 8 + S2_C;         // (g, channel) pairs per entry list + 8 that may be read, never used
constexpr int S2L_PI = 132;              // words per sample row of the fp16 images: 64 hi pairs, 64 lo pairs, 4 pad
constexpr int sa2_bwdl_lds() { return (S2_C * S2_K + 32 * S2L_PI + 2 * S2_C * 2 + 2 * S2L_EN * 2 + 2 * S2L_EO + 2 * S2L_WAVES) * 4; }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for the wave's outstanding GLOBAL accesses
// (s_waitcnt vmcnt(0)): with one workgroup per CU every barrier would sit out the write acknowledgements of the d a0
// stores before it.  Nothing below communicates through global memory inside a launch.
__device__ __forceinline__ void s2_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int MODE>   // 0 = shipped; 1 / 2 / 3: without phase 1 / phase 2 / the stores (tools/ub/sa2_ub.hip)
__global__ __launch_bounds__(64 * S2L_WAVES) void sa2_bwdl_kernel(Sa2BwdLArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char s2_sm[];
  float* s_w2 = reinterpret_cast<float*>(s2_sm);                                          // [256][128]
  unsigned* s_img = reinterpret_cast<unsigned*>(s_w2 + S2_C * S2_K);                      // [32 samples][S2L_PI]
  unsigned long long* s_gt = reinterpret_cast<unsigned long long*>(s_img + 32 * S2L_PI);  // [2][128 m1 words, 128 m0 words]
  float2v* s_en = reinterpret_cast<float2v*>(s_gt + 2 * S2_C);                            // [2][S2L_EN] (g, channel word)
  int* s_eo = reinterpret_cast<int*>(s_en + 2 * S2L_EN);                                  // [2][S2L_EO]
  float* s_red = reinterpret_cast<float*>(s_eo + 2 * S2L_EO);                             // [2 halves][16]: wave maxima
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = wave & 7, cg = wave >> 3, q4 = lane >> 4, p16 = lane & 15;
  const long centres = (long)a.B * a.M;
  const int ldY = a.M * 64;

  // ---- once: W2 -> LDS; W1^T -> scale, split, this wave's fragments
  for (int e = tid; e < S2_C * S2_K / 4; e += 64 * S2L_WAVES)
    reinterpret_cast<float4*>(s_w2)[e] = reinterpret_cast<const float4*>(a.W2)[e];
  {
    float m = 0.f;
    for (int e = tid; e < S2_K * S2_K / 4; e += 64 * S2L_WAVES) {
      const float4 v = reinterpret_cast<const float4*>(a.W1T)[e];
      m = fmaxf(fmaxf(m, fmaxf(__builtin_fabsf(v.x), __builtin_fabsf(v.y))), fmaxf(__builtin_fabsf(v.z), __builtin_fabsf(v.w)));
    }
    m = wave_max(m);
    if (lane == 0) s_red[wave] = m;
  }
  // a centre's lists: entries (threads 0..255), sample offsets (256..320), gate words (384..639), through two registers
  // (ONE two-register vector for all three kinds: with separate scalars the 8-byte gate load lands in a register pair
  // that has to be copied into them at once, and the copy waits out the whole memory latency)
  typedef unsigned uint2v __attribute__((ext_vector_type(2)));
  typedef uint2v Lists;   // entry: (g, channel word); offset: (-, offset); gate word: (lo, hi)
  const int t2 = tid - 384;
  auto fetch = [&](long c, Lists& r) {
    if (tid < S2_C) {
      r[0] = __float_as_uint(a.ent_g[c * S2_C + tid]);
      r[1] = (unsigned)a.ent_c[c * S2_C + tid];
    } else if (tid < S2_C + 64) {
      r[1] = a.ent_o[c * 64 + tid - S2_C];
    } else if (t2 >= 0 && t2 < S2_C) {
      r = *reinterpret_cast<const uint2v*>((t2 < S2_K ? a.m1 + c * S2_K : a.m0 + c * S2_K - S2_K) + t2);
    }
  };
  auto stage = [&](const Lists& r, int buf) {
    if (tid < S2_C) {
      float2v e;
      e[0] = __uint_as_float(r[0]);
      e[1] = __uint_as_float((r[1] & 0xffffu) << 9);   // the channel as the byte offset of its row of W2 in LDS
      s_en[buf * S2L_EN + 8 + tid] = e;
    } else if (tid < S2_C + 64) {
      s_eo[buf * S2L_EO + tid - S2_C] = (int)r[1];
    } else if (tid == S2_C + 64) {
      s_eo[buf * S2L_EO + 64] = S2_C;
    } else if (t2 >= 0 && t2 < S2_C) {
      *reinterpret_cast<uint2v*>(s_gt + buf * S2_C + t2) = r;
    }
  };
  // The lists of centre c + 2 G are staged at the END of iteration c (behind its last barrier: nobody reads the lists of
  // c any more) from registers fetched an iteration earlier; the wait in front of the staging is then for loads only:
  // the d a0 stores of the iteration are YOUNGER than the fetch (counted s_waitcnt), where a wait at the top of the
  // iteration sat out the write acknowledgement of the stores just issued (3000 cycles per centre).
  Lists nl = {0u, 0u};
  {
    const long G = gridDim.x, c0 = blockIdx.x;
    if (c0 < centres) {
      fetch(c0, nl);
      stage(nl, 0);
      fetch(c0 + G < centres ? c0 + G : c0, nl);
      stage(nl, 1);
      fetch(c0 + 2 * G < centres ? c0 + 2 * G : c0, nl);
    }
  }
  __syncthreads();
  half8 wh[4], wl[4];
  float unW;
  {
    float m = 0.f;
#pragma unroll
    for (int w = 0; w < S2L_WAVES; ++w) m = fmaxf(m, s_red[w]);
    const unsigned Ew = s2_exp(m);
    const float sw = s2_scale(Ew);
    unW = s2_unscale(Ew);
    const float* row = a.W1T + (size_t)(16 * rg + p16) * S2_K + 8 * q4;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const float4 b0 = *reinterpret_cast<const float4*>(row + 32 * s);
      const float4 b1 = *reinterpret_cast<const float4*>(row + 32 * s + 4);
      const float x[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
      s2_split8(x, sw, wh[s], wl[s]);
    }
  }
  __syncthreads();   // s_red is reused below

  const unsigned char* wcol = reinterpret_cast<const unsigned char*>(s_w2 + 2 * lane);
  long long tsec[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#define S2L_T(i)                                                   \
  if (MODE == 4) {                                                 \
    const long long now = (long long)__builtin_amdgcn_s_memtime(); \
    tsec[i] += now - tlast;                                        \
    tlast = now;                                                   \
  }
  // A wave owns samples w, w + 16 (first half of the centre) and w + 32, w + 48 (second half): interleaved, because
  // ball queries with few distinct points put all the arg-max samples at the front.

  // ---- phase 1 of one half (samples w + 32 h, w + 16 + 32 h): acc = sum over a sample's entries of
  // g * W2[channel][2 lane, 2 lane + 1], ascending channel; gated; the wave's maximum -> s_red[h].
  // Six entries at a time, rows first, then the fma chain, each as NESTED uniform tests (mm entries cost one taken
  // branch per pass; a fall-through switch is torn apart by the control-flow structuriser).
#define S2L_NEST(X)                  \
  X(0) if (mm > 1) {                 \
    X(1) if (mm > 2) {               \
      X(2) if (mm > 3) {             \
        X(3) if (mm > 4) {           \
          X(4) if (mm > 5) { X(5) }  \
        }                            \
      }                              \
    }                                \
  }
  constexpr int CH = 6;   // entries per pass
  // What phase 1 reads that does not depend on the interval it runs in -- the two samples' (first entry, count), their
  // first eight entries (BROADCAST reads of (g, row offset): no cross-lane VALU work) and the gate words -- is requested
  // in front of the barrier before it: the dependent chain inside the interval starts at the rows of W2.
  struct P1In {
    int first[2], n[2];
    float2v e[2][CH];
    uint4v kw;       // gate words of rows 2 lane, 2 lane + 1: (lo, hi) each
  };
  auto p1_fetch = [&](int h, int bufx, P1In& in) {
    const float2v* en = s_en + bufx * S2L_EN + 8;
    const int* eox = s_eo + bufx * S2L_EO + wave + 32 * h;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      in.first[kk] = __builtin_amdgcn_readfirstlane(eox[16 * kk]);
      in.n[kk] = __builtin_amdgcn_readfirstlane(eox[16 * kk + 1]) - in.first[kk];
    }
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
      for (int t = 0; t < CH; ++t) in.e[kk][t] = en[in.first[kk] + t];   // (reads past the centre's 256 entries stay inside LDS)
    in.kw = *reinterpret_cast<const uint4v*>(reinterpret_cast<const unsigned*>(s_gt + bufx * S2_C) + 4 * lane);
  };
  auto phase1 = [&](auto hsel, int bufx, P1In& in, float2v (&v2)[2]) {
    constexpr int h = decltype(hsel)::value;
    const float2v* en = s_en + bufx * S2L_EN + 8;
    v2[0] = float2v{0.f, 0.f};
    v2[1] = float2v{0.f, 0.f};
    if (MODE != 1) {
      // an entry costs two vector instructions: the row's address and one packed fma (both k of the lane)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        float2v acc = {0.f, 0.f};
        for (int base = 0; base < in.n[kk]; base += CH) {
          const int mm = in.n[kk] - base;
          if (base > 0) {
#pragma unroll
            for (int t = 0; t < CH; ++t) in.e[kk][t] = en[in.first[kk] + base + t];
          }
          float2v w[CH];
#define S2L_ROW(t) w[t] = *reinterpret_cast<const float2v*>(wcol + __float_as_int(in.e[kk][t][1]));
#define S2L_FMA(t) acc = __builtin_elementwise_fma(float2v{in.e[kk][t][0], in.e[kk][t][0]}, w[t], acc);
          S2L_NEST(S2L_ROW)
          S2L_NEST(S2L_FMA)
#undef S2L_ROW
#undef S2L_FMA
        }
        v2[kk] = acc;
      }
    }
    const uint4v kw = in.kw;
    // a1 > 0 of (row, sample): bit `sample` of the row's gate word -> 0 / -1 by a bit-field extract, AND
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const unsigned sh = (unsigned)(wave + 16 * kk);
      v2[kk][0] = __int_as_float(__float_as_int(v2[kk][0]) & __builtin_amdgcn_sbfe((int)kw[h], sh, 1u));
      v2[kk][1] = __int_as_float(__float_as_int(v2[kk][1]) & __builtin_amdgcn_sbfe((int)kw[2 + h], sh, 1u));
    }
    float mx = fmaxf(fmaxf(__builtin_fabsf(v2[0][0]), __builtin_fabsf(v2[0][1])), fmaxf(__builtin_fabsf(v2[1][0]), __builtin_fabsf(v2[1][1])));
    mx = wave_max(mx);
    if (lane == 0) s_red[h * S2L_WAVES + wave] = mx;
  };
  // ---- between the barriers: the half's power-of-two scale; this wave's two rows as split-fp16 images; the gate words
  // of the epilogue
  auto split_rows = [&](int hb, const float2v (&v2)[2], int bufx, unsigned& Ex, int (&gw)[4]) {
    float mx = 0.f;
#pragma unroll
    for (int w4 = 0; w4 < S2L_WAVES / 4; ++w4) {
      const float4 v = *reinterpret_cast<const float4*>(s_red + hb * S2L_WAVES + 4 * w4);
      mx = fmaxf(fmaxf(mx, fmaxf(v.x, v.y)), fmaxf(v.z, v.w));
    }
    Ex = s2_exp(mx);
    const float sx = s2_scale(Ex);
    unsigned* irow = s_img + wave * S2L_PI + lane;
    unsigned hh, ll;
    s2_split2(v2[0][0], v2[0][1], sx, hh, ll);
    irow[0] = hh;
    irow[64] = ll;
    s2_split2(v2[1][0], v2[1][1], sx, hh, ll);
    irow[16 * S2L_PI] = hh;
    irow[16 * S2L_PI + 64] = ll;
    const unsigned* gt = reinterpret_cast<const unsigned*>(s_gt + bufx * S2_C);
#pragma unroll
    for (int r = 0; r < 4; ++r) gw[r] = (int)gt[2 * (S2_K + 16 * rg + 4 * q4 + r) + hb];
  };
  // ---- phase 2: rows 16 rg .. + 15 of d a0 for samples 32 hb + 16 cg .. + 15 of centre (b, m)
  auto phase2 = [&](int hb, unsigned Ex, const int (&gw)[4], int b, int m) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const unsigned* brow = s_img + (16 * cg + p16) * S2L_PI + 4 * q4;
    if (MODE != 2) {
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const half8 xh = *reinterpret_cast<const half8*>(brow + 16 * s);
        const half8 xl = *reinterpret_cast<const half8*>(brow + 64 + 16 * s);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], xh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh[s], xl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl[s], xh, acc, 0, 0, 0);
      }
    }
    const float unscale = s2_unscale(Ex) * unW;
    const unsigned sh = (unsigned)(16 * cg + p16);
    float* Y = a.da0 + ((size_t)b * S2_K + 16 * rg + 4 * q4) * ldY + (size_t)m * 64 + 32 * hb + sh;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = __int_as_float(__float_as_int(acc[r] * unscale) & __builtin_amdgcn_sbfe(gw[r], sh, 1u));
      if (MODE != 3 || v == 12345.f) Y[(size_t)r * ldY] = v;
    }
  };
  typedef std::integral_constant<int, 0> H0;
  typedef std::integral_constant<int, 1> H1;

  // Schedule per centre c (A, B0, C, B1 = workgroup barriers; between B and the next barrier the matrix-core work of one
  // half runs beside the VALU work of another, odd waves in the opposite order to even ones):
  //   A | split 0 | B0 | phase 2 (0) + phase 1 (second half of c) | C | split 1 | B1 | phase 2 (1) + phase 1 (first
  //   half of c + G) | A ...
  float2v va[2], vb[2];
  P1In pin;
  if ((long)blockIdx.x < centres) {
    p1_fetch(0, 0, pin);
    phase1(H0{}, 0, pin, va);
  }
  const int par = wave & 1;
  if (MODE == 4) tlast = (long long)__builtin_amdgcn_s_memtime();
  int buf = 0;
  for (long c = blockIdx.x; c < centres; c += gridDim.x, buf ^= 1) {
    const bool more = c + gridDim.x < centres;
    const int b = (int)(c / a.M), m = (int)(c - (long)b * a.M);
    unsigned Ex;
    int gw[4];
    S2L_T(0)
    s2_lds_barrier();   // A: the first half's maxima and the lists staged last are visible; the images are free
    S2L_T(1)
    split_rows(0, va, buf, Ex, gw);
    p1_fetch(1, buf, pin);
    S2L_T(2)
    s2_lds_barrier();   // B0
    S2L_T(3)
    if (par) {
      phase1(H1{}, buf, pin, vb);
      S2L_T(4)
      phase2(0, Ex, gw, b, m);
    } else {
      phase2(0, Ex, gw, b, m);
      S2L_T(4)
      phase1(H1{}, buf, pin, vb);
    }
    S2L_T(5)
    s2_lds_barrier();   // C
    split_rows(1, vb, buf, Ex, gw);
    p1_fetch(0, buf ^ 1, pin);
    s2_lds_barrier();   // B1
    S2L_T(6)
    if (par) {
      if (more) phase1(H0{}, buf ^ 1, pin, va);
      phase2(1, Ex, gw, b, m);
    } else {
      phase2(1, Ex, gw, b, m);
      if (more) phase1(H0{}, buf ^ 1, pin, va);
    }
    S2L_T(7)
    stage(nl, buf);   // the lists of c + 2 G
    const long cn = c + 3 * (long)gridDim.x < centres ? c + 3 * (long)gridDim.x : c;
    fetch(cn, nl);
  }
#undef S2L_NEST
  if (MODE == 4 && blockIdx.x == 0 && lane == 0)
    for (int i = 0; i < 8; ++i) a.dbg[wave * 8 + i] = tsec[i];
#undef S2L_T
}


}  // namespace
