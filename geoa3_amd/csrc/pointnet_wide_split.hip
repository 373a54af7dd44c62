// The 1024-wide layers of pointnet_wide.hip on the f16 matrix pipe with SPLIT fp32 operands (gfx950).
//
// gfx950 has no tf32/xf32 MFMA and its fp32 MFMA runs at 1/16 of the 16-bit rate.  Every fp32 operand v (after a
// power-of-two scaling that puts the largest magnitude of its tensor / tile into [2^13, 2^14)) is carried as two fp16
// values,
//     hi = rn16(v),   lo = rn16(v - hi)                   (|v - hi - lo| <= 2^-24 |v| while lo is a normal fp16),
// and a product a*w is evaluated as  a_hi*w_hi + a_hi*w_lo + a_lo*w_hi  with fp32 accumulation in the matrix core
// (`v_mfma_f32_32x32x16_f16`; a product of two fp16 values is exact in fp32).  The dropped a_lo*w_lo term is 2^-24
// relative.  Three 32-cycle MFMAs do the work of eight 64-cycle fp32 ones.  Error model: an operand smaller than
// 2^-16 of the largest one of its tile has a subnormal (possibly flushed) lo part and keeps 11 bits -- an absolute
// error below 2^-28 of the largest term, i.e. 1/16 ulp of a sum that contains it.  Measured (tools/wide_accuracy.py):
// the logits' and the input gradient's error against a float64 evaluation equals that of the fp32 MFMA kernel and of
// the fp32 CPU oracle.
// The activation scale is chosen PER WORK UNIT from the tile's own maximum while it is staged (no range restriction;
// a non-finite activation poisons the instance's features with NaN); the weights are scaled on the host
// (geoa3_amd/pointnet.py pack_wide_split); the maxima are scaled back before they are published.
//
// Structure (differences from wide_max2_kernel): a work unit is (instance, 128-point tile, GROUPS x 128 channels); the
// activation tile is split while it is staged and stored POINT-major in LDS ([piece][point + halo][128 ci] fp16, rows
// padded to 272 B), so the A operand of a k-step (8 consecutive ci of one point) is one conflict-free ds_read_b128 and
// the taps of conv5 are row offsets; each wave owns CB x 32 channels x 128 points (4 point tiles per channel tile);
// the weight fragments stream from L2 (2 x 16 B per lane per channel tile and k-step of 16) through a register ring
// across the channel groups.  Epilogue, keys and the finalize kernel are those of the fp32 path.
#include "pointnet_kernels.h"
#include "profile.h"

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

constexpr int SP_THREADS = 256;
constexpr int SP_PTS = 128;                  // points per unit
constexpr int SP_ROWS = SP_PTS + 2;          // + one halo point on either side (conv5: kernel 3, pad 1)
constexpr int SP_ROWB = 272;                 // bytes per LDS row: 128 fp16 + 16 B pad (row stride = 4 banks mod 64)
constexpr int SP_PIECEB = SP_ROWS * SP_ROWB; // 35,360
constexpr int SP_LDS = 2 * SP_PIECEB;        // 70,720 B: two workgroups per CU

__device__ __forceinline__ void split16(float v, _Float16& hi, _Float16& lo) {
  hi = (_Float16)v;
  lo = (_Float16)(v - (float)hi);
}

// CB: channel tiles of 32 per wave (1: a workgroup covers 128 channels per group step, 2: 256)
// FUSE: the 128-channel input tile is COMPUTED while it is staged instead of being read (WideArgs::Xin etc.):
//   1: tile = relu(W2 h + b2) from the 64-channel activation h = Xin [B][64][N]   (T-Net(64) conv2, trunk conv4)
//   2: the same with h = relu(w1 x + b1) evaluated from the 3-channel cloud x3    (T-Net(3) conv1 + conv2)
// Wave w owns the tile's points 32 w .. 32 w + 31: its B operands are the wave's own 64 x 32 block of h (split with a
// per-wave power-of-two scale), the A operands the pre-split fragments of W2 (host: pack_wide_split, K = 64) streamed
// from L2 tile by tile; 48 MFMAs per wave against 1152 (conv5) / 768 (T-Net) of the layer itself.  The relu gate of
// the tile goes out as the bit mask the backward reads (32 bits per (channel, wave)); the [B][128][N] activation is
// never written.  conv5's two halo points are evaluated on the VALU (fp32): they differ from the neighbouring tile's
// MFMA values in the last bit at most, like any two fp32 evaluations.
template <int TAPS, int OCC, int GROUPS, int CB, int PIPE, bool DESYNC, int FUSE>
__global__ __launch_bounds__(SP_THREADS, OCC) void wide_split_kernel(WideArgs a, int slots_per_xcd) {
  constexpr int KS = TAPS * 8;               // k-steps of 16 per channel tile
  constexpr int PF = CB == 1 ? 4 : 2;        // k-steps of weight fragments in flight; 8 % PF == 0
  constexpr int NBLK = KS / PF, BPT = 8 / PF;   // blocks of PF k-steps; blocks per tap
  constexpr int GSTEPS = GROUPS / CB;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  __shared__ float s_max[4];
  __shared__ float4 s_w1[FUSE == 2 ? 64 : 1];      // (w1 row, b1) of the 3-channel first layer
  // packed maxima of the current unit, published (atomicMax) while the NEXT unit is being staged: keeps the atomics
  // out of the in-order vmcnt queue in front of the weight stream (on its own within run-to-run noise)
  __shared__ unsigned long long s_keys[4][GROUPS][32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, l31 = lane & 31;
  const int N = a.N, tiles = (N + SP_PTS - 1) / SP_PTS;
  constexpr int SPLIT = 8 / GROUPS;
  const int per_inst = tiles * SPLIT;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int inst_x = (a.B - xcd + 7) / 8;
  const int units = inst_x * per_inst;
  const half8* Wall = reinterpret_cast<const half8*>(a.Wh);
  // The two workgroups that share a CU start together and do identical work.  DESYNC: the workgroup in the odd wave
  // slot of its SIMD (HW_ID.wave_id) runs half of its first unit's channel groups first and the other half at the very
  // end (one extra staging pass), which shifts its phases by half a period, so one workgroup's staging and epilogues
  // run under the other's MFMAs.  (s_memtime trace, tools/bench_wide.py --stamps: per unit 7 % waiting for the tile,
  // 7 % splitting it, 4 x 20.7 % channel groups of which 2 % epilogue.)  Worth 0-10 % depending on the device: the
  // kernel sits at the clock-limited ceiling of the 16-bit matrix pipe (1.25 PFLOP/s executed at 1.86 GHz, 66 % busy).
  bool late = false;
  if (DESYNC) {
    if (tid == 0) s_max[0] = __int_as_float(__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1);   // HW_REG_HW_ID[3:0]
    __syncthreads();
    late = __float_as_int(s_max[0]) != 0;
  }
  if (FUSE == 2) {
    if (tid < 64) s_w1[tid] = make_float4(a.w1[3 * tid], a.w1[3 * tid + 1], a.w1[3 * tid + 2], a.b1[tid]);
    __syncthreads();
  }
  const int nmine = units > slot ? (units - slot + slots_per_xcd - 1) / slots_per_xcd : 0;
  late = late && nmine > 0 && GSTEPS > 1;
  int nstamp = 0;
  auto stamp = [&]() {   // diagnostic build only (a.stamps != null): wave 0 of workgroup 0 records s_memtime
    if (a.stamps && blockIdx.x == 0 && tid == 0 && nstamp < 255) a.stamps[1 + nstamp++] = __builtin_amdgcn_s_memtime();
  };
  int pend_b = -1, pend_co = 0, pend_n = 0;     // unit whose maxima wait in s_keys: instance, first channel, tiles
  auto flush = [&]() {
    if (pend_b < 0) return;
    for (int i = kh; i < pend_n; i += 2)       // the two half-waves publish two channel tiles per instruction
      atomicMax(a.keys + (size_t)pend_b * a.Co + pend_co + (i / CB) * (128 * CB) + (i % CB) * 32 + l31, s_keys[wave][i][l31]);
    pend_b = -1;
  };
  for (int it = 0; it < nmine + (late ? 1 : 0); ++it) {
    const int u = slot + (it == nmine ? 0 : it) * slots_per_xcd;
    const int g_begin = late && it == nmine ? GSTEPS / 2 : 0;
    const int g_end = late && it == 0 ? GSTEPS / 2 : GSTEPS;
    const int q = u / per_inst, r = u - q * per_inst;
    const int b = xcd + 8 * q, tile = r / SPLIT, half = r - tile * SPLIT;
    const int n0 = tile * SP_PTS;
    const float* X = a.X + (size_t)b * a.sXb;
    stamp();
    // ---- stage: every value of the tile goes through registers once: maximum -> scale -> split -> LDS.
    // Wave w takes the channel octets w, w+4, ..; a lane one point per pass (rows 1..128 of the image = points
    // n0 .. n0+127): 8 coalesced row reads per octet.  conv5's two halo rows (points n0-1, n0+128): one value per
    // thread.
    float xv[2][4][8], xhalo = 0.f;
    float yt[FUSE ? 4 : 1][16];          // FUSE: channel 32 t + (r & 3) + 8 (r >> 2) + 4 kh of point 32 wave + l31
    float m = 0.f;
    if constexpr (FUSE != 0) {
      const int n = n0 + 32 * wave + l31;
      const bool in = n < N;
      float hv[4][8];                    // [k-step][i]: channel 16 s + 8 kh + i of the wave's point l31
      if (FUSE == 1) {
        int ldi = a.ldXin;
        asm volatile("" : "+s"(ldi));
        const float* ph = a.Xin + (size_t)b * a.sXinb + (in ? n : 0);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float v = ph[(size_t)((16 * s4 + 8 * kh + i) * ldi)];
            hv[s4][i] = in ? v : 0.f;
          }
      } else {
        const float* px = a.x3 + (size_t)b * 3 * N + (in ? n : 0);
        const float x0 = px[0], x1 = px[N], x2 = px[2 * (size_t)N];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float4 w = s_w1[16 * s4 + 8 * kh + i];
            const float v = fmaxf(w.x * x0 + w.y * x1 + w.z * x2 + w.w, 0.f);   // first_layer of pointnet_conv_split.hip
            hv[s4][i] = in ? v : 0.f;
          }
      }
      if (TAPS == 3) {   // halo points n0 - 1 (threads 0..127) and n0 + 128 (128..255): channel tid & 127, on the VALU
        const int nh = tid < 128 ? n0 - 1 : n0 + SP_PTS;
        if (nh >= 0 && nh < N) {
          const int c = tid & 127;
          const float* wr = a.W2f + (size_t)c * 64;
          const float* ph = a.Xin + (size_t)b * a.sXinb + nh;
          float acc = a.b2[c];
#pragma unroll 8
          for (int k = 0; k < 64; ++k) acc += wr[k] * ph[(size_t)k * a.ldXin];
          xhalo = fmaxf(acc, 0.f);
        }
      }
      float hm = 0.f;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int i = 0; i < 8; ++i) hm = fmaxf(hm, __builtin_fabsf(hv[s4][i]));
      hm = wave_max(hm);
      unsigned Eh = (__float_as_uint(hm) >> 23) & 0xffu;
      Eh = Eh < 14u ? 14u : (Eh > 254u ? 254u : Eh);
      const float hscale = __uint_as_float((267u - Eh) << 23);
      const float hun = a.w2_unscale * __uint_as_float((Eh - 13u) << 23);
      half8 bh[4], bl[4];
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          _Float16 h, l;
          split16(hv[s4][i] * hscale, h, l);
          bh[s4][i] = h;
          bl[s4][i] = l;
        }
      const half8* W2 = reinterpret_cast<const half8*>(a.W2h) + lane;   // [t][s][piece][lane]
      unsigned mk0 = 0u, mk1 = 0u;       // lane L collects the 32-bit gate words of channels L and 64 + L
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
          const half8 wh = W2[(size_t)((t * 4 + s4) * 2) * 64], wl = W2[(size_t)((t * 4 + s4) * 2 + 1) * 64];
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bh[s4], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, bl[s4], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, bh[s4], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ch = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * kh;
          float v = fmaxf(acc[r] * hun + a.b2[ch], 0.f);
          v = in ? v : 0.f;
          yt[t][r] = v;
          m = fmaxf(m, v);
          if (a.Ymask) {
            const unsigned long long bal = __ballot(v > 0.f);
            const int rowA = (32 * t + (r & 3) + 8 * (r >> 2)) & 63;     // the channel of the kh = 0 half, mod 64
            if (lane == rowA) (t < 2 ? mk0 : mk1) = (unsigned)bal;
            if (lane == rowA + 4) (t < 2 ? mk0 : mk1) = (unsigned)(bal >> 32);
          }
        }
      }
      if (a.Ymask && (TAPS == 1 || half == 0) && n0 + 32 * wave < N) {
        unsigned* mk = reinterpret_cast<unsigned*>(a.Ymask) +
                       (((size_t)b * ((N + 63) >> 6) + (size_t)((n0 >> 6) + (wave >> 1))) * 128) * 2 + (wave & 1);
        mk[2 * lane] = mk0;
        mk[2 * (64 + lane)] = mk1;
      }
      m = fmaxf(m, __builtin_fabsf(xhalo));
    } else {
    {
      int ldx = a.ldX;
      asm volatile("" : "+s"(ldx));              // opaque: keeps the row offsets from being hoisted out of the unit
                                                 // loop (they were spilled to scratch)
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int n = n0 + pass * 64 + lane;
        const bool in = n < N;
        const float* px = X + (in ? n : 0);
#pragma unroll
        for (int oc = 0; oc < 4; ++oc)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const float v = px[(size_t)(((wave + 4 * oc) * 8 + i) * ldx)];
            xv[pass][oc][i] = in ? v : 0.f;
          }
      }
      if (TAPS == 3) {
        const int n = tid < 128 ? n0 - 1 : n0 + SP_PTS;
        if (n >= 0 && n < N) xhalo = X[(size_t)((tid & 127) * ldx) + n];
      }
    }
    m = __builtin_fabsf(xhalo);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass)
#pragma unroll
      for (int oc = 0; oc < 4; ++oc)
#pragma unroll
        for (int i = 0; i < 8; ++i) m = fmaxf(m, __builtin_fabsf(xv[pass][oc][i]));   // NaN is caught through xs below
    }
    m = wave_max(m);
    stamp();
    flush();           // the previous unit's maxima, behind this unit's loads
    __syncthreads();   // every wave is done with the previous tile (LDS image and s_max)
    if (lane == 0) s_max[wave] = m;
    __syncthreads();
    m = fmaxf(fmaxf(s_max[0], s_max[1]), fmaxf(s_max[2], s_max[3]));
    // scale = 2^e with max * 2^e in [2^13, 2^14)
    unsigned E = (__float_as_uint(m) >> 23) & 0xffu;
    bool bad = E == 255u;    // inf (a NaN does not survive fmaxf: caught below)
    E = E < 14u ? 14u : (E > 254u ? 254u : E);
    const float scale = __uint_as_float((267u - E) << 23), unscale = a.unscale * __uint_as_float((E - 13u) << 23);
    if constexpr (FUSE != 0) {
      typedef _Float16 half4 __attribute__((ext_vector_type(4)));
      const int p = 1 + 32 * wave + l31;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          half4 hi, lo;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float xs = yt[t][4 * j + i] * scale;
            bad |= xs != xs;
            _Float16 h, l;
            split16(xs, h, l);
            hi[i] = h;
            lo[i] = l;
          }
          unsigned char* dst = smem_raw + p * SP_ROWB + (32 * t + 8 * j + 4 * kh) * 2;
          *reinterpret_cast<half4*>(dst) = hi;
          *reinterpret_cast<half4*>(dst + SP_PIECEB) = lo;
        }
    } else {
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int p = 1 + pass * 64 + lane;
#pragma unroll
      for (int oc = 0; oc < 4; ++oc) {
        half8 hi, lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xs = xv[pass][oc][i] * scale;
          bad |= xs != xs;
          _Float16 h, l;
          split16(xs, h, l);
          hi[i] = h;
          lo[i] = l;
        }
        unsigned char* dst = smem_raw + p * SP_ROWB + (wave + 4 * oc) * 16;
        *reinterpret_cast<half8*>(dst) = hi;
        *reinterpret_cast<half8*>(dst + SP_PIECEB) = lo;
      }
    }
    }
    if (TAPS == 3) {
      const float xs = xhalo * scale;
      bad |= xs != xs;
      _Float16 h, l;
      split16(xs, h, l);
      unsigned char* dst = smem_raw + (tid < 128 ? 0 : SP_ROWS - 1) * SP_ROWB + (tid & 127) * 2;
      *reinterpret_cast<_Float16*>(dst) = h;
      *reinterpret_cast<_Float16*>(dst + SP_PIECEB) = l;
    }
    if (__syncthreads_or(bad))   // loud, not silently wrong: key ~0 decodes to NaN in wide_finalize_kernel
      for (int c = tid; c < a.Co; c += SP_THREADS) atomicMax(a.keys + (size_t)b * a.Co + c, ~0ull);
    stamp();
    // A operand of lane (r = l31, h = kh), tile t, k-step s (tap = s / 8, ci0 = 16 (s % 8)):
    //   row 32t + r + tap (+1 without taps), bytes (ci0 + 8h) * 2
    const unsigned char* abase = smem_raw + (l31 + (TAPS == 1 ? 1 : 0)) * SP_ROWB + kh * 16;
    // fragments of the wave's channel tile c (< CB) of group step g: [T][s][piece][lane]
    auto wbase = [&](int g, int c) {
      const int co = (half * GROUPS + g * CB) * 128 + wave * 32 * CB + 32 * c;
      return Wall + (size_t)(co / 32) * KS * 2 * 64 + lane;
    };
    half8 wf[PF][CB][2];
#pragma unroll
    for (int c = 0; c < CB; ++c) {
      const half8* W0 = wbase(g_begin, c);
#pragma unroll
      for (int f = 0; f < PF; ++f) {
        wf[f][c][0] = W0[(size_t)(2 * f) * 64];
        wf[f][c][1] = W0[(size_t)(2 * f + 1) * 64];
      }
    }
#pragma unroll 1
    for (int g = g_begin; g < g_end; ++g) {
      const half8 *Wp[CB], *Wn[CB];
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        Wp[c] = wbase(g, c);
        Wn[c] = wbase(g + 1 < g_end ? g + 1 : g, c);
      }
      f32x16 acc[CB][4];
#pragma unroll
      for (int c = 0; c < CB; ++c)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) acc[c][t][i] = 0.f;
      auto lds_rd = [&](const unsigned char* base, int f, int t, half8& h, half8& l) {
        const unsigned char* ap = base + 32 * t * SP_ROWB + f * 32;
        h = *reinterpret_cast<const half8*>(ap);
        l = *reinterpret_cast<const half8*>(ap + SP_PIECEB);
      };
      if (PIPE == 2) {
        // software pipeline over k-steps: the eight A fragments (4 point tiles x hi, lo) of k-step s+1 are requested
        // from LDS before the 12 x CB MFMAs of k-step s are issued (the compiler's own schedule reused one register
        // for the lo fragments: read, wait, MFMA, four exposed LDS latencies per k-step)
        half8 A[2][4][2];
#pragma unroll
        for (int t = 0; t < 4; ++t) lds_rd(abase, 0, t, A[0][t][0], A[0][t][1]);
#pragma unroll 1
        for (int sb = 0; sb < NBLK; ++sb) {
          const unsigned char* ap0 = abase + (sb / BPT) * SP_ROWB + (sb % BPT) * (PF * 32);
          const int sn = sb + 1 < NBLK ? sb + 1 : sb;
          const unsigned char* ap1 = abase + (sn / BPT) * SP_ROWB + (sn % BPT) * (PF * 32);
#pragma unroll
          for (int f = 0; f < PF; ++f) {
            const int cur = f & 1, nxt = cur ^ 1;          // PF is even: the buffers line up across blocks
#pragma unroll
            for (int t = 0; t < 4; ++t) {
              if (f + 1 < PF) lds_rd(ap0, f + 1, t, A[nxt][t][0], A[nxt][t][1]);
              else lds_rd(ap1, 0, t, A[nxt][t][0], A[nxt][t][1]);
            }
            half8 wh[CB], wl[CB];
#pragma unroll
            for (int c = 0; c < CB; ++c) {
              wh[c] = wf[f][c][0];
              wl[c] = wf[f][c][1];
              const half8* src = sb + 1 < NBLK ? Wp[c] + (size_t)(2 * PF * 64) * (sb + 1) : Wn[c];
              wf[f][c][0] = src[(2 * f) * 64];
              wf[f][c][1] = src[(2 * f + 1) * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
              for (int c = 0; c < CB; ++c) {
                acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[cur][t][0], wh[c], acc[c][t], 0, 0, 0);
                acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[cur][t][0], wl[c], acc[c][t], 0, 0, 0);
                acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[cur][t][1], wh[c], acc[c][t], 0, 0, 0);
              }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      } else {
      // software pipeline over (k-step, tile): the A fragments of the next tile-step are read from LDS while the
      // MFMAs of the current one run
      // tile-step j = 4 f + t of the current block (j >= 4 PF: the next block's first step)
      auto rd_step = [&](const unsigned char* ap0, const unsigned char* ap1, int j, half8& h, half8& l) {
        if (j < 4 * PF) lds_rd(ap0, j >> 2, j & 3, h, l);
        else lds_rd(ap1, (j - 4 * PF) >> 2, (j - 4 * PF) & 3, h, l);
      };
      half8 qh[2], ql[2];
      rd_step(abase, abase, 0, qh[0], ql[0]);
#pragma unroll 1
      for (int sb = 0; sb < NBLK; ++sb) {
        // k-steps PF sb .. PF sb + PF - 1: tap = sb / BPT, ci0 = 16 PF (sb % BPT) + 16 f
        const unsigned char* ap0 = abase + (sb / BPT) * SP_ROWB + (sb % BPT) * (PF * 32);
        const int sn = sb + 1 < NBLK ? sb + 1 : sb;
        const unsigned char* ap1 = abase + (sn / BPT) * SP_ROWB + (sn % BPT) * (PF * 32);
#pragma unroll
        for (int f = 0; f < PF; ++f) {
          half8 wh[CB], wl[CB];
#pragma unroll
          for (int c = 0; c < CB; ++c) {
            wh[c] = wf[f][c][0];
            wl[c] = wf[f][c][1];
            const half8* src = sb + 1 < NBLK ? Wp[c] + (size_t)(2 * PF * 64) * (sb + 1) : Wn[c];
            wf[f][c][0] = src[(2 * f) * 64];
            wf[f][c][1] = src[(2 * f + 1) * 64];
          }
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            const int j = 4 * f + t;
            rd_step(ap0, ap1, j + 1, qh[(j + 1) & 1], ql[(j + 1) & 1]);
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
            const half8 xh = qh[j & 1], xl = ql[j & 1];
            // operands swapped as in the fp32 kernel: rows = points, columns = channels
#pragma unroll
            for (int c = 0; c < CB; ++c) {
              acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh[c], acc[c][t], 0, 0, 0);
              acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl[c], acc[c][t], 0, 0, 0);
              acc[c][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh[c], acc[c][t], 0, 0, 0);
            }
            if (PIPE) __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      }
      stamp();
      // lane: channel co0 + l31; acc[t][r]: point n0 + 32t + (r&3) + 8(r>>2) + 4kh.  Ascending point order, strict >
      const bool full = n0 + SP_PTS <= N;
#pragma unroll
      for (int c = 0; c < CB; ++c) {
        float v = -__builtin_inff();
        int col = 0;
        if (full) {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) {
              const bool gt = acc[c][t][r16] > v;
              v = gt ? acc[c][t][r16] : v;
              col = gt ? 32 * t + mfma_row(r16, 0) : col;
            }
          col += n0 + 4 * kh;
        } else {
#pragma unroll
          for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r16 = 0; r16 < 16; ++r16) {
              const int n = n0 + 32 * t + mfma_row(r16, lane);
              const bool gt = n < N && acc[c][t][r16] > v;
              v = gt ? acc[c][t][r16] : v;
              col = gt ? n : col;
            }
        }
        const float ov = __shfl_xor(v, 32, 64);
        const int oc = __shfl_xor(col, 32, 64);
        const bool take = ov > v || (ov == v && oc < col);
        v = take ? ov : v;
        col = take ? oc : col;
        if (lane < 32) s_keys[wave][(g - g_begin) * CB + c][lane] = wide_key(v * unscale, col);
      }
    }
    stamp();
    pend_b = b;
    pend_co = (half * GROUPS + g_begin * CB) * 128 + wave * 32 * CB;
    pend_n = (g_end - g_begin) * CB;
  }
  flush();
  if (a.stamps && blockIdx.x == 0 && tid == 0) a.stamps[0] = nstamp;
}

template <int TAPS, int OCC, int GROUPS, int CB, int PIPE, bool DESYNC, int FUSE = 0>
void launch_variant(const WideArgs& a, hipStream_t s) {
  auto kern = wide_split_kernel<TAPS, OCC, GROUPS, CB, PIPE, DESYNC, FUSE>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS);
  constexpr int SLOTS = 32 * OCC;
  hipLaunchKernelGGL(kern, dim3(SLOTS * 8), dim3(SP_THREADS), SP_LDS, s, a, SLOTS);
}

}  // namespace

int launch_wide_max_split(const WideArgs& a, hipStream_t s) {
  if (a.Co != 1024 || (a.taps != 1 && a.taps != 3) || !a.keys || !a.Wh) return GEOA3_ENOSUPPORT;
  const int tag = a.taps == 3 ? GEOA3_PROF_CONV5 : GEOA3_PROF_TNETWIDE;
  geoa3_prof_begin(tag, s);
  if (!a.keys_clean &&
      hipMemsetAsync(a.keys, 0, (size_t)a.B * a.Co * sizeof(unsigned long long), s) != hipSuccess)
    return GEOA3_ELAUNCH;
  // variants measured on hardware (tools/bench_wide.py 0 1 2): within +-5 % of each other and of run-to-run noise
  // (round 4 again: two channel tiles per wave, CB = 2 -- half the LDS reads per MFMA -- 186.2 against 185.2 us at one
  //  tap, 542 against 508 us at three: LDS bandwidth is not what holds these kernels at ~1.1-1.2 PF on the f16 pipe)
  if (a.W2h) {   // the 64 -> 128 layer in front is computed while the tile is staged
    if (!a.b2 || (a.x3 ? (!a.w1 || !a.b1 || a.taps != 1) : !a.Xin) || (a.taps == 3 && !a.W2f)) return GEOA3_EINVAL;
    if (a.taps == 1 && a.x3) launch_variant<1, 2, 8, 1, 0, true, 2>(a, s);
    else if (a.taps == 1) launch_variant<1, 2, 8, 1, 0, true, 1>(a, s);
    else launch_variant<3, 2, 4, 1, 2, true, 1>(a, s);
  } else if (a.taps == 1) {
    switch (a.variant) {
      case 1: launch_variant<1, 2, 8, 1, 2, true>(a, s); break;
      case 2: launch_variant<1, 2, 8, 1, 0, false>(a, s); break;
      default: launch_variant<1, 2, 8, 1, 0, true>(a, s);
    }
  } else {
    switch (a.variant) {
      case 1: launch_variant<3, 2, 4, 1, 0, true>(a, s); break;
      case 2: launch_variant<3, 2, 4, 1, 2, false>(a, s); break;
      default: launch_variant<3, 2, 4, 1, 2, true>(a, s);
    }
  }
  launch_wide_finalize(a, s);
  geoa3_prof_end(tag, s);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

// One 1024-wide layer in isolation (tools/bench_wide.py, tests): Wp = fp32 fragments, Wh = split fragments or NULL.
extern "C" int geoa3_debug_wide_fwd(const float* X, const float* Wp, const void* Wh, float unscale, const float* bias,
                                    float* out, int32_t* arg, void* keys, int B, int N, int taps, int variant,
                                    void* stamps, void* stream) {
  WideArgs a{};
  a.X = X; a.sXb = (long)128 * N; a.ldX = N;
  a.W = Wp; a.Wh = Wh; a.unscale = unscale; a.bias = bias;
  a.out = out; a.arg = arg; a.keys = (unsigned long long*)keys;
  a.Co = 1024; a.N = N; a.B = B; a.taps = taps;
  a.stamps = (unsigned long long*)stamps;
  a.variant = variant;
  return launch_wide_max(a, geoa3_stream(stream));
}
