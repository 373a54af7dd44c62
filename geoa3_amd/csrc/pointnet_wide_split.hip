// The 1024-wide layers of pointnet_wide.hip on the f16 matrix pipe with SPLIT fp32 operands (opt-in, gfx950).
//
// gfx950 has no tf32/xf32 MFMA and its fp32 MFMA runs at 1/16 of the 16-bit rate.  Every fp32 operand x is therefore
// carried as two fp16 values,
//     hi = rn16(x),   lo = rn16((x - hi) * 2^11)            (|x - hi - lo * 2^-11| <= 2^-24 |x|: half an fp32 ulp),
// and a product a*w is evaluated as  a_hi*w_hi  +  2^-11 * (a_hi*w_lo + a_lo*w_hi)  with fp32 accumulation in the
// matrix core (`v_mfma_f32_32x32x16_f16`; products of fp16 values are exact in fp32); the dropped a_lo*w_lo term is
// 2^-24 relative.  The hi*hi sums and the cross sums run in SEPARATE accumulators (the 2^11 keeps the low parts out
// of fp16's subnormal range) and are combined once per output.  Three 32-cycle MFMAs do the work of eight 64-cycle
// fp32 ones.  Range: |activation| < 65504 (larger values overflow fp16 to inf and surface as NaN logits); weights are
// pre-scaled by a power of two on the host (geoa3_amd/pointnet.py pack_wide_split) and the result is scaled back.
//
// Structure (differences from wide_max2_kernel): a work unit is (instance, 128-point tile, GROUPS x 128 channels); the
// activation tile is split while it is staged and stored POINT-major in LDS ([piece][point + halo][128 ci] fp16, rows
// padded to 272 B), so the A operand of a k-step (8 consecutive ci of one point) is one conflict-free ds_read_b128 and
// the taps of conv5 are row offsets; each wave owns 32 channels x 128 points (4 tiles x 2 accumulators); the weight
// fragments stream from L2 (2 x 16 B per lane per k-step of 16) through a register ring across the channel groups.
// Epilogue, keys and the finalize kernel are those of the fp32 path.
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
constexpr float SP_LO = 2048.f, SP_ILO = 1.f / 2048.f;

template <int TAPS, int OCC, int GROUPS>
__global__ __launch_bounds__(SP_THREADS, OCC) void wide_split_kernel(WideArgs a, int slots_per_xcd) {
  constexpr int KS = TAPS * 8;               // k-steps of 16 per channel tile
  constexpr int PF = 4;                      // k-steps of weight fragments in flight (2 x 16 B each); KS % PF == 0
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, kh = lane >> 5, l31 = lane & 31;
  const int N = a.N, tiles = (N + SP_PTS - 1) / SP_PTS;
  constexpr int SPLIT = 8 / GROUPS;
  const int per_inst = tiles * SPLIT;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int inst_x = (a.B - xcd + 7) / 8;
  const int units = inst_x * per_inst;
  const half8* Wall = reinterpret_cast<const half8*>(a.Wh);
  for (int u = slot; u < units; u += slots_per_xcd) {
    const int q = u / per_inst, r = u - q * per_inst;
    const int b = xcd + 8 * q, tile = r / SPLIT, half = r - tile * SPLIT;
    const int n0 = tile * SP_PTS;
    const float* X = a.X + (size_t)b * a.sXb;
    __syncthreads();   // every wave is done with the previous tile
    // stage + split: wave w takes the channel octets w, w+4, ..; a lane one point: 8 coalesced row reads, two 16-B
    // LDS writes (row p of the image = point n0 - 1 + p)
#pragma unroll 1
    for (int pass = 0; pass < 3; ++pass) {
      const int p = pass * 64 + lane;
      if (p >= SP_ROWS) break;                     // pass 2: the two rows of the right halo only (wave-uniform exit
      const int n = n0 - 1 + p;                    //         for lanes >= 2 would diverge: they idle instead)
      const bool in = n >= 0 && n < N;
      int ldx = a.ldX;
      asm volatile("" : "+s"(ldx));                // opaque: keeps the 32 row offsets from being hoisted out of the
                                                   // unit loop (they were spilled to scratch)
      const float* px = X + (in ? n : 0);
#pragma unroll
      for (int oc = 0; oc < 4; ++oc) {
        const int c8 = (wave + 4 * oc) * 8;
        float x[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = px[(size_t)((c8 + i) * ldx)];
        half8 hi, lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const float xv = in ? x[i] : 0.f;
          const _Float16 h = (_Float16)xv;
          hi[i] = h;
          lo[i] = (_Float16)((xv - (float)h) * SP_LO);
        }
        unsigned char* dst = smem_raw + p * SP_ROWB + c8 * 2;
        *reinterpret_cast<half8*>(dst) = hi;
        *reinterpret_cast<half8*>(dst + SP_PIECEB) = lo;
      }
    }
    __syncthreads();
    // A operand of lane (r = l31, h = kh), tile t, k-step s (tap = s / 8, ci0 = 16 (s % 8)):
    //   row 32t + r + tap (+1 without taps), bytes (ci0 + 8h) * 2
    const unsigned char* abase = smem_raw + (l31 + (TAPS == 1 ? 1 : 0)) * SP_ROWB + kh * 16;
    auto wbase = [&](int g) {
      const int co = (half * GROUPS + g) * 128 + wave * 32;
      return Wall + (size_t)(co / 32) * KS * 2 * 64 + lane;
    };
    half8 wf[PF][2];
    {
      const half8* W0 = wbase(0);
#pragma unroll
      for (int f = 0; f < PF; ++f) {
        wf[f][0] = W0[(size_t)(2 * f) * 64];
        wf[f][1] = W0[(size_t)(2 * f + 1) * 64];
      }
    }
#pragma unroll 1
    for (int g = 0; g < GROUPS; ++g) {
      const int co0 = (half * GROUPS + g) * 128 + wave * 32;
      const half8* Wp = wbase(g);
      const half8* Wn = wbase(g + 1 < GROUPS ? g + 1 : g);
      f32x16 ah[4], ax[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          ah[t][i] = 0.f;
          ax[t][i] = 0.f;
        }
      // software pipeline over (k-step, tile): the A fragments of the next tile-step are read from LDS while the
      // three MFMAs of the current one run; the scheduling barriers keep the compiler from hoisting every read of
      // the unrolled body to its top (that version spilled 800 registers)
      auto lds_rd = [&](const unsigned char* base, int f, int t, half8& h, half8& l) {
        const unsigned char* ap = base + 32 * t * SP_ROWB + f * 32;
        h = *reinterpret_cast<const half8*>(ap);
        l = *reinterpret_cast<const half8*>(ap + SP_PIECEB);
      };
      half8 xh, xl, nh, nl;
      lds_rd(abase, 0, 0, xh, xl);
#pragma unroll 1
      for (int s4 = 0; s4 < KS / PF; ++s4) {
        // k-steps 4 s4 .. 4 s4 + 3: tap = s4 / 2, ci0 = 64 (s4 & 1) + 16 f
        const unsigned char* ap0 = abase + (s4 >> 1) * SP_ROWB + (s4 & 1) * 128;
        const int s5 = s4 + 1 < KS / PF ? s4 + 1 : s4;
        const unsigned char* ap1 = abase + (s5 >> 1) * SP_ROWB + (s5 & 1) * 128;
        const half8* src = s4 + 1 < KS / PF ? Wp + (size_t)(2 * PF * 64) * (s4 + 1) : Wn;
#pragma unroll
        for (int f = 0; f < PF; ++f) {
          const half8 wh = wf[f][0], wl = wf[f][1];
          wf[f][0] = src[(2 * f) * 64];
          wf[f][1] = src[(2 * f + 1) * 64];
#pragma unroll
          for (int t = 0; t < 4; ++t) {
            if (t < 3) lds_rd(ap0, f, t + 1, nh, nl);
            else if (f < PF - 1) lds_rd(ap0, f + 1, 0, nh, nl);
            else lds_rd(ap1, 0, 0, nh, nl);
            __builtin_amdgcn_sched_barrier(0);
            // operands swapped as in the fp32 kernel: rows = points, columns = channels
            ah[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wh, ah[t], 0, 0, 0);
            ax[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xh, wl, ax[t], 0, 0, 0);
            ax[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(xl, wh, ax[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            xh = nh;
            xl = nl;
          }
        }
      }
      // lane: channel co0 + l31; acc[t][r]: point n0 + 32t + (r&3) + 8(r>>2) + 4kh.  Ascending point order, strict >
      float v = -__builtin_inff();
      int col = 0;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r16 = 0; r16 < 16; ++r16) {
          const int n = n0 + 32 * t + mfma_row(r16, lane);
          const float val = (ah[t][r16] + ax[t][r16] * SP_ILO) * a.unscale;
          const bool gt = n < N && val > v;
          v = gt ? val : v;
          col = gt ? n : col;
        }
      const float ov = __shfl_xor(v, 32, 64);
      const int oc = __shfl_xor(col, 32, 64);
      const bool take = ov > v || (ov == v && oc < col);
      v = take ? ov : v;
      col = take ? oc : col;
      if (lane < 32) atomicMax(a.keys + (size_t)b * a.Co + co0 + lane, wide_key(v, col));
    }
  }
}

template <int TAPS, int OCC, int GROUPS>
void launch_variant(const WideArgs& a, hipStream_t s) {
  auto kern = wide_split_kernel<TAPS, OCC, GROUPS>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, SP_LDS);
  constexpr int SLOTS = 32 * OCC;
  hipLaunchKernelGGL(kern, dim3(SLOTS * 8), dim3(SP_THREADS), SP_LDS, s, a, SLOTS);
}

}  // namespace

int launch_wide_max_split(const WideArgs& a, hipStream_t s) {
  if (a.Co != 1024 || (a.taps != 1 && a.taps != 3) || !a.keys || !a.Wh) return GEOA3_ENOSUPPORT;
  const int tag = a.taps == 3 ? GEOA3_PROF_CONV5 : GEOA3_PROF_TNETWIDE;
  geoa3_prof_begin(tag, s);
  if (hipMemsetAsync(a.keys, 0, (size_t)a.B * a.Co * sizeof(unsigned long long), s) != hipSuccess) return GEOA3_ELAUNCH;
  if (a.taps == 1) launch_variant<1, 2, 8>(a, s);
  else launch_variant<3, 2, 4>(a, s);
  launch_wide_finalize(a, s);
  geoa3_prof_end(tag, s);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
