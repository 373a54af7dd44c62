// fp32 MFMA GEMMs for the narrow PointNet layers (gfx950): the channel-major 1x1 convolutions /
// per-instance feature transforms (Model/PointNet.py:79-80,138-145) and the fully connected heads
// (:82-84,150-152), forward and input-gradient.  v_mfma_f32_32x32x2_f32 is an exact k-ordered fmaf
// chain, so results differ from the CPU reference only by summation order.
#include "pointnet_kernels.h"
#include <cstdlib>
#include <type_traits>

namespace {

// ------------------------------------------------------------------------------------------
// conv_cm64: one wavefront owns 64 consecutive columns (points) and 64 output channels.  Every global access is a
// 256-byte row (lane = column): measured on MI355X, a store instruction covering ONE 256-B row runs at 5.7 TB/s
// while the same bytes as two 128-B segments in two rows (a 32x32 accumulator register as it stands) reach only
// 2.2 TB/s (tools/bench_conv.py), which made the 32-column form of this kernel store-bound.  The MFMA operand
// layouts (two rows x 32 columns per register) are produced in registers by v_permlane32_swap: swapping the upper
// half of row 2s with the lower half of row 2s+1 turns two 64-column rows into the B operands of the two 32-column
// blocks, and the same swap on a pair of accumulator registers turns them into two 64-column output rows.
// K is consumed in chunks of 32 rows, double-buffered in registers: chunk c+1 is in flight while chunk c feeds the
// matrix core; chunk 0 is requested before the weights are staged into LDS.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void swap32(float& a, float& b) {   // a[32..63] <-> b[0..31]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}

// FIRST: the input rows are the 3-channel first layer, computed on the fly (ConvArgs::produce_first);
// GFIRST: the output's relu gate is that first layer's sign, recomputed (ConvArgs::gate_first).  Both evaluate
// relu(w1 . (T^T x) + b1) with ONE expression (first_layer), so the forward activation and the backward mask agree.
__device__ __forceinline__ float first_layer(const float4 w, float p0, float p1, float p2) {
  return w.x * p0 + w.y * p1 + w.z * p2 + w.w;
}

template <int NCH, bool FIRST, bool GFIRST, bool BWD3>  // K = CH * NCH; BWD3 needs GFIRST
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 5))) void conv_cm64_kernel(ConvArgs a) {
  extern __shared__ __attribute__((aligned(16))) float s_w[];  // [64][K+1]: the 64 output rows of this row block
  constexpr int CH = 16;   // rows of X per chunk (two chunks in registers)
  constexpr int K = CH * NCH, pitch = K + 1;
  const int b = blockIdx.y, rb = blockIdx.z, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int col = blockIdx.x * 256 + wave * 64 + lane;
  const bool live = col < a.N;
  const float* X = FIRST ? nullptr : a.X + (size_t)b * a.sXb + (live ? col : a.N - 1);
  float4* s_w1 = reinterpret_cast<float4*>(s_w + 64 * pitch + 4);   // [64] (w1 row, b1) of the folded first layer
  // per-row operands of the epilogue (bias, gate words): one coalesced load per wave here, read back with v_readlane --
  // 64 uniform loads inside the epilogue are 64 dependent round trips (pointnet_conv_split.hip)
  const size_t mword = ((size_t)b * ((a.N + 63) >> 6) + (size_t)(blockIdx.x * 4 + wave)) * a.Co;
  const bool wave_live = blockIdx.x * 256 + wave * 64 < a.N;
  const float biasv = a.bias ? a.bias[rb * 64 + lane] : 0.f;
  const unsigned long long zmv = (a.Zmask && wave_live) ? a.Zmask[mword + rb * 64 + lane] : 0ull;
  float p0 = 0.f, p1 = 0.f, p2 = 0.f;                                // T^T x of this lane's point
  float x0 = 0.f, x1 = 0.f, x2 = 0.f;
  if (FIRST || GFIRST) {
    const float* xp = a.x3 + (size_t)b * 3 * a.N + (live ? col : a.N - 1);
    x0 = xp[0];
    x1 = xp[a.N];
    x2 = xp[2 * (size_t)a.N];
    p0 = x0;
    p1 = x1;
    p2 = x2;
    if (a.T3) {   // bmm(pc^T, T)^T : x'[c] = sum_d x[d] T[d][c]   (Model/PointNet.py:138)
      const float* t = a.T3 + (size_t)b * 9;
      // (one scalar chain per coordinate, each behind an opaque barrier: the SLP vectoriser otherwise pairs them into
      //  v_pk_fma_f32 with op_sel -- a LOW result reading the HIGH half of a register pair, the form NOTEBOOK 5a's stand-alone
      //  reproducer shows going wrong beside matrix-core wavefronts; same multiply-adds, same order, same bits)
      p0 = __builtin_fmaf(x2, t[6], __builtin_fmaf(x1, t[3], x0 * t[0]));
      asm volatile("" : "+v"(p0), "+v"(x0), "+v"(x1), "+v"(x2));
      p1 = __builtin_fmaf(x2, t[7], __builtin_fmaf(x1, t[4], x0 * t[1]));
      asm volatile("" : "+v"(p1), "+v"(x0), "+v"(x1), "+v"(x2));
      p2 = __builtin_fmaf(x2, t[8], __builtin_fmaf(x1, t[5], x0 * t[2]));
      asm volatile("" : "+v"(p2));
    }
    if (tid < 64) s_w1[tid] = make_float4(a.w1[3 * tid], a.w1[3 * tid + 1], a.w1[3 * tid + 2], a.b1[tid]);
  }
  float xb[2][CH];
  if (!FIRST) {
#pragma unroll
    for (int u = 0; u < CH; ++u) xb[0][u] = X[(size_t)u * a.ldX];
  }

  const float* W = a.W + (size_t)b * a.sWb + (size_t)rb * 64 * a.sWco;
  if (a.sWk == 1 && (a.sWco & 3) == 0 && (a.sWb & 3) == 0) {   // rows are k-contiguous: 16-byte loads
    for (int e = tid; e < 64 * K / 4; e += 256) {
      const int co = e / (K / 4), k = (e - co * (K / 4)) * 4;
      const float4 w = *reinterpret_cast<const float4*>(W + (size_t)co * a.sWco + k);
      float* d = s_w + co * pitch + k;
      d[0] = w.x; d[1] = w.y; d[2] = w.z; d[3] = w.w;
    }
  } else if (a.sWk == 1) {
    for (int e = tid; e < 64 * K; e += 256) {
      const int co = e / K, k = e - co * K;
      s_w[co * pitch + k] = W[(size_t)co * a.sWco + k];
    }
  } else {  // transposed storage: co is the contiguous index
    for (int e = tid; e < 64 * K; e += 256) {
      const int k = e / 64, co = e - k * 64;
      s_w[co * pitch + k] = W[(size_t)co * a.sWco + (size_t)k * a.sWk];
    }
  }
  __syncthreads();
  const float* wrow = s_w + (lane & 31) * pitch + (lane >> 5);

  f32x16 acc[2][2];   // [column block][row tile]
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;

#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    if (FIRST) {
#pragma unroll
      for (int u = 0; u < CH; ++u) xb[c & 1][u] = fmaxf(first_layer(s_w1[CH * c + u], p0, p1, p2), 0.f);
    } else if (c + 1 < NCH) {
#pragma unroll
      for (int u = 0; u < CH; ++u) xb[(c + 1) & 1][u] = X[(size_t)(CH * (c + 1) + u) * a.ldX];
    }
    float* x = xb[c & 1];
#pragma unroll
    for (int u = 0; u < CH / 2; ++u) {
      swap32(x[2 * u], x[2 * u + 1]);
      const float a0 = wrow[CH * c + 2 * u], a1 = wrow[32 * pitch + CH * c + 2 * u];
      acc[0][0] = mfma32(a0, x[2 * u], acc[0][0]);
      acc[1][0] = mfma32(a0, x[2 * u + 1], acc[1][0]);
      acc[0][1] = mfma32(a1, x[2 * u], acc[0][1]);
      acc[1][1] = mfma32(a1, x[2 * u + 1], acc[1][1]);
    }
  }

  // epilogue: every pair of accumulator registers becomes two 64-column rows (row, row + 4)
  float* Y = BWD3 ? nullptr : a.Y + (size_t)b * a.sYb + col;
  const float* Z = a.Z ? a.Z + (size_t)b * a.sZb + col : nullptr;
  // bit masks: one 64-bit word per (row, this wave's 64 columns), stored [B][column block][row]: the 64 rows of a
  // wave are one 512-byte run (a store of 8 scattered bytes per row cost more than the row reads it saved)
  unsigned long long mymask = 0ull;    // Ymask: lane r collects the word of row rb*64 + r
  float q0 = 0.f, q1 = 0.f, q2 = 0.f;   // BWD3: d/d(T^T x) of this lane's point, summed over the 64 rows
#pragma unroll
  for (int t = 0; t < 2; ++t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      float v[8], z[8], y[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        v[i] = acc[0][t][4 * g + i];
        v[4 + i] = acc[1][t][4 * g + i];
        swap32(v[i], v[4 + i]);    // v[i]: row base+i, v[4+i]: row base+4+i, lane = column
      }
      const int row0 = rb * 64 + t * 32 + 8 * g;   // rows row0 .. row0+7 in the order of v[]
      if (a.Zmask && wave_live) {                  // wave-uniform 8-byte loads: bit `lane` gates this lane's column
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int rl = t * 32 + 8 * g + i;   // the lane that holds this row's gate word
          const unsigned mlo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)zmv, rl);
          const unsigned mhi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(zmv >> 32), rl);
          z[i] = ((lane < 32 ? mlo >> lane : mhi >> (lane - 32)) & 1u) ? 1.f : 0.f;
        }
      }
      if (live) {
        if (Z) {
#pragma unroll
          for (int i = 0; i < 8; ++i) z[i] = Z[(size_t)(row0 + i) * a.ldZ];
        }
        if (a.accumulate) {
#pragma unroll
          for (int i = 0; i < 8; ++i) y[i] = Y[(size_t)(row0 + i) * a.ldY];
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          float o = v[i];
          if (a.bias) o += __int_as_float(__builtin_amdgcn_readlane(__float_as_int(biasv), t * 32 + 8 * g + i));
          if (a.relu) o = fmaxf(o, 0.f);
          if (a.accumulate) o += y[i];
          if (Z || a.Zmask) o = z[i] > 0.f ? o : 0.f;  // gate AFTER accumulation (sum of branches, then relu')
          if (GFIRST) o = first_layer(s_w1[row0 + i], p0, p1, p2) > 0.f ? o : 0.f;
          if (BWD3) {
            const float4 w = s_w1[row0 + i];
            q0 += w.x * o;
            q1 += w.y * o;
            q2 += w.z * o;
          } else {
            Y[(size_t)(row0 + i) * a.ldY] = o;
          }
          v[i] = o;
        }
      }
      if (a.Ymask && wave_live) {   // wave-uniform branch: the whole wave takes part in the ballots
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const unsigned long long m = __ballot(live && v[i] > 0.f);
          if (lane == t * 32 + 8 * g + i) mymask = m;
        }
      }
    }
  }
  if (a.Ymask && wave_live) a.Ymask[mword + rb * 64 + lane] = mymask;
  if (BWD3) {
    // x' = T^T x  =>  dx[d] = sum_c T[d][c] q[c];  dT[d][c] = sum_n x[d][n] q[c][n]
    if (!live) {
      q0 = 0.f;
      q1 = 0.f;
      q2 = 0.f;
    }
    float d0 = q0, d1 = q1, d2 = q2;
    if (a.T3) {
      const float* t = a.T3 + (size_t)b * 9;
      d0 = t[0] * q0 + t[1] * q1 + t[2] * q2;
      d1 = t[3] * q0 + t[4] * q1 + t[5] * q2;
      d2 = t[6] * q0 + t[7] * q1 + t[8] * q2;
    }
    if (live) {
      float* dxp = a.dx3 + (size_t)b * 3 * a.N + col;
      if (a.accumulate) {
        d0 += dxp[0];
        d1 += dxp[a.N];
        d2 += dxp[2 * (size_t)a.N];
      }
      dxp[0] = d0;
      dxp[a.N] = d1;
      dxp[2 * (size_t)a.N] = d2;
    }
    if (a.dTpart) {   // workgroup-uniform
      float* s_part = reinterpret_cast<float*>(s_w1 + 64);   // [4 waves][9]
      const float prod[9] = {x0 * q0, x0 * q1, x0 * q2, x1 * q0, x1 * q1, x1 * q2, x2 * q0, x2 * q1, x2 * q2};
#pragma unroll
      for (int i = 0; i < 9; ++i) {
        const float v = wave_sum(prod[i]);
        if (lane == 0) s_part[wave * 9 + i] = v;
      }
      __syncthreads();
      if (tid < 9) {
        a.dTpart[((size_t)b * gridDim.x + blockIdx.x) * GEOA3_DT_PITCH + tid] =
            s_part[tid] + s_part[9 + tid] + s_part[18 + tid] + s_part[27 + tid];
      }
    }
  }
}

__global__ __launch_bounds__(256) void reduce_dT_kernel(const float* __restrict__ part, int nparts, float* __restrict__ dT,
                                                        int total) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int b = e / 9, i = e - b * 9;
  float v = 0.f;
  for (int w = 0; w < nparts; ++w) v += part[((size_t)b * nparts + w) * GEOA3_DT_PITCH + i];
  dT[e] = v;
}

// ------------------------------------------------------------------------------------------
// fc: Y[m][o] = epi(sum_k X[m][k] W[o][k] + bias[o]) -- the fully connected heads in both directions and the small
// per-instance products.  These layers are LATENCY bound (~1 GFLOP per chain, operands from L2): the time of a launch
// is one dependent chain load -> MFMA -> reduce -> store, so the kernel is shaped to make that chain short:
//   * tile T x T outputs per workgroup, T = 16 (v_mfma_f32_16x16x4_f32) for small M so that a 32-row batch still
//     spreads over >= 64 workgroups, T = 32 (v_mfma_f32_32x32x2_f32) otherwise (half the operand re-reads);
//   * K is split over up to 16 wavefronts so that a wave's share is ONE batch of loads (64 k: every load of the wave is
//     issued before its first MFMA; longer K loops double-buffer batches);
//   * both operands are k-contiguous: a lane pulls 4 consecutive k with one 16-byte load straight into MFMA operand
//     registers (k is consumed in a permuted order, identical for A and B);
//   * the order in which the products of one output are added depends on K alone -- not on the tile shape, hence not on
//     the batch size: rows of a batched run are bit-identical to batch-1 runs (tests/test_gpu_fullsize.py);
//   * the S partial tiles meet in LDS and are summed in wave order (fixed order: results do not depend on timing or on
//     the batch size), each wave finishing one accumulator register (a row group) incl. bias / relu / gate and its store.
// ------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void load4(const float* p, int k, int kend, bool vec, float v[4]) {
  if (vec && k + 3 < kend) {
    const float4 t = *reinterpret_cast<const float4*>(p + k);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (k + i < kend) ? p[k + i] : 0.f;
  }
}

// Q > 1 (K >= 2048, a.kscratch given): the S * Q waves that split K are spread over Q workgroups (grid.z); every wave
// writes its partial tile to kscratch and fc_ksplit_reduce_kernel adds the S * Q partials in wave order and finishes --
// the same partial sums in the same order as ONE workgroup of S * Q waves, so not a bit changes, but a K = 4096 layer with
// 256 output tiles runs on 1024 workgroups instead of 256 (54 -> ~20 us at 250 instances).
template <int T, int S, int Q = 1>
__global__ __launch_bounds__(64 * S) void fc_kernel(FcArgs a) {
  constexpr int R = T == 32 ? 16 : 4;        // accumulator registers per lane
  constexpr int G = 64 / T;                  // lane groups along k inside one MFMA (2 or 4)
  constexpr int KQ = 4 * G;                  // k consumed per 16-byte load of every lane (8 or 16)
  constexpr int U = 64 / KQ;                 // loads per batch: one batch = 64 k
  __shared__ float s_red[S > 1 && Q == 1 ? S * R * 64 : 1];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = (tid >> 6) + (Q > 1 ? (int)blockIdx.z * S : 0);   // among the S * Q waves that split K
  const int o0 = blockIdx.x * T, m0 = blockIdx.y * T, bz = Q > 1 ? 0 : blockIdx.z;
  const int r = lane & (T - 1), g = lane / T;
  const int m = min(m0 + r, a.M - 1), o = min(o0 + r, a.Nout - 1);
  const float* xp = a.X + (size_t)bz * a.sXb + (size_t)m * a.ldX;
  const float* wp = a.W + (size_t)bz * a.sWb + (size_t)o * a.ldW;
  const bool xvec = ((a.ldX | a.sXb) & 3) == 0 && ((uintptr_t)a.X & 15) == 0;
  const bool wvec = ((a.ldW | a.sWb) & 3) == 0 && ((uintptr_t)a.W & 15) == 0;
  // this wave's K range, in units of 16 for both tile shapes
  const int kq = (a.K + 15) / 16;
  const int per = (kq + S * Q - 1) / (S * Q);
  const int kb = wave * per * 16, ke = min(a.K, (wave + 1) * per * 16);

  typename std::conditional<T == 32, f32x16, f32x4>::type acc;
#pragma unroll
  for (int i = 0; i < R; ++i) acc[i] = 0.f;
  float xa[2][U][4], wb[2][U][4];
  auto load_batch = [&](int k0, float (&xd)[U][4], float (&wd)[U][4]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = k0 + KQ * u + 4 * g;
      load4(xp, k, ke, xvec, xd[u]);
      load4(wp, k, ke, wvec, wd[u]);
    }
  };
  // Both tile shapes add the products of an output element in the SAME order -- within every 16 k: k = i, 4 + i, 8 + i,
  // 12 + i for i = 0..3 (an fp32 MFMA is a k-ordered fmaf chain) -- so the choice of the tile, which follows the batch
  // size, never changes a bit of the result: the 16x16x4 form covers the four k of a step in one instruction (lane
  // group g holds k = 4 g + i), the 32x32x2 form in two (lanes h = 0 / 1 hold 4 h + i, then 8 + 4 h + i).
  auto mma_batch = [&](float (&xd)[U][4], float (&wd)[U][4]) {
    if constexpr (T == 32) {
#pragma unroll
      for (int u = 0; u < U; u += 2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc = mfma32(xd[u][i], wd[u][i], acc);
          acc = mfma32(xd[u + 1][i], wd[u + 1][i], acc);
        }
    } else {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(xd[u][i], wd[u][i], acc, 0, 0, 0);
    }
  };
  if (kb < ke) load_batch(kb, xa[0], wb[0]);
  for (int k0 = kb; k0 < ke; k0 += 128) {
    if (k0 + 64 < ke) load_batch(k0 + 64, xa[1], wb[1]);
    mma_batch(xa[0], wb[0]);
    if (k0 + 64 < ke) {
      if (k0 + 128 < ke) load_batch(k0 + 128, xa[0], wb[0]);
      mma_batch(xa[1], wb[1]);
    }
  }
  if constexpr (Q > 1) {   // partial tile of this wave: [tile][wave][register][lane]
    float* ps = a.kscratch + (((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * (S * Q) + wave) * R) * 64 + lane;
#pragma unroll
    for (int i = 0; i < R; ++i) ps[i * 64] = acc[i];
    return;
  }
  const int oc = o0 + r;
  const float bias = (a.bias && oc < a.Nout) ? a.bias[oc] : 0.f;
  auto finish = [&](int i, float v) {   // accumulator register i of this lane -> Y
    const int mr = m0 + (T == 32 ? mfma_row(i, lane) : 4 * g + i);
    if (oc < a.Nout && mr < a.M) {
      v += bias;
      if (a.relu) v = fmaxf(v, 0.f);
      if (a.Z) v = a.Z[(size_t)mr * a.ldZ + oc] > 0.f ? v : 0.f;
      a.Y[(size_t)bz * a.sYb + (size_t)mr * a.ldY + oc] = v;
    }
  };
  if constexpr (S == 1) {
#pragma unroll
    for (int i = 0; i < R; ++i) finish(i, acc[i]);
  } else {
#pragma unroll
    for (int i = 0; i < R; ++i) s_red[(wave * R + i) * 64 + lane] = acc[i];
    __syncthreads();
    for (int i = wave; i < R; i += S) {     // register i of every wave, summed in wave order
      float v = s_red[i * 64 + lane];
#pragma unroll
      for (int w = 1; w < S; ++w) v += s_red[(w * R + i) * 64 + lane];
      finish(i, v);
    }
  }
}

// the S * Q partial tiles of fc_kernel<T, S, Q> summed in wave order, then the layer's tail (one wavefront per tile)
template <int T, int WAVES>
__global__ __launch_bounds__(64) void fc_ksplit_reduce_kernel(FcArgs a) {
  constexpr int R = T == 32 ? 16 : 4;
  const int lane = threadIdx.x, o0 = blockIdx.x * T, m0 = blockIdx.y * T;
  const int r = lane & (T - 1), g = lane / T, oc = o0 + r;
  const float bias = (a.bias && oc < a.Nout) ? a.bias[oc] : 0.f;
  const float* ps = a.kscratch + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * WAVES * R) * 64 + lane;
  float part[R][WAVES];     // every partial value in flight before the first add
#pragma unroll
  for (int i = 0; i < R; ++i)
#pragma unroll
    for (int w = 0; w < WAVES; ++w) part[i][w] = ps[((size_t)w * R + i) * 64];
#pragma unroll
  for (int i = 0; i < R; ++i) {
    float v = part[i][0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) v += part[i][w];
    const int mr = m0 + (T == 32 ? mfma_row(i, lane) : 4 * g + i);
    if (oc < a.Nout && mr < a.M) {
      v += bias;
      if (a.relu) v = fmaxf(v, 0.f);
      if (a.Z) v = a.Z[(size_t)mr * a.ldZ + oc] > 0.f ? v : 0.f;
      a.Y[(size_t)mr * a.ldY + oc] = v;
    }
  }
}

}  // namespace

int launch_conv_cm(const ConvArgs& a, hipStream_t s) {
  if (a.produce_first && (a.K != 64 || !a.x3 || !a.w1 || !a.b1)) return GEOA3_EINVAL;
  if (a.gate_first && (a.Co != 64 || !a.x3 || !a.w1 || !a.b1 || a.produce_first)) return GEOA3_EINVAL;
  if (a.dx3 && !a.gate_first) return GEOA3_EINVAL;
  if (a.split) return launch_conv_cm_split(a, s);
  if ((a.K != 64 && a.K != 128) || (a.Co != 64 && a.Co != 128)) return GEOA3_ENOSUPPORT;
  const size_t lds = ((size_t)64 * (a.K + 1) + 4 + 64 * 4 + 40) * sizeof(float);
  dim3 grid((a.N + 255) / 256, a.B, a.Co / 64);
  if (a.produce_first)
    hipLaunchKernelGGL((conv_cm64_kernel<4, true, false, false>), grid, dim3(256), lds, s, a);
  else if (a.gate_first && a.dx3 && a.K == 64)
    hipLaunchKernelGGL((conv_cm64_kernel<4, false, true, true>), grid, dim3(256), lds, s, a);
  else if (a.gate_first && a.dx3)
    hipLaunchKernelGGL((conv_cm64_kernel<8, false, true, true>), grid, dim3(256), lds, s, a);
  else if (a.gate_first && a.K == 64)
    hipLaunchKernelGGL((conv_cm64_kernel<4, false, true, false>), grid, dim3(256), lds, s, a);
  else if (a.gate_first)
    hipLaunchKernelGGL((conv_cm64_kernel<8, false, true, false>), grid, dim3(256), lds, s, a);
  else if (a.K == 64)
    hipLaunchKernelGGL((conv_cm64_kernel<4, false, false, false>), grid, dim3(256), lds, s, a);
  else
    hipLaunchKernelGGL((conv_cm64_kernel<8, false, false, false>), grid, dim3(256), lds, s, a);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_debug_conv_cm(const float* X, const float* W, const float* bias, const float* Z, float* Y, int B,
                                   int N, int K, int Co, int relu, int split, void* stream) {
  ConvArgs a{};
  a.split = split;
  a.X = X; a.sXb = (long)K * N; a.ldX = N;
  a.W = W; a.sWb = 0; a.sWco = K; a.sWk = 1;
  a.bias = bias;
  a.Z = Z; a.sZb = (long)Co * N; a.ldZ = N;
  a.Y = Y; a.sYb = (long)Co * N; a.ldY = N;
  a.Co = Co; a.K = K; a.N = N; a.B = B;
  a.relu = relu;
  return launch_conv_cm(a, geoa3_stream(stream));
}

int launch_reduce_dT(const float* part, int nparts, float* dT, int B, hipStream_t s) {
  hipLaunchKernelGGL(reduce_dT_kernel, dim3((B * 9 + 255) / 256), dim3(256), 0, s, part, nparts, dT, B * 9);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

template <int T>
static void launch_fc_tile(const FcArgs& a, int waves, hipStream_t s) {
  dim3 grid((a.Nout + T - 1) / T, (a.M + T - 1) / T, a.batch > 1 ? a.batch : 1);
  switch (waves) {
    case 1: hipLaunchKernelGGL((fc_kernel<T, 1>), grid, dim3(64), 0, s, a); break;
    case 2: hipLaunchKernelGGL((fc_kernel<T, 2>), grid, dim3(128), 0, s, a); break;
    case 4: hipLaunchKernelGGL((fc_kernel<T, 4>), grid, dim3(256), 0, s, a); break;
    case 8: hipLaunchKernelGGL((fc_kernel<T, 8>), grid, dim3(512), 0, s, a); break;
    default: hipLaunchKernelGGL((fc_kernel<T, 16>), grid, dim3(1024), 0, s, a);
  }
}

int launch_fc(const FcArgs& a, hipStream_t s) {
  // waves splitting K: one 64-k batch per wave up to 16 waves; a function of K only (or fixed by the call site), so the
  // summation order never depends on the batch size
  int waves = a.ksplit > 0 ? a.ksplit : (a.K + 63) / 64;
  waves = waves >= 16 ? 16 : waves > 4 ? 8 : waves > 2 ? 4 : waves > 1 ? 2 : 1;
  // 16 x 16 tiles while a 32 x 32 tiling would leave most CUs without a workgroup
  const long tiles32 = (long)((a.Nout + 31) / 32) * ((a.M + 31) / 32) * (a.batch > 1 ? a.batch : 1);
  if (a.kscratch && waves == 16 && a.K >= 2048 && a.batch <= 1 && (a.tile == 16 || (a.tile == 0 && tiles32 < 256))) {
    // K spread over four workgroups per tile (same partial sums, same order: see fc_kernel) -- at every batch size: at 32
    // instances one 16-wave workgroup per tile instead makes the iteration 0.4805 against 0.4727 ms
    dim3 grid((a.Nout + 15) / 16, (a.M + 15) / 16, 4);
    hipLaunchKernelGGL((fc_kernel<16, 4, 4>), grid, dim3(256), 0, s, a);
    hipLaunchKernelGGL((fc_ksplit_reduce_kernel<16, 16>), dim3(grid.x, grid.y), dim3(64), 0, s, a);
    GEOA3_CHECK_LAUNCH();
    return GEOA3_OK;
  }
  if (a.tile == 16 || (a.tile == 0 && tiles32 < 256)) launch_fc_tile<16>(a, waves, s);
  else launch_fc_tile<32>(a, waves, s);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

// One fully connected layer in isolation (tools/bench_fc.py): Y[M][Nout] = relu?(X[M][K] W[Nout][K]^T + bias)
extern "C" int geoa3_debug_fc(const float* X, const float* W, const float* bias, float* Y, int M, int Nout, int K,
                              int relu, int ksplit, void* stream) {
  FcArgs a{};
  a.X = X; a.ldX = K;
  a.W = W; a.ldW = K;
  a.bias = bias;
  a.Y = Y; a.ldY = Nout;
  a.M = M; a.Nout = Nout; a.K = K; a.relu = relu;
  a.ksplit = ksplit & 0xff; a.tile = ksplit >> 8;   // tools/bench_fc.py: (tile << 8) | waves, 0 = the shipped choice
  return launch_fc(a, geoa3_stream(stream));
}

// Channel-major 1x1 convolution as an operator (geoa3_amd/pointnet2.py: the shared MLPs of PointNet++ level 2):
// Y[b][co][n] = epi( sum_k W[co][k] X[b][k][n] ),  epi = (+ bias[co]) (relu) (keep where Z[b][co][n] > 0)
extern "C" int geoa3_conv1x1(const float* X, const float* W, const float* bias, const float* Z, float* Y, int B, long N,
                             int K, int Co, int relu, void* stream) {
  if (!X || !W || !Y || B <= 0 || N <= 0 || N > 0x7fffffffL) return GEOA3_EINVAL;
  ConvArgs a{};
  a.split = 1;
  a.X = X; a.sXb = (long)K * N; a.ldX = (int)N;
  a.W = W; a.sWb = 0; a.sWco = K; a.sWk = 1;
  a.bias = bias;
  a.Z = Z; a.sZb = (long)Co * N; a.ldZ = (int)N;
  a.Y = Y; a.sYb = (long)Co * N; a.ldY = (int)N;
  a.Co = Co; a.K = K; a.N = (int)N; a.B = B;
  a.relu = relu;
  return launch_conv_cm(a, geoa3_stream(stream));
}

// geoa3_conv1x1 whose output is max-pooled over each centre's 64 samples in the epilogue (the last layer of a
// set-abstraction MLP + F.max_pool2d, pointnet2_modules.py:57-70): out[b][co][m] = relu(max_s (W x)[co][64 m + s] +
// bias[co]), arg = the first maximal sample.  K = 128, Co a multiple of 64, N = 64 * centres.
extern "C" int geoa3_conv1x1_max64(const float* X, const float* W, const float* bias, float* out, int32_t* arg, int B,
                                   long N, int K, int Co, void* stream) {
  if (!X || !W || !bias || !out || !arg || B <= 0 || N <= 0 || N > 0x7fffffffL) return GEOA3_EINVAL;
  ConvArgs a{};
  a.split = 1;
  a.X = X; a.sXb = (long)K * N; a.ldX = (int)N;
  a.W = W; a.sWb = 0; a.sWco = K; a.sWk = 1;
  a.pool_out = out; a.pool_arg = arg; a.pool_bias = bias;
  a.Co = Co; a.K = K; a.N = (int)N; a.B = B;
  return launch_conv_cm(a, geoa3_stream(stream));
}

// Its input gradient: Y[b][co][64 m + s] = gate( sum_k W[co][k] * ((arg[b][m][k] == s) ? g[b][m][k] : 0) ), the sparse
// gradient of the pooled layer formed in registers (g, arg CENTRE-major [B][centres][K]; g must already carry the pooled
// output's relu gate); gate: keep
// where Z[b][co][n] > 0 (the relu of the layer below).  K = 256.
extern "C" int geoa3_conv1x1_onehot64(const float* g, const int32_t* arg, const float* W, const float* Z, float* Y, int B,
                                      long N, int K, int Co, void* stream) {
  if (!g || !arg || !W || !Y || B <= 0 || N <= 0 || N > 0x7fffffffL) return GEOA3_EINVAL;
  ConvArgs a{};
  a.split = 1;
  a.oh_g = g; a.oh_arg = arg;
  a.W = W; a.sWb = 0; a.sWco = K; a.sWk = 1;
  a.Z = Z; a.sZb = (long)Co * N; a.ldZ = (int)N;
  a.Y = Y; a.sYb = (long)Co * N; a.ldY = (int)N;
  a.Co = Co; a.K = K; a.N = (int)N; a.B = B;
  return launch_conv_cm(a, geoa3_stream(stream));
}
