// PointNet++ set-abstraction operators (gfx950): the six functions the reference's vendored CUDA extension
// `pointnet2_ops._ext` exports to the SSG classifier (Model/pointnet2_ops_lib/pointnet2_ops/_ext-src/src/
// bindings.cpp:6-19): furthest_point_sampling, gather_points(+grad), ball_query, group_points(+grad).
// Same argument layouts ([B,N,3] point-major xyz, [B,C,N] channel-major features, int32 indices) so that the
// Python autograd Functions of pointnet2_utils.py:34-101,194-276 bind to them unchanged.
//
// Semantics kept from the .cu sources:
//   * FPS (sampling_gpu.cu:69-173): starts at index 0; points with |p|^2 <= 1e-3 are skipped (never selected,
//     their running distance is not updated); arg-max ties go to the lowest (k mod T), then the lowest k, where
//     T = opt_n_threads(N) = largest power of two <= min(N, 512) (cuda_utils.h:13-19) is the reference's block.
//   * ball query (ball_query_gpu.cu:9-44): first `nsample` points IN INDEX ORDER with d^2 < r^2; the first hit
//     pre-fills every slot; no hit leaves zeros.
//   * gradients of gather / group are scatter-adds (atomicAdd, sampling_gpu.cu:42, group_points_gpu.cu:60).
// Distances are evaluated un-fused by default, fl(fl(dx*dx + dy*dy) + dz*dz) with each product rounded (the CPU oracle's
// order).  GEOA3_PN2_CONTRACT (the *_ex entry points, geoa3_pn2ssg_weights::flags) selects the form nvcc -O3 most likely
// gave the reference's binary (setup.py:32 passes no -fmad flag; the default contracts): fmaf(dz, dz, fmaf(dy, dy, dx*dx))
// for sampling_gpu.cu:100,103-104 and ball_query_gpu.cu:31-32 -- for users who compare indices with a CUDA run.
#include "pointnet_kernels.h"

namespace {

template <bool CT = false>
__device__ __forceinline__ float sq3(float dx, float dy, float dz) {
#pragma clang fp contract(off)
  if (CT) return __builtin_fmaf(dz, dz, __builtin_fmaf(dy, dy, dx * dx));
  const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
  const float s = xx + yy;
  return s + zz;
}

// ------------------------------------------------------------------------------------------
// FPS: one workgroup of 512 threads per cloud; every thread keeps up to 16 points and their running distances in
// registers, the cloud also sits in LDS so that the coordinates of the point selected in the previous round are an
// LDS read.  The reference's arg-max order -- larger distance, then lower (k mod T), then lower k -- is a max over
// 64-bit keys (distance bits | ~(k mod T) | ~k; distances are >= 0 so their bits order like integers; skipped points
// and padding carry key 0), reduced with DPP row operations inside a wavefront (common.h) and through one LDS slot per
// wave across wavefronts: ONE barrier per round, no global traffic.  0.66 ms -> see DESIGN for 512 of 1024 points.
// ------------------------------------------------------------------------------------------
// FPS_BLOCK threads x FPS_PPT points: 512 x 16 (up to 8192 points) or, for clouds of at most 2048 points, 256 x 8 -- one
// wave per SIMD: a round is a dependent chain (update, DPP reduction, LDS exchange), two waves on a SIMD only take turns
// issuing the same ~100 instructions

template <int FPS_BLOCK, int FPS_PPT, bool CT = false>
__global__ __launch_bounds__(FPS_BLOCK) void fps_kernel(const float* __restrict__ xyz, int N, int m, int T,
                                                        float* __restrict__ temp, int32_t* __restrict__ idxs, int j0, int j1) {
  extern __shared__ __attribute__((aligned(16))) float s_p[];   // [N][3]
  __shared__ unsigned long long s_key[2][FPS_BLOCK / 64];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float* P = xyz + (size_t)b * N * 3;
  for (int e = tid; e < N * 3; e += FPS_BLOCK) s_p[e] = P[e];
  // Straight-line rounds: every register slot is live (the launcher picks FPS_PPT = ceil(N / FPS_BLOCK) rounded up to a power
  // of two), a slot without a point or with a skipped one carries an all-zero select mask (bit-field insert, key 0) instead
  // of a branch around its update -- the branchy form spent most of a round in exec-mask bookkeeping (0.73 -> 0.5 us per round)
  float px[FPS_PPT], py[FPS_PPT], pz[FPS_PPT], td[FPS_PPT];
  unsigned um[FPS_PPT], klo[FPS_PPT];   // ~0 where the slot's point takes part; low key word: ~((k mod T) << 16 | k)
#pragma unroll
  for (int i = 0; i < FPS_PPT; ++i) {
    const int k = tid + i * FPS_BLOCK;
    const bool in = k < N;
    px[i] = in ? P[k * 3] : 0.f;
    py[i] = in ? P[k * 3 + 1] : 0.f;
    pz[i] = in ? P[k * 3 + 2] : 0.f;
    // sampling.cpp:74-76; rounds j0 .. j1 - 1 of the m: a launch that does not start at round 0 resumes from the running
    // distances the launch before it left in `temp` (the same fp32 values: the chunks select the same points bit for bit)
    td[i] = (j0 > 0 && in) ? temp[(size_t)b * N + k] : 1e10f;
    const float mag = sq3<CT>(px[i], py[i], pz[i]);
    um[i] = (in && !(mag <= 1e-3f)) ? 0xFFFFFFFFu : 0u;
    klo[i] = 0xFFFFFFFFu - (((unsigned)(k & (T - 1)) << 16) | (unsigned)k);   // k < 8192, T <= 512
  }
  int old = j0 > 0 ? idxs[(size_t)b * m + j0 - 1] : 0;
  if (j0 == 0 && tid == 0) idxs[(size_t)b * m] = 0;
  __syncthreads();
  for (int j = j0 > 1 ? j0 : 1; j < j1; ++j) {
    const float x1 = s_p[old * 3], y1 = s_p[old * 3 + 1], z1 = s_p[old * 3 + 2];
#ifdef GEOA3_FPS_PRENEG
    float nx1 = -x1, ny1 = -y1, nz1 = -z1;
    asm volatile("" : "+v"(nx1), "+v"(ny1), "+v"(nz1));   // (opaque: not folded back into a modifier)
    float nx1b = nx1, ny1b = ny1, nz1b = nz1;
    asm volatile("" : "+v"(nx1b), "+v"(ny1b), "+v"(nz1b));
#endif
    unsigned long long key = 0ull;
#pragma unroll
    for (int i = 0; i < FPS_PPT; ++i) {
#if defined(GEOA3_FPS_PRENEG) && GEOA3_FPS_PRENEG == 2   // (... and without op_sel: odd slots take a second copy of the negated point)
      const float d = (i & 1) ? sq3<CT>(px[i] + nx1b, py[i] + ny1b, pz[i] + nz1b) : sq3<CT>(px[i] + nx1, py[i] + ny1, pz[i] + nz1);
#elif defined(GEOA3_FPS_PRENEG)   // (tools/ub/pk_fp32_coresidency.hip: the packed subtraction without `neg` modifiers)
      const float d = sq3<CT>(px[i] + nx1, py[i] + ny1, pz[i] + nz1);
#else
      const float d = sq3<CT>(px[i] - x1, py[i] - y1, pz[i] - z1);
#endif
      const float dm = d < td[i] ? d : td[i];     // min(d, temp) (sampling_gpu.cu:118); a NaN distance leaves temp as it is
      // d2 = use ? min(d, td) : td, as a bit select
      const unsigned d2 = (__float_as_uint(dm) & um[i]) | (__float_as_uint(td[i]) & ~um[i]);
      td[i] = __uint_as_float(d2);
      const unsigned long long kk = ((unsigned long long)(d2 & um[i]) << 32) | (klo[i] & um[i]);   // 0 for a slot out of play
      key = kk > key ? kk : key;
    }
    key = wave_max_u64_split(key);
    const int buf = j & 1;   // double buffered: the next round's writes cannot race this round's reads
    if (lane == 0) s_key[buf][wave] = key;
    __syncthreads();
    // every lane reads the waves' keys (broadcast reads) and takes their maximum itself: no second DPP reduction in the
    // round's dependent chain
    unsigned long long k2 = s_key[buf][0];
#pragma unroll
    for (int w = 1; w < FPS_BLOCK / 64; ++w) {
      const unsigned long long kw = s_key[buf][w];
      k2 = kw > k2 ? kw : k2;
    }
    const unsigned lo = (unsigned)k2, hi = (unsigned)(k2 >> 32);
    // every point skipped: the reference's reduction returns its initial index 0
    old = (lo == 0u && hi == 0u) ? 0 : (int)((0xFFFFFFFFu - lo) & 0xFFFFu);
    if (tid == 0) idxs[(size_t)b * m + j] = old;
  }
  if (temp) {
#pragma unroll
    for (int i = 0; i < FPS_PPT; ++i) {
      const int k = tid + i * FPS_BLOCK;
      if (k < N) temp[(size_t)b * N + k] = td[i];
    }
  }
}

// out[b][c][j] = points[b][c][idx[b][j]]
__global__ __launch_bounds__(256) void gather_points_kernel(const float* __restrict__ points,
                                                            const int32_t* __restrict__ idx, float* __restrict__ out,
                                                            int C, int N, int M) {
  const int b = blockIdx.z, c = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  out[((size_t)b * C + c) * M + j] = points[((size_t)b * C + c) * N + idx[(size_t)b * M + j]];
}
__global__ __launch_bounds__(256) void gather_points_grad_kernel(const float* __restrict__ grad_out,
                                                                 const int32_t* __restrict__ idx,
                                                                 float* __restrict__ grad_points, int C, int N, int M) {
  const int b = blockIdx.z, c = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  atomicAdd(grad_points + ((size_t)b * C + c) * N + idx[(size_t)b * M + j], grad_out[((size_t)b * C + c) * M + j]);
}

// ball query: one thread per centre, the cloud of the instance staged in LDS in chunks; all lanes read the
// same point (broadcast) and stop individually once their ball is full.
constexpr int BQ_CHUNK = 1024;
template <bool CT = false>
__global__ __launch_bounds__(256) void ball_query_kernel(const float* __restrict__ new_xyz,
                                                         const float* __restrict__ xyz, int N, int M, float radius2,
                                                         int nsample, int32_t* __restrict__ idx) {
  __shared__ float s_p[BQ_CHUNK * 3];
  const int b = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  const bool live = j < M;
  const float* C = new_xyz + ((size_t)b * M + (live ? j : 0)) * 3;
  const float cx = C[0], cy = C[1], cz = C[2];
  int32_t* out = idx + ((size_t)b * M + (live ? j : 0)) * nsample;
  int cnt = live ? 0 : nsample;
  for (int k0 = 0; k0 < N; k0 += BQ_CHUNK) {
    const int kn = min(BQ_CHUNK, N - k0);
    __syncthreads();
    for (int e = threadIdx.x; e < kn * 3; e += 256) s_p[e] = xyz[((size_t)b * N + k0) * 3 + e];
    __syncthreads();
    if (__syncthreads_and(cnt >= nsample)) break;
    for (int k = 0; k < kn && !__all(cnt >= nsample); ++k) {
      const float d2 = sq3<CT>(cx - s_p[k * 3], cy - s_p[k * 3 + 1], cz - s_p[k * 3 + 2]);
      if (cnt < nsample && d2 < radius2) {
        if (cnt == 0)
          for (int l = 0; l < nsample; ++l) out[l] = k0 + k;
        out[cnt] = k0 + k;
        ++cnt;
      }
    }
  }
}

// nsample <= 64 and the cloud in LDS: one WAVEFRONT per centre, lane = point of a 64-point chunk.  The hits of a chunk
// are a ballot; a hit's slot is the running count + the number of hits below its lane (index order kept), the row of
// `nsample` indices is assembled in LDS and leaves as one coalesced store (no memset, no per-hit global writes; the
// thread-per-centre form above ran 250 us for 512 centres x 1024 points x 250 instances).
constexpr int BQW_CPW = 8;   // centres per wavefront
template <bool CT = false>
__global__ __launch_bounds__(256) void ball_query_wave_kernel(const float* __restrict__ new_xyz,
                                                              const float* __restrict__ xyz, int N, int M, float radius2,
                                                              int nsample, int32_t* __restrict__ idx, int m0, int m1, int cpw) {
  extern __shared__ __attribute__((aligned(16))) float s_bq[];   // x[N], y[N], z[N], rows [4][64]
  float *s_x = s_bq, *s_y = s_bq + N, *s_z = s_bq + 2 * N;
  const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* row = reinterpret_cast<int*>(s_bq + 3 * N) + wave * 64;
  for (int e = tid; e < N; e += 256) {
    const float* p = xyz + ((size_t)b * N + e) * 3;
    s_x[e] = p[0];
    s_y[e] = p[1];
    s_z[e] = p[2];
  }
  __syncthreads();
  const int j0 = m0 + __builtin_amdgcn_readfirstlane((blockIdx.x * 4 + wave) * cpw);   // centres m0 .. m1 - 1 of the M
  for (int c = 0; c < cpw; ++c) {
    const int j = j0 + c;
    if (j >= m1) break;
    const float* C = new_xyz + ((size_t)b * M + j) * 3;
    const float cx = C[0], cy = C[1], cz = C[2];
    int cnt = 0;
    for (int k0 = 0; k0 < N && cnt < nsample; k0 += 64) {
      const int k = k0 + lane;
      bool hit = false;
      if (k < N) hit = sq3<CT>(cx - s_x[k], cy - s_y[k], cz - s_z[k]) < radius2;
      const unsigned long long mask = __ballot(hit);
      if (mask == 0ull) continue;
      if (cnt == 0) row[lane] = k0 + (int)__builtin_ctzll(mask);     // the first hit pre-fills every slot
      const int slot = cnt + (int)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
      if (hit && slot < nsample) row[slot] = k;
      cnt += (int)__builtin_popcountll(mask);
    }
    if (lane < nsample) idx[((size_t)b * M + j) * nsample + lane] = cnt ? row[lane] : 0;
  }
}

// out[b][c][j][k] = points[b][c][idx[b][j][k]]
__global__ __launch_bounds__(256) void group_points_kernel(const float* __restrict__ points,
                                                           const int32_t* __restrict__ idx, float* __restrict__ out,
                                                           int C, int N, int MS) {
  const int b = blockIdx.z, c = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (e >= MS) return;
  out[((size_t)b * C + c) * MS + e] = points[((size_t)b * C + c) * N + idx[(size_t)b * MS + e]];
}
// out[b][c][j][k] = relu(points[b][c][idx[b][j][k]] + shift[b][c][j])  (geoa3_pn2_group_shift_relu)
// one wavefront per (instance, centre row j): the index row is loaded once and serves all C channels (one thread per
// output element re-read it C times: 0.74 ms for [250,128,128,64]); lanes cover the row's samples in steps of 64
__global__ __launch_bounds__(256) void group_shift_relu_kernel(const float* __restrict__ points,
                                                               const int32_t* __restrict__ idx,
                                                               const float* __restrict__ shift, float* __restrict__ out,
                                                               int C, int N, int M, int S) {
  const int b = blockIdx.y, j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (j >= M) return;
  const float* P = points + (size_t)b * C * N;
  const float* sh = shift + (size_t)b * C * M + j;
  float* O = out + ((size_t)b * C * M + j) * S;
  for (int k = lane; k < S; k += 64) {
    const int i = idx[((size_t)b * M + j) * S + k];
#pragma unroll 8
    for (int c = 0; c < C; ++c) O[(size_t)c * M * S + k] = fmaxf(P[(size_t)c * N + i] + sh[(size_t)c * M], 0.f);
  }
}
__global__ __launch_bounds__(256) void group_points_grad_kernel(const float* __restrict__ grad_out,
                                                                const int32_t* __restrict__ idx,
                                                                float* __restrict__ grad_points, int C, int N, int MS) {
  const int b = blockIdx.z, c = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (e >= MS) return;
  atomicAdd(grad_points + ((size_t)b * C + c) * N + idx[(size_t)b * MS + e], grad_out[((size_t)b * C + c) * MS + e]);
}

// nsample == 64: one workgroup per (instance, GPG_CT channels) accumulates into LDS (ds_add_f32) and writes every
// output row once with plain stores -- no global atomics, no memset.  A wavefront takes one (centre) row of 64
// samples: the index row is loaded once and reused for the GPG_CT channels.  The ball query pads a short ball by
// repeating its FIRST index, so up to 63 lanes of a row would hit one address; entries equal to idx[row][0] are
// summed with wave shuffles first and leave as ONE add.  (Valid for any index table: only entries equal to the
// row's first one are merged.)  Measured at [250,128,128,64] -> [250,128,512]: 4.5 ms with global atomics.
constexpr int GPG_CT = 8;
// PRIV: every wavefront accumulates into its OWN copy of the accumulators ([4][GPG_CT][N]) -- its rows in a fixed order,
// no other wave touching the copy -- and the four copies are added in wave order at the end: the sum no longer depends
// on how the waves' atomics interleave (deterministic; the reference's scatter-add, group_points_gpu.cu:60, is not).
// The loads are software-pipelined GPG_PF rows deep (a wave that loads one row and waits keeps 2 KB in flight: 1.25 TB/s
// at two workgroups per CU), and the 2 x GPG_CT row reductions (totals, padded entries) are done TRANSPOSED: two
// half-swaps leave every lane with a quarter of the values, four DPP steps finish them -- 40 instructions instead of 16
// full wave reductions.
__device__ __forceinline__ void gpg_swap32(float& a, float& b) {   // a[32..63] <-> b[0..31]
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ void gpg_swap16(float& a, float& b) {   // a's odd rows of 16 <-> b's even rows
  const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
  a = __uint_as_float(r[0]);
  b = __uint_as_float(r[1]);
}
__device__ __forceinline__ float gpg_row_sum(float v) {            // sum over the 16 lanes of a DPP row, in every lane
  v += dpp_f32<0xB1, 0xF>(v, 0.f);
  v += dpp_f32<0x4E, 0xF>(v, 0.f);
  v += dpp_f32<0x141, 0xF>(v, 0.f);
  v += dpp_f32<0x140, 0xF>(v, 0.f);
  return v;
}
template <bool PRIV, int MODE = 0, int GPG_PF = 8, int CT = GPG_CT>   // MODE != 0, GPG_PF: variants for tools/ub/gpg_ub.hip;
                                                                 // CT: channels per workgroup (4 or 8: the same sums, bit for bit --
                                                                 // every value goes through the same reduction tree)
__global__ __launch_bounds__(256) void group_points_grad64_kernel(const float* __restrict__ grad_out,
                                                                  const int32_t* __restrict__ idx,
                                                                  float* __restrict__ grad_points, int C, int N, int M,
                                                                  float* __restrict__ rowsum) {
  extern __shared__ __attribute__((aligned(16))) float s_all[];   // [PRIV ? 4 : 1][CT][N]
  const int b = blockIdx.y, c0 = blockIdx.x * CT, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nc = min(CT, C - c0);
  float* s_acc = s_all + (PRIV ? wave * CT * N : 0);
  const float* G = grad_out + ((size_t)b * C + c0) * M * 64;
  const int32_t* I = idx + (size_t)b * M * 64;
  float v[GPG_PF][CT];
  int iv[GPG_PF];
  auto load_row = [&](int j, float (&d)[CT], int& i) {
    const int jj = j < M ? j : M - 1;
    i = I[(size_t)jj * 64 + lane];
#pragma unroll
    for (int c = 0; c < CT; ++c) d[c] = c < nc ? G[((size_t)c * M + jj) * 64 + lane] : 0.f;
  };
#pragma unroll
  for (int p = 0; p < GPG_PF; ++p) load_row(wave + 4 * p, v[p], iv[p]);   // in flight while the accumulators are cleared
  for (int e = tid; e < (PRIV ? 4 : 1) * CT * N; e += 256) s_all[e] = 0.f;
  __syncthreads();
  for (int j0 = wave; j0 < M; j0 += 4 * GPG_PF) {
#pragma unroll
    for (int p = 0; p < GPG_PF; ++p) {
      const int j = j0 + 4 * p;
      if (j < M) {   // wave-uniform
        const int i = iv[p];
        const int i0 = __builtin_amdgcn_readfirstlane(i);
        const bool dup = lane > 0 && i == i0;
        // A ball query's row -- indices ascending, then repeats of the first -- has distinct live entries: plain
        // read-modify-writes then (ds_add_f32 runs at a fraction of the plain LDS rate: 905 us against 401 us for this
        // kernel at [250,128,128,64]); any other table takes the atomics.  Same sums either way.
        const int iprev = __shfl_up(i, 1, 64);
        const bool dprev = __shfl_up((int)dup, 1, 64) != 0;
        const bool distinct = __all(lane == 0 || dup || (i > iprev && !dprev)) != 0;
        // r[0..7]: the row totals, r[8..15]: the sums over the padded entries -- reduced transposed
        float r[2 * CT];
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          r[c] = v[p][c];
          r[CT + c] = dup ? v[p][c] : 0.f;
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {      // lanes 0-31 keep value c, lanes 32-63 value 8 + c
          gpg_swap32(r[c], r[CT + c]);
          r[c] += r[CT + c];
        }
#pragma unroll
        for (int c = 0; c < CT / 2; ++c) {  // even rows keep value c, odd rows value 4 + c (of their half's eight)
          gpg_swap16(r[c], r[CT / 2 + c]);
          r[c] += r[CT / 2 + c];
        }
#pragma unroll
        for (int c = 0; c < CT / 2; ++c) r[c] = gpg_row_sum(r[c]);
        // row 0 (lane 0): totals 0-3, row 1 (lane 16): totals 4-7, row 2 (lane 32): padded sums 0-3, row 3 (lane 48): 4-7
        float dsum[CT];
#pragma unroll
        for (int c = 0; c < CT / 2; ++c) {
          dsum[c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r[c]), 32));
          dsum[CT / 2 + c] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r[c]), 48));
        }
#pragma unroll
        for (int c = 0; c < CT; ++c) {
          if (c >= nc) break;
          // (plain updates only on a wave's OWN copy: with one shared copy other waves' rows hit the same points)
          if (MODE == 1 || (MODE == 0 && PRIV && distinct)) {
            if (!dup) s_acc[c * N + i] += lane == 0 ? v[p][c] + dsum[c] : v[p][c];
          } else if (MODE == 2) {
            if (v[p][c] + dsum[c] == 123.456f) s_acc[c * N + i] = 1.f;
          } else {
            if (lane == 0) atomicAdd(&s_acc[c * N + i], v[p][c] + dsum[c]);
            else if (!dup) atomicAdd(&s_acc[c * N + i], v[p][c]);
          }
        }
        if (rowsum && (lane == 0 || lane == 16)) {   // d shift of the pre-transformed first layer: the row totals
          const int cb = lane ? CT / 2 : 0;
#pragma unroll
          for (int c = 0; c < CT / 2; ++c)
            if (cb + c < nc) rowsum[((size_t)b * C + c0 + cb + c) * M + j] = r[c];
        }
      }
      load_row(j + 4 * GPG_PF, v[p], iv[p]);   // (clamped past the end)
    }
  }
  __syncthreads();
  float* dst = grad_points + ((size_t)b * C + c0) * N;
  const int P = CT * N;
  for (int e = tid; e < nc * N; e += 256)
    dst[e] = PRIV ? ((s_all[e] + s_all[P + e]) + s_all[2 * P + e]) + s_all[3 * P + e] : s_all[e];
}

}  // namespace

// rounds j0 .. j1 - 1 of the m (0 <= j0 < j1 <= m); j0 > 0 resumes from `temp` [B,N] (required then), which every launch
// with temp != null leaves behind
int launch_pn2_fps_range(const float* xyz, int B, int N, int m, int j0, int j1, float* temp, int32_t* idx, hipStream_t s,
                         bool contract) {
  if (!xyz || !idx || B <= 0 || N <= 0 || m <= 0 || j0 < 0 || j1 <= j0 || j1 > m || (j0 > 0 && !temp)) return GEOA3_EINVAL;
  if (N > 512 * 16) return GEOA3_ENOSUPPORT;
  int T = 1;
  while (T * 2 <= N && T * 2 <= 512) T *= 2;  // opt_n_threads(N), cuda_utils.h:13-19
  const size_t lds = (size_t)N * 3 * sizeof(float);
#define GEOA3_FPS(BLK, PPT)                                                                                              \
  do {                                                                                                                   \
    if (contract) {                                                                                                      \
      if (lds > 48 * 1024)                                                                                               \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel<BLK, PPT, true>),                             \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
      hipLaunchKernelGGL((fps_kernel<BLK, PPT, true>), dim3(B), dim3(BLK), lds, s, xyz, N, m, T, temp, idx, j0, j1);      \
    } else {                                                                                                             \
      if (lds > 48 * 1024)                                                                                               \
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fps_kernel<BLK, PPT, false>),                            \
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                 \
      hipLaunchKernelGGL((fps_kernel<BLK, PPT, false>), dim3(B), dim3(BLK), lds, s, xyz, N, m, T, temp, idx, j0, j1);     \
    }                                                                                                                    \
  } while (0)
  if (N <= 512) GEOA3_FPS(256, 2);
  else if (N <= 1024) GEOA3_FPS(256, 4);
  else if (N <= 2048) GEOA3_FPS(256, 8);
  else GEOA3_FPS(512, 16);
#undef GEOA3_FPS
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_pn2_furthest_point_sampling(const float* xyz, int B, int N, int m, float* temp, int32_t* idx,
                                                 void* stream) {
  return launch_pn2_fps_range(xyz, B, N, m, 0, m, temp, idx, geoa3_stream(stream), false);
}
extern "C" int geoa3_pn2_furthest_point_sampling_ex(const float* xyz, int B, int N, int m, float* temp, int32_t* idx,
                                                    int flags, void* stream) {
  if (flags & ~GEOA3_PN2_CONTRACT) return GEOA3_EINVAL;
  return launch_pn2_fps_range(xyz, B, N, m, 0, m, temp, idx, geoa3_stream(stream), (flags & GEOA3_PN2_CONTRACT) != 0);
}

extern "C" int geoa3_pn2_gather_points(const float* points, const int32_t* idx, int B, int C, int N, int M, float* out,
                                       void* stream) {
  if (!points || !idx || !out || B <= 0 || C <= 0 || N <= 0 || M <= 0) return GEOA3_EINVAL;
  hipLaunchKernelGGL(gather_points_kernel, dim3((M + 255) / 256, C, B), dim3(256), 0, geoa3_stream(stream), points, idx,
                     out, C, N, M);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_pn2_gather_points_grad(const float* grad_out, const int32_t* idx, int B, int C, int N, int M,
                                            float* grad_points, void* stream) {
  if (!grad_out || !idx || !grad_points || B <= 0 || C <= 0 || N <= 0 || M <= 0) return GEOA3_EINVAL;
  if (hipMemsetAsync(grad_points, 0, (size_t)B * C * N * sizeof(float), geoa3_stream(stream)) != hipSuccess)
    return GEOA3_ELAUNCH;
  hipLaunchKernelGGL(gather_points_grad_kernel, dim3((M + 255) / 256, C, B), dim3(256), 0, geoa3_stream(stream),
                     grad_out, idx, grad_points, C, N, M);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

// centres m0 .. m1 - 1 of the M (the wave-per-centre kernel; GEOA3_ENOSUPPORT where the public entry point falls back)
int launch_pn2_ball_query_range(const float* new_xyz, const float* xyz, int B, int N, int M, int m0, int m1, float radius,
                                int nsample, int32_t* idx, hipStream_t s, bool contract) {
  if (!new_xyz || !xyz || !idx || B <= 0 || N <= 0 || M <= 0 || nsample <= 0 || m0 < 0 || m1 <= m0 || m1 > M) return GEOA3_EINVAL;
  const size_t lds = (size_t)N * 3 * sizeof(float) + 4 * 64 * sizeof(int);
  if (nsample > 64 || lds > 128 * 1024) return GEOA3_ENOSUPPORT;
  auto kern = contract ? ball_query_wave_kernel<true> : ball_query_wave_kernel<false>;
  if (lds > 48 * 1024)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  // centres per wavefront: 8 for a whole level (the cloud is staged once per 32 centres); a partial range is a short launch
  // beside other kernels (pointnet2_net.hip), where 8 centres in sequence per wave are its whole duration: 2
  const int cpw = (m1 - m0 == M) ? BQW_CPW : 2;
  hipLaunchKernelGGL(kern, dim3((m1 - m0 + 4 * cpw - 1) / (4 * cpw), B), dim3(256), lds, s, new_xyz, xyz,
                     N, M, radius * radius, nsample, idx, m0, m1, cpw);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

static int ball_query_impl(const float* new_xyz, const float* xyz, int B, int N, int M, float radius, int nsample, int32_t* idx,
                           bool contract, void* stream) {
  if (!new_xyz || !xyz || !idx || B <= 0 || N <= 0 || M <= 0 || nsample <= 0) return GEOA3_EINVAL;
  const int rc = launch_pn2_ball_query_range(new_xyz, xyz, B, N, M, 0, M, radius, nsample, idx, geoa3_stream(stream), contract);
  if (rc != GEOA3_ENOSUPPORT) return rc;
  if (hipMemsetAsync(idx, 0, (size_t)B * M * nsample * sizeof(int32_t), geoa3_stream(stream)) != hipSuccess)
    return GEOA3_ELAUNCH;
  auto kern = contract ? ball_query_kernel<true> : ball_query_kernel<false>;
  hipLaunchKernelGGL(kern, dim3((M + 255) / 256, B), dim3(256), 0, geoa3_stream(stream), new_xyz, xyz, N, M,
                     radius * radius, nsample, idx);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
extern "C" int geoa3_pn2_ball_query(const float* new_xyz, const float* xyz, int B, int N, int M, float radius,
                                    int nsample, int32_t* idx, void* stream) {
  return ball_query_impl(new_xyz, xyz, B, N, M, radius, nsample, idx, false, stream);
}
extern "C" int geoa3_pn2_ball_query_ex(const float* new_xyz, const float* xyz, int B, int N, int M, float radius,
                                       int nsample, int32_t* idx, int flags, void* stream) {
  if (flags & ~GEOA3_PN2_CONTRACT) return GEOA3_EINVAL;
  return ball_query_impl(new_xyz, xyz, B, N, M, radius, nsample, idx, (flags & GEOA3_PN2_CONTRACT) != 0, stream);
}

extern "C" int geoa3_pn2_group_points(const float* points, const int32_t* idx, int B, int C, int N, int M, int nsample,
                                      float* out, void* stream) {
  if (!points || !idx || !out || B <= 0 || C <= 0 || N <= 0 || M <= 0 || nsample <= 0) return GEOA3_EINVAL;
  const int MS = M * nsample;
  hipLaunchKernelGGL(group_points_kernel, dim3((MS + 255) / 256, C, B), dim3(256), 0, geoa3_stream(stream), points, idx,
                     out, C, N, MS);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_pn2_group_shift_relu(const float* points, const int32_t* idx, const float* shift, int B, int C, int N,
                                          int M, int nsample, float* out, void* stream) {
  if (!points || !idx || !shift || !out || B <= 0 || C <= 0 || N <= 0 || M <= 0 || nsample <= 0) return GEOA3_EINVAL;
  const int MS = M * nsample;
  (void)MS;
  hipLaunchKernelGGL(group_shift_relu_kernel, dim3((M + 3) / 4, B), dim3(256), 0, geoa3_stream(stream), points, idx, shift,
                     out, C, N, M, nsample);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

static int group_points_grad_impl(const float* grad_out, const int32_t* idx, int B, int C, int N, int M, int nsample,
                                  float* grad_points, float* rowsum, void* stream);

extern "C" int geoa3_pn2_group_points_grad(const float* grad_out, const int32_t* idx, int B, int C, int N, int M,
                                           int nsample, float* grad_points, void* stream) {
  return group_points_grad_impl(grad_out, idx, B, C, N, M, nsample, grad_points, nullptr, stream);
}

// + rowsum[b][c][m] = sum_s grad_out[b][c][m][s] in the same pass (nsample == 64 and C * N small enough for LDS only)
extern "C" int geoa3_pn2_group_points_grad_sums(const float* grad_out, const int32_t* idx, int B, int C, int N, int M,
                                                int nsample, float* grad_points, float* rowsum, void* stream) {
  if (!rowsum || nsample != 64 || (size_t)GPG_CT * N * sizeof(float) > 128 * 1024) return GEOA3_ENOSUPPORT;
  return group_points_grad_impl(grad_out, idx, B, C, N, M, nsample, grad_points, rowsum, stream);
}

static int group_points_grad_impl(const float* grad_out, const int32_t* idx, int B, int C, int N, int M, int nsample,
                                  float* grad_points, float* rowsum, void* stream) {
  if (!grad_out || !idx || !grad_points || B <= 0 || C <= 0 || N <= 0 || M <= 0 || nsample <= 0) return GEOA3_EINVAL;
  const int MS = M * nsample;
  const size_t lds = (size_t)GPG_CT * N * sizeof(float);
  if (nsample == 64 && 4 * lds <= 128 * 1024) {          // per-wave accumulators: deterministic
#ifndef GEOA3_GPG_CT
#define GEOA3_GPG_CT 4
#endif
    // four channels per workgroup: a wave's accumulators are 8 KB instead of 16 (N = 512), 16 waves per CU instead of 8 --
    // the kernel streams 1.1 GB at configs[3] and is bound by the latency of those reads; the index rows are read twice as
    // often (from L2).  Same bits as eight channels per workgroup (-DGEOA3_GPG_CT=8); configs[3] 5.26 -> 5.10 ms; two channels
    // or sixteen rows in flight give nothing more.
    constexpr int CT = GEOA3_GPG_CT;
    auto kern = group_points_grad64_kernel<true, 0, 8, CT>;
    const size_t l4 = (size_t)4 * CT * N * sizeof(float);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)l4);
    hipLaunchKernelGGL(kern, dim3((C + CT - 1) / CT, B), dim3(256), l4, geoa3_stream(stream), grad_out, idx, grad_points, C,
                       N, M, rowsum);
  } else if (nsample == 64 && lds <= 128 * 1024) {
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(group_points_grad64_kernel<false>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(group_points_grad64_kernel<false>, dim3((C + GPG_CT - 1) / GPG_CT, B), dim3(256), lds,
                       geoa3_stream(stream), grad_out, idx, grad_points, C, N, M, rowsum);
  } else {
    if (hipMemsetAsync(grad_points, 0, (size_t)B * C * N * sizeof(float), geoa3_stream(stream)) != hipSuccess)
      return GEOA3_ELAUNCH;
    hipLaunchKernelGGL(group_points_grad_kernel, dim3((MS + 255) / 256, C, B), dim3(256), 0, geoa3_stream(stream),
                       grad_out, idx, grad_points, C, N, MS);
  }
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
