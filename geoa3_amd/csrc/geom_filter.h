// Exact all-pairs 1-NN where a cell structure does not pay (geom_grid.hip decides; geom_filter.hip for clouds of up to 1024
// and of more than 4096 points): the matrix core FILTERS, the vector unit decides.
//
// Where a query's ball holds a large part of the searched cloud (dense clusters, thin rods, iterates far from the
// surface) the grid walk degenerates to the all-pairs work at ~11 vector instructions per (query, candidate) pair.  Here
// one v_mfma_f32_32x32x16_f16 forms 32 x 32 APPROXIMATE squared distances -- |p|^2 - 2 p.q from split-fp16 coordinates
// (hi + lo pieces, three products per coordinate, |p|^2 in three pieces: 12 of the 16 k) -- and a pair is evaluated
// exactly only if its approximation does not exceed the query's threshold: the exact distance to the query's seed (last
// iteration's neighbour) plus a bound on the approximation's error.  The exact evaluation is the un-fused distance of
// every other search, candidates compared lexicographically on (distance, index): the results are BIT-IDENTICAL to the
// all-pairs kernel (tests/test_gpu_geometry.py, tests/test_gpu_cad.py).
//
// Error bound (coordinates centred on the searched cloud's box and scaled so that |u| <= 1 in every coordinate; distances
// in those units): fp16 pieces u = h + l + r with |r| <= 2^-21 even if the matrix core flushes fp16 denormals (pieces are
// taken at 128 u); dropped l.l and r terms <= 6 * 2.5 * 2^-21 = 7.2e-6; fp32 accumulation of 12 terms, sums <= 9, one
// ulp each (truncation allowed) <= 1.3e-5; |p|^2, |q|^2, the scaling and the exact formula's own rounding <= 4e-6: 2.5e-5
// in all, NF_EPS = 2^-15 = 3.05e-5.  A wider band only costs exact evaluations, never a result: on a unit sphere sampled
// with 4096 points the band admits the neighbours within 0.006 of a query's seed distance.
// Clouds whose extent is not finite (or absurd: outside 2^+-60) take the plain exact sweep at the end of search().
#pragma once
#include "geom_internal.h"

namespace nf {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float nf_f32x16 __attribute__((ext_vector_type(16)));

constexpr int NF_B = 4;                   // blocks of 32 queries per wave: one A operand read feeds four products
constexpr int NF_QW = 32 * NF_B;          // queries per wave
constexpr int NF_CH = 1024;               // candidates per chunk (32 tiles): what the workgroup keeps in LDS at a time
constexpr int NF_LIST = 384;              // (query, half tile) items a wave collects before it evaluates them
constexpr float NF_EPS = 1.0f / 32768.0f;
constexpr float NF_UNIT = 16384.0f;       // accumulator units: products of coordinates taken at 128 u
constexpr float NF_INF = __builtin_inff();
template <int NT>   // threads per workgroup: its NT / 64 waves search a slice of NT * 2 queries
struct Cfg {
  static constexpr int W = NT / 64, SLICE = W * NF_QW;
  static constexpr size_t LDS = (size_t)NF_CH * 32 + 3 * NF_CH * 4 + 3 * SLICE * 4 + SLICE * 8 + (size_t)W * NF_LIST * 4 + W * 8 * 4;
};

// This file is compiled with -fno-honor-nans (geoa3_amd/build.py): fminf over matrix-core results then is v_min3_f32 without
// a quieting v_max_f32 v, v, v per operand (15 instead of 8 instructions per 16 accumulators), and no NaN can reach those
// minima (finite operands of bounded magnitude, see `ok`).  What must SEE a NaN therefore looks at the bits.
__device__ __forceinline__ float nf_abs_or_inf(float a) {   // |a|, +inf for NaN / inf
  const unsigned u = geoa3_opaque_bits(a) & 0x7fffffffu;
  return u >= 0x7f800000u ? NF_INF : __uint_as_float(u);
}
__device__ __forceinline__ void nf_split(float u, _Float16& h, _Float16& l) {
  h = (_Float16)u;
  l = (_Float16)(u - (float)h);
}
__device__ __forceinline__ float nf_min16(const nf_f32x16& a) {   // 8 x v_min3_f32
  const float m0 = fminf(fminf(a[0], a[1]), a[2]), m1 = fminf(fminf(a[3], a[4]), a[5]), m2 = fminf(fminf(a[6], a[7]), a[8]);
  const float m3 = fminf(fminf(a[9], a[10]), a[11]), m4 = fminf(fminf(a[12], a[13]), a[14]);
  return fminf(fminf(fminf(m0, m1), m2), fminf(fminf(m3, m4), a[15]));
}

// The queries q0 .. q0 + Cfg<NT>::SLICE - 1 of Q [3][Nq] against all M points of P [3][M]: (distance, index) -> dout / iout.
// prior (optional, one searched-cloud index per query, MAY ALIAS iout): seeds the thresholds (default: the query's own
// index).  Every thread of the workgroup calls it (barriers inside); smem: Cfg<NT>::LDS bytes, 16-byte aligned.
template <int NT>
__device__ __forceinline__ void search(const float* __restrict__ P, const float* __restrict__ Q, int M, int Nq, int q0,
                                       const int32_t* prior, float* dout, int32_t* iout, unsigned char* f_smem) {
  constexpr int NF_T = NT, NF_W = Cfg<NT>::W, NF_FILTER_SLICE = Cfg<NT>::SLICE;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, kh = lane >> 5;
  half8* s_img = reinterpret_cast<half8*>(f_smem);                           // [32 tiles][2 k halves][32 rows]
  float* s_px = reinterpret_cast<float*>(s_img + NF_CH * 2);                 // the chunk's exact coordinates, planar
  float* s_py = s_px + NF_CH;
  float* s_pz = s_py + NF_CH;
  float* s_qx = s_pz + NF_CH;                                                // the workgroup's queries, planar
  float* s_qy = s_qx + NF_FILTER_SLICE;
  float* s_qz = s_qy + NF_FILTER_SLICE;
  unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_qz + NF_FILTER_SLICE);   // [NF_FILTER_SLICE]
  unsigned* s_list = reinterpret_cast<unsigned*>(s_key + NF_FILTER_SLICE);   // [NF_W][NF_LIST]
  float* s_red = reinterpret_cast<float*>(s_list + NF_W * NF_LIST);          // [NF_W * 8]

  // this lane's queries: block k of the wave holds queries q0 + wave * NF_QW + 32 k + (lane % 32), both k halves alike;
  // their seeds (requested first: two dependent trips to memory, hidden behind the pass over the cloud)
  int qt[NF_B], si[NF_B];
#pragma unroll
  for (int k = 0; k < NF_B; ++k) {
    qt[k] = q0 + wave * NF_QW + 32 * k + n;
    const bool valid = qt[k] < Nq;
    si[k] = prior ? prior[valid ? qt[k] : Nq - 1] : qt[k];
    si[k] = si[k] < 0 ? 0 : (si[k] >= M ? M - 1 : si[k]);
  }
  float sx[NF_B], sy[NF_B], sz[NF_B];
#pragma unroll
  for (int k = 0; k < NF_B; ++k) {
    sx[k] = P[si[k]];
    sy[k] = P[M + si[k]];
    sz[k] = P[2 * M + si[k]];
  }
  // ---- centre and scale: the box of the searched cloud AND this workgroup's queries, in one reduction.  The candidates of
  // the first chunk stay in registers for its images (a cloud of up to 1024 points is read once)
  constexpr int CPT = NF_CH / NF_T;      // candidates per thread and chunk
  static_assert(CPT * NF_T == NF_CH && CPT <= 4, "a chunk is a whole number of candidates per thread");
  float kx[CPT], ky[CPT], kz[CPT];
  float lox = NF_INF, loy = NF_INF, loz = NF_INF, hix = -NF_INF, hiy = -NF_INF, hiz = -NF_INF, pbad = 0.f;
  for (int i0 = 0; i0 < M; i0 += 4 * NF_T) {
    float x[4], y[4], z[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {          // twelve loads in flight
      const int i = i0 + u * NF_T + tid, ii = i < M ? i : M - 1;
      x[u] = P[ii];
      y[u] = P[M + ii];
      z[u] = P[2 * M + ii];
    }
    if (i0 == 0) {
#pragma unroll
      for (int u = 0; u < CPT; ++u) {
        kx[u] = x[u];
        ky[u] = y[u];
        kz[u] = z[u];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      lox = fminf(lox, x[u]); hix = fmaxf(hix, x[u]);
      loy = fminf(loy, y[u]); hiy = fmaxf(hiy, y[u]);
      loz = fminf(loz, z[u]); hiz = fmaxf(hiz, z[u]);
      pbad = fmaxf(pbad, fmaxf(fmaxf(nf_abs_or_inf(x[u]), nf_abs_or_inf(y[u])), nf_abs_or_inf(z[u])));
    }
  }
  float qx[NF_B], qy[NF_B], qz[NF_B];
#pragma unroll
  for (int k = 0; k < NF_B; ++k) {
    const int q = qt[k] < Nq ? qt[k] : Nq - 1;
    qx[k] = Q[q];
    qy[k] = Q[Nq + q];
    qz[k] = Q[2 * Nq + q];
    if (kh == 0) {
      s_qx[wave * NF_QW + 32 * k + n] = qx[k];
      s_qy[wave * NF_QW + 32 * k + n] = qy[k];
      s_qz[wave * NF_QW + 32 * k + n] = qz[k];
    }
    lox = fminf(lox, qx[k]); hix = fmaxf(hix, qx[k]);
    loy = fminf(loy, qy[k]); hiy = fmaxf(hiy, qy[k]);
    loz = fminf(loz, qz[k]); hiz = fmaxf(hiz, qz[k]);
    pbad = fmaxf(pbad, fmaxf(fmaxf(nf_abs_or_inf(qx[k]), nf_abs_or_inf(qy[k])), nf_abs_or_inf(qz[k])));
  }
  lox = -wave_max(-lox); loy = -wave_max(-loy); loz = -wave_max(-loz);
  hix = wave_max(hix); hiy = wave_max(hiy); hiz = wave_max(hiz);
  pbad = wave_max(pbad);
  if (lane == 0) {
    float* r = s_red + wave * 8;
    r[0] = lox; r[1] = loy; r[2] = loz; r[3] = hix; r[4] = hiy; r[5] = hiz; r[6] = pbad;
  }
  __syncthreads();
#pragma unroll
  for (int w = 0; w < NF_W; ++w) {
    const float* r = s_red + w * 8;
    lox = fminf(lox, r[0]); loy = fminf(loy, r[1]); loz = fminf(loz, r[2]);
    hix = fmaxf(hix, r[3]); hiy = fmaxf(hiy, r[4]); hiz = fmaxf(hiz, r[5]);
    pbad = fmaxf(pbad, r[6]);
  }
  const float cx = 0.5f * lox + 0.5f * hix, cy = 0.5f * loy + 0.5f * hiy, cz = 0.5f * loz + 0.5f * hiz;
  // (every point and query within m of the centre in every coordinate; 1.0001: the centre's own rounding)
  float m = fmaxf(fmaxf(fmaxf(hix - cx, cx - lox), fmaxf(hiy - cy, cy - loy)), fmaxf(hiz - cz, cz - loz)) * 1.0001f;
  m = geoa3_opaque_bits(pbad) >= 0x7f800000u ? NF_INF : m;
  const unsigned Em = (geoa3_opaque_bits(m) >> 23) & 0xffu;
  const bool ok = Em <= 127u + 60u && Em >= 127u - 60u;   // (workgroup-uniform; inf: 255; all points equal: 0)

  // seeds, keys
  float seed_d[NF_B];
#pragma unroll
  for (int k = 0; k < NF_B; ++k) {
    seed_d[k] = geoa3_sqdist(qx[k], qy[k], qz[k], sx[k], sy[k], sz[k]);
    if (kh == 0 && qt[k] < Nq)
      s_key[wave * NF_QW + 32 * k + n] = ((unsigned long long)__float_as_uint(seed_d[k]) << 32) | (unsigned)si[k];
  }

  if (ok) {
    // u = (x - c) / s with s just above m: |u| <= 1 (s is not a power of two: u's rounding, 2^-24 relative, is part of the
    // error bound above; so is the rounding of the scaled threshold)
    const float inv_s = 1.0f / (m * 1.000001f);
    const float us = inv_s * 128.f, inv_s2 = inv_s * inv_s;
    // ---- query operands and thresholds
    half8 Bq[NF_B];
    float T[NF_B], Tb[NF_B];      // threshold = (distance bound / s^2) * UNIT + Tb
#pragma unroll
    for (int k = 0; k < NF_B; ++k) {
      const float ux = (qx[k] - cx) * us, uy = (qy[k] - cy) * us, uz = (qz[k] - cz) * us;
      _Float16 xh, xl, yh, yl, zh, zl;
      nf_split(ux, xh, xl);
      nf_split(uy, yh, yl);
      nf_split(uz, zh, zl);
      const _Float16 one = (_Float16)1.f, zero = (_Float16)0.f;
      if (kh == 0) {
        Bq[k][0] = xh; Bq[k][1] = xl; Bq[k][2] = xh; Bq[k][3] = yh; Bq[k][4] = yl; Bq[k][5] = yh; Bq[k][6] = zh; Bq[k][7] = zl;
      } else {
        Bq[k][0] = zh; Bq[k][1] = one; Bq[k][2] = one; Bq[k][3] = one; Bq[k][4] = zero; Bq[k][5] = zero; Bq[k][6] = zero; Bq[k][7] = zero;
      }
      const float Qn = ux * ux + uy * uy + uz * uz;
      Tb[k] = qt[k] < Nq ? NF_EPS * NF_UNIT - Qn : -NF_INF;
      T[k] = seed_d[k] * inv_s2 * NF_UNIT + Tb[k];
    }
    unsigned* mylist = s_list + wave * NF_LIST;
    unsigned long long* mykey = s_key + wave * NF_QW;
    int cnt = 0;
    for (int c0 = 0; c0 < M; c0 += NF_CH) {
      __syncthreads();            // (every wave is done with the previous chunk)
      // ---- the chunk: exact coordinates + images: row j of a tile, k half 0 = -2 (xh, xh, xl, yh, yh, yl, zh, zh),
      // k half 1 = (-2 zl, P0, P1, P2, 0, 0, 0, 0)
      {
        float x[CPT], y[CPT], z[CPT];
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
          if (c0 == 0) {       // (workgroup-uniform: in registers since the pass over the cloud)
            x[u] = kx[u];
            y[u] = ky[u];
            z[u] = kz[u];
          } else {
            const int j = c0 + u * NF_T + tid, jj = j < M ? j : M - 1;
            x[u] = P[jj];
            y[u] = P[M + jj];
            z[u] = P[2 * M + jj];
          }
        }
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
          const int jl = u * NF_T + tid;
          s_px[jl] = x[u];
          s_py[jl] = y[u];
          s_pz[jl] = z[u];
          half8 v0, v1;
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            v0[e] = (_Float16)0.f;
            v1[e] = (_Float16)0.f;
          }
          if (c0 + jl < M) {
            const float ux = (x[u] - cx) * us, uy = (y[u] - cy) * us, uz = (z[u] - cz) * us;
            _Float16 xh, xl, yh, yl, zh, zl;
            nf_split(ux, xh, xl);
            nf_split(uy, yh, yl);
            nf_split(uz, zh, zl);
            const _Float16 m2 = (_Float16)-2.f;
            v0[0] = m2 * xh; v0[1] = m2 * xh; v0[2] = m2 * xl;
            v0[3] = m2 * yh; v0[4] = m2 * yh; v0[5] = m2 * yl;
            v0[6] = m2 * zh; v0[7] = m2 * zh;
            const float Pn = ux * ux + uy * uy + uz * uz;
            const _Float16 p0 = (_Float16)Pn;
            const float r1 = Pn - (float)p0;
            const _Float16 p1 = (_Float16)r1;
            v1[0] = m2 * zl; v1[1] = p0; v1[2] = p1; v1[3] = (_Float16)(r1 - (float)p1);
          } else {
            v1[1] = (_Float16)60000.f;   // rows beyond the cloud: never under a threshold
          }
          s_img[(jl >> 5) * 64 + (jl & 31)] = v0;
          s_img[(jl >> 5) * 64 + 32 + (jl & 31)] = v1;
        }
      }
      __syncthreads();
      const int left = M - c0;
      const int ntile = left >= NF_CH ? NF_CH / 32 : (left + 31) >> 5;
      // the exact evaluation of the collected (query, half tile) items: a lane per item, its 16 candidates from LDS
      auto flush = [&]() {
        for (int i0 = 0; i0 < cnt; i0 += 64) {
          const int i = i0 + lane;
          if (i < cnt) {
            const unsigned item = mylist[i];
            const int ql = (int)(item & 127u), h = (int)((item >> 7) & 1u), t = (int)(item >> 8);
            const float x = s_qx[wave * NF_QW + ql], y = s_qy[wave * NF_QW + ql], z = s_qz[wave * NF_QW + ql];
            float best = NF_INF;
            int bi = 0x7fffffff;
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
              float4 X[2], Y[2], Z[2];
#pragma unroll
              for (int g = 0; g < 2; ++g) {
                const int jl = t * 32 + 16 * g2 + 8 * g + 4 * h;
                X[g] = *reinterpret_cast<const float4*>(s_px + jl);
                Y[g] = *reinterpret_cast<const float4*>(s_py + jl);
                Z[g] = *reinterpret_cast<const float4*>(s_pz + jl);
              }
#pragma unroll
              for (int g = 0; g < 2; ++g) {
                const float xs[4] = {X[g].x, X[g].y, X[g].z, X[g].w}, ys[4] = {Y[g].x, Y[g].y, Y[g].z, Y[g].w};
                const float zs[4] = {Z[g].x, Z[g].y, Z[g].z, Z[g].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  const int j = c0 + t * 32 + 16 * g2 + 8 * g + 4 * h + e;
                  const float d = geoa3_sqdist(x, y, z, xs[e], ys[e], zs[e]);
                  const bool take = (j < M) & ((d < best) | ((d == best) & (j < bi)));
                  best = take ? d : best;
                  bi = take ? j : bi;
                }
              }
            }
            if (bi != 0x7fffffff) atomicMin(&mykey[ql], ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bi);
          }
        }
        cnt = 0;
      };
#ifndef GEOA3_NF_STOP
#define GEOA3_NF_STOP 0      // (tools/nn1_filter_check.py --time: 1 = chunk builds only, 2 = no hit recording, 3 = no exact evaluation)
#endif
      half8 a = s_img[lane];
      for (int t = 0; t < (GEOA3_NF_STOP == 1 ? 0 : ntile); ++t) {
        const half8 an = s_img[(t + 1 < ntile ? t + 1 : t) * 64 + lane];      // the next tile's operand while this one multiplies
        nf_f32x16 acc[NF_B];
#pragma unroll
        for (int k = 0; k < NF_B; ++k) {
          nf_f32x16 z;
#pragma unroll
          for (int e = 0; e < 16; ++e) z[e] = 0.f;
          acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, Bq[k], z, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);      // (all four products issued before the first minimum: four accumulator sets)
        unsigned long long mask_any = 0ull;
#pragma unroll
        for (int k = 0; k < NF_B; ++k) {
          const bool hit = nf_min16(acc[k]) <= T[k];
          const unsigned long long mask = __ballot(hit);
          mask_any |= mask;
          if (mask && GEOA3_NF_STOP != 2) {   // (wave-uniform)
            const int pos = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if (hit) mylist[pos] = ((unsigned)t << 8) | ((unsigned)kh << 7) | (unsigned)(32 * k + n);
            cnt += __builtin_popcountll(mask);
          }
        }
        if (GEOA3_NF_STOP == 2) cnt += (int)(mask_any & 1ull);
        if (cnt > NF_LIST - 4 * 64) {
          if (GEOA3_NF_STOP >= 2) cnt = 0; else flush();
          // the thresholds follow the best distances found so far (a seed far from the answer -- no prior, a junk prior --
          // admits most of the cloud; after the first evaluations only what can still win or tie is admitted)
#ifndef GEOA3_NF_NO_TIGHTEN
#pragma unroll
          for (int k = 0; k < NF_B; ++k)
            T[k] = fminf(T[k], __uint_as_float((unsigned)(mykey[32 * k + n] >> 32)) * inv_s2 * NF_UNIT + Tb[k]);
#endif
        }
        a = an;
      }
      if (GEOA3_NF_STOP >= 2) cnt = 0; else flush();
    }
  } else {
    // ---- not finite / absurd extents: every candidate against every query in index order, strict comparison (the first
    // of equal distances wins; NaN distances never win), as nn1_pair_kernel
    __syncthreads();
    if (kh == 0) {
      unsigned best[NF_B];      // distance bits: for sums of squares the unsigned order is the float order, NaN above +inf
      int bi[NF_B];
#pragma unroll
      for (int k = 0; k < NF_B; ++k) {
        best[k] = 0x7f800000u;
        bi[k] = 0;
      }
      for (int j = 0; j < M; ++j) {
        const float px = P[j], py = P[M + j], pz = P[2 * M + j];
#pragma unroll
        for (int k = 0; k < NF_B; ++k) {
          const unsigned d = geoa3_opaque_bits(geoa3_sqdist(qx[k], qy[k], qz[k], px, py, pz)) & 0x7fffffffu;
          const bool lt = d < best[k];
          best[k] = lt ? d : best[k];
          bi[k] = lt ? j : bi[k];
        }
      }
#pragma unroll
      for (int k = 0; k < NF_B; ++k)
        if (qt[k] < Nq) s_key[wave * NF_QW + 32 * k + n] = ((unsigned long long)best[k] << 32) | (unsigned)bi[k];
    }
  }
  if (kh == 0) {
#pragma unroll
    for (int k = 0; k < NF_B; ++k)
      if (qt[k] < Nq) {
        const unsigned long long key = s_key[wave * NF_QW + 32 * k + n];
        dout[qt[k]] = __uint_as_float((unsigned)(key >> 32));
        iout[qt[k]] = (int)(unsigned)key;
      }
  }
}


}  // namespace nf
