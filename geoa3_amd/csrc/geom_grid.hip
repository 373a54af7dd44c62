// Exact nearest-neighbour search through a uniform grid (gfx950).
//
// The all-pairs kernels of geom_nn.hip spend O(N^2) distance evaluations per instance although the clouds are
// surfaces: the nearest neighbour of a point lies a few percent of the cloud's diameter away.  Here one workgroup
// per (instance, direction) bins the searched cloud into a 16^3 grid over its bounding box -- counting sort in
// LDS, rebuilt every call (~2 us).  Every query takes the distance to one real point as its radius (the point with
// its own index: adv_i = ori_i + offset_i in the attack loop) and scans only the cells its ball touches.
// Candidates are compared lexicographically on (distance, index) with the same un-fused distance as everywhere
// else, and the cell range carries a margin far above float rounding, so the result is BIT-IDENTICAL to the
// all-pairs search (tests/test_gpu_geometry.py) for any input: a query far from every point simply scans more
// cells (in the limit: all of them, i.e. the all-pairs work).
#include "geom_filter.h"
#include "profile.h"

namespace {

constexpr int GT = 1024;              // threads per workgroup of the K = 1 kernel
constexpr int GG = 16;                // cells per axis
constexpr int GC = GG * GG * GG;      // 4096 cells
constexpr int GW = GT / GEOA3_WAVE;   // 16 waves
constexpr float G_INF = __builtin_inff();

struct GridGeom {
  float ox, oy, oz, h, inv_h;
};

__device__ __forceinline__ int grid_coord(float f) {
  const int c = (int)floorf(f);
  return c < 0 ? 0 : (c > GG - 1 ? GG - 1 : c);
}

// Bins the M points of P (planar, stride M) into the grid.  On return (after the trailing barrier) s_start[c] ..
// s_start[c+1] delimit the points of cell c = (cx*GG + cy)*GG + cz in s_p4 (x, y, z, index bits).
template <int T, int PPT>
__device__ __forceinline__ GridGeom grid_build(const float* __restrict__ P, int M, int* s_start, int* s_fill,
                                               float4* s_p4, float* s_red) {
  constexpr int GT = T, GW = T / GEOA3_WAVE, CPT = GC / T;   // CPT: cell counters scanned per thread
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float px[PPT], py[PPT], pz[PPT];
  float lox = G_INF, loy = G_INF, loz = G_INF, hix = -G_INF, hiy = -G_INF, hiz = -G_INF;
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int i = tid + p * GT;
    if (i < M) {
      px[p] = P[i];
      py[p] = P[M + i];
      pz[p] = P[2 * M + i];
      lox = fminf(lox, px[p]); hix = fmaxf(hix, px[p]);
      loy = fminf(loy, py[p]); hiy = fmaxf(hiy, py[p]);
      loz = fminf(loz, pz[p]); hiz = fmaxf(hiz, pz[p]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lox = fminf(lox, __shfl_xor(lox, o, 64)); hix = fmaxf(hix, __shfl_xor(hix, o, 64));
    loy = fminf(loy, __shfl_xor(loy, o, 64)); hiy = fmaxf(hiy, __shfl_xor(hiy, o, 64));
    loz = fminf(loz, __shfl_xor(loz, o, 64)); hiz = fmaxf(hiz, __shfl_xor(hiz, o, 64));
  }
  if (lane == 0) {
    float* r = s_red + wave * 6;
    r[0] = lox; r[1] = loy; r[2] = loz; r[3] = hix; r[4] = hiy; r[5] = hiz;
  }
  for (int i = tid; i < GC; i += GT) s_fill[i] = 0;
  __syncthreads();
#pragma unroll
  for (int w = 0; w < GW; ++w) {
    const float* r = s_red + w * 6;
    lox = fminf(lox, r[0]); loy = fminf(loy, r[1]); loz = fminf(loz, r[2]);
    hix = fmaxf(hix, r[3]); hiy = fmaxf(hiy, r[4]); hiz = fmaxf(hiz, r[5]);
  }
  GridGeom g;
  g.ox = lox; g.oy = loy; g.oz = loz;
  const float ext = fmaxf(fmaxf(hix - lox, hiy - loy), hiz - loz);
  g.h = ext * (1.0f / GG) * 1.00001f;      // cubic cells; the far faces lie strictly inside the last cell
  if (!(g.h > 1e-30f)) g.h = 1.0f;         // all points coincide
  g.inv_h = 1.0f / g.h;

  int cell[PPT];
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int i = tid + p * GT;
    if (i < M) {
      cell[p] = (grid_coord((px[p] - g.ox) * g.inv_h) * GG + grid_coord((py[p] - g.oy) * g.inv_h)) * GG +
                grid_coord((pz[p] - g.oz) * g.inv_h);
      atomicAdd(&s_fill[cell[p]], 1);
    }
  }
  __syncthreads();
  // exclusive scan of the 4096 counters: CPT per thread, shuffle scan per wave, wave totals through LDS
  int cnt[CPT];
  int local = 0;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    cnt[i] = s_fill[CPT * tid + i];
    local += cnt[i];
  }
  int incl = local;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o, 64);
    if (lane >= o) incl += v;
  }
  int* s_wsum = reinterpret_cast<int*>(s_red) + GW * 6;
  if (lane == 63) s_wsum[wave] = incl;
  __syncthreads();
  int base = 0;
#pragma unroll
  for (int w = 0; w < GW; ++w) base += (w < wave) ? s_wsum[w] : 0;
  int run = base + incl - local;
#pragma unroll
  for (int i = 0; i < CPT; ++i) {
    s_start[CPT * tid + i] = run;
    run += cnt[i];
    s_fill[CPT * tid + i] = 0;
  }
  if (tid == GT - 1) s_start[GC] = run;
  __syncthreads();
#pragma unroll
  for (int p = 0; p < PPT; ++p) {
    const int i = tid + p * GT;
    if (i < M) {
      const int pos = s_start[cell[p]] + atomicAdd(&s_fill[cell[p]], 1);
      s_p4[pos] = make_float4(px[p], py[p], pz[p], __int_as_float(i));   // one 16-byte record per point: (x, y, z, index bits)
    }
  }
  __syncthreads();
  return g;
}

// Visits, as runs of consecutive sorted points, every cell that can hold a point within `rad` (world units) of the
// query at grid coordinates (fx,fy,fz): the cells of one (x,y) column are adjacent in the cell order, and the z-range
// of a column is the chord of the ball over that column.  A margin of 1e-4 cells covers the float rounding of the
// cell assignment, so no point with a computed distance <= rad^2 is missed.
template <typename Visit>
__device__ __forceinline__ void grid_ball(const int* __restrict__ s_start, float fx, float fy, float fz, float rad,
                                          float inv_h, Visit visit) {
  const float rho = rad * inv_h * 1.00001f + 1e-4f;
  const int x0 = grid_coord(fx - rho), x1 = grid_coord(fx + rho);
  const int y0 = grid_coord(fy - rho), y1 = grid_coord(fy + rho);
  const float rho2 = rho * rho;
  for (int x = x0; x <= x1; ++x) {
    const float ex = fmaxf(fmaxf((float)x - fx, fx - (float)(x + 1)), 0.f);   // distance to the slab [x, x+1]
    for (int y = y0; y <= y1; ++y) {
      const float ey = fmaxf(fmaxf((float)y - fy, fy - (float)(y + 1)), 0.f);
      const float rem = rho2 - ex * ex - ey * ey;
      if (rem < 0.f) continue;
      const float zr = __builtin_amdgcn_sqrtf(rem) + 1e-4f;   // v_sqrt_f32 (1 ulp): far inside the 1e-4-cell margin
      const int col = (x * GG + y) * GG;
      visit(s_start[col + grid_coord(fz - zr)], s_start[col + grid_coord(fz + zr) + 1]);
    }
  }
}

// NaN / inf among three coordinates, by their bits (this file is compiled with -fno-honor-nans for geom_filter.h)
__device__ __forceinline__ bool grid_nonfinite3(float x, float y, float z) {
  return ((int)geoa3_nonfinite(x) | (int)geoa3_nonfinite(y) | (int)geoa3_nonfinite(z)) != 0;
}

// One candidate against a query's running (distance, index) minimum -- lexicographic, the un-fused distance of every search.
__device__ __forceinline__ void nn1_take(float& best, int& bi, float qx, float qy, float qz, const float4 c) {
  const float d = geoa3_sqdist(qx, qy, qz, c.x, c.y, c.z);
  const int i = __float_as_int(c.w);
  const bool take = (d < best) | ((d == best) & (i < bi));   // (no short circuit: the scans stay branch-free)
  best = take ? d : best;
  bi = take ? i : bi;
}
// The candidates [a0, a1) of the sorted cloud, four 16-byte LDS reads in flight.
__device__ __forceinline__ void nn1_scan(const float4* __restrict__ p4, int a0, int a1, float qx, float qy, float qz, float& best,
                                         int& bi) {
  int j = a0;
  for (; j + 4 <= a1; j += 4) {
    const float4 c0 = p4[j], c1 = p4[j + 1], c2 = p4[j + 2], c3 = p4[j + 3];
    nn1_take(best, bi, qx, qy, qz, c0);
    nn1_take(best, bi, qx, qy, qz, c1);
    nn1_take(best, bi, qx, qy, qz, c2);
    nn1_take(best, bi, qx, qy, qz, c3);
  }
  for (; j < a1; ++j) nn1_take(best, bi, qx, qy, qz, p4[j]);
}

constexpr int NN1_WIDE = 16;   // a query whose ball touches more grid columns is walked by the whole wavefront
#ifndef GEOA3_GRID_BRUTE
#define GEOA3_GRID_BRUTE 0.12f
#endif
#ifndef GEOA3_NN1_DIRECT
#define GEOA3_NN1_DIRECT 8
#endif
#ifndef GEOA3_NN1_CHUNK
#define GEOA3_NN1_CHUNK 16
#endif
constexpr float GRID_BRUTE = GEOA3_GRID_BRUTE;   // fraction of the searched cloud inside the queries' boxes beyond which a batch is searched by brute force (in-kernel sweep)
#ifndef GEOA3_GRID_FILTER
#define GEOA3_GRID_FILTER 0.03f
#endif
constexpr float GRID_FILTER = GEOA3_GRID_FILTER; // ... the same when the workgroup then runs the matrix-core filter search (geom_filter.h), the shipped form
constexpr int NN1_DIRECT = GEOA3_NN1_DIRECT;     // candidates of a column run scanned by the (query, column) pair's own lane
#ifndef GEOA3_NN1_LONG
#define GEOA3_NN1_LONG 24
#endif
constexpr int NN1_LONG = GEOA3_NN1_LONG;         // a trip with a run longer than this is dealt a second time (below)
constexpr int NN1_CHUNK = GEOA3_NN1_CHUNK;       // the rest of a run is dealt over the wavefront's lanes in chunks of this size

// (a call, not inlined: the search's registers are allocated apart from the walk's -- inlined, grid_nn1_kernel<4> spilled 41)
__device__ __attribute__((noinline)) void grid_filter_search(const float* P, const float* Q, int M, int Nq, int q0,
                                                             const int32_t* prior, float* dout, int32_t* iout,
                                                             unsigned char* smem) {
  nf::search<GT>(P, Q, M, Nq, q0, prior, dout, iout, smem);
}

// ---- The filter search on the grid's SORTED cloud (clouds of 1025..4096 points, the heavy workgroups): the records stay
// where the counting sort left them (cell order = x-major), so the candidates a block of 32 queries can need -- those whose
// cell x lies within the block's x-range +- its largest seed radius -- are ONE contiguous range of the sorted order, and a
// wave multiplies a tile only for the blocks whose range holds it (same margin as grid_ball: a candidate outside is
// farther than the seed along x alone).  The blocks of a wave are spread over the sorted order (block ids wave + 16 k), so
// every wave finds about the same share of every chunk.  Candidates' images are built from LDS (no second trip to memory),
// exact evaluation reads the same records; ties by the ORIGINAL index, as everywhere.
constexpr int GS_PASS = 2 * GT;                       // queries per pass: four blocks of 32 per wave
constexpr int GS_LIST = 224;                          // items per wave between exact evaluations
static_assert(GT == nf::NF_CH, "grid_filter_sorted builds one candidate image per thread and chunk");
constexpr size_t GS_OFF_P4 = ((size_t)(GC + 4) + GC + GW * 8) * 4;
constexpr size_t GS_TAIL = (size_t)GS_PASS * 8 + 4 * (size_t)GS_PASS * 4 + (size_t)GW * GS_LIST * 4 + (GS_PASS / 32) * 2 * 4 + 32 * 4;
struct GsTail {
  unsigned long long* key;   // [GS_PASS] (distance bits : original index) of the pass's queries, by position
  float *qx, *qy, *qz;       // [GS_PASS]
  int* qidx;                 // [GS_PASS] query index (-1: none)
  unsigned* list;            // [GW][GS_LIST]
  int* blk;                  // [GS_PASS / 32][2] candidate range of a block of 32 positions
  int* slab;                 // [17] first sorted position of every cell x
};
__device__ __forceinline__ GsTail gs_tail(unsigned char* smem, int M) {
  GsTail t;
  t.key = reinterpret_cast<unsigned long long*>(smem + GS_OFF_P4 + (size_t)M * 16);
  t.qx = reinterpret_cast<float*>(t.key + GS_PASS);
  t.qy = t.qx + GS_PASS;
  t.qz = t.qy + GS_PASS;
  t.qidx = reinterpret_cast<int*>(t.qz + GS_PASS);
  t.list = reinterpret_cast<unsigned*>(t.qidx + GS_PASS);
  t.blk = reinterpret_cast<int*>(t.list + GW * GS_LIST);
  t.slab = t.blk + (GS_PASS / 32) * 2;
  return t;
}

__device__ __attribute__((noinline)) void grid_filter_sorted(unsigned char* smem, int M, int nq, float cx, float cy, float cz,
                                                             float inv_s, float* dout, int32_t* iout) {
  using namespace nf;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, n = lane & 31, kh = lane >> 5;
  half8* s_img = reinterpret_cast<half8*>(smem);                                   // [32 tiles][2 k halves][32 rows]
  const float4* s_p4 = reinterpret_cast<const float4*>(smem + GS_OFF_P4);          // the sorted cloud
  const GsTail T0 = gs_tail(smem, M);
  const float us = inv_s * 128.f, inv_s2 = inv_s * inv_s;
  // ---- this lane's queries: block k of the wave = block wave + 16 k of the pass (positions 32 block + lane % 32)
  half8 Bq[NF_B];
  float T[NF_B], Tb[NF_B];      // threshold = (distance bound / s^2) * UNIT + Tb
  int lo[NF_B], hi[NF_B], pos[NF_B], qid[NF_B];
  int wlo = 0x7fffffff, whi = 0;
#pragma unroll
  for (int k = 0; k < NF_B; ++k) {
    const int blk = wave + GW * k;
    pos[k] = blk * 32 + n;
    qid[k] = pos[k] < nq ? T0.qidx[pos[k]] : -1;
    const bool valid = qid[k] >= 0;
    const float qx = T0.qx[pos[k]], qy = T0.qy[pos[k]], qz = T0.qz[pos[k]];
    const float seed_d = __uint_as_float((unsigned)(T0.key[pos[k]] >> 32));
    const float ux = (qx - cx) * us, uy = (qy - cy) * us, uz = (qz - cz) * us;
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
    Tb[k] = valid ? NF_EPS * NF_UNIT - Qn : -NF_INF;
    T[k] = seed_d * inv_s2 * NF_UNIT + Tb[k];
    lo[k] = __builtin_amdgcn_readfirstlane(T0.blk[2 * blk]);
    hi[k] = __builtin_amdgcn_readfirstlane(T0.blk[2 * blk + 1]);
    wlo = lo[k] < hi[k] && lo[k] < wlo ? lo[k] : wlo;
    whi = lo[k] < hi[k] && hi[k] > whi ? hi[k] : whi;
  }
  unsigned* mylist = T0.list + wave * GS_LIST;
  int cnt = 0;
  // the exact evaluation of the collected (query, half tile) items: a lane per item, its 16 candidates' records from LDS
  auto flush = [&]() {
    for (int i0 = 0; i0 < cnt; i0 += 64) {
      const int i = i0 + lane;
      if (i < cnt) {
        const unsigned item = mylist[i];
        const int ql = (int)(item & 127u), h = (int)((item >> 7) & 1u), t = (int)(item >> 8);
        const int qp = (wave + GW * (ql >> 5)) * 32 + (ql & 31);
        const float x = T0.qx[qp], y = T0.qy[qp], z = T0.qz[qp];
        float best = NF_INF;
        int bi = 0x7fffffff;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
          float4 c[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int j = t * 32 + 16 * g2 + 8 * (e >> 2) + 4 * h + (e & 3);
            c[e] = s_p4[j < M ? j : M - 1];
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int j = t * 32 + 16 * g2 + 8 * (e >> 2) + 4 * h + (e & 3);
            const float d = geoa3_sqdist(x, y, z, c[e].x, c[e].y, c[e].z);
            const int ci = __float_as_int(c[e].w);
            const bool take = (j < M) & ((d < best) | ((d == best) & (ci < bi)));
            best = take ? d : best;
            bi = take ? ci : bi;
          }
        }
        if (bi != 0x7fffffff) atomicMin(&T0.key[qp], ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bi);
      }
    }
    cnt = 0;
  };
  for (int c0 = 0; c0 < M; c0 += NF_CH) {
    __syncthreads();            // (every wave is done with the previous chunk's images)
    {
      const int j = c0 + tid;   // GT == NF_CH: one candidate per thread
      half8 v0, v1;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v0[e] = (_Float16)0.f;
        v1[e] = (_Float16)0.f;
      }
      if (j < M) {
        const float4 r = s_p4[j];
        const float ux = (r.x - cx) * us, uy = (r.y - cy) * us, uz = (r.z - cz) * us;
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
      s_img[(tid >> 5) * 64 + (tid & 31)] = v0;
      s_img[(tid >> 5) * 64 + 32 + (tid & 31)] = v1;
    }
    __syncthreads();
    const int left = M - c0;
    const int ntile = left >= NF_CH ? NF_CH / 32 : (left + 31) >> 5;
    int t0 = (wlo - c0) >> 5, t1 = (whi - c0 + 31) >> 5;      // the wave's tiles of this chunk
    t0 = t0 < 0 ? 0 : t0;
    t1 = t1 > ntile ? ntile : t1;
    for (int t = t0; t < t1; ++t) {
      const half8 a = s_img[t * 64 + lane];
      const int j0 = c0 + t * 32;
#pragma unroll
      for (int k = 0; k < NF_B; ++k) {
        if (j0 + 31 >= lo[k] && j0 < hi[k]) {     // (wave-uniform)
          nf_f32x16 z;
#pragma unroll
          for (int e = 0; e < 16; ++e) z[e] = 0.f;
          const nf_f32x16 acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, Bq[k], z, 0, 0, 0);
          const bool hit = nf_min16(acc) <= T[k];
          const unsigned long long mask = __ballot(hit);
          if (mask) {
            const int p = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
            if (hit) mylist[p] = ((unsigned)(j0 >> 5) << 8) | ((unsigned)kh << 7) | (unsigned)(32 * k + n);
            cnt += __builtin_popcountll(mask);
            if (cnt > GS_LIST - 64) {
              flush();
              // (the thresholds follow the best distances found so far: geom_filter.h)
#ifndef GEOA3_NF_NO_TIGHTEN
#pragma unroll
              for (int kk = 0; kk < NF_B; ++kk)
                T[kk] = fminf(T[kk], __uint_as_float((unsigned)(T0.key[pos[kk]] >> 32)) * inv_s2 * NF_UNIT + Tb[kk]);
#endif
            }
          }
        }
      }
    }
  }
  flush();
  if (kh == 0) {
#pragma unroll
    for (int k = 0; k < NF_B; ++k)
      if (qid[k] >= 0) {
        const unsigned long long key = T0.key[pos[k]];
        dout[qid[k]] = __uint_as_float((unsigned)(key >> 32));
        iout[qid[k]] = (int)(unsigned)key;
      }
  }
}

template <int PPT, int MODE = 0>   // MODE (tools/ub/nn1_ub.hip): 1 = build only, 2 = seeds only (no ball walk)
// (two workgroups per CU while the cloud's LDS allows it -- up to 1024 points: at most 64 registers there)
__global__ __launch_bounds__(GT, PPT == 1 ? 8 : 4) void grid_nn1_kernel(const float* __restrict__ A, const float* __restrict__ R, int Na,
                                                      int Nr, const int32_t* prior_ar, const int32_t* prior_ra,
                                                      float* d_ar, int32_t* i_ar, float* d_ra, int32_t* i_ra, int wide_thr,
                                                      float brute_frac, int filter) {
  // brute_frac: the fraction of the searched cloud inside the queries' boxes beyond which the workgroup's queries are
  // searched all-pairs; filter != 0 (clouds of more than 1024 points): through the matrix core (geom_filter.h), in place
  // prior_* (optional, may alias i_*): a searched-cloud index per query -- last iteration's answer -- used as the seed
  extern __shared__ __attribute__((aligned(16))) unsigned char g_smem[];
  const int b = blockIdx.x;
  const bool swap = blockIdx.y != 0;
  const int32_t* prior = swap ? prior_ra : prior_ar;
  const int Nq = swap ? Nr : Na, M = swap ? Na : Nr;
  const float* Q = (swap ? R : A) + (size_t)b * 3 * Nq;
  const float* P = (swap ? A : R) + (size_t)b * 3 * M;
  float* dout = (swap ? d_ra : d_ar) + (size_t)b * Nq;
  int32_t* iout = (swap ? i_ra : i_ar) + (size_t)b * Nq;
  int* s_start = reinterpret_cast<int*>(g_smem);            // [GC + 1]
  int* s_fill = s_start + GC + 4;                            // [GC]
  float* s_red = reinterpret_cast<float*>(s_fill + GC);      // [GW*6 + GW]
  float4* s_p4 = reinterpret_cast<float4*>(s_red + GW * 8);   // [M] (x, y, z, index bits) in cell order
  const GridGeom g = grid_build<GT, PPT>(P, M, s_start, s_fill, s_p4, s_red);

  if (MODE == 1) return;
  // Points per cell as a 3-D summed-area table (in s_fill, free after the build): the number of points in the cells of a
  // query's box is eight reads.  A batch of queries whose boxes hold more than GRID_BRUTE of the searched cloud
  // on average is searched by brute force instead (below): clouds with a dense cluster, thin rods, or iterates far from the
  // surface put hundreds of points into the few cells a ball touches, and a candidate of the grid walk (a lane's serial
  // scan of its own column run, four scattered LDS reads each) costs ~10 candidates of the brute-force loop (broadcast
  // reads, no divergence).  Measured at N = 1024, offsets 0.3 (tools/nn1_cad_probe.py): clusters 1810 us and rods 765 us
  // per launch through the grid against 167 us all-pairs.  Same candidates order-independently: same bits either way.
  int* s_sat = s_fill;                                      // [16][16][16] inclusive prefix sums of the cell counts
  if (MODE == 4) {
    for (int c = threadIdx.x; c < GC; c += GT) s_sat[c] = s_start[c + 1] - s_start[c];
    __syncthreads();
    if (threadIdx.x < GG * GG) {                            // along z
      int run = 0;
      for (int z = 0; z < GG; ++z) {
        run += s_sat[threadIdx.x * GG + z];
        s_sat[threadIdx.x * GG + z] = run;
      }
    }
    __syncthreads();
    if (threadIdx.x < GG * GG) {                            // along y: lines (x, z)
      const int x = threadIdx.x / GG, z = threadIdx.x % GG;
      int run = 0;
      for (int y = 0; y < GG; ++y) {
        run += s_sat[(x * GG + y) * GG + z];
        s_sat[(x * GG + y) * GG + z] = run;
      }
    }
    __syncthreads();
    if (threadIdx.x < GG * GG) {                            // along x: lines (y, z)
      int run = 0;
      for (int x = 0; x < GG; ++x) {
        run += s_sat[x * GG * GG + threadIdx.x];
        s_sat[x * GG * GG + threadIdx.x] = run;
      }
    }
    __syncthreads();
  }
  auto sat = [&](int x, int y, int z) { return (x < 0 || y < 0 || z < 0) ? 0 : s_sat[(x * GG + y) * GG + z]; };
  auto in_box = [&](float fx, float fy, float fz, float rho) {   // points in the cells of the ball's bounding box
    const int x0 = grid_coord(fx - rho), x1 = grid_coord(fx + rho), y0 = grid_coord(fy - rho), y1 = grid_coord(fy + rho);
    const int z0 = grid_coord(fz - rho), z1 = grid_coord(fz + rho);
    return sat(x1, y1, z1) - sat(x0 - 1, y1, z1) - sat(x1, y0 - 1, z1) - sat(x1, y1, z0 - 1) + sat(x0 - 1, y0 - 1, z1) +
           sat(x0 - 1, y1, z0 - 1) + sat(x1, y0 - 1, z0 - 1) - sat(x0 - 1, y0 - 1, z0 - 1);
  };
  if (MODE == 4 && PPT > 1) {
    // Several batches of queries per thread: ONE decision for all of them, and the brute-force sweep keeps a thread's PPT
    // queries in registers -- every candidate is read from LDS once for all of them
    float qx[PPT], qy[PPT], qz[PPT], best[PPT];
    int bi[PPT], qi[PPT];
    float est = 0.f;
#pragma unroll
    for (int p = 0; p < PPT; ++p) {
      const int t = p * GT + threadIdx.x;
      const bool valid = t < Nq;
      qi[p] = valid ? (Nq == M ? __float_as_int(s_p4[t].w) : t) : -1;
      const int q = valid ? qi[p] : 0;
      qx[p] = Q[q];
      qy[p] = Q[Nq + q];
      qz[p] = Q[2 * Nq + q];
      int si = prior ? prior[(size_t)b * Nq + q] : q;
      si = si < 0 ? 0 : (si >= M ? M - 1 : si);
      const float sx = P[si], sy = P[M + si], sz = P[2 * M + si];
      best[p] = geoa3_sqdist(qx[p], qy[p], qz[p], sx, sy, sz);
      bi[p] = si;
      if ((int)grid_nonfinite3(qx[p], qy[p], qz[p]) | (int)grid_nonfinite3(sx, sy, sz)) {   // no radius: (inf, 0), where the all-pairs kernel starts
        best[p] = G_INF;
        bi[p] = 0;
      }
      const float rho = sqrtf(best[p]) * g.inv_h * 1.00001f + 1e-4f;
      if (valid) est += (float)in_box((qx[p] - g.ox) * g.inv_h, (qy[p] - g.oy) * g.inv_h, (qz[p] - g.oz) * g.inv_h, rho);
    }
    est = wave_sum(est);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = est;
    __syncthreads();
    float all = 0.f;
#pragma unroll
    for (int w = 0; w < GW; ++w) all += s_red[w];
    __syncthreads();
    if (all > brute_frac * (float)Nq * (float)M) {
      if (filter) {
        // ---- through the matrix core (geom_filter.h).  Scale: the cloud's cube (every point within 8 h of its centre) and
        // this workgroup's queries
        const float ccx = g.ox + 8.f * g.h, ccy = g.oy + 8.f * g.h, ccz = g.oz + 8.f * g.h;
        float m = 8.f * g.h * 1.0001f;
        // (branch-free on purpose, here and below: with `if (qi[p] >= 0)` around this line hipcc 7.2 placed a register spill of
        // qi[1] INSIDE the divergent region -- lanes without a fourth query reloaded garbage: NOTEBOOK 10)
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
          const float dev = fmaxf(fmaxf(nf::nf_abs_or_inf(qx[p] - ccx), nf::nf_abs_or_inf(qy[p] - ccy)), nf::nf_abs_or_inf(qz[p] - ccz));
          m = fmaxf(m, qi[p] >= 0 ? dev : 0.f);
        }
        m = geoa3_nonfinite(g.h) ? G_INF : m;
        m = wave_max(m);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
        __syncthreads();
#pragma unroll
        for (int w = 0; w < GW; ++w) m = fmaxf(m, s_red[w]);
        const unsigned Em = (geoa3_opaque_bits(m) >> 23) & 0xffu;
        if (!(Em <= 127u + 60u && Em >= 127u - 60u)) {
          // not finite / absurd extents: the search on the unsorted cloud (its own exact sweep).  prior may alias iout: a
          // pass reads its own slice's seeds before it writes that slice's results
          for (int q0 = 0; q0 < Nq; q0 += nf::Cfg<GT>::SLICE) {
            __syncthreads();
            grid_filter_search(P, Q, M, Nq, q0, prior ? prior + (size_t)b * Nq : nullptr, dout, iout, g_smem);
          }
          return;
        }
        const float inv_s = 1.0f / (m * 1.000001f);
        const GsTail tl = gs_tail(g_smem, M);
        if (threadIdx.x <= GG) tl.slab[threadIdx.x] = s_start[threadIdx.x * GG * GG];
#pragma unroll
        for (int pass = 0; pass < (PPT + 1) / 2; ++pass) {
          if (pass * GS_PASS >= Nq) break;        // (workgroup-uniform)
          __syncthreads();                        // (the previous pass is done with the tail; the slab table is written)
#pragma unroll
          for (int pp = 0; pp < 2; ++pp) {
            const int p = 2 * pass + pp;
            if (p < PPT) {
              const int ps = pp * GT + threadIdx.x;
              const bool valid = qi[p] >= 0;
              tl.qx[ps] = qx[p];
              tl.qy[ps] = qy[p];
              tl.qz[ps] = qz[p];
              tl.qidx[ps] = qi[p];
              tl.key[ps] = ((unsigned long long)__float_as_uint(best[p]) << 32) | (unsigned)bi[p];
              // the block's x-range and its largest seed radius -> its candidates' range of the sorted order
              float xmin = valid ? qx[p] : G_INF, xmax = valid ? qx[p] : -G_INF, bmax = valid ? best[p] : 0.f;
#pragma unroll
              for (int o = 16; o > 0; o >>= 1) {
                xmin = fminf(xmin, __shfl_xor(xmin, o, 64));
                xmax = fmaxf(xmax, __shfl_xor(xmax, o, 64));
                bmax = fmaxf(bmax, __shfl_xor(bmax, o, 64));
              }
              {
                const bool any = xmin <= xmax;
                const float rho = sqrtf(any ? bmax : 0.f) * g.inv_h * 1.00001f + 1e-4f;
                const int l0 = tl.slab[grid_coord(((any ? xmin : g.ox) - g.ox) * g.inv_h - rho)];
                const int l1 = tl.slab[grid_coord(((any ? xmax : g.ox) - g.ox) * g.inv_h + rho) + 1];
                if ((threadIdx.x & 31) == 0) {
                  tl.blk[2 * (ps >> 5)] = any ? l0 : 0;
                  tl.blk[2 * (ps >> 5) + 1] = any ? l1 : 0;
                }
              }
            } else {
              const int ps = pp * GT + threadIdx.x;
              tl.qidx[ps] = -1;
              tl.qx[ps] = 0.f; tl.qy[ps] = 0.f; tl.qz[ps] = 0.f;
              tl.key[ps] = 0ull;
              if ((threadIdx.x & 31) == 0) {
                tl.blk[2 * (ps >> 5)] = 0;
                tl.blk[2 * (ps >> 5) + 1] = 0;
              }
            }
          }
          __syncthreads();
          const int left = Nq - pass * GS_PASS;
          grid_filter_sorted(g_smem, M, left < GS_PASS ? left : GS_PASS, ccx, ccy, ccz, inv_s, dout, iout);
        }
        return;
      }
      // every point against every query: the searched cloud is staged again in INDEX order, so that the running minimum
      // needs the strict comparison only (ascending indices: the first of equal distances wins, as in nn1_pair_kernel) --
      // 11 instead of 14 instructions per pair, and the seeds are not needed
#pragma unroll
      for (int p = 0; p < PPT; ++p) {
        const int i = p * GT + threadIdx.x;
        if (i < M) s_p4[i] = make_float4(P[i], P[M + i], P[2 * M + i], 0.f);
        best[p] = G_INF;
        bi[p] = 0;
      }
      __syncthreads();
      for (int j = 0; j < M; ++j) {
        const float4 c = s_p4[j];
#pragma unroll
        for (int p = 0; p < PPT; ++p) {
          const float d = geoa3_sqdist(qx[p], qy[p], qz[p], c.x, c.y, c.z);
          const bool lt = d < best[p];
          best[p] = lt ? d : best[p];
          bi[p] = lt ? j : bi[p];
        }
      }
#pragma unroll
      for (int p = 0; p < PPT; ++p)
        if (qi[p] >= 0) {
          dout[qi[p]] = best[p];
          iout[qi[p]] = bi[p];
        }
      return;
    }
  }
  for (int t0 = 0; t0 < Nq; t0 += GT) {     // (uniform trips: the wave-wide walk below needs every lane)
    const int t = t0 + threadIdx.x;
    const bool valid = t < Nq;
    // equal-sized clouds: walk the queries in the searched cloud's cell order (query i is a perturbation of point i
    // in the attack loop), so that the lanes of a wavefront look at the same few cells
    const int q = valid ? (Nq == M ? __float_as_int(s_p4[t].w) : t) : 0;
    const float qx = Q[q], qy = Q[Nq + q], qz = Q[2 * Nq + q];
    const float fx = (qx - g.ox) * g.inv_h, fy = (qy - g.oy) * g.inv_h, fz = (qz - g.oz) * g.inv_h;
    // seed: any real point gives a valid radius.  Last iteration's nearest neighbour when the caller has it,
    // otherwise the point with the query's own index (adv_i = ori_i + offset_i in the attack loop)
    int si = prior ? prior[(size_t)b * Nq + q] : q;
    si = si < 0 ? 0 : (si >= M ? M - 1 : si);
    const float sx = P[si], sy = P[M + si], sz = P[2 * M + si];
    float best = geoa3_sqdist(qx, qy, qz, sx, sy, sz);
    int bi = si;
    if ((int)grid_nonfinite3(qx, qy, qz) | (int)grid_nonfinite3(sx, sy, sz)) {   // no radius: (inf, 0), where the all-pairs kernel starts
      best = G_INF;
      bi = 0;
    }
    if (MODE == 3) {   // statistics: columns iterated per query, by bucket
      const float rho = sqrtf(best) * g.inv_h * 1.00001f + 1e-4f;
      const int nx = grid_coord(fx + rho) - grid_coord(fx - rho) + 1, ny = grid_coord(fy + rho) - grid_coord(fy - rho) + 1;
      const int n = nx * ny;
      atomicAdd(reinterpret_cast<unsigned long long*>(d_ar) + 2 + (n <= 4 ? 0 : (n <= 16 ? 1 : (n <= 64 ? 2 : 3))), 1ull);
    }
    // A query's ball touches ~5 grid columns on average, but a few queries (outliers: a large distance to their seed)
    // touch up to 64, and a wavefront walks as long as its widest lane: 33 column steps on average for 4.9 useful ones
    // (s_memtime + counters).  So a lane walks its own box only if it is small; the wide boxes are taken one after the
    // other by the WHOLE wave, a column per lane, and reduced with the same lexicographic (distance, index) minimum --
    // the same candidates, the same result.
    const float rad = sqrtf(best);
    const float rho = rad * g.inv_h * 1.00001f + 1e-4f;   // (grid_ball's radius in cells)
    const int bx0 = grid_coord(fx - rho), bx1 = grid_coord(fx + rho), by0 = grid_coord(fy - rho), by1 = grid_coord(fy + rho);
    const int bny = by1 - by0 + 1, bcols = (bx1 - bx0 + 1) * bny;
    bool brute = false;
    if (MODE == 4) {
      const int inbox = valid ? in_box(fx, fy, fz, rho) : 0;
      const float tot = wave_sum((float)inbox);
      float* s_est = s_red;                                 // [GW] (the build's scratch is free)
      __syncthreads();                                      // (every wave is done with the previous batch's use of it)
      if ((threadIdx.x & 63) == 0) s_est[threadIdx.x >> 6] = tot;
      __syncthreads();
      float all = 0.f;
#pragma unroll
      for (int w = 0; w < GW; ++w) all += s_est[w];
      const int nq = Nq - t0 < GT ? Nq - t0 : GT;
      brute = PPT == 1 && all > brute_frac * (float)nq * (float)M;    // (PPT > 1: decided for all batches above)
    }
    if (brute) {
      // (workgroup-uniform: the cloud once more in INDEX order, strict comparison, no seed -- as above)
      __syncthreads();
      if (threadIdx.x < M) s_p4[threadIdx.x] = make_float4(P[threadIdx.x], P[M + threadIdx.x], P[2 * M + threadIdx.x], 0.f);
      __syncthreads();
      best = G_INF;
      bi = 0;
      for (int j = 0; j < M; j += 4) {      // (M <= 1024 here; the tail reads stay inside the workgroup's LDS and are masked)
        const float4 c0 = s_p4[j], c1 = s_p4[j + 1], c2 = s_p4[j + 2], c3 = s_p4[j + 3];
        const float4 cc[4] = {c0, c1, c2, c3};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float d = geoa3_sqdist(qx, qy, qz, cc[u].x, cc[u].y, cc[u].z);
          const bool lt = (d < best) & (j + u < M);
          best = lt ? d : best;
          bi = lt ? j + u : bi;
        }
      }
    } else if (MODE == 4) {
      // BALANCED walk: the (query, column) pairs of the wavefront's 64 queries are dealt evenly over its lanes -- a query's
      // ball touches 4.9 columns on average but up to 64, and with a query per lane the wavefront walks as long as its
      // widest lane.  Prefix sums of the column counts (LDS, one row per wave) map pair t to (query lane, column); the
      // query's box comes over by lane permutes; a pair's best candidate goes into the query's 64-bit key (distance
      // bits : index -- the lexicographic (distance, index) order) with an LDS atomic minimum.  Same candidates, same
      // result.
      unsigned long long* s_key = reinterpret_cast<unsigned long long*>(s_p4 + M);   // [GT]
      int* s_pre = reinterpret_cast<int*>(s_key + GT);                                    // [GW][64]
      const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
      const int nc = valid ? bcols : 0;
      int incl = nc;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
      }
      const int total = __builtin_amdgcn_readlane(incl, 63);
      s_pre[wv * 64 + lane] = incl - nc;
      s_key[threadIdx.x] = ((unsigned long long)__float_as_uint(best) << 32) | (unsigned)bi;
      const int* pre = s_pre + wv * 64;
      int* cpre = s_pre + GW * 64 + wv * 64;           // [GW][64]: chunk prefix of the trip in flight
      for (int t0p = 0; t0p < total; t0p += 64) {      // (uniform trips: the lane permutes below need every source lane)
        const int t = t0p + lane;
        const bool act = t < total;
        int l = 0;                       // the last lane whose exclusive prefix is <= t (lanes without columns repeat the
#pragma unroll                           // prefix of the lane before them: "last" skips them)
        for (int st = 32; st > 0; st >>= 1) l += (pre[l + st] <= t) ? st : 0;
        const int c = act ? t - pre[l] : 0;
        const float wqx = __shfl(qx, l, 64), wqy = __shfl(qy, l, 64), wqz = __shfl(qz, l, 64);
        const float wfx = __shfl(fx, l, 64), wfy = __shfl(fy, l, 64), wfz = __shfl(fz, l, 64), wrho = __shfl(rho, l, 64);
        const int wx0 = __shfl(bx0, l, 64), wy0 = __shfl(by0, l, 64), wny = __shfl(bny, l, 64);
        const int xx = wx0 + c / wny, yy = wy0 + c % wny;
        const float ex = fmaxf(fmaxf((float)xx - wfx, wfx - (float)(xx + 1)), 0.f);
        const float ey = fmaxf(fmaxf((float)yy - wfy, wfy - (float)(yy + 1)), 0.f);
        const float rem = wrho * wrho - ex * ex - ey * ey;
        int js = 0, len = 0;
        if (act && rem >= 0.f) {
          const float zr = __builtin_amdgcn_sqrtf(rem) + 1e-4f;
          const int col = (xx * GG + yy) * GG;
          js = s_start[col + grid_coord(wfz - zr)];
          len = s_start[col + grid_coord(wfz + zr) + 1] - js;
        }
        // A pair's lane scans the first NN1_DIRECT candidates of its column run itself.  What a run holds beyond them is
        // cut into chunks of NN1_CHUNK candidates and the chunks of the trip's 64 runs are dealt evenly over the lanes
        // again: a trip lasts as long as its longest lane, and the runs of one trip differ by an order of magnitude (a
        // thin rod, a table leg or a dense cluster puts hundreds of points into one column of cells beside empty ones; at
        // 4096 points a run holds 30-120 points even on a uniform surface).  Same candidates, same keys, same result.
        {
          // (a trip whose longest run is short is scanned as it is: the second dealing costs ~100 instructions)
          const bool deal = __ballot(len > NN1_LONG) != 0ull;      // (wave-uniform)
          const int nd = (!deal || len < NN1_DIRECT) ? len : NN1_DIRECT;
          float cbest = G_INF;
          int cbi = 0x7fffffff;
          nn1_scan(s_p4, js, js + nd, wqx, wqy, wqz, cbest, cbi);
          if (nd > 0) atomicMin(&s_key[wv * 64 + l], ((unsigned long long)__float_as_uint(cbest) << 32) | (unsigned)cbi);
          const int rest = len - nd;
          const int nch = (rest + NN1_CHUNK - 1) / NN1_CHUNK;
          if (deal) {
            int cin = nch;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
              const int v = __shfl_up(cin, o, 64);
              if (lane >= o) cin += v;
            }
            const int ktot = __builtin_amdgcn_readlane(cin, 63);
            cpre[lane] = cin - nch;
            const int rs = js + nd;                     // the rest of this lane's run: [rs, rs + rest)
            for (int k0 = 0; k0 < ktot; k0 += 64) {     // (uniform trips again)
              const int k = k0 + lane;
              const bool act2 = k < ktot;
              int p = 0;
#pragma unroll
              for (int st = 32; st > 0; st >>= 1) p += (cpre[p + st] <= k) ? st : 0;
              const int ch = act2 ? k - cpre[p] : 0;
              const float pqx = __shfl(wqx, p, 64), pqy = __shfl(wqy, p, 64), pqz = __shfl(wqz, p, 64);
              const int prs = __shfl(rs, p, 64), prest = __shfl(rest, p, 64), pl = __shfl(l, p, 64);
              if (act2) {
                const int a0 = prs + ch * NN1_CHUNK;
                const int a1 = (ch + 1) * NN1_CHUNK < prest ? a0 + NN1_CHUNK : prs + prest;
                float kbest = G_INF;
                int kbi = 0x7fffffff;
                nn1_scan(s_p4, a0, a1, pqx, pqy, pqz, kbest, kbi);
                atomicMin(&s_key[wv * 64 + pl], ((unsigned long long)__float_as_uint(kbest) << 32) | (unsigned)kbi);
              }
            }
          }
        }
      }
      const unsigned long long k = s_key[threadIdx.x];
      best = __uint_as_float((unsigned)(k >> 32));
      bi = (int)(unsigned)k;
    }
    const bool wide = MODE == 0 && valid && bcols > wide_thr;
    if (MODE != 2 && MODE != 4 && !wide && valid) grid_ball(s_start, fx, fy, fz, rad, g.inv_h, [&](int s, int e) {
      if (MODE == 3) {   // statistics: columns visited, candidates
        atomicAdd(reinterpret_cast<unsigned long long*>(d_ar), 1ull);
        atomicAdd(reinterpret_cast<unsigned long long*>(d_ar) + 1, (unsigned long long)(e - s));
      }
      nn1_scan(s_p4, s, e, qx, qy, qz, best, bi);
    });
    if (MODE == 0) {
      constexpr int nact = 64;
      const int myrank = threadIdx.x & 63;
      unsigned long long todo = __ballot(wide);
      while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1ull;
        auto bcf = [&](float v) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), src)); };
        const float wqx = bcf(qx), wqy = bcf(qy), wqz = bcf(qz), wfx = bcf(fx), wfy = bcf(fy), wfz = bcf(fz), wrho = bcf(rho);
        const int wx0 = __builtin_amdgcn_readlane(bx0, src), wy0 = __builtin_amdgcn_readlane(by0, src);
        const int wny = __builtin_amdgcn_readlane(bny, src), wcols = __builtin_amdgcn_readlane(bcols, src);
        float cbest = bcf(best);
        int cbi = __builtin_amdgcn_readlane(bi, src);
        const float rho2 = wrho * wrho;
        for (int c = myrank; c < wcols; c += nact) {      // this lane's columns of the box: the body of grid_ball
          const int xx = wx0 + c / wny, yy = wy0 + c % wny;
          const float ex = fmaxf(fmaxf((float)xx - wfx, wfx - (float)(xx + 1)), 0.f);
          const float ey = fmaxf(fmaxf((float)yy - wfy, wfy - (float)(yy + 1)), 0.f);
          const float rem = rho2 - ex * ex - ey * ey;
          if (rem < 0.f) continue;
          const float zr = __builtin_amdgcn_sqrtf(rem) + 1e-4f;
          const int col = (xx * GG + yy) * GG;
          const int js = s_start[col + grid_coord(wfz - zr)], je = s_start[col + grid_coord(wfz + zr) + 1];
          nn1_scan(s_p4, js, je, wqx, wqy, wqz, cbest, cbi);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {      // lexicographic minimum over the wave
          const float od = __shfl_xor(cbest, o, 64);
          const int oi = __shfl_xor(cbi, o, 64);
          const bool take = od < cbest || (od == cbest && oi < cbi);
          cbest = take ? od : cbest;
          cbi = take ? oi : cbi;
        }
        if ((int)(threadIdx.x & 63) == src) {
          best = cbest;
          bi = cbi;
        }
      }
    }
    if (MODE != 3 && valid) {
      dout[q] = best;
      iout[q] = bi;
    }
  }
}

size_t grid_nn1_lds(int M) {
  const size_t g = ((size_t)(GC + 4) + GC + GW * 8 + 4 * (size_t)M + 2 + 2 * GT + 2 * GW * 64) * 4;
  if (M <= GT) return g;
  // (beyond 1024 points a workgroup may turn to the filter search: on the sorted cloud with its tail, or nf::search)
  const size_t f = GS_OFF_P4 + (size_t)M * 16 + GS_TAIL;
  const size_t h = f > nf::Cfg<GT>::LDS ? f : nf::Cfg<GT>::LDS;
  return g > h ? g : h;
}

}  // namespace

// Same contract as geoa3_launch_nn1 without `only`; returns GEOA3_ENOSUPPORT when a cloud does not fit the
// workgroup (callers then use the all-pairs kernel).
// brute_frac < 0 / filter < 0: the shipped policy
int geoa3_launch_grid_nn1_policy(const float* a, const float* r, int B, int Na, int Nr, const int32_t* prior_ar,
                                 const int32_t* prior_ra, float* d_ar, int32_t* i_ar, float* d_ra, int32_t* i_ra,
                                 float brute_frac, int filter, hipStream_t s) {
  const int M = Na > Nr ? Na : Nr;
  if ((d_ra == nullptr) != (i_ra == nullptr)) return GEOA3_EINVAL;
  const size_t lds = grid_nn1_lds(M);
  const int wide_thr = NN1_WIDE;
  if (filter < 0) filter = 1;
  filter = filter && Na >= 32 && Nr >= 32;
  if (brute_frac < 0.f) brute_frac = filter ? GRID_FILTER : GRID_BRUTE;
  // Up to 1024 points the filter kernel alone beats the walk wherever the iterates have left the surface, and costs the same
  // on every kind of cloud (tools/nn1_filter_check.py --time, 250 instances, us per launch: walk 34 at offsets of 0.02 of the
  // radius, 57 at 0.2, 50-170 on rods / clusters; filter ~45 everywhere): no grid at all
  // ... and beyond 4096 points (no grid fits the workgroup's LDS; the filter works in chunks of 1024 candidates)
  if (filter && (M <= GT || M > 4 * GT) && brute_frac < 1e8f)
    return geoa3_launch_nn1_filter(a, r, B, Na, Nr, prior_ar, prior_ra, d_ar, i_ar, d_ra, i_ra, s);
  if (M > 4 * GT) return GEOA3_ENOSUPPORT;
  dim3 grid(B, d_ra ? 2 : 1);
#define GEOA3_GRID_CASE(PPT)                                                                                    \
  if (M <= PPT * GT) {                                                                                          \
    auto kern = grid_nn1_kernel<PPT, 4>;                                                                         \
    if (lds > 64 * 1024)                                                                                        \
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                (int)lds);                                                                      \
    hipLaunchKernelGGL(kern, grid, dim3(GT), lds, s, a, r, Na, Nr, prior_ar, prior_ra, d_ar, i_ar, d_ra, i_ra, wide_thr,     \
                       brute_frac, filter);                                                                     \
    GEOA3_CHECK_LAUNCH();                                                                                       \
    return GEOA3_OK;                                                                                            \
  }
  GEOA3_GRID_CASE(1)
  GEOA3_GRID_CASE(2)
  GEOA3_GRID_CASE(4)
#undef GEOA3_GRID_CASE
  return GEOA3_ENOSUPPORT;
}

// Same contract as geoa3_launch_nn1 without `only`; returns GEOA3_ENOSUPPORT when a cloud does not fit the
// workgroup (callers then use the all-pairs kernel).
int geoa3_launch_grid_nn1(const float* a, const float* r, int B, int Na, int Nr, const int32_t* prior_ar,
                          const int32_t* prior_ra, float* d_ar, int32_t* i_ar, float* d_ra, int32_t* i_ra,
                          hipStream_t s) {
  return geoa3_launch_grid_nn1_policy(a, r, B, Na, Nr, prior_ar, prior_ra, d_ar, i_ar, d_ra, i_ra, -1.f, -1, s);
}

// geoa3_grid_nn1_pair with the search policy given (tests, tools/nn1_state_probe.py): brute_frac = the fraction of the
// searched cloud inside the queries' boxes beyond which a workgroup's queries are searched all-pairs (0: always), filter =
// by the matrix-core filter kernel (1) or the in-kernel sweep (0); negative = the shipped choice.  Same results all ways.
extern "C" int geoa3_debug_grid_nn1_pair(const float* a, const float* r, int B, int Na, int Nr, const int32_t* prior_ar,
                                         const int32_t* prior_ra, float* d_ar, int32_t* i_ar, float* d_ra, int32_t* i_ra,
                                         float brute_frac, int filter, void* stream) {
  if (!a || !r || !d_ar || !i_ar || B <= 0 || Na <= 0 || Nr <= 0) return GEOA3_EINVAL;
  return geoa3_launch_grid_nn1_policy(a, r, B, Na, Nr, prior_ar, prior_ra, d_ar, i_ar, d_ra, i_ra, brute_frac, filter,
                                      geoa3_stream(stream));
}

extern "C" int geoa3_grid_nn1_pair(const float* a, const float* r, int B, int Na, int Nr, const int32_t* prior_ar,
                                   const int32_t* prior_ra, float* d_ar, int32_t* i_ar, float* d_ra, int32_t* i_ra,
                                   void* stream) {
  if (!a || !r || !d_ar || !i_ar || B <= 0 || Na <= 0 || Nr <= 0) return GEOA3_EINVAL;
  geoa3_prof_begin(GEOA3_PROF_NN1, geoa3_stream(stream));
  const int rc = geoa3_launch_grid_nn1(a, r, B, Na, Nr, prior_ar, prior_ra, d_ar, i_ar, d_ra, i_ra, geoa3_stream(stream));
  geoa3_prof_end(GEOA3_PROF_NN1, geoa3_stream(stream));
  return rc;
}
