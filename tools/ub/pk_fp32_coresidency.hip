// Reproducer (round 5, NOTEBOOK 5a): a kernel whose fp32 arithmetic the SLP vectoriser packed into v_pk_add_f32 / v_pk_mul_f32
// (VECTOR-register operands, op_sel / neg modifiers) is exact and deterministic ALONE -- one wavefront per SIMD -- and
// computes wrong values at ~1e-4 per dependent round as soon as another kernel's wavefronts share its SIMDs.
// The kernel is the library's farthest-point sampler (geoa3_amd/csrc/pointnet2_ops.hip, fps_kernel<256, 4>: 511 dependent
// rounds per cloud, any wrong distance changes every later index); the neighbours: synthetic fillers of 8 waves per CU, and
// the library's own sa1_fwd_kernel (from libgeoa3_hip.so: 8 waves per CU, matrix core + LDS + vector work).
//   L="-L../../geoa3_amd/lib -lgeoa3_hip -Wl,-rpath,\$ORIGIN/../../geoa3_amd/lib"
//   hipcc --offload-arch=gfx950 -O3 -I../../include -I../../geoa3_amd/csrc -o pk_slp   pk_fp32_coresidency.hip $L
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I../../include -I../../geoa3_amd/csrc -o pk_noslp pk_fp32_coresidency.hip $L
//   ./pk_slp ; ./pk_noslp        (tools/gpu_pk_hazard.sh)
#include "../../geoa3_amd/csrc/pointnet2_ops.hip"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

// FILL 0: fp32 fma chains; 1: + one v_mfma_f32_32x32x16_f16 per 16 fmas; 2: LDS traffic instead; 5: PACKED fp32 fma chains
// (v_pk_fma_f32 on explicit two-element vectors); 6: packed multiplies + v_cvt_pk_f16_f32 + v_fma_mix (the operand split
// of the library's matrix-core kernels); 8: packed fma / add WITH neg source modifiers
template <int FILL>
__global__ __launch_bounds__(512) void filler_kernel(float* out, int iters) {
  __shared__ float s_f[512];
  float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f;
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 acc = {};
  half8 h = {};
  s_f[threadIdx.x] = a;
  __syncthreads();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) a = __builtin_fmaf(a, b, c);
    if (FILL == 1) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, acc, 0, 0, 0);
    if (FILL == 2) a += s_f[(threadIdx.x + i) & 511];
    if (FILL == 5) {
      typedef float float2v __attribute__((ext_vector_type(2)));
      float2v p = {a, c}, q = {b, b};
#pragma unroll
      for (int u = 0; u < 8; ++u) p = __builtin_elementwise_fma(p, q, q);
      a = p[0] + p[1];
    }
    if (FILL == 8) {   // PACKED fp32 with NEG source modifiers: v_pk_fma_f32 p, q, -r  and  v_pk_add_f32 p, -q
      typedef float float2v __attribute__((ext_vector_type(2)));
      float2v p = {a, c}, q = {b, b}, r = {c, a};
      asm volatile("" : "+v"(p), "+v"(q), "+v"(r));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        p = __builtin_elementwise_fma(p, q, -r);
        r = p - q;
      }
      a = p[0] + p[1] + r[0];
    }
    if (FILL == 6) {
      typedef float float2v __attribute__((ext_vector_type(2)));
      typedef _Float16 half2v __attribute__((ext_vector_type(2)));
      float2v p = {a, c};
      float sv = b;
      asm volatile("" : "+v"(sv));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const half2v hh = __builtin_convertvector(p * sv, half2v);
        const half2v ll = {(_Float16)__builtin_fmaf(p[0], sv, -(float)hh[0]), (_Float16)__builtin_fmaf(p[1], sv, -(float)hh[1])};
        p[0] += (float)hh[0] + (float)ll[1];
        p[1] += (float)hh[1] + (float)ll[0];
      }
      a = p[0] + p[1];
    }
  }
  if (a == 12345.f || acc[0] == 1.f) out[blockIdx.x] = a;
}

// WHERE the sampler's wavefronts sit in the SIMD's register file: a neighbour with NR live registers per lane at two waves per
// SIMD pushes every later wave behind 2 * NR registers (the sa1 kernels: 188 / 232).  MODE 0: fma chains over all NR
// registers; 1: the registers held, the waves ASLEEP (s_sleep: no vector instruction issues beside the sampler); 2: fma
// chains + dense v_mfma_f32_32x32x16_f16 on non-zero data; 3: the same on ZERO data; 4: ONE matrix instruction per pass of
// the chains instead of four; 5: matrix instructions only (no chains); 6: v_mfma_f32_32x32x2_f32 (the fp32 matrix instruction);
// 7: v_mfma_f32_16x16x32_f16
template <int NR, int MODE>
__global__ __launch_bounds__(512) void bloat_kernel(float* out, int iters) {
  float r[NR];
  const float b = 1.0001f, c = 0.5f;
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    r[i] = threadIdx.x * 1e-3f + i;
    asm volatile("" : "+v"(r[i]));
  }
  typedef _Float16 half8 __attribute__((ext_vector_type(8)));
  typedef float f32x16 __attribute__((ext_vector_type(16)));
  f32x16 acc = {};
  half8 h;
#pragma unroll
  for (int i = 0; i < 8; ++i) h[i] = MODE == 3 ? (_Float16)0.f : (_Float16)(0.01f * (threadIdx.x & 7) + 0.1f * i);
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  f32x4 acc4 = {};
  float hf = 0.01f * (threadIdx.x & 7) + 0.1f;
  asm volatile("" : "+v"(hf));
  for (int it = 0; it < iters; ++it) {
    if (MODE == 1) {
      __builtin_amdgcn_s_sleep(64);
    } else {
      if (MODE != 5) {
#pragma unroll
        for (int i = 0; i < NR; ++i) r[i] = __builtin_fmaf(r[i], b, c);
      }
      if (MODE == 2 || MODE == 3 || MODE == 5) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, acc, 0, 0, 0);
      }
      if (MODE == 4) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(h, h, acc, 0, 0, 0);
      if (MODE == 6) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hf, hf, acc, 0, 0, 0);
      }
      if (MODE == 7) {
#pragma unroll
        for (int u = 0; u < 4; ++u) acc4 = __builtin_amdgcn_mfma_f32_16x16x32_f16(h, h, acc4, 0, 0, 0);
      }
    }
  }
  float a = acc[0] + acc[7] + acc4[1];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    asm volatile("" : "+v"(r[i]));
    a += r[i];
  }
  if (a == 12345.f) out[blockIdx.x] = a;
}

int main() {
  const int B = 250, N = 1024, m = 512;
  std::vector<float> xyz((size_t)B * N * 3);
  srand(5);
  for (auto& v : xyz) v = (rand() % 20001 - 10000) * 1e-4f;
  float *dx, *dout;
  int32_t *idx, *ref;
  hipMalloc(&dx, xyz.size() * 4);
  hipMalloc(&dout, 4096);
  hipMalloc(&idx, (size_t)B * m * 4);
  hipMalloc(&ref, (size_t)B * m * 4);
  hipMemcpy(dx, xyz.data(), xyz.size() * 4, hipMemcpyHostToDevice);
  hipStream_t sa, sb;
  hipStreamCreateWithFlags(&sa, hipStreamNonBlocking);
  hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  launch_pn2_fps_range(dx, B, N, m, 0, m, nullptr, ref, sa);
  hipStreamSynchronize(sa);
  std::vector<int32_t> hr((size_t)B * m), hi((size_t)B * m);
  hipMemcpy(hr.data(), ref, hr.size() * 4, hipMemcpyDeviceToHost);
  // the neighbour from the library: level 1 of PointNet++ on random groups
  const int M = 512;
  std::vector<float> nx((size_t)B * M * 3), wts(64 * 3 + 64 + 64 * 64 + 64 + 128 * 64 + 128);
  std::vector<int32_t> gi((size_t)B * M * 64);
  for (auto& v : nx) v = (rand() % 20001 - 10000) * 1e-4f;
  for (auto& v : wts) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : gi) v = rand() % N;
  float *dnx, *dw, *dsa;
  int32_t* dgi;
  uint8_t* darg;
  hipMalloc(&dnx, nx.size() * 4); hipMalloc(&dw, wts.size() * 4); hipMalloc(&dgi, gi.size() * 4);
  hipMalloc(&dsa, (size_t)B * M * 128 * 4); hipMalloc(&darg, (size_t)B * M * 128);
  hipMemcpy(dnx, nx.data(), nx.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, wts.data(), wts.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dgi, gi.data(), gi.size() * 4, hipMemcpyHostToDevice);
  geoa3_sa1_weights sw{dw, dw + 192, dw + 256, dw + 256 + 4096, dw + 256 + 4096 + 64, dw + 256 + 4096 + 64 + 8192};
  float *dgo, *dgx, *dgn, *dscr;
  hipMalloc(&dgo, (size_t)B * M * 128 * 4); hipMalloc(&dgx, (size_t)B * N * 3 * 4); hipMalloc(&dgn, (size_t)B * M * 3 * 4);
  hipMalloc(&dscr, (size_t)geoa3_pn2_sa1_scratch_bytes(B, M));
  hipMemset(dgo, 0x3c, (size_t)B * M * 128 * 4);
  geoa3_pn2_sa1_forward(dx, dnx, dgi, &sw, B, N, M, dsa, darg, sb);   // out / arg for the backward neighbour
  hipStreamSynchronize(sb);
  auto trial = [&](int fill, int reps, const char* name) {
    long wrong_clouds = 0, wrong_launches = 0;
    for (int r = 0; r < reps; ++r) {
      if (fill == 0) hipLaunchKernelGGL(filler_kernel<0>, dim3(256), dim3(512), 0, sb, dout, 6000);
      if (fill == 1) hipLaunchKernelGGL(filler_kernel<1>, dim3(256), dim3(512), 0, sb, dout, 3000);
      if (fill == 2) hipLaunchKernelGGL(filler_kernel<2>, dim3(256), dim3(512), 0, sb, dout, 4000);
      if (fill == 5) hipLaunchKernelGGL(filler_kernel<5>, dim3(256), dim3(512), 0, sb, dout, 4000);
      if (fill == 6) hipLaunchKernelGGL(filler_kernel<6>, dim3(256), dim3(512), 0, sb, dout, 2500);
      if (fill == 8) hipLaunchKernelGGL(filler_kernel<8>, dim3(256), dim3(512), 0, sb, dout, 4000);
      if (fill == 10) hipLaunchKernelGGL((bloat_kernel<200, 0>), dim3(256), dim3(512), 0, sb, dout, 500);
      if (fill == 11) hipLaunchKernelGGL((bloat_kernel<48, 0>), dim3(256), dim3(512), 0, sb, dout, 2000);
      if (fill == 12) hipLaunchKernelGGL((bloat_kernel<200, 1>), dim3(256), dim3(512), 0, sb, dout, 60000);
      if (fill == 13) hipLaunchKernelGGL((bloat_kernel<200, 2>), dim3(256), dim3(512), 0, sb, dout, 400);
      if (fill == 14) hipLaunchKernelGGL((bloat_kernel<48, 2>), dim3(256), dim3(512), 0, sb, dout, 1500);
      if (fill == 15) hipLaunchKernelGGL((bloat_kernel<120, 0>), dim3(256), dim3(512), 0, sb, dout, 800);
      if (fill == 16) hipLaunchKernelGGL((bloat_kernel<48, 3>), dim3(256), dim3(512), 0, sb, dout, 1500);
      if (fill == 17) hipLaunchKernelGGL((bloat_kernel<48, 4>), dim3(256), dim3(512), 0, sb, dout, 1800);
      if (fill == 18) hipLaunchKernelGGL((bloat_kernel<48, 5>), dim3(256), dim3(512), 0, sb, dout, 4000);
      if (fill == 19) hipLaunchKernelGGL((bloat_kernel<48, 6>), dim3(256), dim3(512), 0, sb, dout, 1000);
      if (fill == 20) hipLaunchKernelGGL((bloat_kernel<48, 7>), dim3(256), dim3(512), 0, sb, dout, 1800);
      if (fill == 3) geoa3_pn2_sa1_forward(dx, dnx, dgi, &sw, B, N, M, dsa, darg, sb);
      if (fill == 7) geoa3_pn2_sa1_backward(dx, dnx, dgi, &sw, B, N, M, dsa, darg, dgo, dgx, dgn, dscr, sb);
      if (fill == 4) geoa3_pn2_ball_query(dnx, dx, B, N, M, 0.2f, 64, dgi, sb);
      launch_pn2_fps_range(dx, B, N, m, 0, m, nullptr, idx, sa);
      hipStreamSynchronize(sa);
      hipStreamSynchronize(sb);
      hipMemcpy(hi.data(), idx, hi.size() * 4, hipMemcpyDeviceToHost);
      long w = 0;
      for (int b = 0; b < B; ++b) {
        bool bad = false;
        for (int j = 0; j < m; ++j) bad |= hi[(size_t)b * m + j] != hr[(size_t)b * m + j];
        w += bad;
      }
      wrong_clouds += w;
      wrong_launches += w > 0;
    }
    printf("%-34s %3d launches: %ld with a wrong cloud, %ld wrong clouds of %d (%.1e per round)\n", name, reps, wrong_launches,
           wrong_clouds, reps * B, (double)wrong_clouds / ((double)reps * B * (m - 1)));
  };
  trial(-1, 40, "alone");
  trial(0, 40, "beside fp32 fma chains");
  trial(1, 40, "beside fma chains + MFMA");
  trial(2, 40, "beside fma chains + LDS reads");
  trial(5, 40, "beside PACKED fp32 fma chains");
  trial(6, 40, "beside packed mul + cvt_pk + fma_mix");
  trial(8, 100, "beside PACKED fp32 with NEG modifiers");
  if (getenv("PK_BLOAT")) {
    trial(11, 40, "beside 48-register fma waves");
    trial(15, 40, "beside 120-register fma waves");
    trial(10, 100, "beside 200-register fma waves");
    trial(12, 100, "beside 200-register SLEEPING waves");
    trial(14, 40, "beside 48-register fma + MFMA");
    trial(13, 100, "beside 200-register fma + MFMA");
    trial(16, 40, "beside fma + 4 MFMA on ZERO data");
    trial(17, 40, "beside fma + 1 MFMA per pass");
    trial(18, 40, "beside MFMA only, back to back");
    trial(19, 40, "beside fma + 4 fp32 MFMA (32x32x2)");
    trial(20, 40, "beside fma + 4 MFMA 16x16x32 f16");
  }
  trial(3, 200, "beside the library's sa1_fwd_kernel");
  trial(7, 100, "beside the library's sa1_bwd_kernel");
  trial(4, 200, "beside the library's ball query");
  trial(-1, 40, "alone again");

  // ---- the library's OWN geometry kernels (compiled with the SLP vectoriser, they hold packed FP32) run beside the sa1
  // kernels in every PointNet++ iteration (second stream of the attack loop): the same comparison for them
  {
    std::vector<float> pa((size_t)B * 3 * N), pr((size_t)B * 3 * N);
    for (size_t e = 0; e < pa.size(); ++e) {
      pr[e] = (rand() % 20001 - 10000) * 1e-4f;
      pa[e] = pr[e] + (rand() % 2001 - 1000) * 1e-5f;
    }
    float *da, *dr, *d1, *d2, *kd, *r1, *r2, *rkd;
    int32_t *i1, *i2, *ki, *ri1, *ri2, *rki, *prior;
    void* scr;
    const int K = 17;
    const size_t nb = (size_t)B * N;
    hipMalloc(&da, pa.size() * 4); hipMalloc(&dr, pr.size() * 4);
    hipMalloc(&d1, nb * 4); hipMalloc(&d2, nb * 4); hipMalloc(&i1, nb * 4); hipMalloc(&i2, nb * 4);
    hipMalloc(&r1, nb * 4); hipMalloc(&r2, nb * 4); hipMalloc(&ri1, nb * 4); hipMalloc(&ri2, nb * 4);
    hipMalloc(&kd, nb * K * 4); hipMalloc(&ki, nb * K * 4); hipMalloc(&rkd, nb * K * 4); hipMalloc(&rki, nb * K * 4);
    hipMalloc(&prior, nb * K * 4);
    hipMalloc(&scr, (size_t)geoa3_knn_self_scratch_bytes(B, N));
    hipMemcpy(da, pa.data(), pa.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dr, pr.data(), pr.size() * 4, hipMemcpyHostToDevice);
    geoa3_grid_nn1_pair(da, dr, B, N, N, nullptr, nullptr, r1, ri1, r2, ri2, sa);
    geoa3_knn(dr, dr, B, N, N, K, nullptr, rkd, prior, sa);            // the clean cloud's table as the prior
    geoa3_knn_self(da, B, N, K, prior, rkd, rki, scr, 0, sa);
    hipStreamSynchronize(sa);
    std::vector<int32_t> h0(nb * K), h1(nb * K);
    auto same = [&](const void* x, const void* y, size_t bytes) {
      hipMemcpy(h0.data(), x, bytes, hipMemcpyDeviceToHost);
      hipMemcpy(h1.data(), y, bytes, hipMemcpyDeviceToHost);
      return memcmp(h0.data(), h1.data(), bytes) == 0;
    };
    for (int nbk = 0; nbk < 2; ++nbk) {
      long wrong_nn1 = 0, wrong_knn = 0;
      const int reps = 150;
      for (int r = 0; r < reps; ++r) {
        if (nbk == 0) geoa3_pn2_sa1_forward(dx, dnx, dgi, &sw, B, N, M, dsa, darg, sb);
        else geoa3_pn2_sa1_backward(dx, dnx, dgi, &sw, B, N, M, dsa, darg, dgo, dgx, dgn, dscr, sb);
        geoa3_grid_nn1_pair(da, dr, B, N, N, nullptr, nullptr, d1, i1, d2, i2, sa);
        geoa3_knn_self(da, B, N, K, prior, kd, ki, scr, 0, sa);
        hipStreamSynchronize(sa);
        hipStreamSynchronize(sb);
        wrong_nn1 += !(same(d1, r1, nb * 4) && same(i1, ri1, nb * 4) && same(d2, r2, nb * 4) && same(i2, ri2, nb * 4));
        wrong_knn += !(same(kd, rkd, nb * K * 4) && same(ki, rki, nb * K * 4));
      }
      printf("library grid 1-NN + self K-NN beside %s: %d launches, %ld / %ld with a wrong table\n",
             nbk == 0 ? "sa1_fwd_kernel" : "sa1_bwd_kernel", reps, wrong_nn1, wrong_knn);
    }
  }
  return 0;
}
