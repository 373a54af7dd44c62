// PointNet++ SSG classifier (eval mode) as ONE native forward and ONE native input-gradient entry point: the launch
// sequence over the set-abstraction kernels of pointnet2_ops.hip / pointnet2_sa.hip / pointnet2_mlp.hip, the split-fp16
// 1x1 convolutions of pointnet_conv_split.hip and the FC kernel of pointnet_gemm.hip.  No torch operator, no library
// GEMM, no host synchronisation in between.
//
// Reference: Model/PointNetPP_ssg.py:51-124 (PointNet2ClassificationSSG: SA(512, 0.2, 64, [3,64,64,128]) ->
// SA(128, 0.4, 64, [128+3,128,128,256]) -> SA(GroupAll, [256+3,256,512,1024]) -> FC 1024-512-256-classes),
// Model/pointnet2_ops_lib/pointnet2_ops/pointnet2_modules.py:29-74 (_PointnetSAModuleBase.forward),
// pointnet2_utils.py:296-333 (QueryAndGroup), :349-379 (GroupAll); the backward is the hand-derived input gradient of
// that graph (the attack never needs weight gradients; the reference's autograd forms them anyway,
// Attacker/geoA3_attack.py:326).
//
// Algebra used (eval mode, BatchNorm folded into W / shift on the host, geoa3_amd/pointnet2.py pack_ssg):
//   * a level's first layer is linear in the grouped input: W [xyz_j - c_m ; f_j] = (W_x xyz + W_f f)_j - (W_x c)_m, so it
//     is applied to the UN-grouped points and the result is gathered (level 2); GroupAll (level 3) has no centre:
//     layer 1 = relu(W_f f + W_x xyz + b);
//   * level 3's 512 -> 1024 layer runs as two K = 256 products accumulated before the bias / relu / max tail.
#include <new>
#include "pointnet_kernels.h"
#include "profile.h"

namespace {

constexpr int M1 = 512, M2 = 128, S = 64;          // PointNetPP_ssg.py:58-76
constexpr float R1 = 0.2f, R2 = 0.4f;
constexpr int C1 = 128, C2 = 256, C3 = 1024;       // output widths of the three levels
constexpr int MIN_N = 32;      // (the sampler picks 512 centroids of ANY cloud, repeating points of a small one: sampling_gpu.cu:69-173)
constexpr int MAX_N = 8192;    // the sampler's register tiling (pointnet2_ops.hip); also keeps N * 3 <= C1 * M1 floats (gx2 in the backward)

#define TRY(expr)               \
  do {                          \
    int rc__ = (expr);          \
    if (rc__ != 0) return rc__; \
  } while (0)

// ---- fragment images of the level-2 / level-3 matrices (launch_frag_image), one buffer: image i at img_off(i), its
// 1 / scale in the 256 bytes behind it
enum ImgId { IM_SA2_WF, IM_SA2_W1, IM_SA2_W2, IM_SA2_W1T, IM_SA2_WFT, IM_SA3_WF, IM_SA3_W1, IM_SA3_W2, IM_SA3_W2T, IM_SA3_W1T,
             IM_SA3_WFT, IM_COUNT };
constexpr int IMG_R[IM_COUNT] = {128, 128, 256, 128, 128, 256, 512, 1024, 512, 256, 256};
constexpr int IMG_K[IM_COUNT] = {128, 128, 128, 128, 128, 256, 256, 512, 1024, 512, 256};
constexpr size_t img_off(int i) {
  size_t o = 0;
  for (int j = 0; j < i; ++j) o += (size_t)IMG_R[j] * IMG_K[j] * 4 + 256;
  return o;
}
struct Img {
  const void* p;
  const float* un;
};
inline Img img_of(const void* base, int i) {
  const char* b = static_cast<const char*>(base) + img_off(i);
  return Img{b, reinterpret_cast<const float*>(b + (size_t)IMG_R[i] * IMG_K[i] * 4)};
}
// every matrix either direction reads (a NULL here would fault inside a kernel instead of returning GEOA3_EINVAL)
bool weights_complete(const geoa3_pn2ssg_weights& p) {
  const void* need[] = {p.sa1.w1, p.sa1.b1, p.sa1.w2, p.sa1.b2, p.sa1.w3, p.sa1.b3, p.sa2_wx, p.sa2_wf, p.sa2_b0, p.sa2_wft,
                        p.sa2_w1, p.sa2_b1, p.sa2_w1t, p.sa2_w2, p.sa2_b2, p.sa2_w2t, p.sa3_wx, p.sa3_wf, p.sa3_b0,
                        p.sa3_wft, p.sa3_w1, p.sa3_b1, p.sa3_w1t, p.sa3_w2, p.sa3_b2, p.f1, p.fb1, p.f1t, p.f2, p.fb2,
                        p.f2t, p.f3, p.fb3, p.f3t};
  for (const void* q : need)
    if (!q) return false;
  return true;
}

int pack_images(const geoa3_pn2ssg_weights& p, void* base, hipStream_t s) {
  const float* W[IM_COUNT] = {p.sa2_wf, p.sa2_w1, p.sa2_w2, p.sa2_w1t, p.sa2_wft, p.sa3_wf, p.sa3_w1, p.sa3_w2, p.sa3_w2t,
                              p.sa3_w1t, p.sa3_wft};
  for (int i = 0; i < IM_COUNT; ++i) {
    if (i == IM_SA3_W2T) continue;   // (slot kept so the offsets stay; the backward walks sa3_w2 sparsely instead)
    const Img im = img_of(base, i);
    TRY(launch_frag_image(W[i], IMG_R[i], IMG_K[i], const_cast<void*>(im.p), const_cast<float*>(im.un), s));
  }
  return GEOA3_OK;
}

struct Ws {
  float *xyz, *nx1, *out1, *f1, *nx2, *r, *shift, *da0, *out2, *h1, *h2, *z3, *p3, *q1, *q2;
  int32_t *idx1, *gidx1, *idx2, *gidx2, *arg2, *arg3;
  uint8_t* arg1;
  int32_t* bad;    // [B]: the cloud holds a NaN / Inf coordinate (its logits and its input gradient are NaN)
  // backward
  float *g256, *g512, *g1024, *dh2, *dh1, *dout2, *dnx2, *d1, *df1, *dnx1, *g1, *gxyz, *gnx1;
  unsigned *m0, *m1;   // relu gates of the level-2 activations a0, a1 as bits per row: [B * M2][64][4]
  void* images;   // fragment images, when the caller passes none (geoa3_pn2ssg_weights::images == NULL)
  size_t total;
};

Ws carve(void* base, int B, int N) {
  Ws w{};
  size_t off = 0;
  char* p = static_cast<char*>(base);
  auto take = [&](size_t bytes) {
    void* r = p ? p + off : nullptr;
    off += (bytes + 255) / 256 * 256;
    return r;
  };
  const size_t b = (size_t)B, f = sizeof(float);
  w.xyz = (float*)take(b * N * 3 * f);
  w.idx1 = (int32_t*)take(b * M1 * 4);
  w.nx1 = (float*)take(b * M1 * 3 * f);
  w.gidx1 = (int32_t*)take(b * M1 * S * 4);
  w.out1 = (float*)take(b * M1 * C1 * f);
  w.arg1 = (uint8_t*)take(b * M1 * C1);
  w.f1 = (float*)take(b * C1 * M1 * f);
  w.idx2 = (int32_t*)take(b * M2 * 4);
  w.nx2 = (float*)take(b * M2 * 3 * f);
  w.gidx2 = (int32_t*)take(b * M2 * S * 4);
  w.r = (float*)take(b * 128 * M1 * f);
  w.shift = (float*)take(b * 128 * M2 * f);
  w.da0 = (float*)take(sa2b_scratch_bytes(B));   // level 2's backward: rows, entries, tiles (the [B,128,128,64] grouped gradient this slot held until round 5 no longer exists)
  w.out2 = (float*)take(b * C2 * M2 * f);
  w.arg2 = (int32_t*)take(b * C2 * M2 * 4);
  w.h1 = (float*)take(b * 256 * M2 * f);
  w.h2 = (float*)take(b * 512 * M2 * f);
  w.z3 = (float*)take(b * C3 * M2 * f);
  w.p3 = (float*)take(b * C3 * f);
  w.arg3 = (int32_t*)take(b * C3 * 4);
  w.q1 = (float*)take(b * 512 * f);
  w.q2 = (float*)take(b * 256 * f);
  w.g256 = (float*)take(b * 256 * f);
  w.g512 = (float*)take(b * 512 * f);
  w.g1024 = (float*)take(b * C3 * f);
  w.dh2 = (float*)take(b * 512 * M2 * f);
  w.dh1 = (float*)take(b * 256 * M2 * f);
  w.dout2 = (float*)take(b * C2 * M2 * f);
  w.dnx2 = (float*)take(b * M2 * 3 * f);
  w.d1 = (float*)take(b * (size_t)(M1 * S * 3 > MAX_N ? M1 * S * 3 : MAX_N) * f);   // [B,N] sampler distances | [B,512,64,3] level-1 contributions
  w.df1 = (float*)take(b * C1 * M1 * f);
  w.dnx1 = (float*)take(b * M1 * 3 * f);
  w.g1 = (float*)take(b * M1 * C1 * f);
  w.gxyz = (float*)take(b * N * 3 * f);
  w.gnx1 = (float*)take(b * M1 * 3 * f);
  w.bad = (int32_t*)take(b * 4);
  w.m0 = (unsigned*)take(b * M2 * S * 16);
  w.m1 = (unsigned*)take(b * M2 * S * 16);
  w.images = take(img_off(IM_COUNT));
  w.total = off;
  return w;
}

// ---- layout helpers -------------------------------------------------------------------------------------------------
// planar [B,3,N] <-> point-major [B,N,3]
__global__ __launch_bounds__(256) void planar_to_points_kernel(const float* __restrict__ x, float* __restrict__ xyz, int N,
                                                               long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const long b = e / N;
  const int n = (int)(e - b * N);
  const float* p = x + b * 3 * N + n;
  float* q = xyz + e * 3;
  q[0] = p[0];
  q[1] = p[N];
  q[2] = p[2 * (size_t)N];
}
// A NaN / Inf coordinate must not come out as finite logits: the level kernels are compiled with -fno-honor-nans (relu / max
// without the canonicalising instruction), the sampler skips what it cannot compare, a ball query never admits a NaN
// distance.  One workgroup per cloud looks at the bits and leaves a flag; the cloud's logits and its input gradient are NaN.
__global__ __launch_bounds__(256) void nonfinite_flag_kernel(const float* __restrict__ x, int32_t* __restrict__ bad, int n3) {
  const unsigned* p = reinterpret_cast<const unsigned*>(x) + (size_t)blockIdx.x * n3;
  bool any = false;
  for (int e = threadIdx.x; e < n3; e += 256) any = any || (p[e] & 0x7f800000u) == 0x7f800000u;
  const int r = __syncthreads_or(any);
  if (threadIdx.x == 0) bad[blockIdx.x] = r;
}
__global__ __launch_bounds__(256) void poison_rows_kernel(float* __restrict__ y, const int32_t* __restrict__ bad, int cols, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < total && bad[e / cols]) y[e] = __uint_as_float(0x7fc00000u);
}
// (xyz2: a second point-major tensor added on the way, or null; bad: clouds whose gradient is NaN, see above)
__global__ __launch_bounds__(256) void points_to_planar_kernel(const float* __restrict__ xyz, const float* __restrict__ xyz2,
                                                               float* __restrict__ x, int N, long total,
                                                               const int32_t* __restrict__ bad) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const long b = e / N;
  const int n = (int)(e - b * N);
  const float* q = xyz + e * 3;
  float v0 = q[0], v1 = q[1], v2 = q[2];
  if (xyz2) {
    const float* r = xyz2 + e * 3;
    v0 += r[0];
    v1 += r[1];
    v2 += r[2];
  }
  if (bad[b]) v0 = v1 = v2 = __uint_as_float(0x7fc00000u);
  float* p = x + b * 3 * N + n;
  p[0] = v0;
  p[N] = v1;
  p[2 * (size_t)N] = v2;
}

// out[b][m][:] = in[b][idx[b][m]][:]   (rows of 3 floats), m in [m0, m0 + Mc) of the M; total = B * Mc
__global__ __launch_bounds__(256) void gather_rows3_kernel(const float* __restrict__ in, const int32_t* __restrict__ idx,
                                                           float* __restrict__ out, int N, int M, int m0, int Mc, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const long b = e / Mc, r = b * M + m0 + (e - b * Mc);
  const float* p = in + (b * N + idx[r]) * 3;
  float* q = out + r * 3;
  q[0] = p[0];
  q[1] = p[1];
  q[2] = p[2];
}

// dst[b][n][:] (+)= sum over m with idx[b][m] == n of src[b][m][:], in ascending m: the gradient of gather_rows3 as an
// OWNER-side scan (every destination row sums its own sources in a fixed order: deterministic, no atomics; FPS indices
// are distinct except for degenerate clouds, where duplicates simply add up)
__global__ __launch_bounds__(256) void scatter_rows3_kernel(const float* __restrict__ src, const int32_t* __restrict__ idx,
                                                            float* __restrict__ dst, int N, int M, int accumulate) {
  extern __shared__ int32_t s_idx[];
  const int b = blockIdx.y, n = blockIdx.x * 256 + threadIdx.x;
  for (int m = threadIdx.x; m < M; m += 256) s_idx[m] = idx[(size_t)b * M + m];
  __syncthreads();
  if (n >= N) return;
  float* q = dst + ((size_t)b * N + n) * 3;
  float a0 = accumulate ? q[0] : 0.f, a1 = accumulate ? q[1] : 0.f, a2 = accumulate ? q[2] : 0.f;
  const float* sp = src + (size_t)b * M * 3;
  for (int m = 0; m < M; ++m)
    if (s_idx[m] == n) {
      a0 += sp[3 * m];
      a1 += sp[3 * m + 1];
      a2 += sp[3 * m + 2];
    }
  q[0] = a0;
  q[1] = a1;
  q[2] = a2;
}

// [B][R][C] -> [B][C][R] through a 32 x 33 LDS tile.  GATE: the value is zeroed where gate[b][r][c] <= 0 (the pooled
// output's relu folded into the transpose that produces the centre-major gradient of geoa3_conv1x1_onehot64).
template <typename T, bool GATE>
__global__ __launch_bounds__(256) void transpose_kernel(const T* __restrict__ in, const float* __restrict__ gate,
                                                        T* __restrict__ out, int R, int C) {
  __shared__ T tile[32][33];
  const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const size_t base = (size_t)b * R * C;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = r0 + ty + 8 * j, c = c0 + tx;
    if (r < R && c < C) {
      T v = in[base + (size_t)r * C + c];
      if (GATE) v = gate[base + (size_t)r * C + c] > 0.f ? v : (T)0;
      tile[ty + 8 * j][tx] = v;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = c0 + ty + 8 * j, r = r0 + tx;
    if (r < R && c < C) out[base + (size_t)c * R + r] = tile[tx][ty + 8 * j];
  }
}

// ---- the 3-channel (xyz) part of a level's first layer -------------------------------------------------------------
// out[b][co][m] = bias[co] + sign * <Wx[co], p[b][m]>        (shift of the pre-transformed level: b0 - Wx c_m)
__global__ __launch_bounds__(256) void affine3_kernel(const float* __restrict__ Wx, const float* __restrict__ bias,
                                                      const float* __restrict__ p, float sign, float* __restrict__ out,
                                                      int Co, int M, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int m = (int)(e % M);
  const long t = e / M;
  const int co = (int)(t % Co);
  const long b = t / Co;
  const float* q = p + (b * M + m) * 3;
  const float d = Wx[3 * co] * q[0] + Wx[3 * co + 1] * q[1] + Wx[3 * co + 2] * q[2];
  out[e] = (bias ? bias[co] : 0.f) + sign * d;
}
// Y[b][co][n] = act(Y[b][co][n] + <Wx[co], p[b][n]> + bias[co])   in place (p point-major [B,N,3])
__global__ __launch_bounds__(256) void affine3_add_kernel(float* __restrict__ Y, const float* __restrict__ Wx,
                                                          const float* __restrict__ bias, const float* __restrict__ p,
                                                          int Co, int N, int relu, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int n = (int)(e % N);
  const long t = e / N;
  const int co = (int)(t % Co);
  const long b = t / Co;
  const float* q = p + (b * N + n) * 3;
  float v = Y[e] + (Wx[3 * co] * q[0] + Wx[3 * co + 1] * q[1] + Wx[3 * co + 2] * q[2]);
  if (bias) v += bias[co];
  if (relu) v = fmaxf(v, 0.f);
  Y[e] = v;
}
// dp[b][n][c] (+)= sign * sum_co Wx[co][c] dY[b][co][n]   (one thread per point, co ascending: fixed order)
__global__ __launch_bounds__(256) void affine3_grad_kernel(const float* __restrict__ dY, const float* __restrict__ Wx,
                                                           float sign, float* __restrict__ dp, int Co, int N,
                                                           int accumulate, long total) {
  extern __shared__ float s_wx[];   // [Co][3]
  for (int i = threadIdx.x; i < 3 * Co; i += 256) s_wx[i] = Wx[i];
  __syncthreads();
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const long b = e / N;
  const int n = (int)(e - b * N);
  const float* g = dY + b * Co * N + n;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int co = 0; co < Co; ++co) {
    const float v = g[(size_t)co * N];
    a0 += s_wx[3 * co] * v;
    a1 += s_wx[3 * co + 1] * v;
    a2 += s_wx[3 * co + 2] * v;
  }
  float* q = dp + e * 3;
  if (accumulate) {
    q[0] += sign * a0;
    q[1] += sign * a1;
    q[2] += sign * a2;
  } else {
    q[0] = sign * a0;
    q[1] = sign * a1;
    q[2] = sign * a2;
  }
}
// a[i] += b[i]
__global__ __launch_bounds__(256) void add_inplace_kernel(float* __restrict__ a, const float* __restrict__ b, long total) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e < total) a[e] += b[e];
}

inline dim3 g1d(long total) { return dim3((unsigned)((total + 255) / 256)); }

template <typename T, bool GATE>
int transpose(const T* in, const float* gate, T* out, int B, int R, int C, hipStream_t s) {
  hipLaunchKernelGGL((transpose_kernel<T, GATE>), dim3((C + 31) / 32, (R + 31) / 32, B), dim3(256), 0, s, in, gate, out, R, C);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

// Y[b][co][n] = epi(sum_{k < K} W[co][k0 + k] X[b][k0 + k][n]) over a K-slice of a [Co][Kfull] weight / [B][Kfull][N] input
int conv_slice(const float* X, int Kfull, int k0, int K, const Img& W, const float* bias, const float* Z, float* Y, int Co,
               int B, int N, bool relu, bool accumulate, hipStream_t s) {
  ConvArgs a{};
  a.split = 1;
  a.X = X + (size_t)k0 * N; a.sXb = (long)Kfull * N; a.ldX = N;
  a.Wimg = W.p; a.Wun = W.un; a.img_kc = Kfull / 16; a.img_c0 = k0 / 16;
  a.bias = bias;
  a.Z = Z; a.sZb = (long)Co * N; a.ldZ = N;
  a.Y = Y; a.sYb = (long)Co * N; a.ldY = N;
  a.Co = Co; a.K = K; a.N = N; a.B = B;
  a.relu = relu; a.accumulate = accumulate;
  a.pack2 = N <= 128;   // level 3's [B][C][128] tensors: two instances per workgroup
  return launch_conv_cm(a, s);
}

int fc(const float* X, int K, const float* W, const float* bias, float* Y, int Nout, int M, bool relu, const float* Z,
       hipStream_t s) {
  FcArgs a{};
  a.X = X; a.ldX = K;
  a.W = W; a.ldW = K;
  a.bias = bias;
  a.Z = Z; a.ldZ = Nout;
  a.Y = Y; a.ldY = Nout;
  a.M = M; a.Nout = Nout; a.K = K; a.relu = relu;
  return launch_fc(a, s);
}

}  // namespace

extern "C" int64_t geoa3_pn2ssg_workspace_bytes(int B, int N) {
  if (B <= 0 || N < MIN_N || N > MAX_N) return -1;
  return (int64_t)carve(nullptr, B, N).total;
}

extern "C" int64_t geoa3_pn2ssg_images_bytes(void) { return (int64_t)img_off(IM_COUNT); }

extern "C" int geoa3_pn2ssg_pack_images(const geoa3_pn2ssg_weights* pw, void* images, void* stream) {
  if (!pw || !images || ((uintptr_t)images & 255) != 0) return GEOA3_EINVAL;
  return pack_images(*pw, images, geoa3_stream(stream));
}

namespace {
struct SideQueue {
  hipStream_t stream;
  hipEvent_t ev[4], join;   // ev[c]: centroid chunk c of level 1 is chosen and gathered; join: level 2's geometry is done
};

// A failing launch between a fork and its join must not leave the side stream un-joined (an eager caller would reuse the
// workspace under kernels still running there; a stream capture would be left with an unjoined branch): once armed, leaving the
// scope on any path records the queue's join event and makes `main` wait for it.
struct SideJoin {
  SideQueue* q = nullptr;
  hipStream_t main = nullptr;
  void arm(SideQueue* queue, hipStream_t m) { q = queue; main = m; }
  void disarm() { q = nullptr; }
  ~SideJoin() {
    if (q) {
      (void)hipEventRecord(q->join, q->stream);
      (void)hipStreamWaitEvent(main, q->join, 0);
    }
  }
};
}  // namespace

extern "C" void geoa3_side_queue_destroy(void* side) {
  SideQueue* q = static_cast<SideQueue*>(side);
  if (!q) return;
  if (q->stream) (void)hipStreamSynchronize(q->stream);
  for (hipEvent_t e : q->ev)
    if (e) (void)hipEventDestroy(e);
  if (q->join) (void)hipEventDestroy(q->join);
  if (q->stream) (void)hipStreamDestroy(q->stream);
  delete q;
}

extern "C" void* geoa3_side_queue_create(void) {
  SideQueue* q = new (std::nothrow) SideQueue{};
  if (!q) return nullptr;
  bool ok = hipStreamCreateWithFlags(&q->stream, hipStreamNonBlocking) == hipSuccess;
  for (hipEvent_t& e : q->ev) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&q->join, hipEventDisableTiming) == hipSuccess;
  if (!ok) {
    geoa3_side_queue_destroy(q);
    return nullptr;
  }
  return q;
}

extern "C" int geoa3_pn2ssg_forward(const geoa3_pn2ssg_weights* pw, const float* x, int B, int N, float* logits,
                                    void* workspace, void* stream) {
  if (!pw || !x || !logits || !workspace || B <= 0 || N < MIN_N || N > MAX_N || pw->classes <= 0) return GEOA3_EINVAL;
  if (((uintptr_t)workspace & 255) != 0 || !weights_complete(*pw)) return GEOA3_EINVAL;
  hipStream_t s = geoa3_stream(stream);
  const geoa3_pn2ssg_weights& p = *pw;
  Ws w = carve(workspace, B, N);
  const void* im = p.images;
  if (!im) {   // no packed images: build them into the workspace (eleven one-workgroup launches per call)
    TRY(pack_images(p, w.images, s));
    im = w.images;
  }
  // ---- level 1 (PointNetPP_ssg.py:58-66): FPS 512, ball 0.2 x 64, MLP 3 -> 64 -> 64 -> 128, max
  hipLaunchKernelGGL(nonfinite_flag_kernel, dim3(B), dim3(256), 0, s, x, w.bad, 3 * N);
  hipLaunchKernelGGL(planar_to_points_kernel, g1d((long)B * N), dim3(256), 0, s, x, w.xyz, N, (long)B * N);
  // With a side queue in the weights the level is PIPELINED.  The sampler is a chain of 511 dependent rounds in one workgroup
  // per cloud (0.33 ms whatever the batch), and nothing of the level can start before its first centroid: so the rounds run
  // in four launches of 128 (the running distances pass through memory: same bits), each followed by the gather and the
  // ball query of its 128 centroids, on the queue's stream, and the MLP of a chunk runs on `stream` beside the next chunk's.  Behind the last sampler
  // launch the side stream continues with level 2's sampling, ball query and shift (functions of the level-1 centroids
  // only) and is joined in front of level 2's MLP.  The call is still ordered on `stream` as a whole (the side stream
  // starts behind an event of it and is waited for); twelve launches more per forward.  (A launch failure between the fork
  // and the join still joins the side stream: SideJoin.)
  SideQueue* sq = static_cast<SideQueue*>(p.side);
  const bool ct = (p.flags & GEOA3_PN2_CONTRACT) != 0;   // the _ext distances as nvcc's default contraction forms them
  // (while bench.py samples sa1_fwd_kernel's time the level runs as one launch per kernel)
  const bool use_side = sq != nullptr && !geoa3_prof_tag_on(GEOA3_PROF_SA1_FWD) && (size_t)N * 3 * sizeof(float) + 1024 <= 128 * 1024;
  hipStream_t s2 = use_side ? sq->stream : s;
  SideJoin fork;
  constexpr int CH = 4;   // (1 / 2 / 4 / 8 equal launches measured: 4.228 / 4.221 / 4.196 / 4.205 ms per iteration)
  // (a smaller first launch -- 32 / 64 / 96 centroids -- to start the MLP earlier: 4.01-4.02 / 4.00-4.01 / 3.99 ms against 3.97)
  constexpr int cut[CH + 1] = {0, M1 / 4, M1 / 2, 3 * M1 / 4, M1};
  float* fps_td = w.d1;   // [B,N] running distances between the sampler's launches (a backward buffer, free in forward)
  auto rows = [&](int m0, int m1, hipStream_t st) {   // the centroid rows nx1[:, m0 .. m1 - 1]
    hipLaunchKernelGGL(gather_rows3_kernel, g1d((long)B * (m1 - m0)), dim3(256), 0, st, w.xyz, w.idx1, w.nx1, N, M1, m0, m1 - m0,
                       (long)B * (m1 - m0));
  };
  if (use_side) {
    TRY(launch_pn2_fps_range(w.xyz, B, N, M1, 0, cut[1], fps_td, w.idx1, s, ct));
    rows(0, cut[1], s);
    if (hipEventRecord(sq->ev[0], s) != hipSuccess || hipStreamWaitEvent(s2, sq->ev[0], 0) != hipSuccess) return GEOA3_ELAUNCH;
    fork.arm(sq, s);   // from here to the join below every early return still joins the side stream
    TRY(launch_pn2_ball_query_range(w.nx1, w.xyz, B, N, M1, 0, cut[1], R1, S, w.gidx1, s, ct));
    for (int c = 1; c < CH; ++c) {   // sampler, centroid rows and ball query of chunk c: all on the side stream
      TRY(launch_pn2_fps_range(w.xyz, B, N, M1, cut[c], cut[c + 1], fps_td, w.idx1, s2, ct));
      rows(cut[c], cut[c + 1], s2);
      TRY(launch_pn2_ball_query_range(w.nx1, w.xyz, B, N, M1, cut[c], cut[c + 1], R1, S, w.gidx1, s2, ct));
      if (hipEventRecord(sq->ev[c], s2) != hipSuccess) return GEOA3_ELAUNCH;
    }
    for (int c = 0; c < CH; ++c) {
      if (c > 0 && hipStreamWaitEvent(s, sq->ev[c], 0) != hipSuccess) return GEOA3_ELAUNCH;
      TRY(launch_sa1_forward_range(w.xyz, w.nx1, w.gidx1, &p.sa1, B, N, M1, cut[c], cut[c + 1], w.out1, w.arg1, s));
    }
  } else {
    TRY(launch_pn2_fps_range(w.xyz, B, N, M1, 0, M1, nullptr, w.idx1, s, ct));
    rows(0, M1, s);
    TRY(geoa3_pn2_ball_query_ex(w.nx1, w.xyz, B, N, M1, R1, S, w.gidx1, ct ? GEOA3_PN2_CONTRACT : 0, stream));
    TRY(launch_sa1_forward_range(w.xyz, w.nx1, w.gidx1, &p.sa1, B, N, M1, 0, M1, w.out1, w.arg1, s));
  }
  // ---- level 2's geometry (:68-76): FPS 128, ball 0.4 x 64, b0 - W_x c -- on the side stream behind the sampler (every
  // centroid row is written by then: the first 128 on `stream` before the fork, the others on the side stream itself)
  TRY(launch_pn2_fps_range(w.nx1, B, M1, M2, 0, M2, nullptr, w.idx2, s2, ct));
  hipLaunchKernelGGL(gather_rows3_kernel, g1d((long)B * M2), dim3(256), 0, s2, w.nx1, w.idx2, w.nx2, M1, M2, 0, M2, (long)B * M2);
  TRY(launch_pn2_ball_query_range(w.nx2, w.nx1, B, M1, M2, 0, M2, R2, S, w.gidx2, s2, ct));
  hipLaunchKernelGGL(affine3_kernel, g1d((long)B * 128 * M2), dim3(256), 0, s2, p.sa2_wx, p.sa2_b0, w.nx2, -1.f, w.shift,
                     128, M2, (long)B * 128 * M2);                                                         // b0 - W_x c
  if (use_side && hipEventRecord(sq->join, s2) != hipSuccess) return GEOA3_ELAUNCH;
  // ---- level 2 (:68-76): MLP (128 + 3) -> 128 -> 128 -> 256, max.  r = W_f f + W_x xyz per POINT, point-major in and out
  // (one kernel: the features leave level 1 point-major and the gather below wants r point-major)
  float* rt = w.df1;   // [B,512,128] (a backward buffer, free in forward)
  {
    const Img wfi = img_of(im, IM_SA2_WF);
    TRY(launch_sa2_pre(w.out1, false, 0, w.nx1, wfi.p, wfi.un, p.sa2_wx, rt, (long)B * M1, s));
  }
  if (use_side && hipStreamWaitEvent(s, sq->join, 0) != hipSuccess) return GEOA3_ELAUNCH;
  fork.disarm();
  // gather + shift + relu, W1, W2 + max in one kernel; the activations a0 / a1 exist only as gate bits (m0 / m1)
  {
    const Img i1 = img_of(im, IM_SA2_W1), i2 = img_of(im, IM_SA2_W2);
    geoa3_prof_begin(GEOA3_PROF_SA2_FWD, s);
    TRY(launch_sa2_fwd(rt, w.gidx2, w.shift, i1.p, i1.un, p.sa2_b1, i2.p, i2.un, p.sa2_b2, w.out2, w.arg2, w.m0, w.m1, B, M1,
                       M2, s));
    geoa3_prof_end(GEOA3_PROF_SA2_FWD, s);
  }
  // ---- level 3 (:78-82, GroupAll): MLP (256 + 3) -> 256 -> 512 -> 1024 on the 128 points, max over them
  TRY(conv_slice(w.out2, 256, 0, 256, img_of(im, IM_SA3_WF), nullptr, nullptr, w.h1, 256, B, M2, false, false, s));
  hipLaunchKernelGGL(affine3_add_kernel, g1d((long)B * 256 * M2), dim3(256), 0, s, w.h1, p.sa3_wx, p.sa3_b0, w.nx2, 256, M2,
                     1, (long)B * 256 * M2);
  TRY(conv_slice(w.h1, 256, 0, 256, img_of(im, IM_SA3_W1), p.sa3_b1, nullptr, w.h2, 512, B, M2, true, false, s));
  {   // 512 -> 1024 with the bias / relu / max over the 128 points in the epilogue (the [B,1024,128] tensor never exists)
    ConvArgs a{};
    a.split = 1;
    a.X = w.h2; a.sXb = (long)512 * M2; a.ldX = M2;
    const Img i2 = img_of(im, IM_SA3_W2);
    a.Wimg = i2.p; a.Wun = i2.un; a.img_kc = 32; a.img_c0 = 0;
    a.pool_out = w.p3; a.pool_arg = w.arg3; a.pool_bias = p.sa3_b2;
    a.Co = C3; a.K = 512; a.N = M2; a.B = B;
    a.pack2 = 1;
    TRY(launch_conv_cm(a, s));
  }
  // ---- FC head (:84-98): Linear (no bias) + BatchNorm1d (folded) + ReLU twice, Dropout = identity, Linear
  TRY(fc(w.p3, C3, p.f1, p.fb1, w.q1, 512, B, true, nullptr, s));
  TRY(fc(w.q1, 512, p.f2, p.fb2, w.q2, 256, B, true, nullptr, s));
  TRY(fc(w.q2, 256, p.f3, p.fb3, logits, p.classes, B, false, nullptr, s));
  hipLaunchKernelGGL(poison_rows_kernel, g1d((long)B * p.classes), dim3(256), 0, s, logits, w.bad, p.classes, (long)B * p.classes);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}

extern "C" int geoa3_pn2ssg_backward(const geoa3_pn2ssg_weights* pw, const float* x, const float* dlogits, int B, int N,
                                     float* dx, void* workspace, void* stream) {
  if (!pw || !x || !dlogits || !dx || !workspace || B <= 0 || N < MIN_N || N > MAX_N || pw->classes <= 0) return GEOA3_EINVAL;
  static_assert((size_t)MAX_N * 3 <= (size_t)C1 * M1, "gx2 borrows the forward's f1 buffer");
  if (((uintptr_t)workspace & 255) != 0 || !weights_complete(*pw)) return GEOA3_EINVAL;
  hipStream_t s = geoa3_stream(stream);
  const geoa3_pn2ssg_weights& p = *pw;
  Ws w = carve(workspace, B, N);
  const void* im = p.images ? p.images : w.images;   // the forward of this workspace built them
  // ---- FC head
  TRY(fc(dlogits, p.classes, p.f3t, nullptr, w.g256, 256, B, false, w.q2, s));
  TRY(fc(w.g256, 256, p.f2t, nullptr, w.g512, 512, B, false, w.q1, s));
  TRY(fc(w.g512, 512, p.f1t, nullptr, w.g1024, C3, B, false, w.p3, s));   // gated by the pooled layer's relu
  // ---- level 3: the pooled gradient reaches one point per channel: d h2 = relu'(h2) . sum_{ch : arg = point} g W2[ch]
  // (the sparse walk of the PointNet 1024-wide layers, in four 128-row slices of h2), then W1^T (two), Wf^T / Wx^T
  SideQueue* sq = static_cast<SideQueue*>(p.side);
  SideJoin fork;
  for (int k0 = 0; k0 < 512; k0 += 128) {
    WideBwdArgs a{};
    a.g = w.g1024; a.arg = w.arg3;
    a.W = p.sa3_w2 + k0; a.ldW = 512;
    a.Z = w.h2 + (size_t)k0 * M2; a.sZb = (long)512 * M2; a.ldZ = M2;
    a.dX = w.dh2 + (size_t)k0 * M2; a.sXb = (long)512 * M2; a.ldX = M2;
    a.Co = C3; a.N = M2; a.B = B; a.taps = 1;
    TRY(launch_wide_max_bwd(a, s));   // (two of the four independent slices on the side queue: 3.966 ms against 3.965)
  }
  for (int k0 = 0; k0 < 512; k0 += 256)
    TRY(conv_slice(w.dh2, 512, k0, 256, img_of(im, IM_SA3_W1T), nullptr, w.h1, w.dh1, 256, B, M2, false, k0 > 0, s));
  TRY(conv_slice(w.dh1, 256, 0, 256, img_of(im, IM_SA3_WFT), nullptr, nullptr, w.dout2, C2, B, M2, false, false, s));
  {   // level 3's coordinate gradient: first read by the side queue's kernels below, so it runs there too (in order)
    hipStream_t s3 = sq ? sq->stream : s;
    if (sq && (hipEventRecord(sq->ev[3], s) != hipSuccess || hipStreamWaitEvent(s3, sq->ev[3], 0) != hipSuccess)) return GEOA3_ELAUNCH;
    if (sq) fork.arm(sq, s);   // the side stream holds work of this call until the join at the end: every early return joins it
    hipLaunchKernelGGL(affine3_grad_kernel, g1d((long)B * M2), dim3(256), 3 * 256 * sizeof(float), s3, w.dh1, p.sa3_wx, 1.f,
                       w.dnx2, 256, M2, 0, (long)B * M2);
  }
  // ---- level 2: pooled layer's sparse gradient -> W2^T -> W1^T -> the grouping's scatter-add, in ONE pass over the rows
  // ordered by the point they gather from (pointnet2_sa2b.hip): the [B,128,128,64] grouped gradient never exists
  float* dr = w.r;          // [B][512][128] POINT-major
  void* sb = w.da0;         // the pass's scratch (0.4 MB per cloud)
  geoa3_prof_begin(GEOA3_PROF_SA2_GRAD, s);
  TRY(launch_sa2b_prep(w.dout2, w.out2, w.arg2, w.gidx2, sb, dr, B, s));
  geoa3_prof_end(GEOA3_PROF_SA2_GRAD, s);
  {
    const Img i1t = img_of(im, IM_SA2_W1T);
    geoa3_prof_begin(GEOA3_PROF_SA2_BWD, s);
    TRY(launch_sa2b_bwd(sb, w.m1, w.m0, p.sa2_w2, i1t.p, i1t.un, p.sa2_wx, dr, B, s));
    geoa3_prof_end(GEOA3_PROF_SA2_BWD, s);
  }
  {   // d f1 = W_f^T dr, written centroid-major for level 1's backward
    const Img wti = img_of(im, IM_SA2_WFT);
    TRY(launch_sa2_pre(dr, false, 0, nullptr, wti.p, wti.un, nullptr, w.g1, (long)B * M1, s));
  }
  // the coordinate gradients of the level (three small kernels, needed only behind level 1's backward) on the side queue
  // beside sa1_bwd_kernel
  const bool side_tail = sq != nullptr;   // (3.934 ms against 3.965)
  hipStream_t st = side_tail ? sq->stream : s;
  if (side_tail && (hipEventRecord(sq->ev[1], s) != hipSuccess || hipStreamWaitEvent(st, sq->ev[1], 0) != hipSuccess))
    return GEOA3_ELAUNCH;
  TRY(launch_affine3_grad_pm(dr, p.sa2_wx, w.dnx1, (long)B * M1, st));                                 // d xyz1 = W_x^T dr
  TRY(launch_sa2b_centre(sb, w.dnx2, B, st));                                                          // d c -= W_x^T d shift
  hipLaunchKernelGGL(scatter_rows3_kernel, dim3((M1 + 255) / 256, B), dim3(256), M2 * sizeof(int32_t), st, w.dnx2, w.idx2,
                     w.dnx1, M1, M2, 1);                                                               // gather(new_xyz1, idx2)
  // ---- level 1
  // (the [B,512,64,3] contributions go through the level-2 buffer d1, free by now)
  TRY(launch_sa1_backward(w.xyz, w.nx1, w.gidx1, &p.sa1, B, N, M1, w.out1, w.arg1, w.g1, w.gxyz, w.gnx1, w.d1,
                          side_tail ? sq->ev[2] : nullptr, stream));
  // the centroids' own gradient rows (d nx1 += level 1's share; scattered to their points) beside level 1's scatter kernel,
  // into a buffer of their own; the two are added on the way to the planar output
  float* gx2 = side_tail ? w.f1 : w.gxyz;   // (w.f1: [B,128,512] floats of the forward, free here)
  if (side_tail && hipStreamWaitEvent(st, sq->ev[2], 0) != hipSuccess) return GEOA3_ELAUNCH;
  hipLaunchKernelGGL(add_inplace_kernel, g1d((long)B * M1 * 3), dim3(256), 0, st, w.dnx1, w.gnx1, (long)B * M1 * 3);
  hipLaunchKernelGGL(scatter_rows3_kernel, dim3((N + 255) / 256, B), dim3(256), M1 * sizeof(int32_t), st, w.dnx1, w.idx1,
                     gx2, N, M1, side_tail ? 0 : 1);                                                   // gather(xyz, idx1)
  if (side_tail && (hipEventRecord(sq->join, st) != hipSuccess || hipStreamWaitEvent(s, sq->join, 0) != hipSuccess))
    return GEOA3_ELAUNCH;
  fork.disarm();
  hipLaunchKernelGGL(points_to_planar_kernel, g1d((long)B * N), dim3(256), 0, s, w.gxyz, side_tail ? gx2 : (const float*)nullptr, dx,
                     N, (long)B * N, w.bad);
  GEOA3_CHECK_LAUNCH();
  return GEOA3_OK;
}
