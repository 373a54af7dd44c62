// PointNet eval forward and input-gradient: launch sequence over the kernels of pointnet_*.hip.
// Reference: Model/PointNet.py:56-94 (transform_net) and :96-160 (PointNet.forward); the backward is the
// hand-derived input-gradient of that graph (weights never receive gradients; the reference's autograd
// also forms the unused weight gradients, Attacker/geoA3_attack.py:326).
#include <cstdlib>
#include "pointnet_kernels.h"

namespace {

struct Ws {  // workspace carve-up; every buffer starts on a 256-byte boundary
  // T-Net 3 (input transform)
  unsigned long long* keys;   // [B][1024] packed running maxima of the 1024-wide layers (reused by all three)
  float *a2, *p3, *tf4, *tf5, *T3;
  int* i3;
  // trunk + T-Net 64 (feature transform)
  float *h2, *c1, *c2, *q3, *qf4, *qf5, *T64;
  int* iq3;
  float *W3eff, *h3, *h4, *p5, *f6, *f7;   // W3eff [B][64][64] = W3 T64^T
  int* i5;
  // backward temporaries
  float *G128, *G64a, *P64, *dh2, *g1024, *g512, *g256, *gT64, *gT3, *dTpart;   // P64 [B][64][64] = h2 G64a^T
  // relu gates of the stored activations as bit masks [B][ceil(N/64)][C] x 64 bit (ConvArgs::Ymask / Zmask)
  unsigned long long *m_a2, *m_h2, *m_c1, *m_c2, *m_h3, *m_h4;
  // the sparse backward's hit lists, built by the forward's finalize passes (WideArgs::hits / hoff): lists [B][1024 taps],
  // column offsets [B][N + 1] -- T-Net 3, T-Net 64, conv5
  int *hl3, *ho3, *hlq, *hoq, *hl5, *ho5;
  size_t total;
};

Ws carve(void* base, int B, int N, int classes) {
  Ws w{};
  size_t off = 0;
  char* p = static_cast<char*>(base);
  auto take = [&](size_t nfloats) {
    void* r = p ? p + off : nullptr;
    off += ((nfloats * 4 + 255) / 256) * 256;
    return r;
  };
  const size_t s64 = (size_t)B * 64 * N, s128 = (size_t)B * 128 * N, b = (size_t)B;
  w.a2 = (float*)take(s128);
  w.keys = (unsigned long long*)take(b * 1024 * 2);
  w.p3 = (float*)take(b * 1024);
  w.i3 = (int*)take(b * 1024);
  w.tf4 = (float*)take(b * 512);
  w.tf5 = (float*)take(b * 256);
  w.T3 = (float*)take(b * 9);
  w.h2 = (float*)take(s64);
  w.c1 = (float*)take(s64);
  w.c2 = (float*)take(s128);
  w.q3 = (float*)take(b * 1024);
  w.iq3 = (int*)take(b * 1024);
  w.qf4 = (float*)take(b * 512);
  w.qf5 = (float*)take(b * 256);
  w.T64 = (float*)take(b * 4096);
  w.W3eff = (float*)take(b * 4096);
  w.h3 = (float*)take(s64);
  w.h4 = (float*)take(s128);
  w.p5 = (float*)take(b * 1024);
  w.i5 = (int*)take(b * 1024);
  w.f6 = (float*)take(b * 512);
  w.f7 = (float*)take(b * 256);
  w.G128 = (float*)take(s128);
  w.G64a = (float*)take(s64);
  w.P64 = (float*)take(b * 4096);
  w.dh2 = (float*)take(s64);
  w.g1024 = (float*)take(b * 1024);
  w.g512 = (float*)take(b * 512);
  w.g256 = (float*)take(b * 256);
  w.gT64 = (float*)take(b * 4096);
  w.gT3 = (float*)take(b * 16);
  w.dTpart = (float*)take(b * GEOA3_DT_PITCH * (size_t)((N + 255) / 256));
  const size_t n64 = (size_t)(N + 63) / 64;
  w.m_a2 = (unsigned long long*)take(b * 128 * n64 * 2);
  w.m_h2 = (unsigned long long*)take(b * 64 * n64 * 2);
  w.m_c1 = (unsigned long long*)take(b * 64 * n64 * 2);
  w.m_c2 = (unsigned long long*)take(b * 128 * n64 * 2);
  w.m_h3 = (unsigned long long*)take(b * 64 * n64 * 2);
  w.m_h4 = (unsigned long long*)take(b * 128 * n64 * 2);
  w.hl3 = (int*)take(b * 1024);
  w.ho3 = (int*)take(b * ((size_t)N + 1));
  w.hlq = (int*)take(b * 1024);
  w.hoq = (int*)take(b * ((size_t)N + 1));
  w.hl5 = (int*)take(b * 3072);
  w.ho5 = (int*)take(b * ((size_t)N + 1));
  w.total = off;
  (void)classes;
  return w;
}

// arithmetic of the narrow convolutions for the call in flight: 1 = split-fp16 operands (set from the weights struct:
// the callers that hand over split wide-layer fragments get the whole network on the f16 matrix pipe)
thread_local int tl_split = 0;

#define TRY(expr)             \
  do {                        \
    int rc__ = (expr);        \
    if (rc__ != 0) return rc__; \
  } while (0)

// geoa3_pointnet_weights.flags (A/B switches, set by the caller: the library reads no environment):
// GEOA3_PN_NO_FUSE_BWD: the sparse backward and the 128 -> 64 layer behind it as two kernels;
// GEOA3_PN_NO_CHAIN: the trunk's 64-input layers as one kernel each instead of chains.  Same bits either way.
// GEOA3_PN_KEYS_CLEAN: see include/geoa3_hip.h.
thread_local int tl_flags = 0;
bool fuse_bwd() { return !(tl_flags & GEOA3_PN_NO_FUSE_BWD); }
bool fuse_chain() { return !(tl_flags & GEOA3_PN_NO_CHAIN); }
// the sparse backward's hit lists come from the forward's finalize pass (GEOA3_PN_NO_PRE_LISTS: every backward workgroup
// builds its own, as before round 4; same bits)
bool pre_lists(int N) { return tl_split && fuse_bwd() && N <= 4096 && !(tl_flags & GEOA3_PN_NO_PRE_LISTS); }

// Y = act(W X + bias) over [B][K][N] -> [B][Co][N]; shared weights [Co][K]
int conv(const float* X, int K, const float* W, const float* bias, float* Y, int Co, int B, int N, bool relu,
         const float* Z, bool accumulate, hipStream_t s, unsigned long long* Ymask = nullptr,
         const unsigned long long* Zmask = nullptr) {
  ConvArgs a{};
  a.split = tl_split;
  a.Ymask = Ymask; a.Zmask = Zmask;
  a.X = X; a.sXb = (long)K * N; a.ldX = N;
  a.W = W; a.sWb = 0; a.sWco = K; a.sWk = 1;
  a.bias = bias;
  a.Z = Z; a.sZb = (long)Co * N; a.ldZ = N;
  a.Y = Y; a.sYb = (long)Co * N; a.ldY = N;
  a.Co = Co; a.K = K; a.N = N; a.B = B;
  a.relu = relu; a.accumulate = accumulate;
  return launch_conv_cm(a, s);
}

// Y = act(W relu(w1 (T^T x) + b1) + bias): the 3-channel first layer (Model/PointNet.py:79,137-139) folded into the
// 64-input convolution that follows it; its activation is never written
int conv_first(const float* x, const float* T, const float* w1, const float* b1, const float* W, const float* bias,
               float* Y, int Co, int B, int N, hipStream_t s, unsigned long long* Ymask) {
  ConvArgs a{};
  a.split = tl_split;
  a.Ymask = Ymask;
  a.x3 = x; a.T3 = T; a.w1 = w1; a.b1 = b1; a.produce_first = 1;
  a.W = W; a.sWb = 0; a.sWco = 64; a.sWk = 1;
  a.bias = bias;
  a.Y = Y; a.sYb = (long)Co * N; a.ldY = N;
  a.Co = Co; a.K = 64; a.N = N; a.B = B;
  a.relu = 1;
  return launch_conv_cm(a, s);
}

// G64 = relu'(first layer) * (W X): the input-gradient convolution whose mask is the first layer's sign, recomputed from x
// ... and, with dx != null, the first layer's own backward in the same kernel (no G64 in memory): dx (+)= T w1^T G64,
// dTpart = per-workgroup partial sums of d/dT
int conv_gate_first(const float* X, int K, const float* W, float* Y, const float* x, const float* T, const float* w1,
                    const float* b1, int B, int N, hipStream_t s, float* dx = nullptr, int accumulate = 0,
                    float* dTpart = nullptr) {
  ConvArgs a{};
  a.split = tl_split;
  a.dx3 = dx; a.accumulate = accumulate; a.dTpart = dTpart;
  a.X = X; a.sXb = (long)K * N; a.ldX = N;
  a.W = W; a.sWb = 0; a.sWco = K; a.sWk = 1;
  a.x3 = x; a.T3 = T; a.w1 = w1; a.b1 = b1; a.gate_first = 1;
  a.Y = Y; a.sYb = (long)64 * N; a.ldY = N;
  a.Co = 64; a.K = K; a.N = N; a.B = B;
  return launch_conv_cm(a, s);
}

int fc(const float* X, int K, const float* W, const float* bias, float* Y, int Nout, int M, bool relu, const float* Z,
       hipStream_t s, float* kscratch = nullptr) {
  FcArgs a{};
  a.kscratch = kscratch;
  a.X = X; a.ldX = K;
  a.W = W; a.ldW = K;
  a.bias = bias;
  a.Z = Z; a.ldZ = Nout;
  a.Y = Y; a.ldY = Nout;
  a.M = M; a.Nout = Nout; a.K = K; a.relu = relu;
  return launch_fc(a, s);
}

// fused: the 64 -> 128 layer in front (weights w2 / fragments w2h / bias b2) is evaluated in the wide kernel's staging pass
// from h64 [B][64][N] (or, with x3, from relu(w1 x3 + b1)); its relu gate goes to m128, its activation nowhere
struct FrontLayer {
  const void* w2h; float w2_unscale; const float* w2; const float* b2;
  const float* h64; const float* x3; const float* w1; const float* b1;
  unsigned long long* m128;
};
int wide(const float* X, const float* W, const void* Wh, float unscale, const float* bias, float* out, int* arg,
         unsigned long long* keys, int taps, int B, int N, hipStream_t s, const FrontLayer* f = nullptr,
         const void* Wh16 = nullptr, int* hits = nullptr, int* hoff = nullptr) {
  WideArgs a{};
  a.Wh16 = Wh16;
  if (pre_lists(N)) { a.hits = hits; a.hoff = hoff; }
  if (f) {
    a.W2h = f->w2h; a.w2_unscale = f->w2_unscale; a.W2f = f->w2; a.b2 = f->b2;
    a.Xin = f->h64; a.sXinb = (long)64 * N; a.ldXin = N;
    a.x3 = f->x3; a.w1 = f->w1; a.b1 = f->b1;
    a.Ymask = f->m128;
  }
  a.keys = keys;
  a.keys_clean = 1;   // zeroed once per forward; every finalize leaves them zero
  a.Wh = Wh; a.unscale = unscale;
  a.X = X; a.sXb = (long)128 * N; a.ldX = N;
  a.W = W; a.bias = bias; a.out = out; a.arg = arg;
  a.Co = 1024; a.N = N; a.B = B; a.taps = taps;
  return launch_wide_max(a, s);
}

int wide_bwd(const float* g, const int* arg, const float* W, const float* Z, const unsigned long long* Zmask, float* dX,
             int taps, int B, int N, hipStream_t s) {
  WideBwdArgs a{};
  a.Zmask = Zmask;
  a.g = g; a.arg = arg; a.W = W;
  a.Z = Z; a.sZb = (long)128 * N; a.ldZ = N;
  a.dX = dX; a.sXb = (long)128 * N; a.ldX = N;
  a.Co = 1024; a.N = N; a.B = B; a.taps = taps;
  return launch_wide_max_bwd(a, s);
}

// sparse backward of a 1024-wide layer + the gated 128 -> 64 layer behind it in one kernel (split mode: bit gates)
int wide_bwd_conv(const float* g, const int* arg, const float* W, const unsigned long long* Zmask, const float* W2t,
                  const unsigned long long* Zmask2, float* dY, int taps, int B, int N, hipStream_t s,
                  const int* hits, const int* hoff, const void* W2th, float w2th_unscale) {
  WideBwdArgs a{};
  a.W2th = W2th; a.w2th_unscale = w2th_unscale;
  if (pre_lists(N)) { a.hits = hits; a.hoff = hoff; }
  a.Zmask = Zmask;
  a.g = g; a.arg = arg; a.W = W;
  a.W2t = W2t; a.Zmask2 = Zmask2;
  a.dY = dY; a.sYb = (long)64 * N; a.ldY = N;
  a.Co = 1024; a.N = N; a.B = B; a.taps = taps;
  return launch_wide_bwd_conv(a, s);
}

// transform_net.forward (Model/PointNet.py:78-87) after its first layer
// act64 == nullptr: the T-Net reads the cloud itself (K = 3) and its first layer is folded into conv2 (x3 given)
int tnet_tail_fwd(const geoa3_tnet_weights& t, const float* act64, const float* x3, float* act128,
                  unsigned long long* m128, float* pooled,
                  int* arg, float* f4, float* f5, float* T, unsigned long long* keys, int B, int N, hipStream_t s,
                  int* hits, int* hoff, bool have128 = false) {
  if (have128) {   // act128 (and m128) already produced by the trunk's chain kernel
    TRY(wide(act128, t.w3p, t.w3h, t.w3h_unscale, t.b3, pooled, arg, keys, 1, B, N, s, nullptr, t.w3h16, hits, hoff));
  } else if (tl_split && t.w2h) {   // conv2 (behind conv1 for the 3-channel T-Net) inside the wide kernel: act128 is never written
    FrontLayer f{t.w2h, t.w2h_unscale, t.w2, t.b2, act64, act64 ? nullptr : x3, t.w1, t.b1, m128};
    TRY(wide(nullptr, t.w3p, t.w3h, t.w3h_unscale, t.b3, pooled, arg, keys, 1, B, N, s, &f, nullptr, hits, hoff));
  } else {
    if (act64) TRY(conv(act64, 64, t.w2, t.b2, act128, 128, B, N, true, nullptr, false, s, m128));
    else if (tl_split && fuse_chain()) {   // conv1 + conv2 of the 3-channel T-Net: the chain kernel with one stage
      ConvChainArgs a{};
      a.x3 = x3; a.w1 = t.w1; a.b1 = t.b1;
      a.N = N; a.B = B; a.ns = 1;
      a.st[0] = ChainStage{t.w2, 0, t.b2, act128, (long)128 * N, m128, 128};
      TRY(launch_conv_chain(a, s));
    } else TRY(conv_first(x3, nullptr, t.w1, t.b1, t.w2, t.b2, act128, 128, B, N, s, m128));
    TRY(wide(act128, t.w3p, t.w3h, t.w3h_unscale, t.b3, pooled, arg, keys, 1, B, N, s, nullptr, t.w3h16, hits, hoff));
  }
  TRY(fc(pooled, 1024, t.f1, t.fb1, f4, 512, B, true, nullptr, s));
  TRY(fc(f4, 512, t.f2, t.fb2, f5, 256, B, true, nullptr, s));
  TRY(fc(f5, 256, t.f3, t.fb3, T, t.K * t.K, B, false, nullptr, s));
  return 0;
}

// d/d(transform) [B][K*K] -> gradient w.r.t. the pre-activation of the T-Net's first layer (G64 out)
int tnet_bwd(const geoa3_tnet_weights& t, const float* gT, const float* act64, const unsigned long long* m64,
             const float* x3, const float* act128, const unsigned long long* m128,
             const float* pooled, const int* arg, const float* f4, const float* f5, Ws& w, float* G64out, int B, int N,
             hipStream_t s, const int* hits, const int* hoff) {
  // (K = 4096 for the feature transform: split over four workgroups per tile through G128, which is written only below,
  //  when its B * 128 * N floats hold the ceil(B / 16) * 16 tiles x 4096 partial values)
  const bool ks = (size_t)((B + 15) / 16) * 16 * 4096 <= (size_t)B * 128 * N;
  TRY(fc(gT, t.K * t.K, t.f3t, nullptr, w.g256, 256, B, false, f5, s, ks ? w.G128 : nullptr));
  TRY(fc(w.g256, 256, t.f2t, nullptr, w.g512, 512, B, false, f4, s));
  TRY(fc(w.g512, 512, t.f1t, nullptr, w.g1024, 1024, B, false, pooled, s));
  if (act64 && tl_split && m128 && m64 && fuse_bwd() && t.w2th) {
    TRY(wide_bwd_conv(w.g1024, arg, t.w3, m128, t.w2t, m64, G64out, 1, B, N, s, hits, hoff, t.w2th, t.w2th_unscale));
    return 0;
  }
  if (!act64 && tl_split && m128 && fuse_bwd() && t.w2th) {   // the 3-channel T-Net: ... and its first layer's backward: dx += w1^T (..)
    WideBwdArgs a{};
    if (pre_lists(N)) { a.hits = hits; a.hoff = hoff; }
    a.Zmask = m128;
    a.g = w.g1024; a.arg = arg; a.W = t.w3;
    a.W2th = t.w2th; a.w2th_unscale = t.w2th_unscale;
    a.x3 = x3; a.w1 = t.w1; a.b1 = t.b1; a.dx3 = G64out;
    a.Co = 1024; a.N = N; a.B = B; a.taps = 1;
    return launch_wide_bwd_conv(a, s);
  }
  TRY(wide_bwd(w.g1024, arg, t.w3, act128, m128, w.G128, 1, B, N, s));
  if (act64) TRY(conv(w.G128, 128, t.w2t, nullptr, G64out, 64, B, N, false, nullptr, false, s, nullptr, m64));
  else TRY(conv_gate_first(w.G128, 128, t.w2t, nullptr, x3, nullptr, t.w1, t.b1, B, N, s, G64out /* = dx */, 1));
  return 0;
}

}  // namespace

extern "C" int64_t geoa3_pointnet_workspace_bytes(int B, int N, int classes) {
  if (B <= 0 || N <= 0 || classes <= 0) return -1;
  return (int64_t)carve(nullptr, B, N, classes).total;
}

// diagnostics (geoa3_hip_debug.h): names and byte offsets of the workspace's buffers, in address order
extern "C" int geoa3_debug_pointnet_workspace_layout(int B, int N, int classes, const char** names, int64_t* offsets,
                                                     int cap) {
  if (B <= 0 || N <= 0 || classes <= 0) return -1;
  char* const base = reinterpret_cast<char*>(static_cast<uintptr_t>(1) << 40);
  const Ws w = carve(base, B, N, classes);
  const struct { const char* name; const void* p; } f[] = {
      {"a2", w.a2}, {"keys", w.keys}, {"p3", w.p3}, {"i3", w.i3}, {"tf4", w.tf4}, {"tf5", w.tf5}, {"T3", w.T3},
      {"h2", w.h2}, {"c1", w.c1}, {"c2", w.c2}, {"q3", w.q3}, {"iq3", w.iq3}, {"qf4", w.qf4}, {"qf5", w.qf5},
      {"T64", w.T64}, {"W3eff", w.W3eff}, {"h3", w.h3}, {"h4", w.h4}, {"p5", w.p5}, {"i5", w.i5}, {"f6", w.f6},
      {"f7", w.f7}, {"G128", w.G128}, {"G64a", w.G64a}, {"P64", w.P64}, {"dh2", w.dh2}, {"g1024", w.g1024},
      {"g512", w.g512}, {"g256", w.g256}, {"gT64", w.gT64}, {"gT3", w.gT3}, {"dTpart", w.dTpart}, {"m_a2", w.m_a2},
      {"m_h2", w.m_h2}, {"m_c1", w.m_c1}, {"m_c2", w.m_c2}, {"m_h3", w.m_h3}, {"m_h4", w.m_h4},
      {"hl3", w.hl3}, {"ho3", w.ho3}, {"hlq", w.hlq}, {"hoq", w.hoq}, {"hl5", w.hl5}, {"ho5", w.ho5}};
  const int n = (int)(sizeof(f) / sizeof(f[0]));
  for (int i = 0; i < n && i < cap; ++i) {
    names[i] = f[i].name;
    offsets[i] = (int64_t)(static_cast<const char*>(f[i].p) - base);
  }
  return n;
}

extern "C" int geoa3_pointnet_forward(const geoa3_pointnet_weights* pw, const float* x, int B, int N, float* logits,
                                      void* workspace, void* stream) {
  if (!pw || !x || !logits || !workspace || B <= 0 || N <= 0) return GEOA3_EINVAL;
  if (pw->t3.K != 3 || pw->t64.K != 64 || pw->classes <= 0) return GEOA3_EINVAL;
  if (((uintptr_t)workspace & 255) != 0) return GEOA3_EINVAL;
  hipStream_t s = geoa3_stream(stream);
  Ws w = carve(workspace, B, N, pw->classes);
  const geoa3_pointnet_weights& p = *pw;
  tl_split = p.w5h != nullptr;
  tl_flags = p.flags;
  if (!(p.flags & GEOA3_PN_KEYS_CLEAN) &&
      hipMemsetAsync(w.keys, 0, (size_t)B * 1024 * sizeof(unsigned long long), s) != hipSuccess)
    return GEOA3_ELAUNCH;
  // input transform (Model/PointNet.py:137-138)
  TRY(tnet_tail_fwd(p.t3, nullptr, x, w.a2, w.m_a2, w.p3, w.i3, w.tf4, w.tf5, w.T3, w.keys, B, N, s, w.hl3, w.ho3));
  const bool chain = tl_split && fuse_chain();
  if (chain && !p.t64.w2h) {
    // trunk conv1, conv2 (:139-140) and the feature transform's conv1, conv2 (:78-80) in one kernel: h2 is written (the
    // backward and conv3 read it), c1 exists only as gate bits, c2 is the T-Net's 1024-wide layer's input
    ConvChainArgs a{};
    a.x3 = x; a.T3 = w.T3; a.w1 = p.w1; a.b1 = p.b1;
    a.N = N; a.B = B; a.ns = 3;
    a.st[0] = ChainStage{p.w2, 0, p.b2, w.h2, (long)64 * N, w.m_h2, 64};
    a.st[1] = ChainStage{p.t64.w1, 0, p.t64.b1, nullptr, 0, w.m_c1, 64};
    a.st[2] = ChainStage{p.t64.w2, 0, p.t64.b2, w.c2, (long)128 * N, w.m_c2, 128};
    TRY(launch_conv_chain(a, s));
    TRY(tnet_tail_fwd(p.t64, w.c1, nullptr, w.c2, w.m_c2, w.q3, w.iq3, w.qf4, w.qf5, w.T64, w.keys, B, N, s, w.hlq, w.hoq, true));
  } else {
    // trunk conv1, conv2 (:139-140)
    TRY(conv_first(x, w.T3, p.w1, p.b1, p.w2, p.b2, w.h2, 64, B, N, s, w.m_h2));
    // feature transform (:142-143)
    TRY(conv(w.h2, 64, p.t64.w1, p.t64.b1, w.c1, 64, B, N, true, nullptr, false, s, w.m_c1));
    TRY(tnet_tail_fwd(p.t64, w.c1, nullptr, w.c2, w.m_c2, w.q3, w.iq3, w.qf4, w.qf5, w.T64, w.keys, B, N, s, w.hlq, w.hoq));
  }
  // feature transform folded into conv3 (:143-144): W3 (T64^T h2) = (W3 T64^T) h2 -- one 64^3 product per instance
  // instead of a pass over [B,64,N]
  {
    FcArgs g{};   // W3eff[b][o][i] = sum_j W3[o][j] T64[b][i][j]
    g.X = p.w3; g.ldX = 64; g.sXb = 0;
    g.W = w.T64; g.ldW = 64; g.sWb = 4096;
    g.Y = w.W3eff; g.ldY = 64; g.sYb = 4096;
    g.M = 64; g.Nout = 64; g.K = 64; g.batch = B;
    TRY(launch_fc(g, s));
  }
  const bool chain34 = chain && !p.w4h;
  if (chain34) {   // conv3, conv4 in one kernel: h3 exists only as gate bits
    ConvChainArgs a{};
    a.X = w.h2; a.sXb = (long)64 * N; a.ldX = N;
    a.N = N; a.B = B; a.ns = 2;
    a.st[0] = ChainStage{w.W3eff, 4096, p.b3, nullptr, 0, w.m_h3, 64};
    a.st[1] = ChainStage{p.w4, 0, p.b4, w.h4, (long)128 * N, w.m_h4, 128};
    TRY(launch_conv_chain(a, s));
  } else {
    ConvArgs a{};
    a.split = tl_split;
    a.X = w.h2; a.sXb = (long)64 * N; a.ldX = N;
    a.W = w.W3eff; a.sWb = 4096; a.sWco = 64; a.sWk = 1;
    a.bias = p.b3;
    a.Y = w.h3; a.sYb = (long)64 * N; a.ldY = N;
    a.Ymask = w.m_h3;
    a.Co = 64; a.K = 64; a.N = N; a.B = B; a.relu = 1;
    TRY(launch_conv_cm(a, s));
  }
  // conv4, conv5 + max (:145-147)
  if (chain34) {
    TRY(wide(w.h4, p.w5p, p.w5h, p.w5h_unscale, p.b5, w.p5, w.i5, w.keys, 3, B, N, s, nullptr, p.w5h16, w.hl5, w.ho5));
  } else if (tl_split && p.w4h) {   // conv4 inside conv5's staging pass: h4 is never written
    FrontLayer f{p.w4h, p.w4h_unscale, p.w4, p.b4, w.h3, nullptr, nullptr, nullptr, w.m_h4};
    TRY(wide(nullptr, p.w5p, p.w5h, p.w5h_unscale, p.b5, w.p5, w.i5, w.keys, 3, B, N, s, &f, nullptr, w.hl5, w.ho5));
  } else {
    TRY(conv(w.h3, 64, p.w4, p.b4, w.h4, 128, B, N, true, nullptr, false, s, w.m_h4));
    TRY(wide(w.h4, p.w5p, p.w5h, p.w5h_unscale, p.b5, w.p5, w.i5, w.keys, 3, B, N, s, nullptr, p.w5h16, w.hl5, w.ho5));
  }
  // classifier head (:150-152), dropout is the identity in eval mode
  TRY(fc(w.p5, 1024, p.f1, p.fb1, w.f6, 512, B, true, nullptr, s));
  TRY(fc(w.f6, 512, p.f2, p.fb2, w.f7, 256, B, true, nullptr, s));
  TRY(fc(w.f7, 256, p.f3, p.fb3, logits, p.classes, B, false, nullptr, s));
  return GEOA3_OK;
}

extern "C" int geoa3_pointnet_backward(const geoa3_pointnet_weights* pw, const float* x, const float* dlogits, int B,
                                       int N, float* dx, void* workspace, void* stream) {
  if (!pw || !x || !dlogits || !dx || !workspace || B <= 0 || N <= 0) return GEOA3_EINVAL;
  hipStream_t s = geoa3_stream(stream);
  Ws w = carve(workspace, B, N, pw->classes);
  const geoa3_pointnet_weights& p = *pw;
  tl_split = p.w5h != nullptr;
  tl_flags = p.flags;
  // classifier head
  TRY(fc(dlogits, p.classes, p.f3t, nullptr, w.g256, 256, B, false, w.f7, s));
  TRY(fc(w.g256, 256, p.f2t, nullptr, w.g512, 512, B, false, w.f6, s));
  TRY(fc(w.g512, 512, p.f1t, nullptr, w.g1024, 1024, B, false, w.p5, s));
  // max + conv5 (sparse), conv4, conv3
  if (tl_split && fuse_bwd() && p.w4th) {
    TRY(wide_bwd_conv(w.g1024, w.i5, p.w5, w.m_h4, p.w4t, w.m_h3, w.G64a, 3, B, N, s, w.hl5, w.ho5, p.w4th, p.w4th_unscale));
  } else {
    TRY(wide_bwd(w.g1024, w.i5, p.w5, w.h4, w.m_h4, w.G128, 3, B, N, s));
    TRY(conv(w.G128, 128, p.w4t, nullptr, w.G64a, 64, B, N, false, nullptr, false, s, nullptr, w.m_h3));
  }
  // conv3 + feature transform, merged as in forward (G64a = d/d(pre-activation of h3)):
  //   dT64[b][i][j] = sum_n h2[i][n] (W3^T G64a)[j][n] = ((h2 G64a^T) W3)[i][j];   dh2 = T64 W3^T G64a = W3eff^T G64a
  {
    FcArgs g{};   // P[b][i][o] = sum_n h2[b][i][n] G64a[b][o][n]: the batched NT product of the FC kernel (K = N)
    g.X = w.h2; g.ldX = N; g.sXb = (long)64 * N;
    g.W = w.G64a; g.ldW = N; g.sWb = (long)64 * N;
    g.Y = w.P64; g.ldY = 64; g.sYb = 4096;
    g.M = 64; g.Nout = 64; g.K = N; g.batch = B;
    // f16 matrix pipe; partial sums of the four workgroups per instance go through dh2 (written only afterwards:
    // 8 * 4096 floats per instance fit its 64 * N from N = 512 on, and gram64_parts(N) is at most 4 below N = 897)
    if (tl_split) TRY(launch_gram64(w.h2, w.G64a, B, N, w.P64, w.dh2, s));
    else TRY(launch_fc(g, s));
    FcArgs q{};   // dT64[b][i][j] = sum_o P[b][i][o] W3[o][j]
    q.X = w.P64; q.ldX = 64; q.sXb = 4096;
    q.W = p.w3t; q.ldW = 64; q.sWb = 0;
    q.Y = w.gT64; q.ldY = 64; q.sYb = 4096;
    q.M = 64; q.Nout = 64; q.K = 64; q.batch = B;
    TRY(launch_fc(q, s));
    if (!(tl_split && fuse_chain())) {
    ConvArgs a{};   // dh2[b][i][n] = sum_o W3eff[b][o][i] G64a[b][o][n]
    a.split = tl_split;
    a.X = w.G64a; a.sXb = (long)64 * N; a.ldX = N;
    a.W = w.W3eff; a.sWb = 4096; a.sWco = 1; a.sWk = 64;
    a.Y = w.dh2; a.sYb = (long)64 * N; a.ldY = N;
    a.Co = 64; a.K = 64; a.N = N; a.B = B;
    TRY(launch_conv_cm(a, s));
    }
  }
  const int nparts = (N + 255) / 256;
  if (tl_split && fuse_chain()) {
    // the T-Net branch's gradient goes to the dh2 buffer (the Gram kernel's scratch is consumed by now); then ONE kernel:
    // dh2 = gate_h2(W3eff^T G64a + W_t64.conv1^T G), conv2's backward, trunk conv1 + input transform backward (dx, dT3
    // partials) -- dh2 is never written
    TRY(tnet_bwd(p.t64, w.gT64, w.c1, w.m_c1, nullptr, w.c2, w.m_c2, w.q3, w.iq3, w.qf4, w.qf5, w, w.dh2, B, N, s, w.hlq, w.hoq));
    ConvBwdChainArgs a{};
    a.Xa = w.G64a; a.Wa = w.W3eff; a.sWa = 4096;
    a.Xb = w.dh2; a.Wb = p.t64.w1;
    a.Zmask = w.m_h2;
    a.W2t = p.w2t;
    a.x3 = x; a.T3 = w.T3; a.w1 = p.w1; a.b1 = p.b1;
    a.dx = dx; a.dTpart = w.dTpart;
    a.N = N; a.B = B;
    TRY(launch_conv_bwd_chain(a, s));
  } else {
  TRY(tnet_bwd(p.t64, w.gT64, w.c1, w.m_c1, nullptr, w.c2, w.m_c2, w.q3, w.iq3, w.qf4, w.qf5, w, w.G64a, B, N, s, w.hlq, w.hoq));
  // dh2 += W_t64.conv1^T G64a, then the relu gate of h2
  {
    ConvArgs a{};
    a.split = tl_split;
    a.X = w.G64a; a.sXb = (long)64 * N; a.ldX = N;
    a.W = p.t64.w1; a.sWb = 0; a.sWco = 1; a.sWk = 64;  // W^T through the strided loader
    a.Zmask = w.m_h2;
    a.Y = w.dh2; a.sYb = (long)64 * N; a.ldY = N;
    a.Co = 64; a.K = 64; a.N = N; a.B = B; a.accumulate = 1;
    TRY(launch_conv_cm(a, s));
  }
  // conv2 backward fused with trunk conv1 + input transform backward: dx, dT3 (partials per 256-point workgroup)
  TRY(conv_gate_first(w.dh2, 64, p.w2t, nullptr, x, w.T3, p.w1, p.b1, B, N, s, dx, 0, w.dTpart));
  }
  TRY(launch_reduce_dT(w.dTpart, nparts, w.gT3, B, s));
  // T-Net(3) backward; its last kernel adds the T-Net branch into dx
  TRY(tnet_bwd(p.t3, w.gT3, nullptr, nullptr, x, w.a2, w.m_a2, w.p3, w.i3, w.tf4, w.tf5, w, dx, B, N, s, w.hl3, w.ho3));
  return GEOA3_OK;
}
