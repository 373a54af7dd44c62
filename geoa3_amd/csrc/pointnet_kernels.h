// Internal launch interface between pointnet.hip (orchestration) and the kernel translation units.
#pragma once
#include "common.h"

constexpr int GEOA3_DT_PITCH = 32;   // floats per row of the dT3 partial sums (ConvArgs::dTpart)

typedef float f32x16 __attribute__((ext_vector_type(16)));

// exact-fp32 matrix core op: D(32x32) += A(32x2) * B(2x32).  Lane l supplies A[l&31][l>>5] and
// B[l>>5][l&31]; D[i][j]: j = l&31, i = (reg&3) + 8*(reg>>2) + 4*(l>>5).
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ int mfma_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// Y[b][co][n] = epi( sum_k W[b][co][k] * X[b][k][n] ),  n contiguous ("channel-major" 1x1 convolution,
// per-instance transforms, and their transposed/backward forms).
struct ConvArgs {
  const float* X; long sXb; int ldX;
  const float* W; long sWb; int sWco, sWk;   // element (co,k) at W + b*sWb + co*sWco + k*sWk
  const float* bias;                          // [Co] or null
  const float* Z; long sZb; int ldZ;          // relu mask source (keep where Z > 0) or null
  // the same gate as one BIT per element, [B][ceil(N/64)][Co] 64-bit words (bit j of word (w, row) = column 64w + j): written
  // by the forward layer that produces the activation (Ymask, with relu), read by the backward layer instead of the
  // fp32 activation (Zmask): 32x fewer gate bytes
  const unsigned long long* Zmask;
  unsigned long long* Ymask;
  float* Y; long sYb; int ldY;
  int Co, K, N, B;
  int relu, accumulate;
  // First layer folded in (Model/PointNet.py:79,137-139: the 3-channel input layer behind the 3x3 input transform):
  //   h[k][n] = relu( w1[k][:] . (T[b]^T x[b][:,n]) + b1[k] ),  k < 64.
  // produce_first: the K = 64 input rows are COMPUTED from x3 instead of being read from X (X unused);
  // gate_first:    the relu mask of the Co = 64 output rows is h > 0, recomputed instead of being read from Z.
  const float* x3;                            // [B][3][N]
  const float* T3;                            // [B][9] or null (identity)
  const float* w1; const float* b1;           // [64][3], [64]
  int produce_first, gate_first;
  // with gate_first: finish the first layer's backward in the epilogue instead of writing Y (Model/PointNet.py:79,
  // 137-139 backward): q = w1^T (gated result), dx[b][d][n] (+)= sum_c T[d][c] q[c]  (`accumulate` selects +=), and,
  // when dTpart != null, per-workgroup partial sums of dT[d][c] = sum_n x[d][n] q[c][n] into dTpart[b][gridDim.x][GEOA3_DT_PITCH]
  // (9 values in a 128-byte row of their own: no two workgroups, i.e. no two XCDs' L2 caches, write to one cache line)
  float* dx3;                                 // [B][3][N] or null (= write Y as usual)
  float* dTpart;
  int split;                                  // 1: split-fp16 operands on the f16 matrix pipe (pointnet_conv_split.hip)
  int pack2;                                  // split kernel, N <= 128, shared weights: two instances per workgroup
  // PointNet++ shared-MLP forms (split kernel only; a wave's 64 columns = the 64 samples of ONE centre, N % 64 == 0):
  // pool_out: the output is max-pooled over each centre's samples in the epilogue instead of being written,
  //           pool_out[b][co][m] = relu(max_s y + pool_bias[co]), pool_arg = first maximal sample (Y unused);
  // oh_g:     the input is the pooled layer's sparse gradient, X[b][k][64 m + s] = (oh_arg[b][m][k] == s) ? oh_g[b][m][k]
  //           : 0 (oh_* CENTRE-major [B][centres][K]), formed in registers (X unused)
  float* pool_out; int32_t* pool_arg; const float* pool_bias;
  const float* oh_g; const int32_t* oh_arg;
  // split kernel, shared weights (sWb == 0), K = 128 / 256: the weights as a pre-split fragment image of the FULL matrix
  // [Co][16 img_kc] (geoa3_pn2ssg_pack_images: [row >> 5][k >> 4][piece][lane][8 halves]); this launch's K-slice starts
  // at k-step img_c0; Wun[0] = 1 / the image's power-of-two scale.  W is unused then.
  const void* Wimg; const float* Wun; int img_kc, img_c0;
};
int launch_conv_cm(const ConvArgs& a, hipStream_t s);         // dispatches on a.split

// A chain of 64-input layers with relu in one kernel (pointnet_conv_chain.hip; split arithmetic, the bits of the
// layer-by-layer launches): stage i reads the 64 rows stage i - 1 produced (stage 0: X [B][64][N], or, with x3, the folded
// first layer as in ConvArgs); the stages before the last have 64 outputs, the last 128.
struct ChainStage {
  const float* W; long sWb;                   // [Co][64], k contiguous; sWb: per-instance weights (0 = shared)
  const float* bias;                          // [Co]
  float* Y; long sYb;                         // [B][Co][N] or null: the activation is not written (gate bits only)
  unsigned long long* Ymask;                  // relu gate bits (ConvArgs::Ymask layout)
  int Co;
};
struct ConvChainArgs {
  const float* X; long sXb; int ldX;
  const float* x3; const float* T3; const float* w1; const float* b1;
  int N, B, ns;
  ChainStage st[3];
};
int launch_conv_chain(const ConvChainArgs& a, hipStream_t s);
// the backward counterpart (pointnet_conv_chain.hip): dh2 = gate( Wa[b]^T Xa + Wb^T Xb ), then conv_gate_first's form with
// W2t -- dx [B][3][N] written, dTpart [B][ceil(N/256)][GEOA3_DT_PITCH] partial sums; dh2 is never written
struct ConvBwdChainArgs {
  const float* Xa; const float* Wa; long sWa;   // [B][64][N]; [B][64 o][64 i] applied transposed (sWa: instance stride)
  const float* Xb; const float* Wb;             // [B][64][N]; [64 o][64 i] applied transposed
  const unsigned long long* Zmask;              // relu gate bits of the 64-channel activation the sum belongs to
  const float* W2t;                             // [64][64] k contiguous
  const float* x3; const float* T3; const float* w1; const float* b1;   // as ConvArgs (gate_first)
  float* dx; float* dTpart;
  int N, B;
};
int launch_conv_bwd_chain(const ConvBwdChainArgs& a, hipStream_t s);
int launch_conv_cm_split(const ConvArgs& a, hipStream_t s);

// Y[m][o] = epi( sum_k X[m][k] * W[o][k] + bias[o] )   (fully connected layers, both directions)
struct FcArgs {
  const float* X; int ldX;
  const float* W; int ldW;
  long sXb, sWb, sYb; int batch;              // optional batch (blockIdx.z): per-instance X, W, Y (0 / 1 = none)
  int ksplit;                                 // waves splitting K (0 = by K: one 64-k batch per wave, at most 16); never
                                              // a function of the batch size, so the summation order is fixed
  int tile;                                   // 16 / 32: output tile (0 = by the number of tiles)
  const float* bias;
  const float* Z; int ldZ;                    // relu mask source or null
  float* Y; int ldY;
  int M, Nout, K;
  int relu;
  float* kscratch;                            // optional: ceil(Nout/16) * ceil(M/16) * 4096 floats -- K >= 2048 is then split
                                              // over four workgroups per tile (same sums, same order)
};
int launch_fc(const FcArgs& a, hipStream_t s);

// out[b][co] = relu( max_n ( sum_{tap,ci} W[co][tap*Ci+ci] * X[b][ci][n+tap-(TAPS/2)] ) + bias[co] ),
// arg[b][co] = the maximising n.  Ci = 128; TAPS = 1 (T-Net conv3) or 3 (conv5, zero padded).
struct WideArgs {
  const float* X; long sXb; int ldX;          // [B][128][N]
  const float* W;                             // MFMA A-fragment order of [Co][TAPS*128] (pointnet_wide.hip)
  const float* bias;                          // [Co]
  float* out; int* arg;                       // [B][Co]
  unsigned long long* keys;                   // [B][Co] scratch of the column-major kernel (packed running maxima)
  int Co, N, B, taps;
  const void* Wh;                             // split-fp16 fragments (pointnet_wide_split.hip) or null = fp32 MFMA
  const void* Wh16;                           // the same weights as 16x16x32 fragments (pointnet_wide16.hip); takes precedence
  float unscale;                              // 1 / (power-of-two scale of Wh)
  int keys_clean;                             // 1: keys are already zero (wide_finalize_kernel leaves them zero): no memset
  int variant;                                // tuning variant of the split kernel (geoa3_debug_wide_fwd; 0 = shipped)
  // split kernel only: the layer in front of the wide one folded into its staging pass -- X[b][c][n] =
  // relu(W2 h + b2)[c][n] is computed per tile (X unused, never written) from the 64-channel activation h = Xin, or,
  // with x3, from h = relu(w1 x3 + b1); Ymask receives the relu gate of X as bits ([B][ceil(N/64)][128] 64-bit words)
  const void* W2h; float w2_unscale;          // pack_wide_split fragments of W2 [128][64] and 1 / their scale
  const float* W2f;                           // the same weights [128][64] fp32 (conv5's two halo points per tile)
  const float* b2;                            // [128]
  const float* Xin; long sXinb; int ldXin;    // [B][64][N]
  const float* x3; const float* w1; const float* b1;   // [B][3][N], [64][3], [64]
  unsigned long long* Ymask;
  unsigned long long* stamps;                 // diagnostics (tools/bench_wide.py --stamps): s_memtime trace of workgroup 0
  // with both (Co = 1024, N <= 4096): the finalize pass also builds the sparse backward's hit lists -- hits [B][Co taps]
  // = (co * taps + tap) | (column << 16) sorted by (column, chunk of 64 channels, tap, channel), hoff [B][N + 1] the
  // columns' start offsets (WideBwdArgs::hits / hoff)
  int* hits; int* hoff;
};
int launch_wide_max(const WideArgs& a, hipStream_t s);          // dispatches on a.Wh
int launch_wide_max_split(const WideArgs& a, hipStream_t s);    // pointnet_wide_split.hip
int launch_wide_max_split16(const WideArgs& a, hipStream_t s);  // pointnet_wide16.hip
void launch_wide_finalize(const WideArgs& a, hipStream_t s);    // keys -> out (bias + relu), arg

// packed running maximum of the 1024-wide layers: (order-preserving value bits, ~point index) under a 64-bit atomicMax
__device__ __forceinline__ unsigned long long wide_key(float v, int col) {
  unsigned u = __float_as_uint(v);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);            // order-preserving map of the float
  return ((unsigned long long)u << 32) | (unsigned)(0xFFFFFFFFu - (unsigned)col);   // ties: the LOWER point index wins
}

// dX[b][ci][m] = relu'(Z) * sum_{co,tap : arg[b][co]+tap-(TAPS/2) == m} W[co][tap*128+ci] * g[b][co]
struct WideBwdArgs {
  const float* g; const int* arg;             // [B][Co]
  const float* W;                             // [Co][TAPS*128]
  int ldW;                                    // elements between the 128-channel rows (co * TAPS + tap); 0 = 128.  A
                                              // [Co][K] matrix is walked in 128-column slices with ldW = K, W + k0
  const float* Z; long sZb; int ldZ;          // activation whose relu gates the result ([B][128][N]), or
  const unsigned long long* Zmask;            // its bit mask [B][ceil(N/64)][128] (ConvArgs::Ymask)
  float* dX; long sXb; int ldX;               // [B][128][N]
  int Co, N, B, taps;
  // launch_wide_bwd_conv (pointnet_wide_bwdconv.hip): the gated 128 -> 64 layer behind the sparse gradient in the same
  // kernel: dY[b][o][n] = gate2 . sum_ci W2t[o][ci] dX[b][ci][n] (dX stays in LDS; needs Zmask).  W2t [64][128];
  // Zmask2: relu bits of the 64-channel activation, [B][ceil(N/64)][64] words (ConvArgs::Ymask layout)
  const float* W2t; const unsigned long long* Zmask2;
  const void* W2th; float w2th_unscale;       // W2t as split-fp16 fragments (pack_wide_split: [row >> 5][k >> 4][hi / lo][lane][8])
                                              // and 1 / their scale: the A operands (required)
  float* dY; long sYb; int ldY;
  // ... or, with dx3: that layer is the one behind the 3-channel first layer (gate recomputed from x3 [B][3][N] with
  // w1 [64][3], b1 [64]) and the first layer's backward finishes in the same kernel: dx3[b][d][n] += sum_o w1[o][d] dY[o][n]
  const float* x3; const float* w1; const float* b1; float* dx3;
  const int* hits; const int* hoff;           // launch_wide_bwd_conv: the lists built by the forward's finalize pass (or null:
                                              // every workgroup builds its tile's lists from g / arg)
  int form;                                   // 0 = register accumulation over per-column lists (default), 1 = the first
                                              // form (LDS accumulation); same sums in the same order
};
int launch_wide_max_bwd(const WideBwdArgs& a, hipStream_t s);
int launch_wide_bwd_conv(const WideBwdArgs& a, hipStream_t s);

// P[b][i][o] = sum_n A[b][i][n] G[b][o][n], A and G [B][64][N], split-fp16 operands (pointnet_gram.hip)
int gram64_parts(int N);
int launch_gram64(const float* A, const float* G, int B, int N, float* P, float* scratch, hipStream_t s);

// dT[b][i] = sum_w part[b][w][i], i < 9, in a fixed order (the partial sums of ConvArgs::dTpart)
int launch_reduce_dT(const float* part, int nparts, float* dT, int B, hipStream_t s);


// PointNet++ level 2 (pointnet2_sa2.hip).  Weight matrices as split-fp16 FRAGMENT IMAGES: W[R][K] -> [row >> 5][k >> 4]
// [hi / lo][lane][8 halves], un[0] = 1 / the image's power-of-two scale (R * K * 4 bytes; one workgroup, run once per
// set of weights)
int launch_frag_image(const float* W, int R, int K, void* img, float* un, hipStream_t s);
// the level's forward in one kernel: out = relu(max_s W2 relu(W1 relu(rT[gidx] + shift) + b1) + b2), arg, gate bits m0 / m1
// per ROW: [B * M][64 samples][4 words of 32 channels]
int launch_sa2_fwd(const float* rT, const int32_t* gidx, const float* shift, const void* w1_img, const float* w1_un,
                   const float* b1, const void* w2_img, const float* w2_un, const float* b2, float* out, int32_t* arg,
                   unsigned* m0, unsigned* m1, int B, int N1, int M, hipStream_t s);

// PointNet++ level 1 in centroid ranges (pointnet2_net.hip pipelines the sampler's rounds against the MLP of the centroids
// it has already chosen): the sampler's rounds j0 .. j1 - 1 (resuming from `temp`), ball query and forward MLP of centroids
// m0 .. m1 - 1 of every cloud
int launch_pn2_fps_range(const float* xyz, int B, int N, int m, int j0, int j1, float* temp, int32_t* idx, hipStream_t s,
                         bool contract = false);
int launch_pn2_ball_query_range(const float* new_xyz, const float* xyz, int B, int N, int M, int m0, int m1, float radius,
                                int nsample, int32_t* idx, hipStream_t s, bool contract = false);
int launch_sa1_forward_range(const float* xyz, const float* new_xyz, const int32_t* idx, const geoa3_sa1_weights* w, int B, int N,
                             int M, int m0, int m1, float* out, uint8_t* arg, hipStream_t s);
// level 2's pre-transformed first layer per point: Y[p][co] = sum_k X[p][k] W[co][k] (+ sum_d Wx[co][d] xyz[p][d]), Y point-major;
// X point-major [P][128], or channel-major [P / Np][128][Np] (the backward's d f = W_f^T d r with W = the image of W_f^T)
int launch_sa2_pre(const float* X, bool x_channel_major, int Np, const float* xyz, const void* wf_img, const float* wf_un,
                   const float* Wx, float* Y, long P, hipStream_t s);
// level 2's backward with the rows in destination order (pointnet2_sa2b.hip): prep (rows, entries, gate words per cloud) ->
// the pass (d r [B][512][128] point-major, the rows' coordinate terms) -> d c of the centres; W_x^T of a point-major tensor
size_t sa2b_scratch_bytes(int B);
int launch_sa2b_prep(const float* dout, const float* outp, const int32_t* arg, const int32_t* gidx, void* scratch, float* dr, int B,
                     hipStream_t s);
int launch_sa2b_bwd(const void* scratch, const unsigned* m1, const unsigned* m0, const float* W2, const void* w1t_img,
                    const float* w1t_un, const float* Wx, float* dr, int B, hipStream_t s);
int launch_sa2b_centre(const void* scratch, float* dnx2, int B, hipStream_t s);
int launch_affine3_grad_pm(const float* dY, const float* Wx, float* dp, long points, hipStream_t s);
// geoa3_pn2_sa1_backward with an event recorded between its two kernels (grad_new_xyz complete, grad_xyz not yet)
int launch_sa1_backward(const float* xyz, const float* new_xyz, const int32_t* idx, const geoa3_sa1_weights* w, int B, int N, int M,
                        const float* out, const uint8_t* arg, const float* grad_out, float* grad_xyz, float* grad_new_xyz,
                        float* scratch, hipEvent_t after_bwd, void* stream);
