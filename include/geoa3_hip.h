/*
 * geoa3_hip.h -- C ABI of libgeoa3_hip.so: the MI355X (gfx950) implementation of the GeoA3
 * inner attack loop (Gorilla-Lab-SCUT/GeoA3: Attacker/geoA3_attack.py + Lib/loss_utils.py +
 * Model/PointNet.py).  Plain pointers and sizes only; no torch types.
 *
 * Conventions:
 *   - every pointer is a DEVICE pointer unless the name ends in _host;
 *   - point tensors are planar fp32 [B,3,N] (the reference's logical layout): plane c of
 *     instance b starts at base + (b*3 + c)*N, so a wavefront reads 64 consecutive floats;
 *   - indices are int32; distances are squared L2, evaluated un-fused as
 *       d = fl(fl(fl(dx*dx)+fl(dy*dy))+fl(dz*dz))   (bit-identical to the CPU oracle);
 *     exact distance ties go to the LOWER index;
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream); every entry
 *     point only enqueues work: no allocation, no host synchronisation, no global state
 *     (the opt-in diagnostics of geoa3_hip_debug.h -- event timers, single-kernel entry points for the
 *     tools/ benchmarks -- are declared apart from this ABI and never change results);
 *   - return value: 0 on success, a negative GEOA3_E* code otherwise (geoa3_strerror()).
 *     The reference prints and exit(-1)s on a launch failure
 *     (Model/pointnet2_ops_lib/pointnet2_ops/_ext-src/include/cuda_utils.h:30-39); callers of
 *     this library are expected to raise instead.
 *
 * Each declaration cites the reference interface it replaces (file:line under the reference
 * repository).  INTEGRATION.md shows the ctypes binding a maintainer of the reference adds.
 */
#ifndef GEOA3_HIP_H
#define GEOA3_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GEOA3_OK 0
#define GEOA3_EINVAL (-1)   /* bad argument (null pointer, size out of the supported range) */
#define GEOA3_ELAUNCH (-2)  /* hipLaunchKernel / hipGetLastError reported a failure */
#define GEOA3_ENOSUPPORT (-3)

/* The ABI version of this header: bumped on EVERY change of a struct layout or a signature.  geoa3_version() returns the
 * value the library was built with; a binding must refuse a library whose version differs (geoa3_amd/_lib.py does: a
 * stale or variant .so would misread the argument structs silently). */
#define GEOA3_ABI_VERSION 600
int geoa3_version(void);
const char* geoa3_strerror(int code);

/* ------------------------------------------------------------------------------------------
 * Operator level: pytorch3d.ops.knn_points as called from
 * Lib/loss_utils.py:32,33,41,48,57,70,77,92 and Attacker/geoA3_attack.py:65,80.
 * ------------------------------------------------------------------------------------------ */

/* K = 1, both directions in ONE launch ("CD kernel"): for every point of `a` its nearest
 * point of `r` (d_ar, i_ar: [B,Na]) and for every point of `r` its nearest point of `a`
 * (d_ra, i_ra: [B,Nr]).  Replaces the knn_points(K=1) pairs at Lib/loss_utils.py:32-33 (and the
 * repeated adv->ori queries at :48,:70,:92).  d_ra/i_ra may be NULL (one direction only,
 * pseudo_chamfer_loss, Lib/loss_utils.py:41). */
int geoa3_nn1_pair(const float* a, const float* r, int B, int Na, int Nr,
                   float* d_ar, int32_t* i_ar, float* d_ra, int32_t* i_ra, void* stream);

/* geoa3_nn1_pair, pruned -- bit-identical results for any input.  Clouds of 1025..4096 points: a uniform grid over the
 * searched cloud (16^3 cells, rebuilt per call in LDS), every query scans the cells its seed's ball touches: O(N) instead
 * of O(N^2) work per instance for surface-like clouds; where the balls hold a large part of the cloud (dense clusters,
 * thin parts, iterates far from the surface), and for every other cloud size (with fewer than 32 points: all pairs), the
 * matrix core forms approximate distances of all pairs and only the pairs under the seed's radius (plus a bound on the
 * approximation's error) are evaluated exactly (csrc/geom_filter.hip).  prior_ar [B,Na] / prior_ra [B,Nr] (optional, MAY ALIAS i_ar / i_ra): an index into the
 * searched cloud per query, e.g. last iteration's answer; it only seeds the search radius (default: the point with
 * the query's own index), any value gives the exact result. */
int geoa3_grid_nn1_pair(const float* a, const float* r, int B, int Na, int Nr, const int32_t* prior_ar,
                        const int32_t* prior_ra, float* d_ar, int32_t* i_ar, float* d_ra, int32_t* i_ra, void* stream);

/* General K (1..GEOA3_KNN_MAX_K): dists/idx [B,Nq,K] ascending by (distance, index).
 * `prior` (optional, [B,Nq,K], int32, K DISTINCT valid indices per query, e.g. the previous
 * iteration's result) only seeds the pruning radius; the result is exact for any prior.
 * Replaces knn_points(K=k+1) at Lib/loss_utils.py:57,77. */
#define GEOA3_KNN_MAX_K 64
int geoa3_knn(const float* q, const float* r, int B, int Nq, int Nr, int K,
              const int32_t* prior, float* dists, int32_t* idx, void* stream);

/* Self K-NN of one cloud (q == r == pc, [B,3,N]) == geoa3_knn(pc, pc, B, N, N, K, prior, ...) bit for bit, pruned by
 * the radius `prior` gives (the K-th distance among last iteration's neighbours), with one of two searches:
 *   method 1 (slab): the cloud is counting-sorted along its longest axis and each workgroup of queries scans only the
 *            contiguous run of sorted points whose coordinate can lie within the queries' radius;
 *   method 2 (grid): the cloud is counting-sorted into a 16^3 grid and ONE WAVEFRONT per query walks the cell rows its
 *            ball touches, collects candidates by ballot and picks the K smallest by rank counting (no per-thread
 *            lists: the form for K > 20 or N >= 2048);
 *   method 0: picks by (K, N) -- and, for the slab search, by launch size between two bit-identical kernels: candidates
 *            kept as (distance, index) pairs (method 3 forces it) or as 2-byte positions (method 4; denser, the choice
 *            for launches of more than 512 workgroups).
 * Without `prior` or `scratch`, or for N > 8192, it runs the all-pairs kernel.  scratch:
 * geoa3_knn_self_scratch_bytes(B, N) bytes, 256-byte aligned, contents irrelevant.
 * Replaces knn_points(adv, adv, K=k+1) at Lib/loss_utils.py:77. */
int64_t geoa3_knn_self_scratch_bytes(int B, int N);
int geoa3_knn_self(const float* pc, int B, int N, int K, const int32_t* prior, float* dists, int32_t* idx,
                   void* scratch, int method, void* stream);


/* _get_kappa_ori (Lib/loss_utils.py:52-62) given the self K-NN table knn_idx [B,N,k+1]
 * (column 0, the nearest hit, is dropped exactly as the reference's [:, :, :, 1:] slice):
 *   kappa[b,i] = mean_m | < normalize(p[knn_idx[b,i,m]] - p_i), n_i > |,  m = 1..k.
 * nn_idx (optional, [B,N]): normals are taken from normal[:, nn_idx[b,i]] -- the
 * _get_kappa_adv form (Lib/loss_utils.py:64-82); NULL = normal[:, i].  normal is [B,3,Nn] (Nn = 0 means N;
 * Nn != N only with nn_idx: the dense-cloud path, where pc is an npoint-sample of a cloud of Nn points). */
int geoa3_kappa(const float* pc, const float* normal, const int32_t* knn_idx, const int32_t* nn_idx,
                int B, int N, int Nn, int k, float* kappa, void* stream);

/* ------------------------------------------------------------------------------------------
 * Objective level: the geometric part of _forward_step (Attacker/geoA3_attack.py:131-166) and its
 * gradient, fused.  Consumes the tables produced by geoa3_nn1_pair / geoa3_knn.
 * ------------------------------------------------------------------------------------------ */
typedef struct geoa3_geo_args {
  /* inputs */
  const float* adv;        /* [B,3,N] current iterate                                  */
  const float* ori;        /* [B,3,Nr] clean cloud                                     */
  const float* normal_ori; /* [B,3,Nr]         (may be NULL when w_curv == 0)          */
  const float* kappa_ori;  /* [B,Nr]           (may be NULL when w_curv == 0)          */
  const float* d_ao;       /* [B,N] adv->ori squared distance                          */
  const int32_t* i_ao;     /* [B,N] adv->ori index                                     */
  const float* d_oa;       /* [B,Nr] ori->adv      (NULL when single_side or dis_type != CD) */
  const int32_t* i_oa;     /* [B,Nr]                                                   */
  const int32_t* knn_adv;  /* [B,N,k+1] self K-NN of adv (NULL when w_curv == 0)       */
  const float* dkappa;     /* [B,N] optional upstream d L / d kappa_adv: when given, the curvature part of
                              `grad` is the vector-Jacobian product of _get_kappa_adv with it (operator-level
                              autograd) instead of the curvature_loss gradient                */
  int32_t B, N, k;
  int32_t Nr;              /* points of ori / normal_ori / kappa_ori / d_oa / i_oa; 0 = N.  Nr > N is the dense-cloud
                              path (--is_subsample_opt, geoA3_attack.py:283-284): adv is the npoint-sample */
  int32_t dis_type;        /* 0 = none, 1 = CD (chamfer_loss / pseudo_chamfer_loss), 2 = L2 (norm_l2_loss) */
  int32_t single_side;     /* --is_cd_single_side                                      */
  float w_dis, w_hd, w_curv; /* dis_loss_weight, hd_loss_weight, curv_loss_weight     */
  /* outputs (each may be NULL) */
  float* dis_loss;         /* [B] chamfer_loss / pseudo_chamfer_loss / norm_l2_loss, Lib/loss_utils.py:25-43 */
  float* hd_loss;          /* [B] hausdorff_loss, Lib/loss_utils.py:45-50              */
  float* curv_loss;        /* [B] curvature_loss, Lib/loss_utils.py:84-97              */
  float* constrain;        /* [B] w_dis*dis + w_hd*hd + w_curv*curv (geoA3_attack.py:137,153,162) */
  float* kappa_adv;        /* [B,N] _get_kappa_adv, Lib/loss_utils.py:64-82            */
  float* grad;             /* [B,3,N] d constrain[b] / d adv[b]                        */
  /* deterministic != 0: every point's gradient is summed by its owner in a fixed order (own terms, then the pulls it
   * receives sorted by source) instead of with LDS float atomics: bit-for-bit reproducible and independent of the rest
   * of the batch (what the reference's scatter-adds -- knn_gather / index backward -- do not promise either).
   * Clouds of at most 1024 points take the pair-parallel kernel (a lane per (centre, neighbour) pair, reverse lists as
   * fixed-capacity rows in LDS, rows sorted in registers; a row beyond its ~50 slots -- a dense cluster, a hub of the
   * K-NN graph -- through 64-bit fixed-point sums in a small pool, order-free); 1025..4096 points with `scratch`: the
   * pair-parallel kernel with fixed-point sums; otherwise up to ~4800 points the one-workgroup kernel with LDS reverse lists; beyond (up to ~5800)
   * the sums fall back to LDS float atomics (free order) whatever this flag says. */
  int32_t deterministic;
  /* optional workspace of 16 * B * N bytes (one float4 record per point).  Given, clouds of 1025..4096 points (and smaller
   * ones with k > 32) with the curvature term take the pair-parallel kernel with 64-bit fixed-point gradient sums (geo_big_kernel: order-free and
   * therefore reproducible; the neighbour table is read once); NULL = the one-workgroup kernel.  Same values to rounding.
   * The fixed-point sums take two power-of-two scales per instance from its largest coefficient X (2 w_curv / (N k) or
   * max |dkappa| / k, and the Chamfer coefficients): terms up to 2^10 X at 2^-40 X per unit, terms up to 2^34 X (pairs down
   * to ~1e-10 apart) at 2^-16 X through a pool of 256 destinations.  A term beyond that, a NaN, or a full pool writes NaN
   * into the gradient of the destination point -- there is no silent saturation. */
  void* scratch;
} geoa3_geo_args;
int geoa3_geo_loss_grad(const geoa3_geo_args* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * Victim model: PointNet eval forward and input-gradient (Model/PointNet.py:56-160).
 * Weights: BatchNorm folded and repacked by the host (geoa3_amd/pointnet.py) into the arrays
 * below; all fp32, row-major [C_out, K] with K contiguous unless noted.
 * ------------------------------------------------------------------------------------------ */
typedef struct geoa3_tnet_weights {   /* transform_net, Model/PointNet.py:56-94 */
  int32_t K;                /* 3 or 64 */
  const float *w1, *b1;     /* conv1+bn1 folded  [64,K]           */
  const float *w2, *b2;     /* conv2+bn2         [128,64]         */
  const float *w3, *b3;     /* conv3+bn3         [1024,128]       */
  const float *w3p;         /* w3 in MFMA A-fragment order (see w5p) */
  const float *w2t;         /* [64,128] = w2^T (input-gradient; conv3's is sparse and reads w3) */
  const float *f1, *fb1;    /* fc1+bn4 [512,1024] */
  const float *f2, *fb2;    /* fc2+bn5 [256,512]  */
  const float *f3, *fb3;    /* fc3     [K*K,256]  */
  const float *f1t, *f2t, *f3t; /* transposes: [1024,512], [512,256], [256,K*K] */
  const void *w3h;          /* optional (NULL = fp32 MFMA): w3 as split-fp16 fragments, see w5h */
  float w3h_unscale;
  const void *w2h;          /* optional, with w3h: w2 as split-fp16 fragments (same packing, K = 64).  Given, conv2 (and
                               conv1 of the 3-channel T-Net) is evaluated inside conv3's staging pass: its [B,128,N]
                               activation is never written, only its relu gate bits */
  float w2h_unscale;
  const void *w3h16;        /* optional, with w3h: the same split weights in 16x16x32 fragment order
                               [T = co/16][s = k/32][piece][lane][j] = piece(w3[16T + (lane&15)][32s + 8(lane>>4) + j])
                               (same scale: w3h_unscale): the layer then runs on `v_mfma_f32_16x16x32_f16`
                               (csrc/pointnet_wide16.hip) -- same arithmetic, higher sustained clock */
  const void *w2th;         /* optional, with w3h: w2t [64,128] as split-fp16 fragments (the packing of w5h, T = row/32, K = 128):
                               the A operands of the backward's fused kernel (sparse gradient of conv3 + conv2's backward in
                               one launch).  NULL: the two run as separate kernels (same bits, slower) */
  float w2th_unscale;
} geoa3_tnet_weights;

#define GEOA3_PN_NO_FUSE_BWD 1 /* sparse backward and the 128 -> 64 layer behind it as two kernels (same bits) */
#define GEOA3_PN_NO_CHAIN 2    /* the 64-input layers one kernel each instead of chains (same bits) */
#define GEOA3_PN_NO_PRE_LISTS 8 /* the sparse backward builds its hit lists per workgroup instead of reading the forward's (same bits) */
#define GEOA3_PN_KEYS_CLEAN 4  /* geoa3_pointnet_forward: the caller vouches that the LAST call that used `workspace` was a
                                  geoa3_pointnet_forward with the same (B, N) (backward calls in between are fine) -- the
                                  arg-max keys in it are then zero already (every layer's finalize leaves them so) and the
                                  forward does not clear them again: one launch less per iteration of a loop */
typedef struct geoa3_pointnet_weights {  /* PointNet, Model/PointNet.py:96-160 */
  int32_t classes;
  geoa3_tnet_weights t3, t64;
  const float *w1, *b1;     /* conv1+bn1 [64,3]    */
  const float *w2, *b2;     /* conv2+bn2 [64,64]   */
  const float *w3, *b3;     /* conv3+bn3 [64,64]   */
  const float *w4, *b4;     /* conv4+bn4 [128,64]  */
  const float *w5, *b5;     /* conv5+bn5 [1024, 3*128]: k = tap*128 + ci (kernel 3, pad 1, PointNet.py:110) */
  const float *w5p;         /* w5 in MFMA A-fragment order: [(T*3+tap)*16+j][lane 0..63][i 0..3] =
                               w5[32T + (lane&31)][tap*128 + 8j + 4(lane>>5) + i] -- one coalesced 16-byte load per lane */
  const float *w4t, *w3t, *w2t;  /* [64,128] [64,64] [64,64] (transposes for the input-gradient) */
  const float *f1, *fb1;    /* fc1+bn6 [512,1024] */
  const float *f2, *fb2;    /* fc2+bn7 [256,512]  */
  const float *f3, *fb3;    /* fc3     [classes,256] */
  const float *f1t, *f2t, *f3t; /* [1024,512] [512,256] [256,classes] */
  const void *w5h;          /* optional (NULL = the 1024-wide layers run on the fp32 MFMA): v = w5 * 2^e as TWO fp16 values
                               per weight, hi = rn16(v), lo = rn16(v - hi), in MFMA 32x32x16 fragment order
                               [T = co/32][s = k/16][piece hi,lo][lane 0..63][j 0..7] = piece(w5[32T + (lane&31)][16s +
                               8(lane>>5) + j]); the layer then evaluates a*w = a_hi*w_hi + a_hi*w_lo + a_lo*w_hi on the
                               f16 matrix pipe with fp32 accumulation (csrc/pointnet_wide_split.hip).  Both or neither
                               of the T-Nets' w3h must be given with it.  Given, the 64/128-wide convolutions and the
                               Gram product of the backward use the same arithmetic (they split their fp32 weights
                               themselves: csrc/pointnet_conv_split.hip, pointnet_gram.hip); NULL = fp32 MFMA throughout. */
  float w5h_unscale;        /* 2^-e */
  const void *w4h;          /* optional, with w5h: w4 as split-fp16 fragments (K = 64): conv4 inside conv5's staging pass */
  float w4h_unscale;
  const void *w5h16;        /* optional, with w5h: w5 in 16x16x32 fragment order (see t3.w3h16) */
  const void *w4th;         /* optional, with w5h: w4t [64,128] as split-fp16 fragments (see t3.w2th) */
  float w4th_unscale;
  int32_t flags;            /* GEOA3_PN_* bits; 0 = the default kernels */
} geoa3_pointnet_weights;

/* bytes of scratch the forward+backward pair needs for a batch of B clouds of N points */
int64_t geoa3_pointnet_workspace_bytes(int B, int N, int classes);

/* logits[B,classes] = net(x[B,3,N]); keeps the activations backward needs in `workspace`.
 * Replaces `net(input_curr_iter)` at Attacker/geoA3_attack.py:103 (and the b batch-1 calls at :297). */
int geoa3_pointnet_forward(const geoa3_pointnet_weights* w, const float* x, int B, int N,
                           float* logits, void* workspace, void* stream);

/* dx[B,3,N] = d( sum_b <dlogits[b], logits[b]> ) / d x, for the forward that last filled
 * `workspace`.  Replaces the autograd walk of loss.backward() through the network
 * (Attacker/geoA3_attack.py:326); weight gradients are never formed. */
int geoa3_pointnet_backward(const geoa3_pointnet_weights* w, const float* x, const float* dlogits,
                            int B, int N, float* dx, void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Driver level: device-resident state of attack() (Attacker/geoA3_attack.py:182-386).
 * ------------------------------------------------------------------------------------------ */
typedef struct geoa3_attack_state {
  int32_t B, N, classes;
  int32_t targeted;          /* cfg.attack_label != 'Untarget' (geoA3_attack.py:189-192) */
  int32_t cls_loss_type;     /* 0 = None, 1 = CE, 2 = Margin (geoA3_attack.py:105-127)   */
  float confidence;          /* cfg.confidence                                          */
  float inv_global_batch;    /* 1 / b of `loss_n.mean()` (geoA3_attack.py:178); the GLOBAL b when sharded */
  const int32_t* gt;         /* [B] ground-truth labels                                 */
  const int32_t* target;     /* [B] attack targets (== gt when untargeted, geoA3_attack.py:211-214) */
  float* scale_const;        /* [B] */
  float* lower_bound;        /* [B] */
  float* upper_bound;        /* [B] */
  float* best_loss;          /* [B] init 1e10 */
  float* best_attack;        /* [B,3,N] init 1.0 (geoA3_attack.py:226) */
  int32_t* best_step;        /* [B] init -1 */
  int32_t* best_bs;          /* [B] init -1 */
  float* iter_best_loss;     /* [B] reset to 1e10 per binary step */
  int32_t* iter_best_score;  /* [B] reset to -1 per binary step   */
  float* prev_constrain;     /* [B] reset to 1e10 per binary step (geoA3_attack.py:233,301) */
  int32_t* label;            /* [B] arg-max label of the current iterate */
  float* cls_loss;           /* [B] */
  float* loss_n;             /* [B] */
  float* loss_hist;          /* [iter_max_steps, B] (all_loss_list, geoA3_attack.py:229,321) or NULL */
  int32_t* last_label;       /* [1] label of the LAST instance at the LAST step (the `output_label`
                                reused for every k at geoA3_attack.py:375) */
} geoa3_attack_state;

/* Per step, after the forward: classification loss + d loss/d logits, arg-max label, the success
 * bookkeeping of geoA3_attack.py:288-310 for the CURRENT iterate x with the PREVIOUS step's
 * constrain loss, loss_n = cls + scale_const*constrain, and prev_constrain <- constrain.
 * dlogits[B,classes] already carries inv_global_batch. */
int geoa3_attack_head(const geoa3_attack_state* st, const float* logits, const float* constrain,
                      const float* x, int step, int search_step, float* dlogits, void* stream);
/* The dense-cloud form (--is_subsample_opt with more points than cfg.npoint, geoA3_attack.py:283-296): `logits` are
 * those of the npoint-sample the objective is evaluated on (classification loss, dlogits); success and
 * output_label come from vote_logits [B, eval_num, classes], the victim's answers on eval_num independent
 * farthest-point resamplings of the full iterate: success = more than half of them satisfy _compare, label = their
 * mode (smallest on ties, as torch.mode).  eval_num <= 64.  x / best_attack stay the FULL cloud (st->N points). */
int geoa3_attack_head_vote(const geoa3_attack_state* st, const float* logits, const float* vote_logits, int eval_num,
                           const float* constrain, const float* x, int step, int search_step, float* dlogits,
                           void* stream);
/* The same step in two launches, for callers that overlap the geometry kernels with the victim's backward:
 * _classify = the part that needs only the logits (cls_loss, label, last_label, d loss / d logits, the success flag of
 * geoA3_attack.py:297-300 -> ok [B] int32); _finish = loss_n / loss history and the best-so-far bookkeeping of
 * :301-310 once the constrain loss is known.  _classify + _finish == geoa3_attack_head_vote bit for bit. */
int geoa3_attack_head_classify(const geoa3_attack_state* st, const float* logits, const float* vote_logits, int eval_num,
                               float* dlogits, int32_t* ok, void* stream);
int geoa3_attack_head_finish(const geoa3_attack_state* st, const int32_t* ok, const float* constrain, const float* x,
                             int step, int search_step, void* stream);

/* Per step, after both backward passes: g = g_cls + scale_const[b]*inv_global_batch*g_geo, then the
 * optimiser update of `offset` (torch.optim.Adam defaults or plain SGD, geoA3_attack.py:269-272,
 * 323-331), optional lp_clip (geoA3_attack.py:88-98,349-352), and x = ori + offset for the next step.
 * step_size = lr/(1-beta1^t) and sqrt_bc2 = sqrt(1-beta2^t) are formed on the host in double
 * exactly as torch.optim.Adam does.  optim: 0 = adam, 1 = sgd (step_size = lr). */
int geoa3_attack_update(const geoa3_attack_state* st, const float* g_cls, const float* g_geo,
                        const float* ori, float* offset, float* adam_m, float* adam_v, float* x,
                        int optim, float step_size, float sqrt_bc2, float cc_linf, void* stream);

/* --is_partial_var (geoA3_attack.py:239-262,278-281, Lib/utility.py:211-216): the optimised variable is `part`
 * [B,3,kr] on the kr points pidx [B,kr] of every instance, x = periodical + pad(part).  optim -1: only writes
 * x[:, :, pidx] = periodical + part (after a re-draw); 0: Adam step (scalars as geoa3_attack_update); 1: SGD with
 * `momentum` (first != 0: the momentum buffer pm is initialised with the gradient, torch.optim.SGD). */
int geoa3_attack_partial_step(const geoa3_attack_state* st, const float* g_cls, const float* g_geo, const int32_t* pidx,
                              int kr, const float* periodical, float* part, float* pm, float* pv, float* x, int optim,
                              float step_size, float sqrt_bc2, float momentum, int first, void* stream);

/* The flag-gated projections that follow the optimiser step (--is_pro_grad / --is_real_offset,
 * geoA3_attack.py:59-85,341-347), one point per thread.
 *   mode 0 (find_offset, :79-85):  offset = x - ori[:, nn[b,i]]            (nn = nearest original of x)
 *   mode 1 (offset_proj, :59-77):  n = normal_ori[:, nn[b,i]] with nn = nearest original point OF THE OFFSET
 *          VECTOR (the reference queries knn_points with `offset` as the cloud, :65);
 *          offset = <offset, n/(|n|+1e-6)> n/(|n|+1e-6); then lp_clip (cc_linf != 0, :88-98,349-352) and
 *          x = ori + offset. */
int geoa3_attack_project(int mode, const float* ori, const float* normal_ori, const int32_t* nn, float* offset,
                         float* x, int B, int N, float cc_linf, void* stream);

/* End of a binary step: the scale_const / bound update of geoA3_attack.py:374-384, bug-compatible
 * (uses *last_label for every instance).  Also re-arms the per-binary-step state. */
int geoa3_attack_binary_update(const geoa3_attack_state* st, void* stream);

/* Start of a binary step: offset <- init_offset, adam state <- 0, x = ori + offset, per-step state reset. */
int geoa3_attack_begin_search_step(const geoa3_attack_state* st, const float* ori, const float* init_offset,
                                   float* offset, float* adam_m, float* adam_v, float* x, void* stream);

/* ------------------------------------------------------------------------------------------
 * PointNet++ set-abstraction operators: the functions of the reference's vendored CUDA extension
 * `pointnet2_ops._ext` that the SSG classifier uses (Model/pointnet2_ops_lib/pointnet2_ops/_ext-src/src/
 * bindings.cpp:6-19; three_nn / three_interpolate serve only the segmentation FP module and are out of scope).
 * Layouts as in the extension: xyz [B,N,3] point-major, features [B,C,N], indices int32.
 * ------------------------------------------------------------------------------------------ */
/* sampling.cpp:66-87 / sampling_gpu.cu:69-229.  temp: optional [B,N] scratch that receives the final running
 * distances (the extension's `tmp` tensor); idx [B,m]. */
int geoa3_pn2_furthest_point_sampling(const float* xyz, int B, int N, int m, float* temp, int32_t* idx, void* stream);
/* sampling.cpp:25-43 / sampling_gpu.cu:8-30: out[b,c,j] = points[b,c,idx[b,j]] */
int geoa3_pn2_gather_points(const float* points, const int32_t* idx, int B, int C, int N, int M, float* out,
                            void* stream);
/* sampling.cpp:45-64 / sampling_gpu.cu:34-57: scatter-add into grad_points [B,C,N] (zeroed here) */
int geoa3_pn2_gather_points_grad(const float* grad_out, const int32_t* idx, int B, int C, int N, int M,
                                 float* grad_points, void* stream);
/* ball_query.cpp:10-32 / ball_query_gpu.cu:9-54: idx [B,M,nsample] (zeroed here) */
int geoa3_pn2_ball_query(const float* new_xyz, const float* xyz, int B, int N, int M, float radius, int nsample,
                         int32_t* idx, void* stream);
/* The same two with `flags`.  GEOA3_PN2_CONTRACT: the squared distances (and the sampler's |p|^2 <= 1e-3 skip test) as
 * fmaf(dz, dz, fmaf(dy, dy, dx * dx)) -- what nvcc's default -fmad=true makes of sampling_gpu.cu:100,103-104 and
 * ball_query_gpu.cu:31-32 (setup.py:32 builds with -O3 and no -fmad flag) -- instead of the un-fused default.  For
 * comparing indices with a CUDA run of the reference; the two forms differ only at near-ties. */
#define GEOA3_PN2_CONTRACT 1
int geoa3_pn2_furthest_point_sampling_ex(const float* xyz, int B, int N, int m, float* temp, int32_t* idx, int flags,
                                         void* stream);
int geoa3_pn2_ball_query_ex(const float* new_xyz, const float* xyz, int B, int N, int M, float radius, int nsample,
                            int32_t* idx, int flags, void* stream);
/* group_points.cpp:13-34 / group_points_gpu.cu:8-39: out[b,c,j,k] = points[b,c,idx[b,j,k]] */
int geoa3_pn2_group_points(const float* points, const int32_t* idx, int B, int C, int N, int M, int nsample,
                           float* out, void* stream);
/* group_points.cpp:36-58 / group_points_gpu.cu:43-75: scatter-add into grad_points [B,C,N] (zeroed here) */
int geoa3_pn2_group_points_grad(const float* grad_out, const int32_t* idx, int B, int C, int N, int M, int nsample,
                                float* grad_points, void* stream);

/* Tails of the shared MLPs (pointnet2_modules.py:9-19: Conv2d 1x1 + BatchNorm2d + ReLU; :62-70: max over nsample)
 * around the channel GEMMs, eval mode, BatchNorm scale already folded into the GEMM weights:
 *   bias_relu:           z [B,C,L] <- relu(z + shift[c])                         (in place)
 *   relu_grad:           dz = (y > 0) ? g : 0                                     (dz may alias g)
 *   bias_relu_max:       out [B,C,M] = relu(max_s z[B,C,M,S] + shift[c]), arg = first maximising s (F.max_pool2d)
 *   bias_relu_max_grad:  dz [B,C,M,S] = (s == arg && out > 0) ? g : 0 */
int geoa3_pn2_bias_relu(float* z, const float* shift, int B, int C, long L, void* stream);
/* The first layer of a set-abstraction MLP is linear in the gathered inputs, W [xyz_j - c_m ; f_j] = (W_x xyz + W_f f)_j
 * - (W_x c)_m: it is applied to the npoint UN-grouped columns once and the result gathered (geoa3_pn2_group_points);
 * what remains per grouped element is a shift per (instance, channel, centre) and the relu:
 *   shift_relu:       z[row][s] <- relu(z[row][s] + shift[row]),  row = (b, c, m), s < S   (in place)
 *   shift_relu_grad:  dz = y > 0 ? g : 0,  dshift[row] = sum_s dz[row][s]
 * (replaces the grouping of xyz, the cat and the K = C + 3 GEMM over npoint * nsample columns of
 * pointnet2_utils.py:318-331 + pointnet2_modules.py:57-64). */
int geoa3_pn2_shift_relu(float* z, const float* shift, long rows, int S, void* stream);
/* gather + shift + relu in one pass: out[b][c][m][s] = relu(points[b][c][idx[b][m][s]] + shift[b][c][m]);
 * shift_relu_grad with y == NULL: g is already gated, only dshift[row] = sum_s g[row][s] is formed */
/* group_points_grad + rowsum[b][c][m] = sum_s grad_out[b][c][m][s] in the same pass (nsample == 64; GEOA3_ENOSUPPORT
 * otherwise: use geoa3_pn2_shift_relu_grad with y == NULL) */
int geoa3_pn2_group_points_grad_sums(const float* grad_out, const int32_t* idx, int B, int C, int N, int M, int nsample,
                                     float* grad_points, float* rowsum, void* stream);
int geoa3_pn2_group_shift_relu(const float* points, const int32_t* idx, const float* shift, int B, int C, int N, int M,
                               int nsample, float* out, void* stream);
int geoa3_pn2_shift_relu_grad(const float* y, const float* g, float* dz, float* dshift, long rows, int S, void* stream);
int geoa3_pn2_relu_grad(const float* y, const float* g, float* dz, long total, void* stream);
int geoa3_pn2_bias_relu_max(const float* z, const float* shift, int B, int C, long M, int S, float* out, int32_t* arg,
                            void* stream);
int geoa3_pn2_bias_relu_max_grad(const float* g, const float* out, const int32_t* arg, int B, int C, long M, int S,
                                 float* dz, void* stream);

/* Channel-major 1x1 convolution with fused epilogue, the shared-MLP layer of pointnet2_modules.py:57-64 (Conv2d 1x1 +
 * eval BatchNorm2d folded into W / bias + ReLU) on [B,K,N] -> [B,Co,N] (N = npoint * nsample, contiguous):
 *   Y[b][co][n] = epi( sum_k W[co][k] X[b][k][n] ),  epi = (+ bias[co] if bias) (relu if relu) (0 where Z[b][co][n] <= 0 if Z)
 * K in {64, 128, 256}, Co a multiple of 64; fp32 in and out, split-fp16 operands on the f16 matrix pipe inside
 * (csrc/pointnet_conv_split.hip).  The input-gradient of such a layer is the same call with W^T and Z = the layer's
 * input activation. */
int geoa3_conv1x1(const float* X, const float* W, const float* bias, const float* Z, float* Y, int B, long N, int K,
                  int Co, int relu, void* stream);

/* The last layer of a set-abstraction MLP with F.max_pool2d over the 64 samples of a centre in its epilogue (the
 * [B,Co,npoint,64] activation is never written): out[b][co][m] = relu(max_s (W x)[co][64 m + s] + bias[co]), arg = the
 * first maximal sample (pointnet2_modules.py:57-70).  K = 128, Co a multiple of 64, N = 64 * npoint. */
int geoa3_conv1x1_max64(const float* X, const float* W, const float* bias, float* out, int32_t* arg, int B, long N, int K,
                        int Co, void* stream);
/* ... and its input gradient: the pooled layer's sparse gradient ((arg == s) ? g : 0; g and arg CENTRE-major
 * [B][npoint][K]; g must carry the pooled output's relu gate) is formed in registers, multiplied by W ([Co][K] = the transposed layer weight, K = 256) and gated by the
 * relu of the layer below (keep where Z > 0). */
int geoa3_conv1x1_onehot64(const float* g, const int32_t* arg, const float* W, const float* Z, float* Y, int B, long N,
                           int K, int Co, void* stream);

/* First set-abstraction level of the SSG classifier, fused (PointNetPP_ssg.py:58-66: npoint 512, radius 0.2, nsample 64,
 * mlp [3, 64, 64, 128]; pointnet2_modules.py:29-74, pointnet2_utils.py:296-333): grouped xyz (xyz[idx] - new_xyz) ->
 * three Conv2d 1x1 + eval BatchNorm2d (folded into w / shift b by the host) + ReLU -> max over the 64 samples.
 * xyz [B,N,3], new_xyz [B,M,3] point-major as in the reference; idx [B,M,64] from geoa3_pn2_ball_query;
 * out [B,M,128] CENTROID-major (one contiguous 512-byte row per centroid; the caller transposes to the reference's
 * [B,128,M]); arg [B,M,128] (u8: the arg-max sample, first on ties like F.max_pool2d) is what backward needs. */
typedef struct geoa3_sa1_weights {
  const float *w1, *b1;   /* [64,3],   [64]  */
  const float *w2, *b2;   /* [64,64],  [64]  */
  const float *w3, *b3;   /* [128,64], [128] */
} geoa3_sa1_weights;
int geoa3_pn2_sa1_forward(const float* xyz, const float* new_xyz, const int32_t* idx, const geoa3_sa1_weights* w, int B,
                          int N, int M, float* out, uint8_t* arg, void* stream);
/* grad_xyz [B,N,3] (the scatter-add over idx) and grad_new_xyz [B,M,3] (= minus the per-centroid sum) from
 * grad_out [B,M,128] (same layout as out); the hidden activations are recomputed, not stored.
 * scratch: geoa3_pn2_sa1_scratch_bytes(B, M) bytes -> the scatter-add as order-free 64-bit fixed-point sums at the
 * instance's own scale (deterministic, batch-independent; a NaN / infinite contribution gives NaN, not a clamp);
 * NULL -> global float atomics as the reference (group_points_gpu.cu:60). */
int64_t geoa3_pn2_sa1_scratch_bytes(int B, int M);
int geoa3_pn2_sa1_backward(const float* xyz, const float* new_xyz, const int32_t* idx, const geoa3_sa1_weights* w, int B,
                           int N, int M, const float* out, const uint8_t* arg, const float* grad_out, float* grad_xyz,
                           float* grad_new_xyz, float* scratch, void* stream);

/* ------------------------------------------------------------------------------------------
 * The whole PointNet++ SSG classifier (eval mode) as a native victim: forward and input gradient, the counterpart of
 * geoa3_pointnet_forward / _backward for BASELINE configs[3].  Replaces `PointNet2ClassificationSSG.forward`
 * (Model/PointNetPP_ssg.py:106-124: SA(512, r 0.2, 64 samples, [3,64,64,128]) -> SA(128, r 0.4, 64, [131,128,128,256])
 * -> SA(GroupAll, [259,256,512,1024]) -> Linear/BN/ReLU 1024-512-256 -> Linear classes; modules
 * pointnet2_modules.py:29-74, groupers pointnet2_utils.py:296-379) and torch autograd's backward through it.
 * Every Conv2d 1x1 / Linear (no bias) + eval BatchNorm is passed FOLDED: W' = diag(bn.weight / sqrt(var + eps)) W,
 * shift = bn.bias - mean * scale.  The first layer of levels 2 and 3 is split into its xyz columns (w?_wx [Co,3], the
 * first three input channels: QueryAndGroup / GroupAll cat xyz first) and its feature columns (w?_wf); *t = the
 * transposed matrix ([K,Co] row-major) used by the input gradient.
 * x, dx: planar [B,3,N] (the attack's layout), 32 <= N <= 8192 (a cloud of fewer than 512 points has its points repeated
 * by the sampler, as in the reference); workspace: geoa3_pn2ssg_workspace_bytes(B, N) bytes, 256-byte
 * aligned, the SAME buffer for forward and the backward that follows it (backward reads the forward's activations).
 * ------------------------------------------------------------------------------------------ */
typedef struct geoa3_pn2ssg_weights {
  int classes;
  geoa3_sa1_weights sa1;
  const float *sa2_wx, *sa2_wf, *sa2_b0, *sa2_wft;   /* [128,3], [128,128], [128], [128,128] */
  const float *sa2_w1, *sa2_b1, *sa2_w1t;            /* [128,128], [128], [128,128] */
  const float *sa2_w2, *sa2_b2, *sa2_w2t;            /* [256,128], [256], [128,256] */
  const float *sa3_wx, *sa3_wf, *sa3_b0, *sa3_wft;   /* [256,3], [256,256], [256], [256,256] */
  const float *sa3_w1, *sa3_b1, *sa3_w1t;            /* [512,256], [512], [256,512] */
  const float *sa3_w2, *sa3_b2, *sa3_w2t;            /* [1024,512], [1024], [512,1024] */
  const float *f1, *fb1, *f1t;                       /* [512,1024], [512], [1024,512] */
  const float *f2, *fb2, *f2t;                       /* [256,512], [256], [512,256] */
  const float *f3, *fb3, *f3t;                       /* [classes,256], [classes], [256,classes] */
  const void* images;   /* geoa3_pn2ssg_pack_images of THESE weights, or NULL (the forward then rebuilds them per call) */
  void* side;           /* geoa3_side_queue_create(), or NULL: one stream.  With it the forward is pipelined: the sampler's rounds
                           of level 1 in four launches, each followed by the gather + ball query of its 128 centroids, and then
                           level 2's sampling, ball query and shift (functions of the level-1 centroids only) on the queue's
                           stream, beside level 1's MLP on `stream`; joined in front of level 2's MLP.  Same bits; the call is
                           still ordered on `stream` as a whole.  One queue per concurrent caller (the events are re-recorded by
                           every call). */
  int32_t flags;        /* GEOA3_PN2_CONTRACT: the sampler's and the ball queries' distances contracted (see above); 0 = default */
} geoa3_pn2ssg_weights;
/* A HIP stream + five events owned by the caller's module object (the library keeps no global state).  Destroy after the
 * last call that used it has been enqueued (destroy synchronises the side stream). */
void* geoa3_side_queue_create(void);
void geoa3_side_queue_destroy(void* side);
/* The level-2 / level-3 matrices as split-fp16 fragment images (the order the matrix-core loops read them; one
 * power-of-two scale per matrix): built once per set of weights into `images` (geoa3_pn2ssg_images_bytes() bytes,
 * 256-byte aligned; the `images` member of *w is not read). */
int64_t geoa3_pn2ssg_images_bytes(void);
int geoa3_pn2ssg_pack_images(const geoa3_pn2ssg_weights* w, void* images, void* stream);
int64_t geoa3_pn2ssg_workspace_bytes(int B, int N);
int geoa3_pn2ssg_forward(const geoa3_pn2ssg_weights* w, const float* x, int B, int N, float* logits, void* workspace,
                         void* stream);
int geoa3_pn2ssg_backward(const geoa3_pn2ssg_weights* w, const float* x, const float* dlogits, int B, int N, float* dx,
                          void* workspace, void* stream);

/* ------------------------------------------------------------------------------------------
 * Dense-cloud attack path, point-removal defence and smoothness measurement.
 * ------------------------------------------------------------------------------------------ */
/* farthest_points_sample, Lib/utility.py:175-187: m-1 rounds of dists = min(dists, |p - p_last|), next = first
 * arg-max.  start [B]: the index the reference draws with torch.randint (:179).  idx [B,m]; pts [B,3,m] (optional)
 * = pc gathered at idx.  N <= 16384. */
int geoa3_fps_sample(const float* pc, int B, int N, int m, const int32_t* start, int32_t* idx, float* pts,
                     void* stream);
/* estimate_normal_via_ori_normal, Lib/utility.py:91-108, from (knn_d, knn_i) = geoa3_knn(adv, ori, K) [B,Nq,K]:
 * out [B,3,Nq] = normal of the nearest original point when knn_d[..,0] < 1e-6, else the normalised mean of the K
 * gathered normals (evaluated per instance, as main_attack.py:244-246 calls it). */
int geoa3_knn_normal(const float* knn_d, const int32_t* knn_i, const float* normal_ori, int B, int Nq, int Nr, int K,
                     float* out, void* stream);
/* Local frames: eigen-decomposition of the covariance (factor 1/(k-1)) of the k = K1-1 nearest neighbours listed in
 * knn_idx [B,N,K1] (column 0, the point itself, dropped) -- Lib/utility.py:122-133,
 * Measurement/compute_data_smoothness.py:52-61.  evals [B,3,N] ascending; evecs [B,3(e),3(xyz),N], unit length,
 * largest-magnitude component positive. */
int geoa3_local_frames(const float* pc, const int32_t* knn_idx, int B, int N, int K1, float* evals, float* evecs,
                       void* stream);
/* estimate_perpendicular's tail, Lib/utility.py:146-149: out [B,3,N] = clamp(v_max*aux1, +-clip) +
 * clamp(v_mid*aux2, +-clip); aux1/aux2 [B,N] = sigma * N(0,1) drawn by the caller. */
int geoa3_perp_jitter(const float* evecs, const float* aux1, const float* aux2, int B, int N, float clip, float* out,
                      void* stream);
/* compute_data_smoothness.py:63-67: out[b] = max_i mean_{m=1..K1-1} |<p[knn_idx[b,i,m]] - p_i, n_i>| with n_i the
 * smallest-eigenvalue eigenvector from geoa3_local_frames; per_point [B,N] optional. */
int geoa3_smoothness(const float* pc, const int32_t* knn_idx, const float* evecs, int B, int N, int K1,
                     float* per_point, float* out, void* stream);
/* defense.py:27-28: dis [B,N] = mean distance to the K nearest neighbours, distances evaluated as
 * sqrt(sum_c (p_j - p_i + 1e-10)^2) like the reference. */
int geoa3_sor_statistic(const float* pc, int B, int N, int K, float* dis, void* stream);
/* defense.py:31-45: the kept indices of every cloud, ascending, idx [B,N] padded with -1, count [B].
 * mode 0 = outliers_fixNum (keep the N-drop_num smallest dis), 1 = outliers_variance (dis < mean + alpha*std);
 * stats [B,2] (optional) = (mean, std). */
int geoa3_sor_select(const float* dis, int B, int N, int mode, int drop_num, float alpha, int32_t* idx,
                     int32_t* count, float* stats, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* GEOA3_HIP_H */
