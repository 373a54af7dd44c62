/*
 * geoa3_hip_debug.h -- diagnostics of libgeoa3_hip.so, declared APART from the product ABI (geoa3_hip.h):
 * event timers around selected kernels (bench.py's `roofline` figures) and single-kernel entry points for the
 * tools/ micro-benchmarks and unit tests.  Nothing here is needed to run the attack loop, and nothing here
 * changes a result.  The event timer is the one piece of process-global state in the library (a table of
 * hipEvent_t, empty unless geoa3_profile_enable() is called); no entry point of geoa3_hip.h reads or writes any
 * other global.
 */
#ifndef GEOA3_HIP_DEBUG_H
#define GEOA3_HIP_DEBUG_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------
 * Event timers (bench.py): per-launch durations of selected kernels, taken with HIP events recorded on the
 * launch stream.  Off by default; never changes results.
 * tag: 0 = conv5+max (wide_max_kernel<3>), 1 = geoa3_nn1_pair ("CD kernel"), 2 = geoa3_knn,
 *      3 = T-Net conv3+max (wide_max_kernel<1>), 4 / 5 = PointNet++ level 1 backward / forward (sa1_*_kernel),
 *      6 = the fully connected chains, 7 = geo_loss_grad_kernel, 8 / 9 = PointNet++ level 2 forward / backward
 *      (sa2_fwd8_kernel, sa2b_bwd_kernel), 10 = the backward's preparation pass (sa2b_prep_kernel).
 * ------------------------------------------------------------------------------------------ */
int geoa3_profile_enable(int capacity);                 /* events for `capacity` launches per tag; 0 = off */
int geoa3_profile_select(unsigned mask);                /* bit t set = tag t is recorded (default: all); an event
                                                           pair costs ~6 us of stream time around the kernel */
int geoa3_profile_read(int tag, float* ms_host, int cap); /* waits for the recorded launches; returns count */

/* ------------------------------------------------------------------------------------------
 * Single kernels of the PointNet path in isolation (tools/bench_*.py, tests/test_gpu_pointnet.py).
 * ------------------------------------------------------------------------------------------ */
/* The sparse arg-max backward of a 1024-wide layer (tools/bench_widebwd.py): g [B,1024], arg [B,1024],
 * W [1024, taps*128], Z / dX [B,128,N]. */
int geoa3_debug_wide_bwd(const float* g, const int32_t* arg, const float* W, const float* Z, float* dX, int B, int N,
                         int taps, int form /* 0: shipped (register accumulation), 1: LDS accumulation */, void* stream);
/* One 1024-wide layer + max (tools/bench_wide.py): Wp = fp32 MFMA fragments, Wh = split-fp16 fragments or NULL;
 * variant selects a tuning variant of the split kernel for THIS call (0 = the shipped configuration). */
int geoa3_debug_wide_fwd(const float* X, const float* Wp, const void* Wh, float unscale, const float* bias, float* out,
                         int32_t* arg, void* keys, int B, int N, int taps, int variant, void* stamps, void* stream);
/* One fully connected layer Y[M,Nout] = relu?(X[M,K] W[Nout,K]^T + bias) (tools/bench_fc.py). */
int geoa3_debug_fc(const float* X, const float* W, const float* bias, float* Y, int M, int Nout, int K, int relu,
                   int ksplit, void* stream);
/* One channel-major 1x1 convolution Y[B,Co,N] = act(W[Co,K] X[B,K,N] + bias) (gated by Z > 0 when given) of the
 * PointNet trunk (K, Co in {64, 128}), for tools/bench_conv.py. */
int geoa3_debug_conv_cm(const float* X, const float* W, const float* bias, const float* Z, float* Y, int B, int N, int K,
                        int Co, int relu, int split /* 1: split-fp16 operands */, void* stream);

/* geoa3_grid_nn1_pair (geoa3_hip.h) with its search policy given: brute_frac = the fraction of the searched cloud inside
 * the queries' boxes beyond which a workgroup searches all pairs (0 = always), filter = through the matrix-core filter
 * kernel (1) or the in-kernel sweep (0); negative values = the shipped choice.  Every policy returns the same bits. */
int geoa3_debug_grid_nn1_pair(const float* a, const float* r, int B, int Na, int Nr, const int32_t* prior_ar,
                              const int32_t* prior_ra, float* d_ar, int32_t* i_ar, float* d_ra, int32_t* i_ra,
                              float brute_frac, int filter, void* stream);

/* Names and byte offsets (address order) of the buffers geoa3_pointnet_forward / _backward keep in their workspace:
 * tools/iteration_replay_soak.py attributes a run-to-run difference to the kernel that wrote it.  Returns the number
 * of buffers (names[i] are static strings). */
int geoa3_debug_pointnet_workspace_layout(int B, int N, int classes, const char** names, int64_t* offsets, int cap);

#ifdef __cplusplus
}
#endif
#endif /* GEOA3_HIP_DEBUG_H */
