// Internal launchers of the exact searches, shared between geom_nn.hip (all pairs), geom_grid.hip and geom_slab.hip.
#pragma once
#include "common.h"
// `only`: optional per-query byte flags ([ndir][B][max(Na,Nr)] for nn1, [B][Nq] for knn): search ONLY those queries
// (the fallback form for a pruned search; the shipped callers pass nullptr).
int geoa3_launch_nn1(const float* a, const float* r, int B, int Na, int Nr, float* d_ar, int32_t* i_ar, float* d_ra,
                     int32_t* i_ra, const uint8_t* only, hipStream_t s);
int geoa3_launch_knn(const float* q, const float* r, int B, int Nq, int Nr, int K, const int32_t* prior, float* dists,
                     int32_t* idx, const uint8_t* only, hipStream_t s);
// Grid-accelerated exact K=1 search (geom_grid.hip); GEOA3_ENOSUPPORT when a cloud exceeds 4096 points.
int geoa3_launch_grid_nn1(const float* a, const float* r, int B, int Na, int Nr, const int32_t* prior_ar,
                          const int32_t* prior_ra, float* d_ar, int32_t* i_ar, float* d_ra, int32_t* i_ra, hipStream_t s);
// The matrix-core filter + exact refinement as a kernel of its own (geom_filter.hip; the search itself: geom_filter.h):
// every query of every instance, clouds of any size.
int geoa3_launch_nn1_filter(const float* a, const float* r, int B, int Na, int Nr, const int32_t* prior_ar,
                            const int32_t* prior_ra, float* d_ar, int32_t* i_ar, float* d_ra, int32_t* i_ra, hipStream_t s);
