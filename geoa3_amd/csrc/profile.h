#pragma once
#include <hip/hip_runtime.h>
#define GEOA3_PROF_TAGS 12
#define GEOA3_PROF_CONV5 0   // wide_max_kernel<3>: conv5 + bn5 + relu + max
#define GEOA3_PROF_NN1 1     // nn1_pair_kernel: the "CD kernel"
#define GEOA3_PROF_KNN 2     // knn_kernel
#define GEOA3_PROF_TNETWIDE 3 // wide_max_kernel<1>
#define GEOA3_PROF_SA1_BWD 4  // sa1_bwd_kernel: PointNet++ level 1, input gradient
#define GEOA3_PROF_SA1_FWD 5  // sa1_fwd_kernel
#define GEOA3_PROF_FC 6       // fully connected chains (all launches of one forward or backward chain)
#define GEOA3_PROF_GEO 7      // geo_loss_grad_kernel
#define GEOA3_PROF_SA2_FWD 8  // sa2_fwd8_kernel: PointNet++ level 2, gather + MLP + max
#define GEOA3_PROF_SA2_BWD 9  // sa2b_bwd_kernel: level 2, pooled gradient -> per-point input gradient (rows in destination order)
#define GEOA3_PROF_SA2_GRAD 10 // sa2b_prep_kernel: the rows / entries / tiles of that pass
bool geoa3_prof_on();
bool geoa3_prof_tag_on(int tag);   // this tag is being sampled (bench.py's per-kernel figures)
void geoa3_prof_begin(int tag, hipStream_t s);
void geoa3_prof_end(int tag, hipStream_t s);
