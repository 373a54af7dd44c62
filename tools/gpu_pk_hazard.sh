#!/bin/bash
# the packed-FP32 co-residency reproducer (tools/ub/pk_fp32_coresidency.hip), both builds
cd $GRAFT_REPO_ROOT/tools/ub
echo "== SLP vectoriser on (v_pk_add_f32 / v_pk_mul_f32 in the sampler's round)"; timeout 300 ./pk_slp
echo "== -fno-slp-vectorize (no packed FP32)"; timeout 300 ./pk_noslp
