#!/bin/bash
# the packed-FP32 co-residency reproducer (tools/ub/pk_fp32_coresidency.hip), all builds; PK_BLOAT=1: the synthetic neighbours
# that bisect the NEIGHBOUR's share (register-file position, matrix instructions: density, data, type)
cd $GRAFT_REPO_ROOT/tools/ub
export PK_BLOAT=1
echo "== SLP vectoriser on (v_pk_add_f32 / v_pk_mul_f32 in the sampler's round)"; timeout 300 ./pk_slp
echo "== -fno-slp-vectorize (no packed FP32)"; timeout 300 ./pk_noslp | grep -v "^library"
# which form of the packed instruction?  the sampler's subtraction (a) as the compiler writes it: v_pk_add_f32 with neg_lo /
# neg_hi + op_sel (above), (b) on a pre-negated point: op_sel only, (c) on two copies of it: no modifier at all
if [ -x ./pk_slp_preneg ]; then echo "== SLP on, no neg modifiers (op_sel only)"; timeout 300 ./pk_slp_preneg | grep "alone\|sa1_\|MFMA"; fi
if [ -x ./pk_slp_preneg2 ]; then echo "== SLP on, no neg modifiers, no op_sel"; timeout 300 ./pk_slp_preneg2 | grep "alone\|sa1_\|MFMA"; fi
