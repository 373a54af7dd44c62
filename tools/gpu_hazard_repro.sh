#!/bin/bash
# NOTEBOOK 5a, everything in one GPU call:  tools/gpu_hazard_repro.sh [launch pairs, default 100000]
#   1. tools/ub/xcd_visibility: the store -> kernel boundary -> load pattern on its own (expected: 0 wrong values);
#   2. tools/ub/dtpart_pair: conv_bwd_chain_kernel + reduce_dT_kernel alone, against the two-wave build
#      (tools/ub/lib_two_wave: `python -m geoa3_amd.build --variant tools/ub/lib_two_wave --no-file-flags` = the chain kernels
#      WITH packed-FP32 instructions; expected: wrong dx / dT3 in lanes 48-63 of ~1e-4 of the workgroups) and against the
#      product build (the same kernel without them; expected: 0);
#   3. the loop soak (80 runs of 30 iterations of configs[1]) on both builds.
# Build the two binaries first (in the container):
#   cd tools/ub && hipcc --offload-arch=gfx950 -O3 -o xcd_visibility xcd_visibility.hip && hipcc --offload-arch=gfx950 -O3 -o regstress regstress.hip
#               && hipcc --offload-arch=gfx950 -O3 -std=c++17 -o dtpart_pair dtpart_pair.hip -L ../../geoa3_amd/lib -lgeoa3_hip
cd $GRAFT_REPO_ROOT
N=${1:-100000}
( cd tools/ub && timeout 600 ./xcd_visibility $N 1000; [ -x ./regstress ] && timeout 600 ./regstress 20000 )
for L in tools/ub/lib_two_wave geoa3_amd/lib; do
  echo "== dtpart_pair against $L"
  LD_LIBRARY_PATH=$PWD/$L:$LD_LIBRARY_PATH timeout 600 tools/ub/dtpart_pair $N 2 32
done
echo "[two-wave build] $(GEOA3_LIB_PATH=$PWD/tools/ub/lib_two_wave/libgeoa3_hip.so python3 tools/loop_determinism_final.py 30 80 2>&1 | tail -1)"
echo "[product] $(python3 tools/loop_determinism_final.py 30 80 2>&1 | tail -1)"
