#!/bin/bash
# A/B: the product library against a build with -fno-slp-vectorize on EVERY file (no compiler-formed packed FP32 anywhere):
#   python -m geoa3_amd.build --variant geoa3_amd/lib_noslp -fno-slp-vectorize
# and, to price the flag file by file (how FILE_FLAGS of geoa3_amd/build.py was chosen), variants with the flag on one group:
#   GEOA3_EXTRA_FILE_FLAGS="pointnet.hip,pointnet_gemm.hip,pointnet_gram.hip,pointnet2_mlp.hip:-fno-slp-vectorize" \
#       python -m geoa3_amd.build --variant geoa3_amd/lib_g3        (round 5: +5.5 % on configs[1]; every other group: 0 to +1 us)
# then list the libraries to compare in the loop below.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { GEOA3_LIB_PATH=$PWD/$1 python3 bench.py --no-cpu-baseline --single-mode $2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernels_ms') or {}; print('$1 [$2]', d['ms_per_step'], {a: round(b, 4) for a, b in k.items() if isinstance(b, float)})"; }
for rep in 1 2; do for l in geoa3_amd/lib/libgeoa3_hip.so geoa3_amd/lib_noslp/libgeoa3_hip.so; do
  run $l "--steps 200 --warmup 20"
  run $l "--instances 32 --no-proxy-full --steps 300 --warmup 20"
  run $l "--arch PointNetPP --steps 40 --warmup 5 --presteps 20"
  run $l "--npoint 4096 --knn 32 --steps 30 --warmup 5 --presteps 40"
done; done
