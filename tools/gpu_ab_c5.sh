#!/bin/bash
# A/B of builds of the library on configs[4] (N = 4096, k = 32): tools/gpu_ab_c5.sh libA.so libB.so ...
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for l in "$@"; do
  r=$(GEOA3_LIB_PATH=$PWD/$l python3 bench.py --npoint 4096 --knn 32 --no-cpu-baseline --single-mode --steps 40 --warmup 5 --presteps 60 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], (d.get('kernels_ms') or {}).get('knn'))")
  echo "$l ms_per_step knn_ms $r"
done; done
