#!/bin/bash
# A/B of builds of the library on the PointNet++ bench line (configs[3]): tools/gpu_ab_pn2.sh libA.so libB.so ...
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for l in "$@"; do
  r=$(GEOA3_LIB_PATH=$PWD/$l python3 bench.py --arch PointNetPP --no-cpu-baseline --single-mode --steps 40 --warmup 5 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$l ms_per_step $r"
done; done
