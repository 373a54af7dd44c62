#!/bin/bash
# A/B of builds of the library on the bench lines: tools/gpu_ab_lib.sh libA.so libB.so ...   (paths relative to the repo;
# variants come from `python -m geoa3_amd.build --variant DIR -D...`).  Three interleaved rounds, 250 and 32 instances.
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for l in "$@"; do
  for cfgs in "--steps 300 --warmup 20" "--instances 32 --no-proxy-full --steps 300 --warmup 20"; do
    r=$(GEOA3_LIB_PATH=$PWD/$l python3 bench.py --no-cpu-baseline --single-mode $cfgs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernels_ms') or {}; print(d['ms_per_step'], 'knn', k.get('knn'), 'nn1', k.get('nn1_pair'))")
    echo "$l [$cfgs] ms_per_step $r"
  done
done; done
