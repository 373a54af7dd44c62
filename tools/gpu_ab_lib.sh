#!/bin/bash
# A/B of two builds of the library on the bench lines: tools/gpu_ab_lib.sh libA.so libB.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do for l in "$@"; do
  cp $l geoa3_amd/lib/libgeoa3_hip.so
  for cfgs in "--steps 300 --warmup 20" "--instances 32 --no-proxy-full --steps 300 --warmup 20"; do
    r=$(python3 bench.py --no-cpu-baseline --single-mode $cfgs 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "$l [$cfgs] ms_per_step $r"
  done
done; done
