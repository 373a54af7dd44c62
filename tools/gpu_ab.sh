#!/bin/bash
# A/B of an environment switch on the bench lines: tools/gpu_ab.sh VAR "v1 v2 ..." [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
V=$1; VALS=$2; shift; shift
for rep in 1 2; do for v in $VALS; do
  for cfgs in "--steps 200 --warmup 20" "--instances 32 --no-proxy-full --steps 300 --warmup 20"; do
    r=$(env $V=$v python3 bench.py --no-cpu-baseline --single-mode $cfgs "$@" 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
    echo "$V=$v [$cfgs] ms_per_step $r"
  done
done; done
