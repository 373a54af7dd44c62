#!/bin/bash
# A/B of an environment switch on the bench lines: bash tools/gpu_ab.sh VAR "v1 v2 .."
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
VAR=$1; shift
for v in $1; do
  echo "$VAR=$v"
  env $VAR=$v python3 bench.py --no-cpu-baseline --single-mode --steps 200 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' c2 ', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
  env $VAR=$v python3 bench.py --no-cpu-baseline --instances 32 --no-proxy-full --steps 300 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(' p32', d['value'], d['ms_per_step'])"
done
