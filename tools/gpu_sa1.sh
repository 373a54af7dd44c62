#!/bin/bash
# PointNet++ level-1 kernels: tests, then per-kernel averages + the bench line under builds of the library
#   tools/gpu_sa1.sh libA.so libB.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_pointnet2.py -x -q -m gpu 2>&1 | tail -5
export GEOA3_GEO_STREAM=0
for l in "$@"; do
  d=gpurun_out/sa1_$(echo $l | tr '/.' '__')
  rm -rf $d
  GEOA3_LIB_PATH=$PWD/$l timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d -o k -- python3 bench.py --no-cpu-baseline --single-mode --arch PointNetPP --steps 20 --warmup 5 --presteps 20 > $d.log 2>&1
  echo "== $l"
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, re, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print("%-90s %6s %8.1f us %s" % (re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:90], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  find $d -name '*kernel_trace.csv' -delete
done
unset GEOA3_GEO_STREAM
for rep in 1 2; do for l in "$@"; do
  r=$(GEOA3_LIB_PATH=$PWD/$l python3 bench.py --arch PointNetPP --no-cpu-baseline --single-mode --steps 40 --warmup 5 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'])")
  echo "$l ms_per_step $r"
done; done
