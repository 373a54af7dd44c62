#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats, one stream) of the bench line under builds of the library:
#   tools/gpu_kstats_lib.sh PATTERN libA.so libB.so ...      (PATTERN: grep -E on the kernel names)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp GEOA3_GEO_STREAM=0
pat=$1; shift
for l in "$@"; do
  d=gpurun_out/kstats_$(echo $l | tr '/.' '__')
  rm -rf $d
  GEOA3_LIB_PATH=$PWD/$l rocprofv3 --kernel-trace --stats --output-format csv -d $d -o k -- python3 bench.py --no-cpu-baseline --single-mode --steps 100 --warmup 10 > /dev/null 2>&1
  echo "== $l"
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  python3 - "$f" "$pat" <<'PY'
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("%-90s %6s %8.1f us" % (re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
