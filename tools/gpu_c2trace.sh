#!/bin/bash
# one-stream kernel statistics of configs[1] (top kernels) + the two-stream bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/c2t}; mkdir -p $O
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/t -o t -- python3 bench.py --no-cpu-baseline --single-mode --no-proxy-full --steps 40 --warmup 5 > $O/log.txt 2>&1
find $O -name '*kernel_trace.csv' -delete
python3 - $O <<'P'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/t_kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:int(16)]:
    print("%-86s %5s %8.1f %6s"%(r['Name'].replace("(anonymous namespace)::","")[:86], r['Calls'], float(r['AverageNs'])/1000, r['Percentage']))
P
for i in 1 2; do python3 bench.py --no-cpu-baseline --single-mode --steps 200 --warmup 20 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])"; done
