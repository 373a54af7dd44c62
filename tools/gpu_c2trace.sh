#!/bin/bash
# one-stream kernel statistics of configs[1]: bash tools/gpu_c2trace.sh <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/c2t}; mkdir -p $O
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --no-cpu-baseline --single-mode --steps 60 --warmup 10 > $O/trace.log 2>&1
python3 tools/trace_timeline.py $O/trace > $O/timeline.txt
find $O -name '*kernel_trace.csv' -delete
python3 - $O <<'P'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/trace/**/t_kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:22]:
    print("%-84s %6s %9.1f %6s"%(r['Name'][:84], r['Calls'], float(r['AverageNs'])/1000, r['Percentage']))
P
tail -1 $O/timeline.txt
