#!/bin/bash
# one-stream kernel statistics of the geometry kernels at 250 / 32 instances and 4096 points
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/geo}; mkdir -p $O
run() {  # name, bench args
  n=$1; shift
  GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n -o t -- python3 bench.py --no-cpu-baseline --single-mode --no-proxy-full "$@" > $O/$n.log 2>&1
  find $O/$n -name '*kernel_trace.csv' -delete
  python3 - $O/$n <<'P'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/**/t_kernel_stats.csv', recursive=True)[0]
print("==", sys.argv[1])
for r in list(csv.DictReader(open(f)))[:int(26)]:
    n=r['Name']
    if any(s in n for s in ("geo_","nn1","knn","slab","kappa","cellsort")) or float(r['Percentage'])>4:
        print("%-90s %6s %9.1f %6s"%(n[:90], r['Calls'], float(r['AverageNs'])/1000, r['Percentage']))
P
}
run c2 --steps 40 --warmup 5
run p32 --instances 32 --steps 40 --warmup 5
if [ "$2" != "noc5" ]; then run c5 --npoint 4096 --knn 32 --steps 20 --warmup 5 --presteps 100; fi
