#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/c4}; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_pointnet2.py -x -q -m gpu 2>&1 | tail -3
python3 bench.py --no-cpu-baseline --arch PointNetPP --steps 40 --warmup 5 --presteps 20 2>$O/err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c4', d['value'], d['ms_per_step'], d['kernels_ms'].get('sa1_bwd'), d['kernels_ms'].get('sa1_fwd'))"
GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --no-cpu-baseline --single-mode --arch PointNetPP --steps 20 --warmup 5 --presteps 20 > $O/trace.log 2>&1
find $O -name '*kernel_trace.csv' -delete
python3 - $O <<'P'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/trace/**/t_kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-84s %6s %9.1f %6s"%(r['Name'][:84], r['Calls'], float(r['AverageNs'])/1000, r['Percentage']))
P
