#!/bin/bash
# kernel start / end times of one PointNet++ forward with the side queue on (which kernels overlap)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pn2tl; rm -rf $O; mkdir -p $O
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --no-cpu-baseline --single-mode --arch PointNetPP --steps 6 --warmup 2 --presteps 4 > $O/log.txt 2>&1
python3 - $O <<'P'
import csv,sys,glob
f=glob.glob(sys.argv[1]+'/trace/**/t_kernel_trace.csv', recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last planar_to_points = start of the last forward
starts=[i for i,r in enumerate(rows) if 'planar_to_points' in r['Kernel_Name']]
i0=starts[len(starts)//3]
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0+34]:
    print("%9.1f %8.1f  q%-3s %s"%((int(r['Start_Timestamp'])-t0)/1000,(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000,r['Queue_Id'],r['Kernel_Name'][:70]))
P
find $O -name '*kernel_trace.csv' -delete
