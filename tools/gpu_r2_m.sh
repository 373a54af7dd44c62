#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2m; mkdir -p $O
for m in 1 2; do
GEOA3_KNN_METHOD=$m GEOA3_GEO_STREAM=0 python3 bench.py --npoint 4096 --knn 32 --steps 40 --warmup 5 --presteps 60 --no-cpu-baseline --single-mode > $O/c5_m${m}_1s.json 2>> $O/bench.err
GEOA3_KNN_METHOD=$m python3 bench.py --npoint 4096 --knn 32 --steps 40 --warmup 5 --presteps 60 --no-cpu-baseline --single-mode > $O/c5_m${m}_2s.json 2>> $O/bench.err
done
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace5 -o t -- python3 bench.py --npoint 4096 --knn 32 --steps 20 --warmup 5 --presteps 60 --no-cpu-baseline --single-mode > $O/trace5.log 2>&1
python3 tools/trace_timeline.py $O/trace5 > $O/timeline5.txt
rm -f $O/trace5/*kernel_trace.csv
