#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/c5t; mkdir -p $O
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --no-cpu-baseline --single-mode --npoint 4096 --knn 32 --steps 10 --warmup 3 --presteps 60 > $O/trace.log 2>&1
find $O -name '*kernel_trace.csv' -delete
head -8 $O/trace/*/t_kernel_stats.csv 2>/dev/null | cut -c1-110 || find $O -name t_kernel_stats.csv -exec head -8 {} \; | cut -c1-110
