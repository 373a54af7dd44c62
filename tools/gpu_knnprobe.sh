#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/knnprobe; mkdir -p $O
for p in ${1:-0 1 2 3}; do
  export GEOA3_KNN_PROBE=$p
  GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p$p -o t -- python3 bench.py --no-cpu-baseline --single-mode --no-proxy-full --steps 30 --warmup 5 > $O/p$p.log 2>&1
  find $O/p$p -name '*kernel_trace.csv' -delete
  echo "probe $p: $(grep knn_slabp $(find $O/p$p -name t_kernel_stats.csv) | awk -F, '{print $(NF-3)}')"
done
