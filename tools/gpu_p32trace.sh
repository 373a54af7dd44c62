#!/bin/bash
# one-stream kernel timeline of a small shard: bash tools/gpu_p32trace.sh <outdir> [instances]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/p32t}; mkdir -p $O
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 bench.py --no-cpu-baseline --single-mode --instances ${2:-32} --no-proxy-full --steps 60 --warmup 10 > $O/trace.log 2>&1
python3 tools/trace_timeline.py $O/trace > $O/timeline.txt
find $O -name '*kernel_trace.csv' -delete
cut -c1-112 $O/timeline.txt
