#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2n; mkdir -p $O
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace32 -o t -- python3 bench.py --instances 32 --steps 60 --warmup 10 --presteps 100 --no-cpu-baseline --single-mode --no-proxy-full > $O/trace32.log 2>&1
python3 tools/trace_timeline.py $O/trace32 15 > $O/timeline32.txt
rm -f $O/trace32/*kernel_trace.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace32b -o t -- python3 bench.py --instances 32 --steps 60 --warmup 10 --presteps 100 --no-cpu-baseline --single-mode --no-proxy-full > $O/trace32b.log 2>&1
python3 tools/trace_timeline.py $O/trace32b 15 > $O/timeline32_2s.txt
rm -f $O/trace32b/*kernel_trace.csv
