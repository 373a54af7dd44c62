#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2j; mkdir -p $O
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace250 -o t -- python3 bench.py --steps 60 --warmup 10 --presteps 100 --no-cpu-baseline --single-mode > $O/trace250.log 2>&1
python3 tools/trace_timeline.py $O/trace250 > $O/timeline250.txt
rm -f $O/trace250/*kernel_trace.csv
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace32 -o t -- python3 bench.py --instances 32 --steps 60 --warmup 10 --presteps 100 --no-cpu-baseline --single-mode > $O/trace32.log 2>&1
python3 tools/trace_timeline.py $O/trace32 > $O/timeline32.txt
rm -f $O/trace32/*kernel_trace.csv
