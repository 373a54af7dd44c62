#!/bin/bash
# timeline of one iteration with the geometry on its own stream (default mode)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/two}; mkdir -p $O; shift
rocprofv3 --kernel-trace --output-format csv -d $O/trace -o t -- python3 bench.py --no-cpu-baseline --single-mode --no-proxy-full --steps 30 --warmup 5 "$@" > $O/log.txt 2>&1
python3 tools/trace_timeline.py $O/trace > $O/timeline.txt
find $O -name '*kernel_trace.csv' -delete
cat $O/timeline.txt
tail -1 $O/log.txt | cut -c1-200
