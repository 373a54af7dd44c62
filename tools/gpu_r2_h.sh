#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_pointnet2.py -x -q -m gpu > $O/tests.log 2>&1
tail -15 $O/tests.log
python3 bench.py --arch PointNetPP --steps 40 --warmup 5 --presteps 20 --no-cpu-baseline > $O/bench_config4.json 2>> $O/bench.err
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace4 -o t -- python3 bench.py --arch PointNetPP --steps 20 --warmup 5 --presteps 10 --no-cpu-baseline > $O/trace4.log 2>&1
python3 tools/trace_timeline.py $O/trace4 > $O/timeline4.txt
rm -f $O/trace4/*kernel_trace.csv
