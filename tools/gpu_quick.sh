#!/bin/bash
# quick A/B: bench lines of configs 2, 4, proxy + pointnet tests
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/q}; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_pointnet.py tests/test_gpu_pointnet2.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
python3 bench.py --no-cpu-baseline --steps 200 --warmup 10 > $O/c2.json 2> $O/err.log
python3 bench.py --no-cpu-baseline --arch PointNetPP --steps 40 --warmup 5 --presteps 20 > $O/c4.json 2>> $O/err.log
python3 bench.py --no-cpu-baseline --instances 32 --no-proxy-full --steps 300 --warmup 10 > $O/p32.json 2>> $O/err.log
for f in c2 c4 p32; do python3 -c "
import json,sys
d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1])
print('$f', d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'))
"; done
# one-stream kernel trace of configs[3]
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4trace -o t -- python3 bench.py --no-cpu-baseline --single-mode --arch PointNetPP --steps 20 --warmup 5 --presteps 20 > $O/c4_trace.log 2>&1
python3 tools/trace_timeline.py $O/c4trace > $O/c4_timeline.txt
find $O -name '*kernel_trace.csv' -delete
head -30 $O/c4trace/*/t_kernel_stats.csv 2>/dev/null | cut -c1-120 || find $O/c4trace -name 't_kernel_stats.csv' -exec head -30 {} \; | cut -c1-120
