#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2f; mkdir -p $O
timeout 2400 python3 -m pytest tests/test_gpu_forward_step.py tests/test_gpu_fullsize.py tests/test_gpu_pointnet.py tests/test_gpu_attack.py tests/test_gpu_attack_aux.py tests/test_gpu_multirank.py -x -q -m gpu > $O/tests.log 2>&1
tail -15 $O/tests.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace4 -o t -- python3 bench.py --arch PointNetPP --steps 20 --warmup 5 --presteps 10 --no-cpu-baseline > $O/trace4.log 2>&1
python3 tools/trace_timeline.py $O/trace4 > $O/timeline4.txt
rm -f $O/trace4/*kernel_trace.csv
