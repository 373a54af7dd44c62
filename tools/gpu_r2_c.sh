#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2c; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_forward_step.py tests/test_gpu_cli.py tests/test_gpu_attack.py -x -q -m gpu > $O/tests.log 2>&1
tail -15 $O/tests.log
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --output-format csv -d $O/trace32 -o t -- python3 bench.py --instances 32 --steps 60 --warmup 20 --no-cpu-baseline --single-mode > $O/trace32.log 2>&1
python3 tools/trace_timeline.py $O/trace32 > $O/timeline32.txt
rocprofv3 --kernel-trace --output-format csv -d $O/trace32b -o t -- python3 bench.py --instances 32 --steps 60 --warmup 20 --no-cpu-baseline --single-mode > $O/trace32b.log 2>&1
python3 tools/trace_timeline.py $O/trace32b > $O/timeline32_2s.txt
rm -rf $O/trace32 $O/trace32b
