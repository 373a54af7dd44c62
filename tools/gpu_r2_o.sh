#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2o; mkdir -p $O
timeout 1800 python3 -m pytest tests/test_gpu_attack.py tests/test_gpu_attack_aux.py tests/test_gpu_forward_step.py tests/test_gpu_geometry.py -x -q -m gpu > $O/tests.log 2>&1
tail -8 $O/tests.log
python3 bench.py --instances 32 --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_proxy32.json 2>> $O/bench.err
GEOA3_LATE_JOIN=0 python3 bench.py --instances 32 --steps 300 --warmup 10 --no-cpu-baseline --no-proxy-full > $O/bench_proxy32_early.json 2>> $O/bench.err
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --single-mode > $O/bench_config2.json 2>> $O/bench.err
GEOA3_LATE_JOIN=1 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --single-mode > $O/bench_config2_late.json 2>> $O/bench.err
