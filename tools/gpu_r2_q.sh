#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2q; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_pointnet.py tests/test_gpu_fullsize.py tests/test_gpu_forward_step.py -x -q -m gpu > $O/tests.log 2>&1
tail -12 $O/tests.log
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --single-mode > $O/bench_config2_s16.json 2>> $O/bench.err
GEOA3_WIDE_SHAPE=32 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --single-mode > $O/bench_config2_s32.json 2>> $O/bench.err
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --single-mode > $O/bench_config2_s16b.json 2>> $O/bench.err
GEOA3_WIDE_SHAPE=32 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --single-mode > $O/bench_config2_s32b.json 2>> $O/bench.err
