#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2r; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_pointnet.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.log 2>&1
tail -5 $O/tests.log
GEOA3_WIDE_SHAPE_TNET=16 timeout 1200 python3 -m pytest tests/test_gpu_pointnet.py -x -q -m gpu > $O/tests16.log 2>&1
tail -3 $O/tests16.log
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --single-mode > $O/bench_config2.json 2>> $O/bench.err
python3 bench.py --instances 32 --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_proxy32.json 2>> $O/bench.err
python3 bench.py --npoint 4096 --knn 32 --steps 40 --warmup 5 --presteps 60 --no-cpu-baseline --single-mode > $O/bench_config5.json 2>> $O/bench.err
