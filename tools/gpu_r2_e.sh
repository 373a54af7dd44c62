#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
timeout 600 python3 tools/bench_fc.py > $O/bench_fc.txt 2>&1
timeout 1200 python3 -m pytest tests/test_gpu_pointnet2.py tests/test_gpu_forward_step.py tests/test_gpu_pointnet.py -x -q -m gpu > $O/tests_a.log 2>&1
tail -15 $O/tests_a.log
timeout 2400 python3 -m pytest tests -x -q -m gpu --deselect tests/test_gpu_pointnet2.py --deselect tests/test_gpu_forward_step.py --deselect tests/test_gpu_pointnet.py > $O/tests_b.log 2>&1
tail -15 $O/tests_b.log
python3 bench.py --instances 32 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_proxy32.json 2>> $O/bench.err
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --single-mode > $O/bench_config2.json 2>> $O/bench.err
python3 bench.py --arch PointNetPP --steps 40 --warmup 5 --presteps 20 --no-cpu-baseline > $O/bench_config4.json 2>> $O/bench.err
tail -5 $O/bench.err
