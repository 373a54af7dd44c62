#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
timeout 2400 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1
tail -15 $O/tests.log
python3 bench.py --steps 200 --warmup 10 > $O/bench_config2.json 2> $O/bench.err
python3 bench.py --instances 32 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_proxy32.json 2>> $O/bench.err
python3 bench.py --arch PointNetPP --steps 40 --warmup 5 --presteps 20 > $O/bench_config4.json 2>> $O/bench.err
python3 bench.py --npoint 4096 --knn 32 --steps 40 --warmup 5 --presteps 60 > $O/bench_config5.json 2>> $O/bench.err
tail -5 $O/bench.err
