#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2i; mkdir -p $O
python3 tools/bench_widebwd.py > $O/widebwd.txt 2>&1
cat $O/widebwd.txt
timeout 1200 python3 -m pytest tests/test_gpu_pointnet.py tests/test_gpu_fullsize.py tests/test_gpu_attack.py -x -q -m gpu > $O/tests.log 2>&1
tail -5 $O/tests.log
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --single-mode > $O/bench_config2.json 2>> $O/bench.err
python3 bench.py --instances 32 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_proxy32.json 2>> $O/bench.err
python3 bench.py --npoint 4096 --knn 32 --steps 40 --warmup 5 --presteps 60 --no-cpu-baseline --single-mode > $O/bench_config5.json 2>> $O/bench.err
