#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2l; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_geometry.py -x -q -m gpu -k "slab or knn" > $O/tests.log 2>&1
tail -12 $O/tests.log
python3 tools/bench_kernels.py > $O/kern_250.txt 2>&1
python3 tools/bench_kernels.py --N 4096 --k 32 --B 250 > $O/kern_4096.txt 2>&1
python3 bench.py --npoint 4096 --knn 32 --steps 40 --warmup 5 --presteps 60 --no-cpu-baseline --single-mode > $O/bench_config5.json 2>> $O/bench.err
GEOA3_KNN_METHOD=2 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --single-mode > $O/bench_config2_grid.json 2>> $O/bench.err
