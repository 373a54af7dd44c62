#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_geometry.py tests/test_gpu_forward_step.py tests/test_gpu_fullsize.py tests/test_gpu_attack.py -x -q -m gpu > $O/tests.log 2>&1
tail -15 $O/tests.log
python3 tools/bench_kernels.py > $O/kern_250.txt 2>&1
python3 tools/bench_kernels.py --B 32 > $O/kern_32.txt 2>&1
python3 tools/bench_kernels.py --N 4096 --k 32 --B 250 > $O/kern_4096.txt 2>&1
tail -2 $O/kern_*.txt
python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --single-mode > $O/bench_config2.json 2>> $O/bench.err
GEOA3_DETERMINISTIC=0 python3 bench.py --steps 200 --warmup 10 --no-cpu-baseline --single-mode > $O/bench_config2_atomic.json 2>> $O/bench.err
python3 bench.py --instances 32 --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_proxy32.json 2>> $O/bench.err
