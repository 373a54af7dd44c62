#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/k}; mkdir -p $O
timeout 1200 python3 -m pytest tests/test_gpu_geometry.py -x -q -m gpu > $O/tests.log 2>&1; tail -3 $O/tests.log
python3 tools/bench_kernels.py --iters 10 2>/dev/null | tr ',' '\n' | grep knn
python3 tools/bench_kernels.py --N 4096 --k 32 --iters 5 2>/dev/null | tr ',' '\n' | grep knn
python3 bench.py --no-cpu-baseline --npoint 4096 --knn 32 --steps 40 --warmup 5 --presteps 60 > $O/c5.json 2> $O/err.log
python3 -c "
import json
d=json.loads(open('$O/c5.json').read().strip().splitlines()[-1])
print('c5', d['value'], d['ms_per_step'])
"
