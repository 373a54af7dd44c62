#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/pn}; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_pointnet.py tests/test_gpu_forward_step.py tests/test_gpu_attack.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -6 $O/tests.log
for tp in 1 0; do
GEOA3_TWO_PASS=$tp python3 bench.py --no-cpu-baseline --single-mode 2>> $O/err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('two_pass=$tp', d['value'], d['ms_per_step'], d['kernels_ms']['conv5_wide_max'], d['roofline']['frac'])"
done
