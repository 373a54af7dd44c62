#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/geo4}; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_geometry.py tests/test_gpu_forward_step.py tests/test_gpu_attack.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -5 $O/tests.log
bash tools/gpu_geotrace.sh $O noc5 | grep -E "==|geo_|nn1|knn|slab"
for a in "--single-mode" "--instances 32 --steps 300"; do
python3 bench.py --no-cpu-baseline $a 2>> $O/err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config']['instances_per_gpu'], d['config']['npoint'], d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), (d.get('strong_scaling_proxy') or {}).get('fraction_of_linear'))"
done
