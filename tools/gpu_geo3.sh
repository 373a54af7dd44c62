#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/geo3}; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_geometry.py tests/test_gpu_forward_step.py tests/test_gpu_attack.py tests/test_gpu_attack_aux.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -25 $O/tests.log
for R in 128 256 64; do
export GEOA3_GEO_R=$R
echo "=== R=$R"
bash tools/gpu_geotrace.sh $O/r$R noc5 | grep -E "==|geo_"
done
unset GEOA3_GEO_R
for a in "--single-mode" "--instances 32 --steps 300"; do
python3 bench.py --no-cpu-baseline $a 2>> $O/err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config']['instances_per_gpu'], d['config']['npoint'], d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), (d.get('strong_scaling_proxy') or {}).get('fraction_of_linear'))"
done
