#!/bin/bash
# the full -m gpu suite + smoke + bench lines (default, shard proxy)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/full}; mkdir -p $O
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
tail -8 $O/tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
for a in "--single-mode" "--instances 32 --steps 300"; do
python3 bench.py --no-cpu-baseline $a 2>> $O/err.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config']['instances_per_gpu'], d['config']['npoint'], d['value'], d['ms_per_step'], d.get('host_enqueue_ms_per_step'), (d.get('strong_scaling_proxy') or {}).get('fraction_of_linear'))"
done
