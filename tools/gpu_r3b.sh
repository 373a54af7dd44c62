#!/bin/bash
# geometry / attack parity + bench lines after a geometry-kernel change
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r3b}; mkdir -p $O
timeout 1500 python3 -m pytest tests/test_gpu_geometry.py tests/test_gpu_forward_step.py tests/test_gpu_attack.py tests/test_gpu_attack_aux.py tests/test_gpu_aux.py tests/test_gpu_fullsize.py -x -q -m gpu > $O/tests.log 2>&1; echo "rc=$?" >> $O/tests.log; tail -25 $O/tests.log
python3 bench.py --no-cpu-baseline --single-mode > $O/c2.json 2> $O/err.log
python3 bench.py --no-cpu-baseline --instances 32 --steps 300 > $O/p32.json 2>> $O/err.log
python3 bench.py --no-cpu-baseline --single-mode --npoint 4096 --knn 32 --steps 40 --warmup 5 --presteps 100 > $O/c5.json 2>> $O/err.log
tail -5 $O/err.log
python3 - <<PY
import json
for f in ("c2","p32","c5"):
    try:
        d=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, d["value"], d["ms_per_step"], d.get("host_enqueue_ms_per_step"), d.get("kernels_ms"), (d.get("strong_scaling_proxy") or {}).get("fraction_of_linear"))
    except Exception as e:
        print(f, "failed", e)
PY
