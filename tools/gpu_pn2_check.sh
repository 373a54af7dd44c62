#!/bin/bash
# PointNet++ path after a kernel / orchestration change: its tests, the replay soak at 10x, the long runs, the bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_pointnet2.py tests/test_gpu_cad.py -x -q -m gpu 2>&1 | tail -3
GEOA3_SOAK_SCALE=10 timeout 1500 python3 -m pytest tests/test_gpu_replay.py -x -q -m gpu -k PointNetPP 2>&1 | tail -3
timeout 1500 python3 tools/iteration_replay_soak.py --arch PointNetPP --b 250 --iters 2000 2>&1 | tail -3
timeout 2400 python3 -m pytest tests/test_gpu_longrun.py tests/test_gpu_attack.py -x -q -m gpu -k "pn2 or pointnetpp or PointNetPP" 2>&1 | tail -3
for i in 1 2; do
python3 bench.py --no-cpu-baseline --arch PointNetPP --steps 40 --warmup 5 --presteps 20 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('c4', d['value'], d['ms_per_step'], d['kernels_ms'].get('sa1_bwd'), d['kernels_ms'].get('sa1_fwd'))"
done
