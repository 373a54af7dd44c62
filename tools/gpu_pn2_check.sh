#!/bin/bash
# everything that exercises the PointNet++ victim, then the configs[3] bench line + one-stream kernel trace
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 2400 python3 -m pytest tests/test_gpu_pointnet2.py tests/test_gpu_attack.py tests/test_gpu_replay.py tests/test_gpu_longrun.py -x -q -m gpu -k "pointnet2 or pn2 or pointnetpp or PointNetPP" 2>&1 | tail -5
bash tools/gpu_c4.sh gpurun_out/c4 2>&1 | tail -20
