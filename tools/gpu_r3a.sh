#!/bin/bash
# round 3, first call: the new tests (group_points_grad at 2048/4096 points, bench self-spawn, long-horizon parity) +
# the default bench line + the 32-instance shard proxy as this round's starting point
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/r3a}; mkdir -p $O
timeout 900 python3 -m pytest tests/test_gpu_pointnet2.py tests/test_gpu_multirank.py -x -q -m gpu > $O/tests_a.log 2>&1; echo "rc=$?" >> $O/tests_a.log; tail -5 $O/tests_a.log
timeout 1500 python3 -m pytest tests/test_gpu_longrun.py -q -m gpu > $O/tests_long.log 2>&1; echo "rc=$?" >> $O/tests_long.log; tail -40 $O/tests_long.log
python3 bench.py --no-cpu-baseline --single-mode > $O/c2.json 2> $O/err.log; tail -c 1500 $O/c2.json
python3 bench.py --no-cpu-baseline --instances 32 --steps 300 > $O/p32.json 2>> $O/err.log; tail -c 900 $O/p32.json
