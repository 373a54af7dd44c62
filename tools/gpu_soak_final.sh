#!/bin/bash
# the replay soak at ten times the suite's counts on the library as built (every hand-over of the loop, both data kinds,
# both victims): one iteration replayed from the same device state, every buffer compared bit for bit with the first replay
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/soak_final}; mkdir -p $O
run() { echo "== $*"; timeout 1500 python3 tools/iteration_replay_soak.py "$@" 2>&1 | tail -1; }
{
run --b 250 --iters 30000
run --b 32 --iters 30000
run --b 250 --n 1024 --data cad --iters 15000 --presteps 100
run --b 40 --n 1024 --data cad --iters 10000 --presteps 100
run --arch PointNetPP --b 250 --iters 3000
run --arch PointNetPP --b 250 --data cad --iters 2000
run --b 250 --n 4096 --k 32 --iters 2000 --presteps 40
} > $O/soak.txt 2>&1
cat $O/soak.txt
