#!/bin/bash
# tools/bench_sa1.py under builds of the library: tools/gpu_sa1_micro.sh libA.so libB.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for rep in 1 2; do for l in "$@"; do
  echo "$l $(GEOA3_LIB_PATH=$PWD/$l python3 tools/bench_sa1.py $([ $rep = 1 ] && echo --check) 2>&1 | tail -1)"
done; done
