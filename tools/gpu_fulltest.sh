#!/bin/bash
# the driver's round-end sequence on the GPU box: the full -m gpu suite, smoke(), the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/full}; mkdir -p $O
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
tail -6 $O/tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
python3 bench.py > $O/bench_default.json 2> $O/bench.err; tail -c 600 $O/bench_default.json
