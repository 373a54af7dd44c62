#!/bin/bash
# what the driver runs at round end: the full -m gpu suite, smoke(), the default bench line
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/final}; mkdir -p $O
timeout 3000 python3 -m pytest tests -x -q -m gpu > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log
tail -4 $O/tests.log
python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
T0=$(date +%s); python3 bench.py > $O/bench_default.json 2> $O/bench.err; echo "bench.py wall: $(( $(date +%s) - T0 )) s"
python3 -c "
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print({k: d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','scaling','dtype','data')})
print('roofline', d['roofline'])
print('cpu_baseline', d['cpu_baseline'])
print('other', d['other_wide_mode']['value'], d['other_wide_mode']['roofline']['frac'], d['other_wide_mode']['roofline'].get('traffic'))
"
