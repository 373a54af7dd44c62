#!/bin/bash
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace32 -o t -- python3 bench.py --instances 32 --steps 100 --warmup 20 --no-cpu-baseline --single-mode > $O/trace32.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace32_2s -o t -- python3 bench.py --instances 32 --steps 100 --warmup 20 --no-cpu-baseline --single-mode > $O/trace32_2s.log 2>&1
ls -la $O/trace32 $O/trace32_2s
# keep the per-dispatch trace of ~3 iterations only
for d in trace32 trace32_2s; do f=$(ls $O/$d/*kernel_trace.csv); head -1 $f > $O/$d/kt_head.csv; awk -F, 'NR>1' $f | sort -t, -k10,10n | sed -n '4000,4400p' >> $O/$d/kt_head.csv; rm $f; done
