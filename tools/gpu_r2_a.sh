#!/bin/bash
# round 2, call A: where the 32-instance shard (configs[2] per-rank regime) stands
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r2a; mkdir -p $O
python3 bench.py --instances 32 --steps 300 --warmup 50 --no-cpu-baseline --single-mode > $O/b32.json 2> $O/b32.err
GEOA3_GEO_STREAM=0 python3 bench.py --instances 32 --steps 300 --warmup 50 --no-cpu-baseline --single-mode > $O/b32_1s.json 2>> $O/b32.err
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats -d $O/trace32 -o t -- python3 bench.py --instances 32 --steps 60 --warmup 15 --no-cpu-baseline --single-mode > $O/trace32.log 2>&1
python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --single-mode > $O/b250.json 2>> $O/b32.err
for b in 16 64 125; do python3 bench.py --instances $b --steps 200 --warmup 30 --no-cpu-baseline --single-mode > $O/b$b.json 2>> $O/b32.err; done
find $O -name '*.csv' ! -name '*kernel_stats.csv' -delete; find $O -name '*.json' -path '*trace32*' -delete; find $O -name '*.rocpd' -delete; find $O -name '*.db' -delete
ls -la $O $O/trace32/* | head -30
cat $O/b32.json | cut -c1-400
