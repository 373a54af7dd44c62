#!/bin/bash
# The round's profile set: kernel traces (one stream: per-kernel durations undisturbed by the overlap), PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs), and the un-profiled bench lines of every configuration.
# usage (on the GPU box): bash tools/gpu_profiles.sh <outdir>
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/prof}; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --single-mode"
# configs[1]
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2/trace -o t -- $B --steps 60 --warmup 10 > $O/c2_trace.log 2>&1
GEOA3_GEO_STREAM=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c2/fetch -o t -- $B --steps 30 --warmup 5 --presteps 100 > $O/c2_fetch.log 2>&1
GEOA3_GEO_STREAM=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c2/write -o t -- $B --steps 30 --warmup 5 --presteps 100 > $O/c2_write.log 2>&1
GEOA3_GEO_STREAM=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/c2/mfma -o t -- $B --steps 30 --warmup 5 --presteps 100 > $O/c2_mfma.log 2>&1
python3 tools/trace_timeline.py $O/c2/trace > $O/c2_timeline.txt
# configs[2] proxy: one rank's 32-instance shard
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p32/trace -o t -- $B --instances 32 --no-proxy-full --steps 60 --warmup 10 > $O/p32_trace.log 2>&1
python3 tools/trace_timeline.py $O/p32/trace > $O/p32_timeline.txt
# configs[3]
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4/trace -o t -- $B --arch PointNetPP --steps 20 --warmup 5 --presteps 20 > $O/c4_trace.log 2>&1
GEOA3_GEO_STREAM=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c4/fetch -o t -- $B --arch PointNetPP --steps 10 --warmup 3 --presteps 10 > $O/c4_fetch.log 2>&1
GEOA3_GEO_STREAM=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c4/write -o t -- $B --arch PointNetPP --steps 10 --warmup 3 --presteps 10 > $O/c4_write.log 2>&1
python3 tools/trace_timeline.py $O/c4/trace > $O/c4_timeline.txt
# configs[4]
GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5/trace -o t -- $B --npoint 4096 --knn 32 --steps 20 --warmup 5 --presteps 60 > $O/c5_trace.log 2>&1
GEOA3_GEO_STREAM=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c5/fetch -o t -- $B --npoint 4096 --knn 32 --steps 10 --warmup 3 --presteps 60 > $O/c5_fetch.log 2>&1
GEOA3_GEO_STREAM=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c5/write -o t -- $B --npoint 4096 --knn 32 --steps 10 --warmup 3 --presteps 60 > $O/c5_write.log 2>&1
python3 tools/trace_timeline.py $O/c5/trace > $O/c5_timeline.txt
# per-dispatch CSVs are large: keep the statistics and the counter collections only
find $O -name '*kernel_trace.csv' -delete
# un-profiled bench lines (both streams, CPU baselines)
python3 bench.py --steps 200 --warmup 10 > $O/bench_config2.json 2> $O/bench.err
python3 bench.py --instances 32 --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_proxy32.json 2>> $O/bench.err
python3 bench.py --arch PointNetPP --steps 40 --warmup 5 --presteps 20 > $O/bench_config4.json 2>> $O/bench.err
python3 bench.py --npoint 4096 --knn 32 --steps 40 --warmup 5 --presteps 60 > $O/bench_config5.json 2>> $O/bench.err
du -sh $O
