#!/bin/bash
# The round's profile set: kernel traces (one stream: per-kernel durations undisturbed by the overlap), PMC passes
# (FETCH_SIZE / WRITE_SIZE, separate runs), and the un-profiled bench lines of every configuration.
# usage (on the GPU box): bash tools/gpu_profiles.sh <outdir>
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/prof}; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --single-mode --no-other-configs"
# configs[1]
GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2/trace -o t -- $B --steps 60 --warmup 10 > $O/c2_trace.log 2>&1
GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c2/fetch -o t -- $B --steps 30 --warmup 5 --presteps 100 > $O/c2_fetch.log 2>&1
GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c2/write -o t -- $B --steps 30 --warmup 5 --presteps 100 > $O/c2_write.log 2>&1
GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/c2/mfma -o t -- $B --steps 30 --warmup 5 --presteps 100 > $O/c2_mfma.log 2>&1
python3 tools/trace_timeline.py $O/c2/trace > $O/c2_timeline.txt
# configs[1] with every convolution on the fp32 MFMA (GEOA3_WIDE_MODE=f32: the `other_wide_mode` leg of the bench line)
GEOA3_WIDE_MODE=f32 GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c2f32/trace -o t -- $B --steps 30 --warmup 5 --presteps 60 > $O/c2f32_trace.log 2>&1
GEOA3_WIDE_MODE=f32 GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c2f32/fetch -o t -- $B --steps 20 --warmup 5 --presteps 60 > $O/c2f32_fetch.log 2>&1
GEOA3_WIDE_MODE=f32 GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c2f32/write -o t -- $B --steps 20 --warmup 5 --presteps 60 > $O/c2f32_write.log 2>&1
# configs[2] proxy: one rank's 32-instance shard
GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p32/trace -o t -- $B --instances 32 --no-proxy-full --steps 60 --warmup 10 > $O/p32_trace.log 2>&1
python3 tools/trace_timeline.py $O/p32/trace > $O/p32_timeline.txt
# configs[3]
GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c4/trace -o t -- $B --arch PointNetPP --steps 20 --warmup 5 --presteps 20 > $O/c4_trace.log 2>&1
GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c4/fetch -o t -- $B --arch PointNetPP --steps 10 --warmup 3 --presteps 10 > $O/c4_fetch.log 2>&1
GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c4/write -o t -- $B --arch PointNetPP --steps 10 --warmup 3 --presteps 10 > $O/c4_write.log 2>&1
GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/c4/mfma -o t -- $B --arch PointNetPP --steps 10 --warmup 3 --presteps 10 > $O/c4_mfma.log 2>&1
python3 tools/trace_timeline.py $O/c4/trace > $O/c4_timeline.txt
# configs[4]
GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5/trace -o t -- $B --npoint 4096 --knn 32 --steps 20 --warmup 5 --presteps 60 > $O/c5_trace.log 2>&1
GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/c5/fetch -o t -- $B --npoint 4096 --knn 32 --steps 10 --warmup 3 --presteps 60 > $O/c5_fetch.log 2>&1
GEOA3_PN2_SIDE=0 GEOA3_GEO_STREAM=0 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c5/write -o t -- $B --npoint 4096 --knn 32 --steps 10 --warmup 3 --presteps 60 > $O/c5_write.log 2>&1
python3 tools/trace_timeline.py $O/c5/trace > $O/c5_timeline.txt
# per-dispatch CSVs are large: keep the statistics and the counter collections only
find $O -name '*kernel_trace.csv' -delete
# un-profiled bench lines (both streams, CPU baselines)
python3 bench.py --steps 200 --warmup 10 > $O/bench_config2.json 2> $O/bench.err
python3 bench.py --instances 32 --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_proxy32.json 2>> $O/bench.err
python3 bench.py --arch PointNetPP --steps 40 --warmup 5 --presteps 20 > $O/bench_config4.json 2>> $O/bench.err
python3 bench.py --npoint 4096 --knn 32 --steps 40 --warmup 5 --presteps 60 > $O/bench_config5.json 2>> $O/bench.err
# the whole CLI run of configs[1] (10 binary steps x 500 iterations, 250 instances; synthetic data and weights) in both
# arithmetic modes: wall time of attack() incl. setup and the result download
for m in f16x2 f32; do
  T0=$(date +%s.%N)
  GEOA3_WIDE_MODE=$m python3 main_attack.py --attack GeoA3 --attack_label Untarget -b 250 --binary_max_steps 10 --iter_max_steps 500 --synthetic --quiet --out_root $O/exps_$m > $O/cli_full_$m.log 2>&1
  echo "wall clock of the whole process (interpreter start, weight packing, 250 .mat + .obj written): $(python3 -c "import time;print(round(time.time()-$T0,1))") s" >> $O/cli_full_$m.log
  rm -rf $O/exps_$m
  tail -4 $O/cli_full_$m.log
done
du -sh $O
