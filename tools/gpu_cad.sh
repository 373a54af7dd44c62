#!/bin/bash
# CAD-like synthetic inputs (geoa3_amd.data.synthetic_cad_clouds): the bit-exactness tests, then every configuration's
# bench line on them beside the ellipsoid line, a one-stream kernel trace of configs[1] on them, and the reference's
# default mode: --attack_label All -b 250 (2 250 attacks in one runner).   bash tools/gpu_cad.sh <outdir> [quick]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/cad}; mkdir -p $O
timeout 1800 python3 -m pytest tests/test_gpu_cad.py -x -q -m gpu 2>&1 | tail -5
B="python3 bench.py --no-cpu-baseline --single-mode --no-other-configs"
for data in ellipsoid cad; do
  $B --data $data --steps 200 --warmup 10 2>/dev/null | tail -1 > $O/bench_c2_$data.json
  $B --data $data --arch PointNetPP --steps 40 --warmup 5 --presteps 20 2>/dev/null | tail -1 > $O/bench_c4_$data.json
  $B --data $data --npoint 4096 --knn 32 --steps 40 --warmup 5 --presteps 60 2>/dev/null | tail -1 > $O/bench_c5_$data.json
  for c in c2 c4 c5; do python3 -c "
import json; d=json.loads(open('$O/bench_${c}_$data.json').read()); k=d.get('kernels_ms') or {}
print('$c $data ms_per_step', d['ms_per_step'], 'cd', (d.get('cd_kernel') or {}).get('avg_launch_us'), 'knn', k.get('knn'), 'nn1', k.get('nn1_pair'))"; done
done
for data in ellipsoid cad; do
  GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c2_$data -o t -- $B --data $data --steps 60 --warmup 10 > $O/trace_c2_$data.log 2>&1
  GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c4_$data -o t -- $B --data $data --arch PointNetPP --steps 20 --warmup 5 --presteps 20 > $O/trace_c4_$data.log 2>&1
  GEOA3_GEO_STREAM=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c5_$data -o t -- $B --data $data --npoint 4096 --knn 32 --steps 20 --warmup 5 --presteps 60 > $O/trace_c5_$data.log 2>&1
done
find $O -name '*kernel_trace.csv' -delete
python3 - $O <<'P'
import csv, glob, re, sys
O = sys.argv[1]
for c in ("c2", "c4", "c5"):
    rows = {}
    for data in ("ellipsoid", "cad"):
        f = glob.glob("%s/trace_%s_%s/**/t_kernel_stats.csv" % (O, c, data), recursive=True)
        if not f:
            continue
        for r in csv.DictReader(open(f[0])):
            rows.setdefault(re.sub(r"\(anonymous namespace\)::", "", r["Name"])[:70], {})[data] = (float(r["AverageNs"]) / 1e3, int(r["Calls"]))
    print("== %s: kernels whose average moves by more than 10 %% (us: ellipsoid -> cad)" % c)
    for k, v in sorted(rows.items(), key=lambda kv: -kv[1].get("cad", (0, 0))[0] * kv[1].get("cad", (0, 0))[1]):
        if "ellipsoid" in v and "cad" in v and v["ellipsoid"][1] > 5 and abs(v["cad"][0] / v["ellipsoid"][0] - 1) > 0.10:
            print("  %-70s %8.1f -> %8.1f  (x%.2f, %d calls)" % (k, v["ellipsoid"][0], v["cad"][0], v["cad"][0] / v["ellipsoid"][0], v["cad"][1]))
P
if [ "$2" != "quick" ]; then
  # the reference's default mode (main_attack.py:330): every instance against its nine other classes, b = 2 250 in one runner
  T0=$(date +%s.%N)
  python3 main_attack.py --attack GeoA3 --attack_label All -b 250 --binary_max_steps 10 --iter_max_steps 500 --synthetic --synthetic_kind cad --quiet --out_root $O/exps_all > $O/cli_all_cad.log 2>&1
  echo "wall clock of the whole process: $(python3 -c "import time;print(round(time.time()-$T0,1))") s" >> $O/cli_all_cad.log
  find $O/exps_all -name '*.mat' | wc -l >> $O/cli_all_cad.log
  rm -rf $O/exps_all
  tail -6 $O/cli_all_cad.log
fi
