#!/bin/bash
# VALU utilisation per kernel of configs[1]: SQ_ACTIVE_INST_VALU (quad-cycles) * 4 / (SQ_BUSY_CYCLES-based SIMD cycles), and
# VALU instructions per wavefront.  bash tools/gpu_valu_util.sh outdir [more bench.py arguments, e.g. --arch PointNetPP]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/valu}; mkdir -p $O; shift
B="python3 bench.py --no-cpu-baseline --single-mode --steps 6 --warmup 2 --presteps 100 $*"
GEOA3_GEO_STREAM=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS --output-format csv -d $O/a -o t -- $B > $O/a.log 2>&1
GEOA3_GEO_STREAM=0 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/b -o t -- $B > $O/b.log 2>&1
GEOA3_GEO_STREAM=0 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/c -o t -- $B > $O/c.log 2>&1
python3 - $O <<'P'
import csv, sys, glob, collections, re
O = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if "GRBM_GUI_ACTIVE" not in m or m.get("SQ_WAVES", 0) == 0:
        continue
    simd_cycles = m["GRBM_GUI_ACTIVE"] / 8 * 256 * 4   # the counter sums the eight XCDs
    rows.append((m["GRBM_GUI_ACTIVE"], re.sub(r"\(anonymous namespace\)::", "", k)[:70], m))
for gui, name, m in sorted(rows, key=lambda r: -r[0])[:24]:
    sc = gui / 8 * 256 * 4   # GRBM_GUI_ACTIVE sums the eight XCDs
    print("%-70s cyc %8.0f  VALU busy %4.1f%%  MFMA busy %4.1f%%  LDS %4.1f%%  SALU %4.1f%%  VALU/wave %6.0f  waves %7.0f" % (
        name, gui, 400 * m.get("SQ_ACTIVE_INST_VALU", 0) / sc, 100 * m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / sc,
        400 * m.get("SQ_ACTIVE_INST_LDS", 0) / sc, 100 * m.get("SQ_INST_CYCLES_SALU", 0) / sc,
        m.get("SQ_INSTS_VALU", 0) / m["SQ_WAVES"], m["SQ_WAVES"]))
P
find $O -name '*counter_collection.csv' -delete
