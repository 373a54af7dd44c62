#!/bin/bash
# SQ counters of the PointNet++ level-1 kernels alone (tools/bench_sa1.py): tools/gpu_sa1_pmc.sh outdir lib.so [lib.so ...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=$1; shift; mkdir -p $O
[ -f $O/counters.txt ] || rocprofv3 -L > $O/counters.txt 2>&1
for l in "$@"; do
  tag=$(echo $l | tr '/.' '__')
  export GEOA3_LIB_PATH=$PWD/$l
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY" \
             "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_INSTS_MFMA" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_IFETCH" \
             "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_FLAT SQ_INSTS_SALU" "GRBM_GUI_ACTIVE SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_COEXEC_CYCLES"; do
    i=$((i+1))
    rocprofv3 --pmc $set --output-format csv -d $O/$tag/$i -o t -- python3 tools/bench_sa1.py --iters 10 > $O/$tag.$i.log 2>&1
  done
  python3 - $O/$tag <<'P'
import csv, sys, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sa1_fwd" in r["Kernel_Name"] or "sa1_bwd" in r["Kernel_Name"]:
            agg["sa1_fwd" if "sa1_fwd" in r["Kernel_Name"] else "sa1_bwd"][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in agg.items():
    print(sys.argv[1], k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())})
P
  find $O/$tag -name '*counter_collection.csv' -delete
done
