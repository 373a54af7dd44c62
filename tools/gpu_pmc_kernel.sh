#!/bin/bash
# SQ counters of one kernel of configs[1]: bash tools/gpu_pmc_kernel.sh <kernel-name-substring> <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
K=${1:-knn_slab}; O=${2:-gpurun_out/pmck}; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --single-mode --steps 6 --warmup 2 --presteps 100"
GEOA3_GEO_STREAM=0 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/a -o t -- $B > $O/a.log 2>&1
GEOA3_GEO_STREAM=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $O/b -o t -- $B > $O/b.log 2>&1
GEOA3_GEO_STREAM=0 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD --output-format csv -d $O/c -o t -- $B > $O/c.log 2>&1
python3 - $O $K <<'P'
import csv,sys,glob,collections
O,K=sys.argv[1],sys.argv[2]
for sub in 'abc':
    fs=glob.glob(O+'/'+sub+'/**/*counter_collection.csv', recursive=True)
    if not fs: print(sub,'no csv'); continue
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if K in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(sub,k,'n',len(v),'avg %.4g'%(sum(v)/len(v)))
P
find $O -name '*counter_collection.csv' -delete
