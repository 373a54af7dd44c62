#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/geo5}; mkdir -p $O
for D in 0 1 2 4 6 7; do
export GEOA3_GEO_DBG=$D
echo "=== DBG=$D"
bash tools/gpu_geotrace.sh $O/d$D noc5 | grep -E "==|geo_"
done
