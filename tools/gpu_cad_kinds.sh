#!/bin/bash
# which CAD kind hurts which geometry kernel: bench.py --data cad restricted to one kind at a time (and without duplicates)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-cpu-baseline --single-mode --steps 60 --warmup 5"
show() { python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernels_ms') or {}
print('$1', 'ms/step', d['ms_per_step'], 'nn1 %.1f knn %.1f geo %.1f us' % (1e3*(k.get('nn1_pair') or 0), 1e3*(k.get('knn') or 0), 1e3*(k.get('geo_loss_grad') or 0)))"; }
$B --data ellipsoid $* 2>/dev/null | show ellipsoid
for kind in box table clusters rod ellipsoid2; do
  $B --data cad --cad-kinds $kind $* 2>/dev/null | show "cad:$kind"
  $B --data cad --cad-kinds $kind --cad-duplicates 0 $* 2>/dev/null | show "cad:$kind,nodup"
done
