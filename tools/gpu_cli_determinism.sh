#!/bin/bash
# the whole configs[1] CLI run twice (10 x 500 iterations, 250 instances): are the 250 adversarial clouds bit-identical?
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=/tmp/clidet; rm -rf $O; mkdir -p $O
for r in a b; do
  python3 main_attack.py --attack GeoA3 --attack_label Untarget -b 250 --binary_max_steps 10 --iter_max_steps 500 --synthetic --quiet --out_root $O/$r > $O/$r.log 2>&1
  tail -1 $O/$r.log
done
python3 - <<'P'
import glob, os, scipy.io as sio, numpy as np
A = sorted(glob.glob("/tmp/clidet/a/**/*.mat", recursive=True)); B = sorted(glob.glob("/tmp/clidet/b/**/*.mat", recursive=True))
print(len(A), len(B), "mat files")
same = 0
for fa, fb in zip(A, B):
    assert os.path.basename(fa) == os.path.basename(fb), (fa, fb)
    a, b = sio.loadmat(fa), sio.loadmat(fb)
    same += int(np.array_equal(a["adversary_point_clouds"], b["adversary_point_clouds"]))
print("bit-identical adversarial clouds: %d of %d" % (same, len(A)))
P
