cd $GRAFT_REPO_ROOT
for rep in 1 2 3 4; do echo "$(python3 tools/loop_determinism_final.py 30 80 2>&1 | tail -1)"; done
for rep in 1 2; do echo "$(NPTS=4096 KNN=32 python3 tools/loop_determinism_final.py 8 24 2>&1 | tail -1)"; done
for rep in 1 2; do echo "$(NB=32 python3 tools/loop_determinism_final.py 40 80 2>&1 | tail -1)"; done
echo "$(GEOA3_WIDE_MODE=f32 python3 tools/loop_determinism_final.py 20 40 2>&1 | tail -1)"
echo "$(ARCH=PointNetPP python3 tools/loop_determinism_final.py 12 40 2>&1 | tail -1)"
