#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 1200 python3 -m pytest tests/test_gpu_library.py tests/test_gpu_geometry.py tests/test_gpu_pointnet.py tests/test_gpu_forward_step.py -x -q -m gpu 2>&1 | tail -25
