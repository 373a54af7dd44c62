#!/bin/bash
# GEOA3_FUSE_CHAIN on/off must give the same bits (logits, input gradient); then tests and the A/B on the bench lines
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
GEOA3_FUSE_CHAIN=$v python3 - <<'P'
import torch, hashlib, os
from geoa3_amd.pointnet import PointNet
torch.manual_seed(0)
net = PointNet(40).cuda().eval()
for B,N in ((3,256),(5,1000),(32,1024)):
    x = (torch.randn(B,3,N,device='cuda')*0.5).requires_grad_(True)
    logits = net(x)
    g = torch.randn(B,40,device='cuda')
    logits.backward(g)
    h = hashlib.sha1()
    h.update(logits.detach().cpu().numpy().tobytes()); h.update(x.grad.cpu().numpy().tobytes())
    print('chain', os.environ['GEOA3_FUSE_CHAIN'], B, N, h.hexdigest())
P
done
python3 -m pytest tests/test_gpu_pointnet.py tests/test_gpu_attack.py -m gpu -x -q 2>&1 | tail -3
bash tools/gpu_ab.sh GEOA3_FUSE_CHAIN "0 1 0 1"
