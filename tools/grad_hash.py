#!/usr/bin/env python3
"""Logits and input gradient of the PointNet victim hashed for a few shapes and both arithmetic modes: run under two builds
of the library (GEOA3_LIB_PATH) to check that a kernel change left every bit alone."""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from geoa3_amd.data import synthetic_clouds, synthetic_state_dict
from geoa3_amd.pointnet import PointNet

dev = torch.device("cuda")
h = hashlib.sha256()
for mode in ("f16x2", "f32"):
    net = PointNet(40)
    net.load_state_dict(synthetic_state_dict(40, seed=0, device=dev))
    net.wide_mode = mode
    net = net.to(dev).eval()
    for B, N in ((3, 256), (5, 1000), (2, 77), (9, 1024), (2, 1500), (1, 4096), (32, 1024), (250, 1024)):
        ori, _ = synthetic_clouds(B, N, seed=B + N)
        x = ori.to(dev).requires_grad_()
        w = torch.randn(B, 40, generator=torch.Generator().manual_seed(1)).to(dev)
        lg = net(x)
        (lg * w).sum().backward()
        h.update(lg.detach().cpu().numpy().tobytes())
        h.update(x.grad.cpu().numpy().tobytes())
print("PointNet", h.hexdigest())
from geoa3_amd.pointnet2 import PointNet2ClassificationSSG
h = hashlib.sha256()
torch.manual_seed(0)
net = PointNet2ClassificationSSG(use_xyz=True, use_normal=False).to(dev).eval()
for B, N in ((3, 1024), (2, 700), (16, 1024)):
    ori, _ = synthetic_clouds(B, N, seed=B + N)
    x = ori.to(dev).requires_grad_()
    w = torch.randn(B, 40, generator=torch.Generator().manual_seed(1)).to(dev)
    lg = net(x)
    (lg * w).sum().backward()
    h.update(lg.detach().cpu().numpy().tobytes())
    h.update(x.grad.cpu().numpy().tobytes())
print("PointNet++", h.hexdigest())
