#!/usr/bin/env python3
"""Timing of the self K-NN kernels on iterates of a real attack run + the size of the candidate sets (points within the
radius the previous iteration's neighbours give)."""
import os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from bench import cfg_full_geoa3
from geoa3_amd import ops
from geoa3_amd.attack import AttackRunner
from geoa3_amd.data import synthetic_clouds, synthetic_state_dict
from geoa3_amd.pointnet import PointNet
from tools.bench_kernels import timeit

dev = torch.device("cuda")
B, N, K = 250, 1024, 17
net = PointNet(40)
net.load_state_dict(synthetic_state_dict(40, seed=0, device=dev))
net = net.to(dev).eval()
ori, nrm = synthetic_clouds(B, N, seed=100)
ori, nrm = ori.to(dev), nrm.to(dev)
gt = net(ori).argmax(1)
r = AttackRunner(net, B, N, cfg_full_geoa3(400), dev, global_batch=250)
r.setup(ori, nrm, gt, gt)
r.begin_search_step((torch.randn(B, 3, N) * 1e-3).to(dev))
for st in range(int(os.environ.get("STEPS", "150"))):
    r.step(st, 0)
x_prev = r.t["x"].clone()
prior = r.t["knn"][r.knn_cur].clone()
r.step(150, 0)
x = r.t["x"].clone()
scratch = ops.knn_self_scratch(B, N, dev)
for m in (1, 3, 4, 2):
    us = timeit(lambda: ops.knn_self_planar(x, K, prior=prior, scratch=scratch, method=m), 10)
    print("method %d: %.1f us" % (m, us))
# candidates within tau
P = x.permute(0, 2, 1)
nb = torch.gather(P.unsqueeze(1).expand(B, N, N, 3), 2, prior.long().unsqueeze(-1).expand(B, N, K, 3))
tau = ((P.unsqueeze(2) - nb) ** 2).sum(-1).amax(2)
d = torch.cdist(P, P) ** 2
n = (d <= tau.unsqueeze(2) * 1.000001).sum(2)
print("candidates within the radius: mean %.1f, p99 %d, max %d; fraction > 32: %.4f, > 48: %.5f" %
      (n.float().mean(), int(n.float().quantile(0.99)), int(n.max()), float((n > 32).float().mean()), float((n > 48).float().mean())))
ext = (P.amax(1) - P.amin(1)).amax(1)
print("extent of the clouds: mean %.2f max %.2f; sqrt(tau) mean %.3f" % (ext.mean(), ext.max(), tau.sqrt().mean()))
