"""How the native PointNet++ input gradient differs from the CPU oracle's on CAD clouds: element mismatches, their share of
the gradient's norm, and whether they are pooling-winner flips (the per-cloud SUM of the difference stays ~0: the same
gradient mass lands on a neighbouring point)."""
import sys, os
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import numpy as np
import torch
from oracle import pointnet2_oracle as P2
from geoa3_amd import pointnet2
from geoa3_amd.data import synthetic_cad_clouds
from tests.test_gpu_cad import _merge_twins

sd = P2.make_pn2_state_dict(0)
net = pointnet2.PointNet2ClassificationSSG(use_xyz=True, use_normal=False)
net.load_state_dict(sd)
net = net.cuda().eval()
for p in net.parameters():
    p.requires_grad_(False)
for seed in (9, 10, 11):
    pts, _ = synthetic_cad_clouds(5, 1024, seed=seed)
    xa = pts.cuda().requires_grad_()
    la = net(xa)
    w = torch.randn(la.shape, generator=torch.Generator().manual_seed(1))
    (la * w.cuda()).sum().backward()
    xo = pts.clone().requires_grad_()
    lo = P2.pointnet2_ssg_forward(sd, xo)
    (lo * w).sum().backward()
    want, got = _merge_twins(xo.grad.numpy(), pts.numpy()), _merge_twins(xa.grad.cpu().numpy(), pts.numpy())
    bad = ~np.isclose(got, want, rtol=2e-3, atol=2e-4 * np.abs(want).max())
    d = got - want
    print("seed", seed, "logits maxdiff %.2e" % np.abs(la.detach().cpu().numpy() - lo.detach().numpy()).max(),
          "bad elements", int(bad.sum()), "of", bad.size, "points", int(bad.any(axis=1).sum()),
          "rel L2 %.2e" % (np.linalg.norm(d) / np.linalg.norm(want)),
          "per-cloud |sum d| / sum |d|:", np.round(np.abs(d.sum(axis=2)).sum(axis=1) / np.abs(d).sum(axis=(1, 2)), 3))
