#!/usr/bin/env python3
"""Stand-alone timing of the seeded K-NN on fixed inputs (the prior is the table of a cloud 0.005 away)."""
import os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from tools.bench_kernels import timeit
from geoa3_amd import ops
from geoa3_amd.data import synthetic_clouds
B, N, K = int(os.environ.get("B", 250)), 1024, 17
ori, _ = synthetic_clouds(B, N, seed=100)
g = torch.Generator().manual_seed(0)
a1 = (ori + 0.02 * torch.randn(B, 3, N, generator=g)).cuda().contiguous()
a2 = (a1 + 0.003 * torch.randn(B, 3, N, generator=g).cuda()).contiguous()
_, prior = ops.knn_planar(a1, a1, K)
scratch = ops.knn_self_scratch(B, N, a1.device)
out = (torch.empty(B, N, K, device="cuda"), torch.empty(B, N, K, device="cuda", dtype=torch.int32))
t = timeit(lambda: ops.knn_self_planar(a2, K, prior=prior, scratch=scratch, out=out, method=4), 50, 5)
print("B=%d: %.1f us (slab_bin + knn)" % (B, t))
