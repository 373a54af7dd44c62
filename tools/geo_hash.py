#!/usr/bin/env python3
"""The objective's values and gradient (ops.geo_loss_grad, deterministic) hashed for a few shapes, the fixed-point kernel's
among them: run under two builds of the library (GEOA3_LIB_PATH) to check that a kernel change left every bit alone."""
import hashlib, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from oracle import geoa3_oracle as O
from geoa3_amd import ops
from tests.test_gpu_geometry import _objective_inputs

h = hashlib.sha256()
for B, N, k in ((3, 4096, 32), (2, 1500, 16), (3, 1024, 40), (4, 1024, 16), (2, 300, 8)):
    ori, nrm = O.make_synthetic_clouds(B, N, seed=N + k)
    adv = ori + 0.02 * torch.randn(B, 3, N, generator=torch.Generator().manual_seed(N))
    kw, advD, oriD = _objective_inputs(ops, adv, ori, nrm, k)
    out = ops.geo_loss_grad(advD, oriD, deterministic=True, want_kappa=True, **kw)
    for n in ("constrain", "grad", "kappa_adv", "dis_loss", "hd_loss", "curv_loss"):
        h.update(out[n].cpu().numpy().tobytes())
print("objective", h.hexdigest())
