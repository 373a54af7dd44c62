#!/usr/bin/env python3
"""grid 1-NN vs all-pairs on CAD kinds at the loop's offset sizes: python tools/nn1_cad_probe.py [--n 1024]"""
import argparse, os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
ap = argparse.ArgumentParser(); ap.add_argument("--n", type=int, default=1024); ap.add_argument("--b", type=int, default=250)
a = ap.parse_args()
from geoa3_amd import ops
from geoa3_amd.data import CAD_KINDS, synthetic_cad_clouds, synthetic_clouds
def timeit(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / it
for kind in ("ellipsoid",) + tuple(CAD_KINDS):
    ori, _ = synthetic_clouds(a.b, a.n, seed=100) if kind == "ellipsoid" else synthetic_cad_clouds(a.b, a.n, seed=100, kinds=(kind,))
    ori = ori.cuda()
    for scale in (0.003, 0.05, 0.3):
        g = torch.Generator().manual_seed(1)
        adv = (ori + scale * torch.randn(ori.shape, generator=g).cuda()).contiguous()
        want = ops.nn1_pair(adv, ori)
        prior = (want[1].clone(), want[3].clone())
        tb = timeit(lambda: ops.nn1_pair(adv, ori))
        tg = timeit(lambda: ops.nn1_pair(adv, ori, method="grid", prior=prior))
        t1 = timeit(lambda: ops.nn1_pair(adv, ori, method="grid", prior=prior, both=False))
        print("%-11s offsets %.3f: all-pairs %7.1f us   grid (exact prior) %7.1f us   adv->ori only %7.1f us" % (kind, scale, tb, tg, t1))
