#!/usr/bin/env python3
"""The grid 1-NN on the attack loop's OWN steady-state iterates, per CAD kind: python tools/nn1_state_probe.py [--n 1024 --knn 16]
Runs `--steps` iterations of configs[1]/[4] on one kind at a time and reports the offsets' size, the 1-NN distances and the
kernel's duration on that state (exact priors), against the all-pairs kernel."""
import argparse, os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench as BN
ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1024); ap.add_argument("--knn", type=int, default=16)
ap.add_argument("--b", type=int, default=50); ap.add_argument("--dump", default=""); ap.add_argument("--steps", type=int, default=160)
a = ap.parse_args()
from geoa3_amd import ops
from geoa3_amd.attack import AttackRunner
from geoa3_amd.data import CAD_KINDS, synthetic_cad_clouds, synthetic_clouds, synthetic_state_dict
from geoa3_amd.pointnet import PointNet
dev = torch.device("cuda", 0)
net = PointNet(40); net.load_state_dict(synthetic_state_dict(40, seed=0, device=dev)); net = net.to(dev).eval()
def timeit(fn, it=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / it
for kind in ("ellipsoid",) + tuple(CAD_KINDS):
    ori, nrm = synthetic_clouds(a.b, a.n, seed=100) if kind == "ellipsoid" else synthetic_cad_clouds(a.b, a.n, seed=100, kinds=(kind,))
    ori, nrm = ori.to(dev), nrm.to(dev)
    with torch.no_grad(): gt = net(ori).argmax(1)
    r = AttackRunner(net, a.b, a.n, BN.cfg_full_geoa3(a.steps + 8, a.n, a.knn), dev, global_batch=250)
    r.setup(ori, nrm, gt, gt)
    g = torch.Generator().manual_seed(7)
    r.begin_search_step((torch.randn(a.b, 3, a.n, generator=g) * 1e-3).to(dev))
    for s in range(a.steps): r.step(s, 0)
    torch.cuda.synchronize()
    x = r.t["x"].clone(); off = (x - ori)
    want = ops.nn1_pair(x, ori)
    prior = (want[1].clone(), want[3].clone())
    tb = timeit(lambda: ops.nn1_pair(x, ori))
    tg = timeit(lambda: ops.nn1_pair(x, ori, method="grid", prior=prior))
    got = ops.nn1_pair(x, ori, method="grid", prior=prior)
    ok = all(torch.equal(p, q) for p, q in zip(want, got))
    on = off.norm(dim=1)
    print("%-11s N=%d b=%d: |offset| mean %.4f p99 %.4f max %.4f | sqrt(d_ao) mean %.4f p99 %.4f | sqrt(d_oa) mean %.4f p99 %.4f | "
          "all-pairs %7.1f us  grid %7.1f us  bit-equal %s" % (kind, a.n, a.b, on.mean(), on.flatten().quantile(0.99), on.max(),
          want[0].sqrt().mean(), want[0].sqrt().flatten().quantile(0.99), want[2].sqrt().mean(), want[2].sqrt().flatten().quantile(0.99), tb, tg, ok), flush=True)
    if a.dump:
        os.makedirs(a.dump, exist_ok=True)
        torch.save({"x": x[:4].cpu(), "ori": ori[:4].cpu(), "i_ao": prior[0][:4].cpu(), "i_oa": prior[1][:4].cpu()},
                   os.path.join(a.dump, "state_%s_%d.pt" % (kind, a.n)))
    del r
