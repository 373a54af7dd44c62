#!/usr/bin/env python3
"""What the loop's iterates look like on CAD-like clouds (geoa3_amd.data.synthetic_cad_clouds), per kind: offset size, the
reverse-list lengths the objective kernel sees (in-degree of the K-NN graph + clean points per adversarial point), points
per 16^3 grid cell of the 1-NN search.   python tools/cad_probe.py [--steps 200] [--b 50]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--b", type=int, default=50)
    ap.add_argument("--n", type=int, default=1024)
    a = ap.parse_args()
    import bench
    from geoa3_amd.attack import AttackRunner
    from geoa3_amd.data import CAD_KINDS, synthetic_cad_clouds, synthetic_clouds, synthetic_state_dict
    from geoa3_amd.pointnet import PointNet
    dev = torch.device("cuda")
    net = PointNet(40)
    net.load_state_dict(synthetic_state_dict(40, seed=0, device=dev))
    net = net.to(dev).eval()
    for kind in ("ellipsoid",) + tuple(CAD_KINDS):
        ori, nrm = (synthetic_clouds(a.b, a.n, seed=100) if kind == "ellipsoid"
                    else synthetic_cad_clouds(a.b, a.n, seed=100, kinds=(kind,)))
        ori, nrm = ori.to(dev), nrm.to(dev)
        with torch.no_grad():
            gt = net(ori).argmax(1)
        cfg = bench.cfg_full_geoa3(a.steps + 16, a.n, 16)
        r = AttackRunner(net, a.b, a.n, cfg, dev, global_batch=250)
        r.setup(ori, nrm, gt, gt)
        g = torch.Generator().manual_seed(7)
        r.begin_search_step((torch.randn(a.b, 3, a.n, generator=g) * 1e-3).to(dev))
        for s in range(a.steps):
            r.step(s, 0)
        torch.cuda.synchronize()
        t = r.t
        off = t["offset"].norm(dim=1)
        knn = t["knn"][0].long()[:, :, 1:]
        other = t["knn"][1].long()[:, :, 1:]
        tot_max, over50, over32 = 0, 0, 0
        for tab in (knn, other):
            for b in range(a.b):
                deg = torch.bincount(tab[b].reshape(-1).clamp(0, a.n - 1), minlength=a.n) + torch.bincount(t["i_oa"][b].long(), minlength=a.n)
                tot_max = max(tot_max, int(deg.max()))
                over50 += int((deg > 50).sum())
                over32 += int((deg > 32).sum())
        # occupancy of the 16^3 grid the 1-NN kernel builds over the clean cloud
        lo = ori.amin(dim=2, keepdim=True)
        ext = (ori.amax(dim=2, keepdim=True) - lo).amax(dim=1, keepdim=True)
        c = ((ori - lo) / (ext / 16 * 1.00001)).floor().clamp(0, 15).long()
        cell = (c[:, 0] * 16 + c[:, 1]) * 16 + c[:, 2]
        occ = torch.stack([torch.bincount(cell[b], minlength=4096) for b in range(a.b)]).float()
        same_cell = (occ * occ).sum(1) / a.n           # candidates per query inside its own cell, on average
        print("%-11s |offset| mean %.4f max %.4f   rows: max %3d, >32: %5.1f / instance, >50: %5.2f / instance   grid: max cell %4d, own-cell candidates %6.1f"
              % (kind, float(off.mean()), float(off.max()), tot_max, over32 / (2 * a.b), over50 / (2 * a.b), int(occ.max()), float(same_cell.mean())))


if __name__ == "__main__":
    main()
