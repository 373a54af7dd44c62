#!/usr/bin/env python3
"""Timing of the objective kernel on fixed inputs (deterministic pair kernel vs the LDS-atomics kernel).
python tools/bench_geo.py [--N 1024] [--k 16]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from tools.bench_kernels import timeit


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--N", type=int, default=1024)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--noise", type=float, default=0.05)
    a = ap.parse_args()
    from geoa3_amd import ops
    from geoa3_amd.data import synthetic_clouds
    N, k = a.N, a.k
    for B in (250, 32, 8):
        ori, nrm = synthetic_clouds(B, N, seed=100)
        ori, nrm = ori.cuda(), nrm.cuda()
        g = torch.Generator().manual_seed(0)
        adv = (ori + a.noise * torch.randn(B, 3, N, generator=g).cuda()).contiguous()
        _, knn_ori = ops.knn_planar(ori, ori, k + 1)
        kap = ops.kappa(ori, nrm, knn_ori)
        d_ao, i_ao, d_oa, i_oa = ops.nn1_pair(adv, ori)
        _, knn_adv = ops.knn_planar(adv, adv, k + 1)
        out, res = {}, {"B": B}
        for dbg in (0,):
            res["deterministic"] = round(timeit(lambda: ops.geo_loss_grad(
                adv, ori, normal_ori=nrm, kappa_ori=kap, d_ao=d_ao, i_ao=i_ao, d_oa=d_oa, i_oa=i_oa, knn_adv=knn_adv, k=k,
                w_dis=1.0, w_hd=0.1, w_curv=1.0, out=out, deterministic=True), 30), 2)
        res["atomic"] = round(timeit(lambda: ops.geo_loss_grad(
            adv, ori, normal_ori=nrm, kappa_ori=kap, d_ao=d_ao, i_ao=i_ao, d_oa=d_oa, i_oa=i_oa, knn_adv=knn_adv, k=k,
            w_dis=1.0, w_hd=0.1, w_curv=1.0, out=out, deterministic=False), 30), 2)
        deg = torch.zeros(B, N, device="cuda").scatter_add_(1, knn_adv[:, :, 1:].reshape(B, -1).long(),
                                                          torch.ones(B, N * k, device="cuda"))
        res["max_in_degree"] = int(deg.max().item())
        res["in_degree_gt32"] = float((deg > 31).float().mean().item())
        print(json.dumps(res))


if __name__ == "__main__":
    main()
