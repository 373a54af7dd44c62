#!/usr/bin/env python3
"""Micro-benchmarks of the individual HIP kernels at the BASELINE config (B=250, N=1024, k=16).
python tools/bench_kernels.py [--B 250] [--N 1024] [--k 16] [--iters 20]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))


def timeit(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters  # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=250)
    ap.add_argument("--N", type=int, default=1024)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    from geoa3_amd import ops
    B, N, k = a.B, a.N, a.k
    g = torch.Generator().manual_seed(0)
    u = torch.randn(B, 3, N, generator=g)
    ori = (u / u.norm(dim=1, keepdim=True)).cuda().contiguous()
    nrm = ori.clone()
    adv = (ori + 0.01 * torch.randn(B, 3, N, generator=g).cuda()).contiguous()
    res = {}
    res["nn1_pair_us"] = timeit(lambda: ops.nn1_pair(adv, ori), a.iters)
    res["nn1_grid_us"] = timeit(lambda: ops.nn1_pair(adv, ori, method="grid"), a.iters)
    far = (ori + 0.3 * torch.randn(B, 3, N, generator=g).cuda()).contiguous()
    res["nn1_pair_far_us"] = timeit(lambda: ops.nn1_pair(far, ori), a.iters)
    res["nn1_grid_far_us"] = timeit(lambda: ops.nn1_pair(far, ori, method="grid"), a.iters)
    flops = 8.0 * B * N * N
    res["nn1_pair_alg_GBps"] = 40.0 * B * N / res["nn1_pair_us"] / 1e3
    res["nn1_pair_valu_frac"] = flops / (res["nn1_pair_us"] * 1e-6) / 157.3e12
    _, knn_ori = ops.knn_planar(ori, ori, k + 1)
    res["knn_noprior_us"] = timeit(lambda: ops.knn_planar(adv, adv, k + 1), a.iters)
    res["knn_prior_us"] = timeit(lambda: ops.knn_planar(adv, adv, k + 1, knn_ori), a.iters)
    scratch = ops.knn_self_scratch(B, N, adv.device)
    res["knn_slab_us"] = timeit(lambda: ops.knn_self_planar(adv, k + 1, knn_ori, scratch, method=1), a.iters)
    res["knn_grid_us"] = timeit(lambda: ops.knn_self_planar(adv, k + 1, knn_ori, scratch, method=2), a.iters)
    _, knn_far = ops.knn_planar(far, far, k + 1)
    res["knn_prior_far_us"] = timeit(lambda: ops.knn_planar(far, far, k + 1, knn_far), a.iters)
    res["knn_slab_far_us"] = timeit(lambda: ops.knn_self_planar(far, k + 1, knn_far, scratch, method=1), a.iters)
    res["knn_grid_far_us"] = timeit(lambda: ops.knn_self_planar(far, k + 1, knn_far, scratch, method=2), a.iters)
    # the attack's regime: a stale table (last iteration's, here the clean cloud's) as the radius
    res["knn_grid_stale_us"] = timeit(lambda: ops.knn_self_planar(far, k + 1, knn_ori, scratch, method=2), a.iters)
    res["knn_slab_stale_us"] = timeit(lambda: ops.knn_self_planar(far, k + 1, knn_ori, scratch, method=1), a.iters)
    kap = ops.kappa(ori, nrm, knn_ori)
    res["kappa_us"] = timeit(lambda: ops.kappa(ori, nrm, knn_ori), a.iters)
    d_ao, i_ao, d_oa, i_oa = ops.nn1_pair(adv, ori)
    _, knn_adv = ops.knn_planar(adv, adv, k + 1, knn_ori)
    out = {}
    for det in (True, False):
        res["geo_loss_grad_%s_us" % ("det" if det else "atomic")] = timeit(lambda: ops.geo_loss_grad(
            adv, ori, normal_ori=nrm, kappa_ori=kap, d_ao=d_ao, i_ao=i_ao, d_oa=d_oa, i_oa=i_oa, knn_adv=knn_adv, k=k,
            w_dis=1.0, w_hd=0.1, w_curv=1.0, out=out, deterministic=det), a.iters)
    print(json.dumps({"B": B, "N": N, "k": k, **{kk: round(v, 3) for kk, v in res.items()}}))


if __name__ == "__main__":
    main()
