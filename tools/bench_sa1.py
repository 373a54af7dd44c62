#!/usr/bin/env python3
"""PointNet++ level-1 kernels alone at configs[3]'s shape (B = 250, N = 1024, 512 centroids x 64 samples): forward and
input gradient through the C ABI, microseconds per launch (CUDA events).  Under `rocprofv3 --pmc ...` the run holds only
these two kernels (+ set-up), so counters are cheap to take:
    python tools/bench_sa1.py [--B 250] [--iters 30] [--check]      (--check: against the layer-by-layer torch evaluation)"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=250)
    ap.add_argument("--N", type=int, default=1024)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    from geoa3_amd import _lib
    from geoa3_amd.data import synthetic_clouds
    from geoa3_amd.pointnet2 import ext, check
    B, N, M = a.B, a.N, 512
    pts, _ = synthetic_clouds(B, N, seed=0)
    xyz = pts.cuda().permute(0, 2, 1).contiguous()                      # [B,N,3]
    idx1 = ext.furthest_point_sampling(xyz, M)
    new_xyz = torch.gather(xyz, 1, idx1.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    gidx = ext.ball_query(new_xyz, xyz, 0.2, 64)
    g = torch.Generator().manual_seed(1)
    w = [torch.randn(64, 3, generator=g) * 0.8, torch.randn(64, generator=g) * 0.1, torch.randn(64, 64, generator=g) * 0.18,
         torch.randn(64, generator=g) * 0.1, torch.randn(128, 64, generator=g) * 0.18, torch.randn(128, generator=g) * 0.1]
    w = [t.cuda().contiguous() for t in w]
    ws = _lib.Sa1Weights(*[t.data_ptr() for t in w])
    out = torch.empty(B, M, 128, device="cuda")
    arg = torch.empty(B, M, 128, device="cuda", dtype=torch.uint8)
    gout = torch.randn(B, M, 128, generator=g).cuda()
    gx, gn = torch.empty_like(xyz), torch.empty_like(new_xyz)
    scratch = torch.empty(B, M, 64, 3, device="cuda")
    L = _lib.load()
    s = torch.cuda.current_stream().cuda_stream

    def fwd():
        check(L.geoa3_pn2_sa1_forward(xyz.data_ptr(), new_xyz.data_ptr(), gidx.data_ptr(), ws, B, N, M, out.data_ptr(),
                                      arg.data_ptr(), s), "sa1_forward")

    def bwd():
        check(L.geoa3_pn2_sa1_backward(xyz.data_ptr(), new_xyz.data_ptr(), gidx.data_ptr(), ws, B, N, M, out.data_ptr(),
                                       arg.data_ptr(), gout.data_ptr(), gx.data_ptr(), gn.data_ptr(), scratch.data_ptr(), s),
              "sa1_backward")

    def timeit(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e3 / a.iters

    res = {"B": B, "fwd_us": round(timeit(fwd), 1), "bwd_plus_scatter_us": round(timeit(bwd), 1)}
    if a.check:
        b = min(B, 8)
        x = xyz[:b].double().requires_grad_(True)
        c = new_xyz[:b].double().requires_grad_(True)
        p = torch.gather(x.unsqueeze(1).expand(-1, M, -1, -1), 2, gidx[:b].long().unsqueeze(-1).expand(-1, -1, -1, 3)) - c.unsqueeze(2)
        h = torch.relu(p @ w[0].double().T + w[1].double())
        h = torch.relu(h @ w[2].double().T + w[3].double())
        h = torch.relu((h @ w[4].double().T + w[5].double()).max(2).values)
        (h * gout[:b].double()).sum().backward()
        fwd(), bwd()
        torch.cuda.synchronize()
        res["fwd_maxerr_rel"] = float((out[:b].double() - h).abs().max() / h.abs().max())
        res["gx_maxerr_rel"] = float((gx[:b].double() - x.grad).abs().max() / x.grad.abs().max())
        res["gn_maxerr_rel"] = float((gn[:b].double() - c.grad).abs().max() / c.grad.abs().max())
    print(json.dumps(res))


if __name__ == "__main__":
    main()
