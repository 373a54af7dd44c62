#!/usr/bin/env python3
"""Feasibility / gain probe for replaying the inner iteration as a HIP graph (32-instance shard): captures two
iterations of AttackRunner.step (both streams) with torch.cuda.CUDAGraph and times replays against eager enqueues.
(The captured by-value step scalars are frozen: timing only.)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from bench import cfg_full_geoa3
from geoa3_amd.attack import AttackRunner
from geoa3_amd.data import synthetic_clouds, synthetic_state_dict
from geoa3_amd.pointnet import PointNet


def main():
    dev = torch.device("cuda")
    for B in [int(v) for v in sys.argv[1:]] or (32, 250):
        net = PointNet(40)
        net.load_state_dict(synthetic_state_dict(40, seed=0, device=dev))
        net = net.to(dev).eval()
        ori, nrm = synthetic_clouds(B, 1024, seed=100)
        ori, nrm = ori.to(dev), nrm.to(dev)
        gt = net(ori).argmax(1)
        r = AttackRunner(net, B, 1024, cfg_full_geoa3(5000), dev, global_batch=250)
        r.setup(ori, nrm, gt, gt)
        r.begin_search_step((torch.randn(B, 3, 1024) * 1e-3).to(dev))
        for s in range(150):
            r.step(s, 0)
        torch.cuda.synchronize()

        def eager(n):
            t0 = time.perf_counter()
            for s in range(n):
                r.step(150 + s, 0)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3
        eager(20)
        e = eager(200)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=side, capture_error_mode="relaxed"):
            r.step(150, 0)
            r.step(151, 0)
        torch.cuda.synchronize()
        for _ in range(10):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(100):
            g.replay()
        torch.cuda.synchronize()
        gr = (time.perf_counter() - t0) / 200 * 1e3
        print("B=%d: eager %.4f ms/iteration, graph replay (2 iterations per graph) %.4f ms/iteration" % (B, e, gr))


if __name__ == "__main__":
    main()
