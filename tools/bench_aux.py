"""Timings of the dense-cloud / defence / measurement kernels and of one dense-cloud attack iteration (GPU)."""
import sys
import time

import torch

sys.path.insert(0, ".")
from geoa3_amd import ops, utility as U  # noqa: E402
from geoa3_amd.data import synthetic_clouds, synthetic_state_dict  # noqa: E402


def timeit(fn, n=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def main():
    dev = torch.device("cuda")
    B = 250
    pc1, _ = synthetic_clouds(B, 1024, 0)
    pc4, nrm4 = synthetic_clouds(B, 4096, 1)
    pc1, pc4, nrm4 = torch.as_tensor(pc1).to(dev), torch.as_tensor(pc4).to(dev), torch.as_tensor(nrm4).to(dev)
    start = torch.zeros(B, dtype=torch.int32, device=dev)
    print("fps_sample B=250 N=4096 -> 1024: %.3f ms" % timeit(lambda: U.fps_indices(pc4, 1024, start)))
    print("fps_sample B=250 N=1024 -> 512 : %.3f ms" % timeit(lambda: U.fps_indices(pc1, 512, start)))
    print("sor_statistic B=250 N=1024 k=2 : %.3f ms" % timeit(lambda: U.sor_statistic(pc1, 2)))
    print("outlier_removal fixNum 128     : %.3f ms" % timeit(
        lambda: U.outlier_removal_indices(pc1, "outliers_fixNum", 128, 1.1, 2)))
    print("outlier_removal variance       : %.3f ms" % timeit(
        lambda: U.outlier_removal_indices(pc1, "outliers_variance", 0, 1.1, 2)))
    _, idx = ops.knn_planar(pc1, pc1, 17)
    print("local_frames B=250 N=1024 k=16 : %.3f ms" % timeit(lambda: U.local_frames(pc1, 16, idx)))
    print("estimate_perpendicular k=16    : %.3f ms" % timeit(lambda: U.estimate_perpendicular(pc1, 16)))
    print("smoothness k=16 k2=16          : %.3f ms" % timeit(lambda: U.smoothness(pc1, 16, 16)))
    adv = pc4[:, :, :1024].contiguous()
    print("estimate_normal_via_ori k=3 1024 vs 4096: %.3f ms" % timeit(
        lambda: U.estimate_normal_via_ori_normal(adv, pc4, nrm4, 3)))
    # one dense-cloud attack iteration: 4096-point clouds, 1024-point victim, eval_num 1
    from oracle.geoa3_oracle import AttackCfg   # plain namespace of the reference defaults (bench leg only)
    from geoa3_amd.attack import AttackRunner
    from geoa3_amd.pointnet import PointNet
    net = PointNet(40)
    net.load_state_dict(synthetic_state_dict(40, seed=0, device=dev))
    net = net.to(dev).eval()
    for E in (1, 3):
        cfg = AttackCfg(is_subsample_opt=True, npoint=1024, eval_num=E, binary_max_steps=1, iter_max_steps=12)
        r = AttackRunner(net, B, 4096, cfg, dev)
        gt = torch.zeros(B, dtype=torch.int64)
        r.setup(pc4, nrm4, gt, gt)
        r.begin_search_step(torch.randn(B, 3, 4096, device=dev) * 1e-3)
        step = [0]

        def one():
            r.step(step[0] % 12, 0)
            step[0] += 1
        print("dense attack iteration B=250 N=4096->1024 eval_num=%d: %.2f ms" % (E, timeit(one, n=8, warm=3)))


if __name__ == "__main__":
    main()
