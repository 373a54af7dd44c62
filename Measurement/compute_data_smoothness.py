#!/usr/bin/env python3
"""Smoothness of the adversarial clouds -- the reference's Measurement/compute_data_smoothness.py (flags, console
lines, metric/k<k>.mat and metric/result.txt), with the neighbour search, the per-point normal estimation (a python
loop of numpy eigen-decompositions in the reference, :56-61) and the statistic running in libgeoa3_hip.so for all
clouds of equal size at once.

    python Measurement/compute_data_smoothness.py --datadir Exps/.../<run> --k 16 --k2 16
"""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np
import scipy.io as sio
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))


# flag, type, default -- the reference's set (compute_data_smoothness.py:10-15); tests/test_cli_data.py checks it
_FLAGS = (("datadir", str, "Data/modelnet40_1024_processed"), ("k", int, 16), ("k2", int, 16), ("print_freq", int, 50))


def build_parser():
    parser = argparse.ArgumentParser(description="Smoothness Computing")
    for name, kind, default in _FLAGS:
        parser.add_argument("--" + name, type=kind, default=default)
    parser.add_argument("--is_not_mat", action="store_true", default=False)
    return parser


def read_off_lines_from_xyz(path, num_points):
    """First three columns of the first num_points lines (-1: all) of a text cloud (compute_data_smoothness.py:19-28)."""
    rows = [ln.split()[:3] for ln in open(path).read().splitlines()]
    rows = rows if num_points == -1 else rows[:num_points]
    return [[float(v) for v in r] for r in rows]


def main(cfg):
    from geoa3_amd import utility as U
    src = cfg.datadir if cfg.is_not_mat else os.path.join(cfg.datadir, "Mat")
    filenames = os.listdir(src)
    clouds = []
    for filename in filenames:
        if cfg.is_not_mat:
            pc = torch.FloatTensor(read_off_lines_from_xyz(os.path.join(src, filename), -1)).t()
        else:
            pc = torch.FloatTensor(sio.loadmat(os.path.join(src, filename))["adversary_point_clouds"])
        clouds.append(pc.contiguous())                                        # [3,n]
    groups = {}
    for i, pc in enumerate(clouds):
        groups.setdefault(int(pc.shape[1]), []).append(i)
    values = [0.0] * len(clouds)
    for n, members in groups.items():
        batch = torch.stack([clouds[i] for i in members]).cuda()
        out = U.smoothness(batch, cfg.k, cfg.k2).cpu().tolist()
        for i, v in zip(members, out):
            values[i] = v
    for i in range(len(values)):
        if (i + 1) % cfg.print_freq == 0:
            print("[{0}/{1}]: {2:.4f}({3:.4f})".format(i + 1, len(filenames), values[i],
                                                       float(np.mean(np.float32(values[:i + 1])))))
    smoothness = torch.FloatTensor(values)
    os.makedirs(os.path.join(cfg.datadir, "metric"), exist_ok=True)
    sio.savemat(os.path.join(cfg.datadir, "metric", "k" + str(cfg.k) + ".mat"), {"smoothness": smoothness.numpy()})
    ma, mi, av = smoothness.max().item(), smoothness.min().item(), smoothness.mean().item()
    with open(os.path.join(cfg.datadir, "metric", "result.txt"), "at") as f:
        info = "k: {0}, avg: {1:.4f}, min: {2:.4f}, max: {3:.4f}\n".format(cfg.k, av, mi, ma)
        print(info)
        f.write(info)
    return smoothness


if __name__ == "__main__":
    cfg = build_parser().parse_args()
    print(cfg)
    main(cfg)
