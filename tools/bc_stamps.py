#!/usr/bin/env python3
"""Phase times inside wide_bwd_conv_kernel (probe build: python -m geoa3_amd.build --variant tools/ub/lib_stamps
-DGEOA3_BC_STAMPS; run with GEOA3_LIB_PATH=tools/ub/lib_stamps/libgeoa3_hip.so): s_memtime (shader cycles) at the phase
boundaries of eight workgroups per kernel, B = 250, N = 1024."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from geoa3_amd import _lib  # noqa: E402
from geoa3_amd.data import synthetic_clouds, synthetic_state_dict  # noqa: E402
from geoa3_amd.pointnet import PointNet  # noqa: E402

NAMES = ["offsets", "list copy + zero", "walk", "barrier", "merge", "gate pass", "W2^T product", "stores"]


def main():
    dev = torch.device("cuda")
    lib = _lib.load()
    net = PointNet(40)
    net.load_state_dict(synthetic_state_dict(40, seed=0, device=dev))
    net = net.to(dev).eval()
    ori, _ = synthetic_clouds(250, 1024, seed=100)
    w = torch.randn(250, 40, device=dev)
    for _ in range(5):
        x = ori.to(dev).clone().requires_grad_()
        (net(x) * w).sum().backward()
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * (3 * 8 * 16))()
    lib.geoa3_debug_bc_stamps.argtypes = [C.c_void_p]
    assert lib.geoa3_debug_bc_stamps(buf) == 0
    t = np.array(buf[:], dtype=np.int64).reshape(3, 8, 16)[:, :, :9]
    for k, name in enumerate(("taps 1 (T-Net 64)", "taps 1, first layer behind (T-Net 3)", "taps 3 (conv5)")):
        d = np.diff(t[k], axis=1)
        print(name, ": total cycles per workgroup", (t[k][:, 8] - t[k][:, 0]).tolist())
        for i, n in enumerate(NAMES):
            print("   %-18s median %6d   %s" % (n, int(np.median(d[:, i])), d[:, i].tolist()))


if __name__ == "__main__":
    main()
