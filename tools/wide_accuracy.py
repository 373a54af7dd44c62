#!/usr/bin/env python3
"""Error of the HIP PointNet against a float64 evaluation of the same network, for both arithmetic modes of the
1024-wide layers (GEOA3_WIDE_MODE f32 = fp32 MFMA, f16x2 = split-fp16 operands), next to the error of the fp32 CPU
oracle: logits and input gradient.   python tools/wide_accuracy.py [--B 8] [--N 1024]"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from oracle import geoa3_oracle as O  # noqa: E402  (checker only: tools/ is not the product path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--N", type=int, default=1024)
    a = ap.parse_args()
    from geoa3_amd.pointnet import PointNet
    sd = O.make_pointnet_state_dict(40, seed=0)
    pc, _ = O.make_synthetic_clouds(a.B, a.N, seed=11)
    w = torch.randn(a.B, 40, generator=torch.Generator().manual_seed(1))

    def cpu(dtype):
        sdd = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
        x = pc.to(dtype).clone().requires_grad_()
        lo = O.pointnet_forward(sdd, x)
        (lo * w.to(dtype)).sum().backward()
        return lo.detach().double(), x.grad.double()

    ref_l, ref_g = cpu(torch.float64)
    res = {}

    def err(name, l, g):
        res[name] = {"logits_max_abs": float((l - ref_l).abs().max()), "logits_rms": float((l - ref_l).pow(2).mean().sqrt()),
                     "grad_max_rel": float((g - ref_g).abs().max() / ref_g.abs().max()),
                     "grad_rms_rel": float((g - ref_g).pow(2).mean().sqrt() / ref_g.pow(2).mean().sqrt())}

    err("cpu_fp32_oracle", *cpu(torch.float32))
    for mode in ("f32", "f16x2"):
        net = PointNet(40)
        net.load_state_dict(sd)
        net.wide_mode = mode
        net = net.cuda().eval()
        x = pc.cuda().requires_grad_()
        lg = net(x)
        (lg * w.cuda()).sum().backward()
        err("hip_" + mode, lg.detach().cpu().double(), x.grad.cpu().double())
    res["logits_scale"] = float(ref_l.abs().max())
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
