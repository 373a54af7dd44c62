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

    # one 1024-wide layer in isolation: pooled pre-activations against float64
    from geoa3_amd import _lib
    from geoa3_amd.pointnet import pack_wide_fragments, pack_wide_split
    lib = _lib.load()
    g = torch.Generator().manual_seed(5)
    for taps in (3, 1):
        B, N = 16, a.N
        X = torch.randn(B, 128, N, generator=g).relu()
        W = torch.randn(1024, taps * 128, generator=g) * 0.05
        ref = torch.nn.functional.conv1d(X.double(), W.double().view(1024, taps, 128).permute(0, 2, 1),
                                         padding=taps // 2).max(dim=2).values
        Wp, (Wh, uns) = pack_wide_fragments(W, taps).cuda(), pack_wide_split(W, taps)
        Wh, Xd = Wh.cuda(), X.cuda()
        bias = torch.full((1024,), 1e6, device="cuda")      # keeps the relu out of the way; subtracted again below
        out = torch.empty(B, 1024, device="cuda")
        arg = torch.empty(B, 1024, device="cuda", dtype=torch.int32)
        keys = torch.empty(B, 1024, device="cuda", dtype=torch.int64)
        zero = torch.zeros(1024, device="cuda")
        for split in (False, True):
            # bias 0: negative maxima are clipped by the relu -- compare where the reference is positive
            lib.geoa3_debug_wide_fwd(Xd.data_ptr(), Wp.data_ptr(), Wh.data_ptr() if split else None, uns,
                                     zero.data_ptr(), out.data_ptr(), arg.data_ptr(), keys.data_ptr(), B, N, taps, 0,
                                     None, torch.cuda.current_stream().cuda_stream)
            o = out.cpu().double()
            pos = ref > 0
            rel = ((o - ref).abs() / ref.abs())[pos]
            res["layer_taps%d_%s" % (taps, "f16x2" if split else "f32")] = {
                "max_rel": float(rel.max()), "rms_rel": float(rel.pow(2).mean().sqrt())}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
