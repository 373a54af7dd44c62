#!/usr/bin/env python3
"""CPU probe for the filter-and-refine idea on the max-fused 1024-wide layers: ONE fp16 product
per term with a rigorous error bound, exact evaluation only for the points that can still be a channel's maximum.

Counts, for conv5 (K = 384: three taps of the 128-channel activation) on synthetic clouds, how many points per
(instance, channel) survive the filter  S_p + eps_p >= max_q (S_q - eps_q)  with
  plain:     S_p = fl16(a_p) . fl16(w),                 eps_p = 1.01 * 2^-10 * |a_p| |w|
  centred:   S_p = mean_t . w (exact) + fl16(a_p - mean_t) . fl16(w),   eps_p = 1.01 * 2^-10 * |a_p - mean_t| |w|
(mean_t = mean input of the point's 128-point tile: the component all points of a tile share carries no information
about WHICH point is the maximum), and how many 128-point tiles hold more than one / more than four survivors.
python tools/filter_refine_probe.py [--noise 0.03]"""
import argparse, os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from oracle import geoa3_oracle as O


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=6)
    ap.add_argument("--noise", type=float, default=0.03)
    a = ap.parse_args()
    torch.manual_seed(0)
    sd = O.make_pointnet_state_dict(40, seed=0)
    ori, _ = O.make_synthetic_clouds(a.B, 1024, seed=100)
    pc = ori + a.noise * torch.randn_like(ori)
    eps = 1e-3
    with torch.no_grad():
        t3 = O._tnet_forward(sd, "input_transform.", pc, 3)
        f = torch.bmm(pc.permute(0, 2, 1), t3).permute(0, 2, 1)
        f = F.relu(O._bn(F.conv1d(f, sd["conv1.weight"], sd["conv1.bias"]), sd, "bn1", eps))
        f = F.relu(O._bn(F.conv1d(f, sd["conv2.weight"], sd["conv2.bias"]), sd, "bn2", eps))
        t64 = O._tnet_forward(sd, "feature_transform.", f, 64)
        f = torch.bmm(f.permute(0, 2, 1), t64).permute(0, 2, 1)
        f = F.relu(O._bn(F.conv1d(f, sd["conv3.weight"], sd["conv3.bias"]), sd, "bn3", eps))
        h4 = F.relu(O._bn(F.conv1d(f, sd["conv4.weight"], sd["conv4.bias"]), sd, "bn4", eps)).double()   # [B,128,N]
        # conv5 + bn5 folded: w [1024, 3*128] with k = tap*128 + ci
        g = (sd["bn5.weight"] / torch.sqrt(sd["bn5.running_var"] + eps)).double()
        w = (sd["conv5.weight"].double() * g[:, None, None]).permute(0, 2, 1).reshape(1024, 384)
        hp = F.pad(h4, (1, 1))
        X = torch.cat([hp[:, :, 0:1024], hp[:, :, 1:1025], hp[:, :, 2:1026]], 1).permute(0, 2, 1)   # [B,N,384]
        exact = X @ w.T                                                                               # [B,N,1024]
        wn = w.norm(dim=1)
        w16 = w.half().double()

        def run(centred, tilemax=False):
            tiles = X.view(a.B, 8, 128, 384)
            mean = tiles.mean(2, keepdim=True) if centred else torch.zeros_like(tiles[:, :, :1])
            d = tiles - mean
            # per-tile power-of-two scale so that the largest |d| sits in [2^13, 2^14) (as the kernels do)
            amax = d.abs().amax((2, 3), keepdim=True).clamp_min(1e-30)
            sc = torch.exp2(13 - torch.floor(torch.log2(amax)))
            d16 = (d * sc).half().double() / sc
            S = (mean @ w.T) + d16 @ w16.T                                      # [B,8,128,1024]
            dn = d.norm(dim=3, keepdim=True)
            if tilemax:
                dn = dn.amax(2, keepdim=True).expand_as(dn)
            e = 1.06 * 2.0 ** -10 * dn * wn             # [B,8,128,1024]
            lo = (S - e).amax((1, 2), keepdim=True)
            cand = (S + e) >= lo
            assert bool((cand | ~(exact.view(a.B, 8, 128, 1024) >= exact.amax(1).view(a.B, 1, 1, 1024))).all()), "bound violated"
            per_chan = cand.sum((1, 2)).double()                                # [B,1024]
            per_tile = cand.sum(2)                                              # [B,8,1024]
            live = per_tile > 0
            return dict(mean_candidates=per_chan.mean().item(), p50=per_chan.median().item(), p99=per_chan.quantile(0.99).item(),
                        max=per_chan.max().item(), tiles_live=live.double().mean().item(),
                        tiles_gt1=(per_tile > 1).double().mean().item(), tiles_gt4=(per_tile > 4).double().mean().item(),
                        ratio_norm=(d.norm(dim=3).mean() / tiles.norm(dim=3).mean()).item())
        for c in (False, True):
            for tm in (False, True):
                print("centred" if c else "plain  ", "tile-max norm" if tm else "per-point norm", {k: round(v, 4) for k, v in run(c, tm).items()})


if __name__ == "__main__":
    main()
