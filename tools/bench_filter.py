#!/usr/bin/env python3
"""conv5 on the shipped three-product kernel against the one-product FILTER pass (pointnet_wide16.hip
wide16_filter_kernel): timing, a check of the records it writes, and the number of points that survive the bound.
python tools/bench_filter.py"""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from geoa3_amd import _lib
from geoa3_amd.pointnet import pack_wide_split16
from tools.bench_conv import timeit


def main():
    lib = _lib.load()
    B, N = 250, 1024
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator().manual_seed(0)
    # activations with a strong common component per instance (as behind conv4): relu(common + noise)
    common = torch.randn(B, 128, 1, generator=g)
    X = (common + 0.25 * torch.randn(B, 128, N, generator=g)).relu_().cuda()
    W = (torch.randn(1024, 384, generator=g) * 0.05)
    if "--real" in sys.argv:
        # the activations and weights of the attack loop: the synthetic victim's conv4 output on iterates of a real run
        from bench import cfg_full_geoa3
        from geoa3_amd.attack import AttackRunner
        from geoa3_amd.data import synthetic_clouds, synthetic_state_dict
        from geoa3_amd.pointnet import PointNet, pack_pointnet
        from oracle import geoa3_oracle as O
        B = 16
        net = PointNet(40)
        sd = synthetic_state_dict(40, seed=0, device=torch.device("cuda"))
        net.load_state_dict(sd)
        net = net.cuda().eval()
        ori, nrm = synthetic_clouds(B, N, seed=100)
        ori, nrm = ori.cuda(), nrm.cuda()
        gt = net(ori).argmax(1)
        r = AttackRunner(net, B, N, cfg_full_geoa3(200), torch.device("cuda"), global_batch=250)
        r.setup(ori, nrm, gt, gt)
        r.begin_search_step((torch.randn(B, 3, N, generator=g) * 1e-3).cuda())
        for st in range(int(os.environ.get("STEPS", "150"))):
            r.step(st, 0)
        x = r.t["x"].cpu()
        sdc = {k: v.cpu() for k, v in sd.items()}
        eps = 1e-3
        with torch.no_grad():
            t3 = O._tnet_forward(sdc, "input_transform.", x, 3)
            f = torch.bmm(x.permute(0, 2, 1), t3).permute(0, 2, 1)
            f = F.relu(O._bn(F.conv1d(f, sdc["conv1.weight"], sdc["conv1.bias"]), sdc, "bn1", eps))
            f = F.relu(O._bn(F.conv1d(f, sdc["conv2.weight"], sdc["conv2.bias"]), sdc, "bn2", eps))
            t64 = O._tnet_forward(sdc, "feature_transform.", f, 64)
            f = torch.bmm(f.permute(0, 2, 1), t64).permute(0, 2, 1)
            f = F.relu(O._bn(F.conv1d(f, sdc["conv3.weight"], sdc["conv3.bias"]), sdc, "bn3", eps))
            X = F.relu(O._bn(F.conv1d(f, sdc["conv4.weight"], sdc["conv4.bias"]), sdc, "bn4", eps)).contiguous().cuda()
        W = pack_pointnet(sdc)["w5"]
        print("real activations after %s attack iterations: fraction of exact zeros %.3f" % (os.environ.get("STEPS", "150"), float((X == 0).float().mean())))
    Wh16, uns = pack_wide_split16(W)
    Wh16 = Wh16.cuda()
    bias = torch.zeros(1024, device="cuda")
    out = torch.empty(B, 1024, device="cuda")
    arg = torch.empty(B, 1024, device="cuda", dtype=torch.int32)
    keys = torch.zeros(B, 1024, device="cuda", dtype=torch.int64)
    T = N // 128
    rec = torch.zeros(B, T, 1024, 4, device="cuda")
    mean = torch.zeros(B, T, 128, device="cuda")
    tile = torch.zeros(B, T, 4, device="cuda")

    Wf = W.float().contiguous().cuda()
    wsumt = W.double().view(1024, 3, 128).sum(1).t().float().contiguous().cuda()                  # [128][1024]
    wnorm = torch.stack((W.double().norm(dim=1).float() * 1.0000002, torch.full((1024,), uns * 2.0 ** -25 * 19.6))).contiguous().cuda()
    flist = torch.zeros(B * T * 2048 + B * T, device="cuda", dtype=torch.int32)

    def run(v):
        rc = lib.geoa3_debug_wide16(X.data_ptr(), Wh16.data_ptr(), uns, bias.data_ptr(), out.data_ptr(), arg.data_ptr(),
                                    keys.data_ptr(), rec.data_ptr(), mean.data_ptr(), tile.data_ptr(), Wf.data_ptr(),
                                    wsumt.data_ptr(), wnorm.data_ptr(), flist.data_ptr(), B, N, v, s)
        assert rc == 0, rc
    print("three products (shipped): %.1f us" % timeit(lambda: run(0)))
    exact, exact_arg = out.clone(), arg.clone()
    print("filter pass, all 8 channel groups per unit: %.1f us" % timeit(lambda: run(8)))
    for occ in (2, 3):
        print("filter pass, %d workgroups per CU: %.1f us" % (occ, timeit(lambda: run(occ))))
    print("filter + decide: %.1f us;  + refine: %.1f us" % (timeit(lambda: run(11)), timeit(lambda: run(12))))
    print("two-pass form (filter + decide + refine + finalize): %.1f us" % timeit(lambda: run(10)))
    run(10)
    torch.cuda.synchronize()
    cnts = flist[B * T * 2048:].view(B, T)
    ent = flist[:B * T * 2048].view(B, T, 2048)
    nmulti = sum(int(((ent[b, t, :int(cnts[b, t])] >> 17) & 1).sum()) for b in range(4) for t in range(T))
    print("two-pass vs three-product kernel: arg-max equal %.5f, max |value diff| %.2e (scale %.2f); survivors per (instance, "
          "channel) %.3f, full-tile evaluations per instance %.1f" % (float((arg == exact_arg).double().mean()),
          float((out - exact).abs().max()), float(exact.abs().max()), float(cnts.double().sum(1).mean()) / 1024, nmulti / 4.0))
    run(2)
    torch.cuda.synchronize()
    # ---- check the records on a few instances against torch (float64)
    nb = 4
    x = X[:nb].double().cpu()
    xp = F.pad(x, (1, 1))
    taps = torch.cat([xp[:, :, 0:N], xp[:, :, 1:N + 1], xp[:, :, 2:N + 2]], 1).permute(0, 2, 1)        # [nb,N,384]
    w = W.double()
    m = x.view(nb, 128, T, 128).mean(3).permute(0, 2, 1)                                              # [nb,T,128]
    assert torch.allclose(mean[:nb].double().cpu(), m, rtol=1e-5, atol=1e-6), "tile means"
    m3 = m.repeat(1, 1, 3)                                                                            # [nb,T,384]
    d = taps.view(nb, T, 128, 384) - m3.unsqueeze(2)
    sval = d @ w.T                                                                                    # exact centred products
    top2 = sval.topk(2, dim=2)
    r = rec[:nb].double().cpu()
    dn = d.norm(dim=3).amax(2)                                                                        # [nb,T]
    wn = w.norm(dim=1)
    bound = 1.1 * 2.0 ** -10 * dn.unsqueeze(-1) * wn                                                  # [nb,T,1024]
    err1 = (r[..., 0] - top2.values[:, :, 0, :]).abs()
    print("records: max |v1 - exact centred max| / bound = %.3f (must be <= 1); tile norm rel err %.2e" %
          (float((err1 / bound).max()), float(((tile[:nb, :, 0].double().cpu() - dn) / dn).abs().max())))
    # ---- survivors: S = m . wsum + v, E = bound
    wsum = w.view(1024, 3, 128).sum(1)                                                                # [1024,128]
    mw = m @ wsum.T                                                                                   # [nb,T,1024]
    U1, U2, Lo = mw + r[..., 0] + bound, mw + r[..., 2] + bound, mw + r[..., 0] - bound
    lo = Lo.amax(1, keepdim=True)
    live, multi = U1 >= lo, U2 >= lo
    print("tiles with a survivor: %.3f per (instance, channel) of %d; tiles needing a full evaluation: %.4f" %
          (float(live.double().sum(1).mean()), T, float(multi.double().sum(1).mean())))
    true_arg = (taps @ w.T).argmax(1)                                                                 # [nb,1024]
    ttile = true_arg // 128
    assert bool(live.gather(1, ttile.unsqueeze(1)).all()), "the true maximum's tile must survive"
    print("agreement of the shipped kernel's arg-max with float64: %.4f" % float((exact_arg[:nb].cpu() == true_arg).double().mean()))


if __name__ == "__main__":
    main()
