import os, sys, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from geoa3_amd.data import synthetic_state_dict, synthetic_clouds
from geoa3_amd.pointnet import PointNet
B, N = 250, 1024
ori, _ = synthetic_clouds(B, N, seed=7)
g = torch.Generator().manual_seed(3)
adv = (ori + 0.01 * torch.randn(B, 3, N, generator=g)).cuda().contiguous()
net = PointNet(40); net.load_state_dict(synthetic_state_dict(40, seed=0)); net = net.cuda().eval()
w = torch.randn(B, 40, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
def grad(x, ww):
    x = x.clone().requires_grad_()
    out = net(x); (out * ww).sum().backward()
    return out.detach().clone(), x.grad.clone()
ref_o, ref_g = grad(adv, w)
for it in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    o, gfull = grad(adv, w)
    d = (gfull != ref_g)
    msg = "run %d: full-vs-full differing elems %d (logits %d)" % (it, int(d.sum()), int((o != ref_o).sum()))
    if d.any():
        idx = d.nonzero()
        import collections
        pts = idx[:, 2].tolist()
        msg += " rows %d distinct; pts/8 histogram %s; chan %s maxdiff %.3e" % (len(set(idx[:, 0].tolist())), sorted(collections.Counter([p // 8 for p in pts]).items())[:12], sorted(set(idx[:, 1].tolist())), float((gfull - ref_g).abs().max()))
    for lo, hi in [(0, 7), (100, 133), (249, 250)]:
        o2, g2 = grad(adv[lo:hi], w[lo:hi])
        d2 = g2 != ref_g[lo:hi]
        if d2.any():
            idx = d2.nonzero()
            msg += " | part(%d,%d) %d elems rows %s pts %s maxdiff %.3e" % (lo, hi, int(d2.sum()), sorted(set(idx[:, 0].tolist()))[:6], sorted(set(idx[:, 2].tolist()))[:10], float((g2 - ref_g[lo:hi]).abs().max()))
    print(msg, flush=True)
