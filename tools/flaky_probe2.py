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
    return x.grad.clone()
gs = [grad(adv, w) for _ in range(6)]
# majority value per element = the median of the runs
st = torch.stack(gs)
med = st.median(0).values
for it, gg in enumerate(gs):
    d = (gg != med).nonzero()
    rows = sorted(set(d[:, 0].tolist()))
    print("run %d: %d elems differ from the median; rows %s" % (it, d.shape[0], rows))
    for e in d[:6].tolist():
        b, c, n = e
        print("   b=%d c=%d n=%d (lane %d): got %.6e median %.6e ratio %.4f | other runs %s" % (b, c, n, n % 64, gg[b, c, n].item(), med[b, c, n].item(), gg[b, c, n].item() / (med[b, c, n].item() + 1e-30), [round(x[b, c, n].item(), 5) for x in gs]))
