import sys, torch
sys.path.insert(0, '.')
from geoa3_amd.data import synthetic_clouds, synthetic_state_dict
from geoa3_amd.pointnet import PointNet
B, N = 250, 1024
ori, _ = synthetic_clouds(B, N, seed=2024)
adv = (ori + 0.01 * torch.randn(B, 3, N, generator=torch.Generator().manual_seed(1))).cuda().contiguous()
net = PointNet(40); net.load_state_dict(synthetic_state_dict(40, seed=0)); net = net.cuda().eval()
w = torch.randn(B, 40, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5))
def run(lo, hi):
    x = adv[lo:hi].clone().requires_grad_()
    l = net(x); (l * w[lo:hi]).sum().backward()
    return x.grad.clone()
g1 = run(0, B); g2 = run(0, B)
print('run-to-run equal:', torch.equal(g1, g2), (g1 - g2).abs().max().item())
for lo, hi in [(0, 7), (0, 32), (0, 64), (100, 133)]:
    gs = run(lo, hi)
    d = (gs - g1[lo:hi]).abs()
    print(lo, hi, torch.equal(gs, g1[lo:hi]), d.max().item(), (d > 0).sum().item(), 'rows differing:', (d.flatten(1).max(1)[0] > 0).nonzero().flatten().tolist()[:10])
