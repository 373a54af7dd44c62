"""Iteration time of a shard with and without the geometry terms of the objective (the victim's launch chain alone):
usage (GPU box): python3 tools/victim_only.py [instances]"""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from geoa3_amd.attack import AttackRunner
from geoa3_amd.pointnet import PointNet
from oracle import geoa3_oracle as O
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0"); N = 1024
ori, nrm = O.make_synthetic_clouds(B, N, seed=100); ori, nrm = ori.to(dev), nrm.to(dev)
net = PointNet(40); net.load_state_dict(O.make_pointnet_state_dict(40, seed=0)); net = net.to(dev).eval()
gt = torch.zeros(B, dtype=torch.long, device=dev)
for geo in (True, False, True, False):
    cfg = bench.cfg_full_geoa3(600, N, 16)
    if not geo:
        cfg.dis_loss_weight = 0.0; cfg.hd_loss_weight = 0.0; cfg.curv_loss_weight = 0.0; cfg.dis_loss_type = "None"
    r = AttackRunner(net, B, N, cfg, dev, global_batch=B)
    r.setup(ori, nrm, gt, gt)
    r.begin_search_step((torch.randn(B, 3, N, generator=torch.Generator().manual_seed(7)) * 1e-3).to(dev))
    for s in range(160): r.step(s, 0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for s in range(160, 460): r.step(s, 0)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 300
    print("instances %d geometry %s: %.4f ms per iteration" % (B, geo, dt * 1e3))
