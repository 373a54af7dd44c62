"""Debug: first iteration at which the graph-searched attack loop departs from the brute-force one."""
import copy, sys
import numpy as np, torch
sys.path.insert(0, ".")
from oracle import geoa3_oracle as O
from geoa3_amd.pointnet import PointNet
from geoa3_amd.attack import AttackRunner, unpack_input
from geoa3_amd import ops
from tests.test_gpu_attack import _loader_batch

net = PointNet(40); net.load_state_dict(O.make_pointnet_state_dict(40, seed=0)); net = net.cuda().eval()
ori, nrm = O.make_synthetic_clouds(5, 700, seed=23)
gt = O.pointnet_forward(O.make_pointnet_state_dict(40, seed=0), ori).argmax(1)
cfg = O.AttackCfg(binary_max_steps=1, iter_max_steps=25, lr=0.001, curv_loss_knn=16)
init = torch.randn(5, 3, 700, generator=torch.Generator().manual_seed(24)) * 1e-3
pc, nm, g, t = unpack_input(_loader_batch(ori, nrm, gt, None, False), False)
rs = []
for gs in (False, True, False):
    c = copy.copy(cfg); c.graph_search = gs
    r = AttackRunner(net, 5, 700, c, torch.device("cuda")); r.setup(pc, nm, g, t); r.begin_search_step(init.cuda())
    rs.append(r)
for step in range(25):
    snaps = []
    for r in rs:
        r.step(step, 0)
        kn = r.t["knn"][r.knn_cur]
        snaps.append({k: v.clone() for k, v in dict(x=r.t["x"], d_ao=r.t["d_ao"], i_ao=r.t["i_ao"], d_oa=r.t["d_oa"],
                      i_oa=r.t["i_oa"], knn=kn, knn_d=r.t["knn_d"], g_geo=r.t["g_geo"], g_cls=r.t["g_cls"]).items()})
    for name in snaps[0]:
        e1 = torch.equal(snaps[0][name], snaps[1][name]); e2 = torch.equal(snaps[0][name], snaps[2][name])
        if not (e1 and e2):
            print("step", step, name, "brute==graph", e1, "brute==brute", e2,
                  (snaps[0][name] != snaps[1][name]).sum().item())
    if any(not torch.equal(snaps[0][n], snaps[1][n]) for n in snaps[0]):
        break
