"""Sequential runs of the attack loop (a fresh runner each, as the test does): per-step copies of the iterate and of the
per-iteration tensors of run 0 against every later run."""
import os, sys, gc, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
import bench
from geoa3_amd.attack import AttackRunner
from geoa3_amd.data import synthetic_state_dict, synthetic_clouds
from geoa3_amd.pointnet import PointNet
B, N, K = 250, 1024, 17
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ori, nrm = synthetic_clouds(B, N, seed=2024)
ori, nrm = ori.cuda(), nrm.cuda()
net = PointNet(40); net.load_state_dict(synthetic_state_dict(40, seed=0)); net = net.cuda().eval()
with torch.no_grad():
    gt = net(ori).argmax(1)
init = (torch.randn(B, 3, N, generator=torch.Generator().manual_seed(11)) * 1e-3).cuda()
names = ("logits", "g_cls", "g_geo", "d_ao", "i_ao", "d_oa", "i_oa", "knn_d", "x")
ref = None
for rep in range(reps):
    cfg = bench.cfg_full_geoa3(steps + 4, N, K - 1)
    r = AttackRunner(net, B, N, cfg, torch.device("cuda"))
    r.setup(ori, nrm, gt, gt)
    r.begin_search_step(init)
    cur = []
    sync_each = os.environ.get("SYNC_EACH", "0") == "1"
    for s in range(steps):
        r.step(s, 0)
        if sync_each:
            torch.cuda.synchronize()
        snap = {n: r.t[n].clone() for n in names}
        snap["knn"] = r.t["knn"][r.knn_cur].clone()
        snap["constrain"] = r.geo_out["constrain"].clone()
        cur.append(snap)
    torch.cuda.synchronize()
    if ref is None:
        ref = cur
        print("run 0: reference", flush=True)
    else:
        msg = "identical"
        for s in range(steps):
            d = [n for n in ref[s] if not torch.equal(ref[s][n], cur[s][n])]
            if d:
                det = []
                for n in d[:4]:
                    dd = (ref[s][n] != cur[s][n]).nonzero()
                    det.append("%s: %d elems, rows %s" % (n, dd.shape[0], sorted(set(dd[:, 0].tolist()))[:5]))
                    if n == "g_cls":
                        import collections
                        ch = collections.Counter(dd[:, 1].tolist()); blk = collections.Counter((dd[:, 2] // 64).tolist())
                        a, b2 = ref[s][n], cur[s][n]
                        rel = ((a - b2).abs() / (a.abs() + 1e-12))[dd[:, 0], dd[:, 1], dd[:, 2]]
                        det.append("g_cls channels %s; 64-point tiles %s; rel diff median %.2e max %.2e" % (dict(ch), sorted(blk.items()), rel.median().item(), rel.max().item()))
                msg = "first difference at step %d in %s | %s" % (s, d, "; ".join(det))
                break
        print("run %d: %s" % (rep, msg), flush=True)
    del r, cur
    gc.collect()
