"""What the K-NN launch sees in the loop: runs the bench's attack to a steady state, then counts (on the CPU) the run of
sorted points a workgroup scans and the candidates within a query's radius, and times the launch on its own.
usage (GPU box): python3 tools/knn_probe.py [instances] [iterations]"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (helpers only)
from geoa3_amd.attack import AttackRunner
from geoa3_amd.pointnet import PointNet
from oracle import geoa3_oracle as O

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 160
dev = torch.device("cuda:0")
N, K = 1024, 17
ori, nrm = O.make_synthetic_clouds(B, N, seed=100)
ori, nrm = ori.to(dev), nrm.to(dev)
net = PointNet(40); net.load_state_dict(O.make_pointnet_state_dict(40, seed=0)); net = net.to(dev).eval()
gt = torch.zeros(B, dtype=torch.long, device=dev)
cfg = bench.cfg_full_geoa3(steps + 16, N, 16)
r = AttackRunner(net, B, N, cfg, dev, global_batch=B)
r.setup(ori, nrm, gt, gt)
g = torch.Generator(device="cpu").manual_seed(7)
r.begin_search_step((torch.randn(B, 3, N, generator=g) * 1e-3).to(dev))
for s in range(steps):
    r.step(s, 0)
torch.cuda.synchronize()
t = r.t
x = t["x"].detach().cpu().numpy() if "x" in t else r.x.detach().cpu().numpy()
prior = t["knn"][r.knn_cur].cpu().numpy()        # [B,N,K]
tot, passes, big = [], [], 0
for b in range(B):
    P = x[b].T
    ext = P.max(0) - P.min(0); ax = int(ext.argmax())
    pr = prior[b]
    d_prior = ((P[:, None, :] - P[pr]) ** 2).sum(-1)      # [N,K]
    tau = d_prior.max(1)
    order = np.argsort(P[:, ax], kind="stable"); a = P[order, ax]; rr = np.sqrt(tau[order])
    D = ((P[:, None, :] - P[None, :, :]) ** 2).sum(-1)
    npass = (D <= tau[:, None]).sum(1)
    passes.append(npass)
    for w in range(4):
        sl = slice(256 * w, 256 * (w + 1))
        lo, hi = (a[sl] - rr[sl]).min(), (a[sl] + rr[sl]).max()
        tot.append(int(((a >= lo) & (a <= hi)).sum()))
passes = np.concatenate(passes)
print("scan range per WG: mean %.0f max %d" % (np.mean(tot), max(tot)))
print("candidates within tau per query: mean %.1f p99 %d max %d; queries with > 36: %d of %d" % (
    passes.mean(), np.percentile(passes, 99), passes.max(), (passes > 36).sum(), passes.size))

# the launch on its own (one stream), timed with events
import ctypes as C
from geoa3_amd import _lib
lib2 = _lib.load()
prior_t, out_t = t["knn"][r.knn_cur], t["knn"][1 - r.knn_cur]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
sg = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
for method in (1, 2):
    for rep in range(3):
        e0.record()
        lib2.geoa3_knn_self(t["x"].data_ptr(), B, N, K, prior_t.data_ptr(), t["knn_d"].data_ptr(), out_t.data_ptr(),
                            t["knn_scratch"].data_ptr(), method, sg)
        e1.record(); torch.cuda.synchronize()
    print("knn_self method %d alone: %.1f us (incl. the sort kernel)" % (method, e0.elapsed_time(e1) * 1e3))
