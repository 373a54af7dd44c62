"""Where does knn_slab_kernel spend its time in the loop?  Runs the bench's attack to a steady state, then looks at the
adversarial cloud + the prior lists the next K-NN launch would see (CPU analysis) and times the launch itself."""
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

# per-wave phase clocks of one launch (s_memtime, 100 MHz constant clock)
import ctypes as C
from geoa3_amd import _lib
lib = C.CDLL(_lib.LIB_PATH) if hasattr(_lib, "LIB_PATH") else _lib.load()
stamps = torch.zeros(B * 4 * 4 * 6, dtype=torch.int64, device=dev)
fn = lib.geoa3_debug_slab_stamps; fn.argtypes = [C.c_void_p]; fn.restype = None
fn(stamps.data_ptr())
r.step(steps, 0); torch.cuda.synchronize()
fn(None)
st = stamps.cpu().numpy().reshape(-1, 6).astype(np.float64)
st = st[st[:, 0] > 0]
d = np.diff(st, axis=1) / 100.0      # us
names = ["prior/tau", "range+stage", "scan", "final compaction", "output"]
for i, n in enumerate(names):
    print("%-18s mean %7.2f us  max %7.2f" % (n, d[:, i].mean(), d[:, i].max()))
print("wave lifetime mean %.2f max %.2f; kernel span %.2f us" % ((st[:, 5] - st[:, 0]).mean() / 100, (st[:, 5] - st[:, 0]).max() / 100, (st[:, 5].max() - st[:, 0].min()) / 100))
s0 = np.sort(st[:, 0]); s5 = np.sort(st[:, 5])
print("start stamps (sorted, minus min)/100:", np.round((s0[::max(1, len(s0)//16)] - s0[0]) / 100, 1))
print("end stamps   (sorted, minus min start)/100:", np.round((s5[::max(1, len(s5)//16)] - s0[0]) / 100, 1))
order = np.argsort(st[:, 0]); so = st[order]
cuts = np.where(np.diff(so[:, 0]) > 1e6)[0] + 1
for grp in np.split(so, cuts):
    print("cluster: %3d waves, starts spread %.0f, first end %.0f, last end %.0f (units since the cluster's first start)" % (
        len(grp), grp[:, 0].max() - grp[:, 0].min(), grp[:, 5].min() - grp[:, 0].min(), grp[:, 5].max() - grp[:, 0].min()))
# the launch on its own (one stream), timed with events, stamps on
fn(stamps.data_ptr()); stamps.zero_()
lib2 = _lib.load()
prior_t, out_t = t["knn"][r.knn_cur], t["knn"][1 - r.knn_cur]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
sg = torch.cuda.current_stream().cuda_stream
torch.cuda.synchronize()
for rep in range(3):
    stamps.zero_()
    e0.record()
    lib2.geoa3_knn_self(t["x"].data_ptr(), B, N, K, prior_t.data_ptr(), t["knn_d"].data_ptr(), out_t.data_ptr(),
                        t["knn_scratch"].data_ptr(), 1, sg)
    e1.record(); torch.cuda.synchronize()
    print("knn_self alone: %.1f us (incl. the bin kernel)" % (e0.elapsed_time(e1) * 1e3))
fn(None)
st = stamps.cpu().numpy().reshape(-1, 6).astype(np.float64)
print("rows with stamps:", int((st[:, 0] > 0).sum()), "of", len(st))
st = st[st[:, 0] > 0]
order = np.argsort(st[:, 0]); so = st[order]
cuts = np.where(np.diff(so[:, 0]) > 3e6)[0] + 1
for grp in np.split(so, cuts):
    print("cluster: %3d waves, starts spread %.0f, lifetime mean %.0f, last end %.0f" % (
        len(grp), grp[:, 0].max() - grp[:, 0].min(), (grp[:, 5] - grp[:, 0]).mean(), grp[:, 5].max() - grp[:, 0].min()))
