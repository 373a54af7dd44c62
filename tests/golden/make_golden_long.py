#!/usr/bin/env python3
"""Long-horizon reference runs -> tests/golden/geoa3_golden_long.npz (build container only).

Run:  python tests/golden/make_golden_long.py       (needs /root/reference; CPU only, a few minutes)

The short trajectories of make_golden.py (<= 8 inner x <= 4 binary steps, N = 64) pin every branch of attack() step by
step.  These runs pin what a USER sees after hundreds of chaotic Adam steps -- which instances end up attacked, how good
the best adversarial cloud is, how the binary search moved the trade-off constant, the level of the objective over
time -- where any two fp32 implementations have long since left each other's trajectory and only statistics can be
compared.  The reference's own attack() (Attacker/geoA3_attack.py:182-386) is imported through make_golden.py's shims
and run with the harness defaults' structure (main_attack.py:345-347: binary search x Adam loop, full objective
CE + CD + HD + curvature) at sizes a CPU finishes in minutes.  Only inputs and the outputs it produced are stored.
"""
from __future__ import annotations

import io
import os
import sys
import time

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, REPO)

from tests.golden import make_golden as MG  # noqa: E402
from oracle import geoa3_oracle as O  # noqa: E402

# tag -> (cfg overrides, b, N, seed).  initial_const / lr are the harness defaults (main_attack.py:343,350) unless a
# case needs the binary search to move both ways within four outer steps.
LONG_CASES = {
    "n256_b8": (dict(curv_loss_knn=16, binary_max_steps=4, iter_max_steps=200, lr=0.01, initial_const=10.0), 8, 256, 501),
    "n256_b8_hard": (dict(curv_loss_knn=16, binary_max_steps=5, iter_max_steps=120, lr=0.003, initial_const=3000.0),
                     8, 256, 502),
    "n1024_b4": (dict(curv_loss_knn=16, binary_max_steps=3, iter_max_steps=150, lr=0.01, initial_const=10.0), 4, 1024, 503),
}


def main():
    if not os.path.isdir(MG.REF):
        sys.exit("needs /root/reference (build container only)")
    MG.install_shims()
    from PointNet import PointNet as RefPointNet
    from Attacker import geoA3_attack as RA
    torch.set_num_threads(8)
    sd = O.make_pointnet_state_dict(40, seed=0)
    net = RefPointNet(40)
    net.load_state_dict(sd)
    net.eval()
    out = {"long/sd_checksum": np.float64(sum(float(v.double().abs().sum()) for v in sd.values()))}
    for tag, (kw, b, N, seed) in LONG_CASES.items():
        cfg = MG.ref_cfg(npoint=N, **kw)
        ori, nrm = O.make_synthetic_clouds(b, N, seed)
        with torch.no_grad():
            gt = net(ori).argmax(1)
        g = torch.Generator().manual_seed(seed + 1000)
        inits = [torch.randn(b, 3, N, generator=g) * 1e-3 for _ in range(cfg.binary_max_steps)]
        it = iter(inits)

        def fake_normal_(t, mean=0.0, std=1.0):
            with torch.no_grad():
                t.copy_(next(it))
            return t

        tr = dict(loss_n=[], constrain=[], cls=[], cd=[], hd=[], curv=[], pred=[], margin=[], scale=[])
        real_fs = RA._forward_step

        def fs_spy(net_, pc_ori, x, normal_ori, kappa, target, scale_const, *a, **k):
            r = real_fs(net_, pc_ori, x, normal_ori, kappa, target, scale_const, *a, **k)
            logits = r[0].detach()
            tr["loss_n"].append(r[3].detach().clone())
            tr["cls"].append(r[4].detach().clone())
            tr["cd"].append(r[5].detach().clone())
            tr["hd"].append(r[6].detach().clone())
            tr["curv"].append(r[7].detach().clone())
            tr["constrain"].append(r[8].detach().clone())
            tr["pred"].append(logits.argmax(1))
            top2 = logits.topk(2, dim=1).values
            tr["margin"].append((top2[:, 0] - top2[:, 1]).clone())
            tr["scale"].append(scale_const.detach().clone())
            return r

        real_normal_ = nn.init.normal_
        nn.init.normal_ = fake_normal_
        RA._forward_step = fs_spy
        data = [ori.permute(0, 2, 1).unsqueeze(1).contiguous(), nrm.permute(0, 2, 1).unsqueeze(1).contiguous(),
                gt.view(b, 1)]
        so, t0 = sys.stdout, time.time()
        sys.stdout = io.StringIO()
        try:
            best, target, succ, best_step, all_loss = RA.attack(net, data, cfg, 0, 1, None)
        finally:
            sys.stdout = so
            nn.init.normal_ = real_normal_
            RA._forward_step = real_fs
        pre = "long/%s/" % tag
        S, T = cfg.binary_max_steps, cfg.iter_max_steps
        st = lambda key: torch.stack(tr[key]).view(S, T, b).numpy()
        out[pre + "ori"], out[pre + "nrm"], out[pre + "gt"] = MG.t2n(ori), MG.t2n(nrm), MG.t2n(gt)
        out[pre + "inits"] = np.stack([MG.t2n(t) for t in inits])
        out[pre + "best_attack"] = MG.t2n(best)
        out[pre + "success"] = np.asarray(succ)
        out[pre + "best_step"] = np.asarray(best_step, dtype=np.int64)
        out[pre + "all_loss"] = np.asarray(all_loss, dtype=np.float32)
        for key in ("loss_n", "constrain", "cls", "cd", "hd", "curv", "margin", "scale"):
            out[pre + "tr_" + key] = st(key).astype(np.float32)
        out[pre + "tr_pred"] = st("pred").astype(np.int16)
        # what the bookkeeping at geoA3_attack.py:301-310 records: the iterate of step s is checked with the constrain loss
        # of step s-1 (1e10 at step 0), strict '<' against the best so far
        pred, con = st("pred"), st("constrain")
        gt_n = gt.numpy()
        best_con = np.full(b, 1e10, dtype=np.float32)
        for s in range(S):
            for t in range(1, T):
                ok = (pred[s, t] != gt_n) & (con[s, t - 1] < best_con)
                best_con = np.where(ok, con[s, t - 1], best_con)
        out[pre + "best_constrain"] = best_con
        assert ((best_con < 1e10) == np.asarray(succ)).all()
        out[pre + "cfg"] = np.array(repr(sorted(vars(cfg).items())))
        print("%s: %d x %d steps, b=%d N=%d in %.0f s; success %s; best_step %s; scale_const per binary step:\n%s"
              % (tag, S, T, b, N, time.time() - t0, np.asarray(succ).astype(int).tolist(), list(best_step),
                 st("scale")[:, 0, :]))
    out["long/cases"] = np.array(list(LONG_CASES))
    path = os.path.join(HERE, "geoa3_golden_long.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), len(out), "arrays")


if __name__ == "__main__":
    main()
