#!/usr/bin/env python3
"""Long-horizon reference runs -> tests/golden/geoa3_golden_long.npz (build container only).

Run:  python tests/golden/make_golden_long.py [--only TAG]    (needs /root/reference; CPU only, ~40 minutes on 8 cores;
                                                                --only regenerates one case and keeps the others)

The short trajectories of make_golden.py (<= 8 inner x <= 4 binary steps, N = 64) pin every branch of attack() step by
step.  These runs pin what a USER sees after hundreds of chaotic Adam steps -- which instances end up attacked, how good
the best adversarial cloud is, how the binary search moved the trade-off constant, the level of the objective over
time -- where any two fp32 implementations have long since left each other's trajectory and only statistics can be
compared.  The reference's own attack() (Attacker/geoA3_attack.py:182-386) is imported through make_golden.py's shims
and run with the harness' structure (main_attack.py:345-347: binary search x Adam loop, full objective
CE + CD + HD + curvature) at sizes a CPU finishes in minutes.  Only inputs and the outputs it produced are stored.

Round 4: every case is HARD -- the reference's best iterate is found tens of steps into a binary step (median best_step
> 10 in every case; the round-3 cases with the harness' default constant fell at step 1-3 and are gone), the binary search
moves the constants both ways, and some instances fail.  What makes a case hard on a synthetic victim:
  * `hard`:    untargeted, trade-off constant 3000 and lr 0.003 (the constrain term dominates; Adam creeps over the boundary);
  * `tgt`:     TARGETED at the clean cloud's least likely class (main_attack.py:330 default `--attack_label All` is a
               targeted mode; geoA3_attack.py:189-192, 211-214), constant 500;
  * `margin`:  the victim's last layer scaled x3 (clean logit gap to the target >= 8, top-2 margin x3), targeted, the
               harness' default constant 10 and lr 0.01;
  * `pn2`:     the PointNet++ SSG victim (Model/PointNetPP_ssg.py:106-124), last layer calibrated as PointNet's, targeted.
Round 5 adds `n256_b8_fail` (targeted at the least likely class of a x3-margin victim under constant 2600 and lr 0.005 (LONG_CASES below): the
reference leaves several of the eight instances un-attacked after 3 x 100 steps) and `pn2_n1024_b8_tgt` (PointNet++,
eight instances, targeted at the least likely class -- an UNTARGETED run on this victim is adversarial from step 1 at any
constant and a run targeted at the 10th-ranked class from step 2: both measured (best step 1-2 for all eight), no test of
anything).
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

# tag -> case.  cfg: overrides of the harness defaults; arch: victim; logit_scale: factor on the victim's last layer;
# target_rank: 0 = untargeted, r > 0 = targeted at the class ranked r-th by the clean logits (39 = least likely).
LONG_CASES = {
    "n256_b8_hard": dict(cfg=dict(curv_loss_knn=16, binary_max_steps=5, iter_max_steps=120, lr=0.003, initial_const=3000.0),
                         b=8, n=256, seed=502, arch="PointNet", logit_scale=1.0, target_rank=0),
    "n256_b8_tgt": dict(cfg=dict(curv_loss_knn=16, binary_max_steps=5, iter_max_steps=100, lr=0.01, initial_const=500.0,
                                 attack_label="All"),
                        b=8, n=256, seed=601, arch="PointNet", logit_scale=1.0, target_rank=39),
    "n1024_b8_hard": dict(cfg=dict(curv_loss_knn=16, binary_max_steps=4, iter_max_steps=150, lr=0.003, initial_const=3000.0),
                          b=8, n=1024, seed=601, arch="PointNet", logit_scale=1.0, target_rank=0),
    "n1024_b4_margin": dict(cfg=dict(curv_loss_knn=16, binary_max_steps=3, iter_max_steps=150, lr=0.01, initial_const=10.0,
                                     attack_label="All"),
                            b=4, n=1024, seed=603, arch="PointNet", logit_scale=3.0, target_rank=39),
    "pn2_n1024_b4_tgt": dict(cfg=dict(curv_loss_knn=16, binary_max_steps=3, iter_max_steps=100, lr=0.01, initial_const=500.0,
                                      attack_label="All"),
                             b=4, n=1024, seed=601, arch="PointNetPP", logit_scale=1.0, target_rank=39),
    # round 5: a case the reference does NOT break everywhere (the "never attacked: all-ones placeholder, the constant only
    # grows" branch of geoA3_attack.py:225-227, 374-386 over a long run), and a second PointNet++ run at b = 8
    "n256_b8_fail": dict(cfg=dict(curv_loss_knn=16, binary_max_steps=3, iter_max_steps=100, lr=0.005, initial_const=2600.0,
                                  attack_label="All"),
                         b=8, n=256, seed=606, arch="PointNet", logit_scale=3.0, target_rank=39),
    "pn2_n1024_b8_tgt": dict(cfg=dict(curv_loss_knn=16, binary_max_steps=3, iter_max_steps=100, lr=0.01, initial_const=500.0,
                                      attack_label="All"),
                             b=8, n=1024, seed=607, arch="PointNetPP", logit_scale=1.0, target_rank=39),
}


def victim_state_dict(case):
    """The synthetic victim of a case (shared with the tests: the fixture stores a checksum of it)."""
    if case["arch"] == "PointNet":
        sd = O.make_pointnet_state_dict(40, seed=0)
        last = "fc3"
    else:
        from oracle import pointnet2_oracle as P2
        sd = P2.calibrate_pn2_state_dict(P2.make_pn2_state_dict(0))
        last = "fc_layer.7"
    if case["logit_scale"] != 1.0:
        sd = dict(sd)
        sd[last + ".weight"] = sd[last + ".weight"] * case["logit_scale"]
        sd[last + ".bias"] = sd[last + ".bias"] * case["logit_scale"]
    return sd


def oracle_net(case, sd=None):
    sd = sd if sd is not None else victim_state_dict(case)
    if case["arch"] == "PointNet":
        return lambda x: O.pointnet_forward(sd, x)
    from oracle import pointnet2_oracle as P2
    return lambda x: P2.pointnet2_ssg_forward(sd, x)


def sd_checksum(sd):
    return float(sum(float(v.double().abs().sum()) for v in sd.values()))


def adversarial(pred, gt, tgt, targeted):
    return (pred == tgt) if targeted else (pred != gt)


def run_case(tag, case):
    from Attacker import geoA3_attack as RA
    sd = victim_state_dict(case)
    if case["arch"] == "PointNet":
        from PointNet import PointNet as RefPointNet
        net = RefPointNet(40)
    else:
        from PointNetPP_ssg import PointNet2ClassificationSSG as RefSSG
        net = RefSSG(use_xyz=True, use_normal=False)
    net.load_state_dict(sd)
    net.eval()
    b, N, seed = case["b"], case["n"], case["seed"]
    cfg = MG.ref_cfg(npoint=N, **case["cfg"])
    targeted = case["target_rank"] > 0
    ori, nrm = O.make_synthetic_clouds(b, N, seed)
    with torch.no_grad():
        clean = net(ori)
    gt = clean.argmax(1)
    tgt = clean.argsort(1, descending=True)[:, case["target_rank"]] if targeted else gt.clone()
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
    if targeted:
        data.append(tgt.view(b, 1))
    so, t0 = sys.stdout, time.time()
    sys.stdout = io.StringIO()
    try:
        best, target, succ, best_step, all_loss = RA.attack(net, data, cfg, 0, 1, None)
    finally:
        sys.stdout = so
        nn.init.normal_ = real_normal_
        RA._forward_step = real_fs
    out = {}
    pre = "long/%s/" % tag
    S, T = cfg.binary_max_steps, cfg.iter_max_steps
    st = lambda key: torch.stack(tr[key]).view(S, T, b).numpy()
    out[pre + "sd_checksum"] = np.float64(sd_checksum(sd))
    out[pre + "ori"], out[pre + "nrm"], out[pre + "gt"], out[pre + "tgt"] = MG.t2n(ori), MG.t2n(nrm), MG.t2n(gt), MG.t2n(tgt)
    out[pre + "clean_logits"] = MG.t2n(clean)
    out[pre + "inits"] = np.stack([MG.t2n(t) for t in inits])
    out[pre + "best_attack"] = MG.t2n(best)
    out[pre + "target"] = MG.t2n(target)
    out[pre + "success"] = np.asarray(succ)
    out[pre + "best_step"] = np.asarray(best_step, dtype=np.int64)
    out[pre + "all_loss"] = np.asarray(all_loss, dtype=np.float32)
    for key in ("loss_n", "constrain", "cls", "cd", "hd", "curv", "margin", "scale"):
        out[pre + "tr_" + key] = st(key).astype(np.float32)
    out[pre + "tr_pred"] = st("pred").astype(np.int16)
    # what the bookkeeping at geoA3_attack.py:301-310 records: the iterate of step s is checked with the constrain loss
    # of step s-1 (1e10 at step 0), strict '<' against the best so far
    pred, con = st("pred"), st("constrain")
    gt_n, tgt_n = gt.numpy(), tgt.numpy()
    best_con = np.full(b, 1e10, dtype=np.float32)
    for s in range(S):
        for t in range(1, T):
            ok = adversarial(pred[s, t], gt_n, tgt_n, targeted) & (con[s, t - 1] < best_con)
            best_con = np.where(ok, con[s, t - 1], best_con)
    out[pre + "best_constrain"] = best_con
    assert ((best_con < 1e10) == np.asarray(succ)).all()
    out[pre + "cfg"] = np.array(repr(sorted(vars(cfg).items())))
    adv = adversarial(pred, gt_n, tgt_n, targeted)
    print("%s: %d x %d steps, b=%d N=%d in %.0f s; success %s; best_step %s (median %.0f); adversarial steps per binary "
          "step:\n%s\nscale_const per binary step:\n%s"
          % (tag, S, T, b, N, time.time() - t0, np.asarray(succ).astype(int).tolist(), list(best_step),
             float(np.median([s for s in best_step if s >= 0] or [-1])), adv.sum(1), st("scale")[:, 0, :]), flush=True)
    return out


def main():
    if not os.path.isdir(MG.REF):
        sys.exit("needs /root/reference (build container only)")
    only = sys.argv[sys.argv.index("--only") + 1] if "--only" in sys.argv else None
    MG.install_shims()
    torch.set_num_threads(8)
    path = os.path.join(HERE, "geoa3_golden_long.npz")
    out = {}
    if only:
        old = np.load(path, allow_pickle=False)
        out = {k: old[k] for k in old.files if k.startswith("long/") and k.split("/")[1] in LONG_CASES
               and k.split("/")[1] != only}
    for tag, case in LONG_CASES.items():
        if only and tag != only:
            continue
        out.update(run_case(tag, case))
    have = sorted({k.split("/")[1] for k in out if k.count("/") >= 2})
    out["long/cases"] = np.array([t for t in LONG_CASES if t in have])
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), len(out), "arrays; cases", have)


if __name__ == "__main__":
    main()
