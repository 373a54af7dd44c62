"""Long-horizon, run-level parity with the reference (SURVEY 7 'run-level statistics'; fixtures of
tests/golden/make_golden_long.py: the reference's own attack() for 5 x 120 / 5 x 100 / 4 x 150 / 3 x 150 / 3 x 100
iterations -- untargeted with a dominating constrain term, targeted at the least likely class, a victim with x3 logit
margins, N = 256 and 1024, the PointNet and the PointNet++ SSG victim.  In every case the reference finds its best
iterate tens of steps into a binary step, the constants move both ways).

After hundreds of Adam steps two fp32 implementations no longer share a trajectory (a 1e-9 difference on a near-zero
gradient becomes a +-lr step, an arg-min flips, ...), so what is compared is what a user of the attack sees:
  * which instances end up attacked (`success`), and the binary search's trade-off constants;
  * the quality of the best adversarial cloud (its constrain loss);
  * the level of the objective over time (window means of loss_n and of the constrain loss).
Both arithmetic modes of the MFMA layers and both summation modes of the objective's gradient are held to the same bars.

The `output_label` quirk (geoA3_attack.py:298,375: the label of the LAST instance at the LAST step decides for every
instance) makes the binary search hinge on one logit margin per binary step (5e-3 in `n256_b8_hard`): the reference's last
labels are replayed through the hook the sharded runs use, so the comparison is about the per-instance state -- and the
run's OWN last labels are held to the reference's wherever the reference's margin there is not a near-tie.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import geoa3_oracle as O
from tests.golden.make_golden_long import LONG_CASES, adversarial, oracle_net, victim_state_dict

pytestmark = pytest.mark.gpu
T = torch.from_numpy
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

# bars (relative unless noted); see DESIGN.md section 2, row "long run"
# Bars per case; see DESIGN.md section 2, row "long run".  They come from the chaos of the loop itself, measured with
# tools/longrun_noise.py (profiles/round4_longrun_noise.txt): 13 deterministic runs whose start is perturbed by k * 1e-7 and
# 13 repeats of the atomics loop, first binary step, against the reference's window means.
#   PointNet victim (four cases, both arithmetic modes): loss_n windows 0.02-0.16 % (median) / 0.34 % (max), constrain windows
#   <= 3.9 %, adversarial fraction equal to 1e-3 -- and the CPU oracle under the same perturbation 0.05-0.25 % / <= 1.6 %.
#   PointNet++ victim (farthest-point sampling and ball queries make the loss DISCONTINUOUS in the iterate; b = 4): loss_n
#   windows 6.0 % (median) / 8.6 % (max), constrain 5.9 % / 8.3 %, adversarial fraction 0.03 / 0.10 -- and the CPU ORACLE
#   perturbed the same way leaves the reference run by 4.1 % / 7.7 %, 4.7 % / 5.8 % and 0.045: the case is that chaotic on any
#   implementation, its bars are wider by that much (windows of the LATER binary steps were seen at 12.3 %: bar 18 %).
BARS = {   # best: best constrain loss per instance; window: 50-step means of loss_n (constrain: twice that); adv: fraction
    "n256_b8_hard": dict(best=0.10, window=0.025, adv=0.05, last_margin=0.05, first=5e-3),
    "n256_b8_tgt": dict(best=0.10, window=0.025, adv=0.05, last_margin=0.05, first=5e-3),
    "n1024_b8_hard": dict(best=0.10, window=0.025, adv=0.05, last_margin=0.05, first=5e-3),
    "n1024_b4_margin": dict(best=0.10, window=0.025, adv=0.05, last_margin=0.05, first=5e-3),
    # (round 6: the WHOLE run 13 times per mode, profiles/round6_longrun_noise_pn2.txt -- the float-atomics mode reached 20.4 %
    #  in one of 13 runs, the perturbed deterministic loop 15.4 % / 0.125: the round-5 bars (0.18 / 0.15) sat INSIDE the loop's own
    #  spread; bars = 1.25 x the largest of the 13 runs)
    "pn2_n1024_b4_tgt": dict(best=0.25, window=0.26, adv=0.16, last_margin=1.0, first=5e-3, first_steps=3),
    # round 5: instances the reference never breaks (their success flag, the all-ones placeholder and the constants are held
    # exactly: they are "robust" by construction), and PointNet++ at b = 8 (twice the instances to average over)
    "n256_b8_fail": dict(best=0.10, window=0.025, adv=0.05, last_margin=0.05, first=5e-3),
    # (eight instances average the PointNet++ case's chaos down: windows seen at 5.7-8.6 % / adversarial fractions within 0.04
    #  in both summation modes, against 5-9 % / 0.06 at b = 4: two thirds of that case's bars)
    #  round 6, measured at b = 8 itself: whole runs reach 12.1 % (perturbed deterministic) / 9.6 % (atomics), adversarial
    #  fractions 0.07: bars = 1.25 x that)
    "pn2_n1024_b8_tgt": dict(best=0.25, window=0.16, adv=0.10, last_margin=1.0, first=5e-3, first_steps=4),
}
ROBUST_STEPS = 3         # an instance counts as robustly (un)successful in a binary step with >= 3 / 0 adversarial steps


@pytest.fixture(scope="module")
def long_golden():
    return np.load(os.path.join(REPO, "tests", "golden", "geoa3_golden_long.npz"), allow_pickle=False)


def _net(mode, case=None):
    case = case or LONG_CASES["n256_b8_hard"]
    sd = victim_state_dict(case)
    if case["arch"] == "PointNet":
        from geoa3_amd.pointnet import PointNet
        n = PointNet(40)
        n.load_state_dict(sd)
        n = n.cuda().eval()
        n.wide_mode = mode
        return n
    from geoa3_amd.pointnet2 import PointNet2ClassificationSSG
    n = PointNet2ClassificationSSG(use_xyz=True, use_normal=False)
    n.load_state_dict(sd)
    return n.cuda().eval()


def _run(net, cfg, g, pre, deterministic, replay_last):
    from geoa3_amd.attack import AttackRunner
    cfg.deterministic = deterministic
    ori, nrm, gt, tgt = T(g[pre + "ori"]), T(g[pre + "nrm"]), T(g[pre + "gt"]), T(g[pre + "tgt"])
    inits = [T(a).cuda() for a in g[pre + "inits"]]
    b, _, n = ori.shape
    r = AttackRunner(net, b, n, cfg, torch.device("cuda"))
    r.setup(ori, nrm, gt, tgt)
    S, Tn = cfg.binary_max_steps, cfg.iter_max_steps
    scale, hist, labels, con = [], [], [], []
    begin = r.begin_search_step

    def begin_spy(init):
        scale.append(r.t["scale_const"].cpu().numpy().copy())
        begin(init)

    r.begin_search_step = begin_spy

    def on_step(s, step):
        labels.append(r.t["label"].clone())
        con.append(r.geo_out["constrain"].clone())
        if step == Tn - 1:
            hist.append(r.t["loss_hist"].cpu().numpy().copy())

    ref_last = g[pre + "tr_pred"][:, -1, -1]
    state = {"s": 0}
    own_last = []

    def sync(last_label):
        own_last.append(int(last_label.item()))
        if replay_last:
            last_label.fill_(int(ref_last[state["s"]]))
        state["s"] += 1

    r.run(inits, on_step=on_step, sync_last_label=sync)
    best, target, succ, best_step, all_loss = r.results()
    return dict(succ=np.asarray(succ), best_loss=r.t["best_loss"].cpu().numpy(), best_step=np.asarray(best_step),
                scale=np.stack(scale), loss_n=np.stack(hist),
                pred=torch.stack(labels).cpu().numpy().reshape(S, Tn, b),
                con=torch.stack(con).cpu().numpy().reshape(S, Tn, b), best=best.cpu().numpy(),
                own_last=np.asarray(own_last))


def _params():
    out = []
    for tag, case in LONG_CASES.items():
        if case["arch"] == "PointNet":
            out += [(tag, "f16x2", True), (tag, "f32", True), (tag, "f16x2", False)]
        else:   # the PointNet++ kernels have one arithmetic mode
            out += [(tag, "native", True), (tag, "native", False)]
    return out


@pytest.mark.timeout(1500)
@pytest.mark.parametrize("tag,mode,deterministic", _params())
def test_long_run_statistics_match_reference(long_golden, tag, mode, deterministic):
    g = long_golden
    case = LONG_CASES[tag]
    b = case["b"]
    cfg = O.AttackCfg(**case["cfg"])
    pre = "long/%s/" % tag
    targeted = case["target_rank"] > 0
    hard = True     # every case: constants compared per instance, the reference's last labels replayed
    out = _run(_net(mode, case), cfg, g, pre, deterministic, replay_last=True)
    S, Tn = cfg.binary_max_steps, cfg.iter_max_steps
    gt, tgt = g[pre + "gt"], g[pre + "tgt"]
    ref_adv = adversarial(g[pre + "tr_pred"], gt, tgt, targeted)     # [S,T,b] adversarial at step t (batched forward)
    got_adv = adversarial(out["pred"], gt, tgt, targeted)
    ref_n, got_n = ref_adv[:, 1:].sum(1), got_adv[:, 1:].sum(1)   # adversarial steps per (binary step, instance)
    report = {"tag": tag, "mode": mode, "deterministic": deterministic}
    fails = []

    def chk(cond, *msg):
        if not cond:
            fails.append(repr(msg))

    # (0) the quirk's input: the run's own label of the last instance at the last step of every binary step, wherever
    # the reference's top-2 margin there is not a near-tie (while the instance still follows the reference's constants)
    ref_margin_last = g[pre + "tr_margin"][:, -1, -1]
    bars = BARS[tag]
    robust_last = ref_margin_last > bars["last_margin"]
    report["last_label"] = [out["own_last"].tolist(), g[pre + "tr_pred"][:, -1, -1].tolist(), ref_margin_last.round(4).tolist()]
    if robust_last[0]:
        chk(out["own_last"][0] == g[pre + "tr_pred"][0, -1, -1], "last label of binary step 0", report["last_label"])

    # (1) success mask: equal for every instance the reference attacks robustly (or never)
    ref_succ = g[pre + "success"]
    robust = (ref_n.sum(0) >= ROBUST_STEPS) | (ref_n.sum(0) == 0)
    chk((out["succ"] == ref_succ)[robust].all(), "success", out["succ"], ref_succ)
    report["success"] = [out["succ"].tolist(), ref_succ.tolist()]
    # ... and an instance that was never attacked keeps the all-ones placeholder (geoA3_attack.py:225-227)
    never = ~out["succ"]
    chk((out["best"][never] == 1.0).all(), "placeholder of the instances never attacked", never)

    # (2) fraction of adversarial steps per binary step (batch level)
    fa_ref, fa_got = ref_adv.mean((1, 2)), got_adv.mean((1, 2))
    report["adv_fraction"] = [fa_got.round(4).tolist(), fa_ref.round(4).tolist()]
    chk(np.abs(fa_got - fa_ref).max() <= bars["adv"], "adversarial fraction", fa_got, fa_ref)

    # (3) the binary search: trade-off constant at the start of every binary step, for instances whose success within
    # the previous binary steps is robust in BOTH runs (>= ROBUST_STEPS adversarial steps, or none)
    ref_scale = g[pre + "tr_scale"][:, 0, :]
    clear = np.ones(b, bool)
    for s in range(S):
        ok = clear.copy()
        report.setdefault("scale_const", []).append([out["scale"][s].tolist(), ref_scale[s].tolist()])
        chk(np.allclose(out["scale"][s][ok], ref_scale[s][ok], rtol=1e-6), "scale_const", s, out["scale"][s], ref_scale[s], ok)
        clear &= ((ref_n[s] >= ROBUST_STEPS) | (ref_n[s] == 0)) & ((got_n[s] >= ROBUST_STEPS) | (got_n[s] == 0)) & \
                 ((ref_n[s] > 0) == (got_n[s] > 0))
    report["scale_const_compared"] = int(clear.sum())
    chk(clear.mean() >= 0.5, "instances compared", clear)

    # (4) quality of the best adversarial cloud
    both = out["succ"] & ref_succ
    ratio = out["best_loss"][both] / g[pre + "best_constrain"][both]
    report["best_constrain_ratio"] = ratio.round(4).tolist()
    rt = bars["best"]
    chk(np.median(np.abs(ratio - 1.0)) <= rt / 2, "best constrain (median)", ratio)
    chk((np.abs(ratio - 1.0) <= rt).mean() >= 0.75, "best constrain", ratio)
    # ... and it IS adversarial with that loss: re-evaluated by the oracle on the returned cloud
    onet = oracle_net(case)
    with torch.no_grad():
        re_pred = onet(T(out["best"][both])).argmax(1).numpy()
    chk(adversarial(re_pred, gt[both], tgt[both], targeted).mean() >= 0.85, "best clouds adversarial", re_pred, gt[both], tgt[both])

    # (5) level of the objective over time: 50-step windows of the batch mean of loss_n and of the constrain loss
    wr = bars["window"]
    ref_ln, got_ln = g[pre + "tr_loss_n"].mean(2), out["loss_n"].mean(2)     # [S,T]
    ref_con, got_con = g[pre + "tr_constrain"].mean(2), out["con"].mean(2)
    rows = []
    for s in range(S):
        if hard and not clear.all() and s > 0:
            # instances whose constants differ follow a different objective from here on: compare the others
            sel = np.isclose(out["scale"][s], ref_scale[s], rtol=1e-6)
            ref_ln[s], got_ln[s] = g[pre + "tr_loss_n"][s][:, sel].mean(1), out["loss_n"][s][:, sel].mean(1)
            ref_con[s], got_con[s] = g[pre + "tr_constrain"][s][:, sel].mean(1), out["con"][s][:, sel].mean(1)
        for w in range(0, Tn, 50):
            a, c = got_ln[s, w:w + 50].mean(), ref_ln[s, w:w + 50].mean()
            a2, c2 = got_con[s, w:w + 50].mean(), ref_con[s, w:w + 50].mean()
            rows.append([s, w, float(a), float(c), float(a2), float(c2)])
            chk(abs(a - c) <= wr * abs(c) + 1e-3, "loss_n window", s, w, a, c)
            chk(abs(a2 - c2) <= 2 * wr * abs(c2) + 1e-6, "constrain window", s, w, a2, c2)
    report["windows"] = rows
    # the first iterations are still a shared trajectory: tight
    dev = np.abs(out["loss_n"][0, :12] - g[pre + "tr_loss_n"][0, :12]) / (np.abs(g[pre + "tr_loss_n"][0, :12]) + 0.1)
    report["first_steps_max_rel_dev"] = dev.max(1).round(6).tolist()
    # (per instance; the median over the instances: one sign flip of Adam's first step on a near-zero gradient moves a
    # coordinate by 2 lr, which a PointNet++ victim turns into another farthest-point sample for THAT instance)
    nfirst = bars.get("first_steps", 6)      # (PointNet++: the median instance leaves the shared trajectory at step 4-6; with float atomics at step 3)
    # PointNet (no discontinuous sampling in the victim): EVERY instance holds the bar; PointNet++: the median instance
    first_dev = np.median(dev[:nfirst], axis=1) if case["arch"] == "PointNetPP" else dev[:nfirst].max(1)
    chk(first_dev.max() <= bars["first"], "first steps", dev.max(1), np.median(dev, axis=1))
    report["fails"] = fails
    outdir = os.path.join(REPO, "gpurun_out")
    if os.path.isdir(outdir):
        with open(os.path.join(outdir, "longrun_%s_%s_%d.json" % (tag, mode, int(deterministic))), "w") as f:
            json.dump(report, f)
    assert not fails, "\n".join(fails)
