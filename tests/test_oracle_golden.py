"""The CPU oracle (oracle/geoa3_oracle.py) against the golden fixtures produced by the
reference's own Python (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import geoa3_oracle as O

T = torch.from_numpy


def _close(a, b, rtol=1e-5, atol=1e-7):
    np.testing.assert_allclose(np.asarray(a), np.asarray(b), rtol=rtol, atol=atol)


def _traj_close(a, b, tight=2e-5, frac=0.995, loose=2e-3):
    """Adam turns a 1e-9 gradient difference on a near-zero gradient into a +-lr step, so a
    handful of coordinates may drift; require `frac` of them within `tight`, all within `loose`."""
    d = np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))
    assert d.max() <= loose, d.max()
    assert (d <= tight).mean() >= frac, (d <= tight).mean()


@pytest.mark.parametrize("tag", ["n64k2", "n64k16", "n256k16", "n128k32", "dup", "zero"])
def test_losses_and_grads(golden, tag):
    pre = "ops/%s/" % tag
    ori, nrm = T(golden[pre + "ori"]), T(golden[pre + "nrm"])
    adv = T(golden[pre + "adv"]).clone().requires_grad_()
    k = int(golden[pre + "k"])
    kap_ori = O.get_kappa_ori(ori, nrm, k)
    _close(kap_ori, golden[pre + "kappa_ori"])
    vals = dict(cd=O.chamfer_loss(adv, ori), pcd=O.pseudo_chamfer_loss(adv, ori), hd=O.hausdorff_loss(adv, ori),
                l2=O.norm_l2_loss(adv, ori))
    kap_adv, nrm_adv = O.get_kappa_adv(adv, ori, nrm, k)
    vals["curv"] = O.curvature_loss(adv, ori, kap_adv, kap_ori)
    _close(kap_adv.detach(), golden[pre + "kappa_adv"])
    _close(nrm_adv.detach(), golden[pre + "normal_adv"])
    for name, v in vals.items():
        _close(v.detach(), golden[pre + name])
        (g,) = torch.autograd.grad(v.sum(), adv, retain_graph=True)
        _close(g, golden[pre + "g_" + name], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("tag", ["n64", "n256", "n1024"])
def test_pointnet(golden, tag):
    sd = O.make_pointnet_state_dict(40, seed=0)
    chk = sum(float(v.double().abs().sum()) for v in sd.values())
    assert abs(chk - float(golden["pn/sd_checksum"])) < 1e-6 * chk, "synthetic weights drifted"
    pre = "pn/%s/" % tag
    pc = T(golden[pre + "pc"]).clone().requires_grad_()
    logits = O.pointnet_forward(sd, pc)
    _close(logits.detach(), golden[pre + "logits"], rtol=1e-5, atol=1e-5)
    (g,) = torch.autograd.grad((logits * T(golden[pre + "w"])).sum(), pc)
    _close(g, golden[pre + "g_pc"], rtol=1e-4, atol=1e-5)
    # eval-mode batch-1 == batched: what lets one batched forward replace b forwards
    assert np.abs(golden[pre + "logits_batch1"] - golden[pre + "logits"]).max() < 2e-4
    assert (golden[pre + "logits_batch1"].argmax(1) == golden[pre + "logits"].argmax(1)).all()


@pytest.mark.parametrize("tag,kw,targeted", [
    ("ce_untarget", dict(cls_loss_type="CE", attack_label="Untarget"), False),
    ("ce_target", dict(cls_loss_type="CE", attack_label="All"), True),
    ("margin_target", dict(cls_loss_type="Margin", attack_label="All", confidence=0.5), True),
    ("margin_untarget", dict(cls_loss_type="Margin", attack_label="Untarget"), False),
    ("l2_nohd", dict(dis_loss_type="L2", hd_loss_weight=0.0, curv_loss_weight=0.0), False),
    ("pcd", dict(is_cd_single_side=True), False)])
def test_forward_step(golden, tag, kw, targeted):
    sd = O.make_pointnet_state_dict(40, seed=0)
    net = lambda x: O.pointnet_forward(sd, x)
    cfg = O.AttackCfg(curv_loss_knn=8, **kw)
    pre = "fs/%s/" % tag
    ori, nrm = T(golden[pre + "ori"]), T(golden[pre + "nrm"])
    x = T(golden[pre + "x"]).clone().requires_grad_()
    target = T(golden[pre + "target"])
    kap = O.get_kappa_ori(ori, nrm, cfg.curv_loss_knn) if cfg.curv_loss_weight != 0 else None
    r = O.forward_step(net, ori, x, nrm, kap, target, T(golden[pre + "scale_const"]), cfg, targeted)
    logits, _, loss, loss_n, cls_loss, dis, hd, curv, constrain = r
    _close(logits.detach(), golden[pre + "logits"], rtol=1e-5, atol=1e-5)
    for name, v in [("loss", loss), ("loss_n", loss_n), ("cls_loss", cls_loss), ("dis_loss", dis),
                    ("hd_loss", hd), ("curv_loss", curv), ("constrain", constrain)]:
        v = v.detach().numpy() if torch.is_tensor(v) else np.float32(v)
        _close(v, golden[pre + name], rtol=2e-5, atol=1e-5)
    (g,) = torch.autograd.grad(loss, x)
    _close(g, golden[pre + "g_x"], rtol=1e-4, atol=1e-6)


def test_adam_matches_torch_optim(golden):
    ps, gs = golden["adam/params"], golden["adam/grads"]
    p = T(ps[0]).clone()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for t in range(gs.shape[0]):
        O.adam_step(p, T(gs[t]), m, v, t + 1, 0.01)
        _close(p, ps[t + 1], rtol=1e-6, atol=1e-9)


from tests.golden.make_golden import ATK_CASES  # configs only; the generator itself needs /root/reference

ATK = {k: (v[0], v[1]) for k, v in ATK_CASES.items()}


@pytest.mark.parametrize("tag", list(ATK))
@pytest.mark.parametrize("faithful", [False, True])
def test_attack_trajectory(golden, tag, faithful):
    kw, targeted = ATK[tag]
    cfg = O.AttackCfg(**kw)
    sd = O.make_pointnet_state_dict(40, seed=0)
    net = lambda x: O.pointnet_forward(sd, x)
    pre = "atk/%s/" % tag
    ori, nrm, gt, tgt = (T(golden[pre + n]) for n in ("ori", "nrm", "gt", "tgt"))
    inits = [T(a) for a in golden[pre + "inits"]]
    tr = {}
    best, target, succ, best_step, all_loss = O.attack(net, ori, nrm, gt, tgt if targeted else None, cfg, inits,
                                                       faithful_success_check=faithful, trace=tr)
    xs = golden[pre + "tr_x"]
    got_x = np.stack([(ori + o).numpy() for o in tr["offsets"]])
    # short trajectories: iterates agree to fp32 round-off (SURVEY §7 'hard parts')
    _traj_close(got_x, xs)
    _close(np.stack([t.numpy() for t in tr["loss_n"]]), golden[pre + "tr_loss_n"], rtol=1e-3, atol=1e-4)
    _close(np.stack([t.numpy() for t in tr["constrain"]]), golden[pre + "tr_constrain"], rtol=1e-3, atol=1e-5)
    assert (np.asarray(tr["labels"]) == golden[pre + "tr_logits"].argmax(-1)).all()
    assert (np.asarray(succ) == golden[pre + "success"]).all()
    assert list(best_step) == list(golden[pre + "best_step"])
    _traj_close(best.numpy(), golden[pre + "best_attack"])
    assert (target.numpy() == golden[pre + "target"]).all()
    _close(np.asarray(all_loss, dtype=np.float32), golden[pre + "all_loss"], rtol=1e-3, atol=1e-4)


def test_shard_invariance_with_global_divisor(golden):
    """rows of a b=6 run == the two b=3 shard runs when the loss divisor is the
    GLOBAL batch and the shards get the global last-instance label (the output_label quirk)."""
    kw, _ = ATK["untarget_mixed"]
    cfg = O.AttackCfg(**kw)
    sd = O.make_pointnet_state_dict(40, seed=0)
    net = lambda x: O.pointnet_forward(sd, x)
    pre = "atk/untarget_mixed/"
    ori, nrm, gt = (T(golden[pre + n]) for n in ("ori", "nrm", "gt"))
    inits = [T(a) for a in golden[pre + "inits"]]
    tr = {}
    full = O.attack(net, ori, nrm, gt, None, cfg, inits, trace=tr)
    for lo, hi in [(0, 3), (3, 6)]:
        part = O.attack(net, ori[lo:hi], nrm[lo:hi], gt[lo:hi], None, cfg, [i[lo:hi] for i in inits],
                        loss_divisor=6, last_label_override=tr["last_label"])
        _close(part[0].numpy(), full[0][lo:hi].numpy(), rtol=0, atol=1e-6)
        assert (part[2] == full[2][lo:hi]).all()
        assert list(part[3]) == list(full[3][lo:hi])
