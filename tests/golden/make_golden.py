#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own Python (build container only).

Run:  python tests/golden/make_golden.py            (needs /root/reference; CPU only)

The reference (Gorilla-Lab-SCUT/GeoA3) is imported unchanged from /root/reference through
shims for what this image lacks: pytorch3d / open3d / ipdb / seaborn /
torchvision are stub modules, ``.cuda()`` is the identity, ``stty size`` is answered, and the
third-party ``pytorch3d.ops.knn_points / knn_gather`` (absent, version unpinned) are supplied
by the dense formulation the reference keeps in comments (Lib/loss_utils.py:30-31,54-56).
Nothing of the reference is copied: only inputs and the outputs it produced are stored.
This script never runs on the GPU box (no /root/reference there); the fixtures travel.
"""
from __future__ import annotations

import argparse
import collections
import io
import os
import sys
import types

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, REPO)

from oracle import geoa3_oracle as O  # noqa: E402  (supplies the absent third-party K-NN)


# ---------------------------------------------------------------------------- shims
def install_shims():
    def stub(name, **attrs):
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    KNN = collections.namedtuple("KNN", ["dists", "idx", "knn"])

    def knn_points(p1, p2, K=1, **kw):
        d, i = O.knn_points(p1, p2, K)
        return KNN(d, i, None)

    stub("pytorch3d")
    stub("pytorch3d.ops", knn_points=knn_points, knn_gather=O.knn_gather,
         sample_points_from_meshes=None)
    stub("pytorch3d.io", load_obj=None, save_obj=None)
    stub("pytorch3d.structures", Meshes=None)
    stub("open3d")
    stub("ipdb", set_trace=lambda: None)
    stub("seaborn", set=lambda *a, **k: None)
    tv = stub("torchvision")
    tv.models = stub("torchvision.models")
    tv.transforms = stub("torchvision.transforms")
    import importlib
    importlib.import_module("torch.autograd.gradcheck")
    sys.modules["torch.autograd.gradcheck"].zero_gradients = lambda x: None
    real_popen = os.popen

    def popen(cmd, *a, **k):
        if cmd.strip().startswith("stty size"):
            return io.StringIO("24 80")
        return real_popen(cmd, *a, **k)

    os.popen = popen
    real_where = torch.where

    def where(cond, *a, **k):   # the reference passes uint8 masks (geoA3_attack.py:63,75); torch >= 2 wants bool
        return real_where(cond.bool() if torch.is_tensor(cond) and cond.dtype == torch.uint8 else cond, *a, **k)

    torch.where = where
    torch.Tensor.cuda = lambda self, *a, **k: self
    nn.Module.cuda = lambda self, *a, **k: self
    # the CUDA-only native module of PointNet++: registered BEFORE the package is imported so that
    # pointnet2_utils.py:7-31 neither JIT-compiles nor hipifies anything inside /root/reference
    from oracle import pointnet2_oracle as P2
    ext = stub("pointnet2_ops._ext")
    for name in ("furthest_point_sampling", "gather_points", "gather_points_grad", "ball_query", "group_points",
                 "group_points_grad"):
        setattr(ext, name, getattr(P2, name))
    import torch.utils.cpp_extension as cpp_ext

    def no_jit(*a, **k):
        raise RuntimeError("JIT build of the reference extension is blocked in the golden generator")

    cpp_ext.load = no_jit
    sys.path.insert(0, os.path.join(REF, "Model", "pointnet2_ops_lib"))
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(REF, "Lib"))
    sys.path.insert(0, os.path.join(REF, "Model"))


def ref_cfg(**kw):
    c = argparse.Namespace(
        arch="PointNet", classes=40, attack_label="Untarget", binary_max_steps=2, initial_const=10.0,
        iter_max_steps=5, optim="adam", lr=0.01, eval_num=1, cls_loss_type="CE", confidence=0.0,
        dis_loss_type="CD", dis_loss_weight=1.0, is_cd_single_side=False, hd_loss_weight=0.1,
        curv_loss_weight=1.0, curv_loss_knn=16, uniform_loss_weight=0.0, is_partial_var=False,
        knn_range=3, is_subsample_opt=False, is_use_lr_scheduler=False, cc_linf=0.0,
        is_real_offset=False, is_pro_grad=False, is_pre_jitter_input=False, is_debug=False,
        npoint=1024, is_save_normal=False)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


# tag -> (cfg overrides, targeted, batch, seed); shared with tests/test_oracle_golden.py
ATK_CASES = {
    "untarget_full": (dict(curv_loss_knn=4, binary_max_steps=3, iter_max_steps=6, lr=0.0015), False, 4, 31),
    "target_full": (dict(attack_label="All", curv_loss_knn=4, binary_max_steps=3, iter_max_steps=6,
                         initial_const=0.5, lr=0.003), True, 4, 32),
    "untarget_cdonly": (dict(hd_loss_weight=0.0, curv_loss_weight=0.0, binary_max_steps=2, iter_max_steps=8,
                             lr=0.002), False, 5, 33),
    "untarget_sched_clip": (dict(curv_loss_knn=4, binary_max_steps=2, iter_max_steps=6, is_use_lr_scheduler=True,
                                 cc_linf=0.004, hd_loss_weight=0.0), False, 3, 34),
    "untarget_mixed": (dict(curv_loss_knn=4, binary_max_steps=4, iter_max_steps=8, lr=0.002,
                            initial_const=2000.0), False, 6, 35),
    "untarget_lowlr": (dict(curv_loss_knn=4, binary_max_steps=4, iter_max_steps=8, lr=0.001,
                            initial_const=2.0), False, 6, 35),
    "pro_grad": (dict(curv_loss_knn=4, binary_max_steps=2, iter_max_steps=6, lr=0.002, is_pro_grad=True,
                      hd_loss_weight=0.0), False, 4, 37),
    "pro_grad_real_clip": (dict(curv_loss_knn=4, binary_max_steps=2, iter_max_steps=6, lr=0.002, is_pro_grad=True,
                                is_real_offset=True, cc_linf=0.01, hd_loss_weight=0.0), False, 4, 38),
    "margin_sgd": (dict(cls_loss_type="Margin", optim="sgd", lr=0.05, curv_loss_knn=4, binary_max_steps=2,
                        iter_max_steps=5), False, 3, 36),
}


def t2n(x):
    return x.detach().cpu().numpy() if torch.is_tensor(x) else np.asarray(x)


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    install_shims()
    import loss_utils as RL            # reference Lib/loss_utils.py
    from PointNet import PointNet as RefPointNet  # reference Model/PointNet.py
    from Attacker import geoA3_attack as RA       # reference Attacker/geoA3_attack.py

    out = {}

    # ---------------------------------------------------------------- (1) op-level losses
    cases = []
    for tag, N, k, seed, mode in [("n64k2", 64, 2, 1, "normal"), ("n64k16", 64, 16, 2, "normal"),
                                  ("n256k16", 256, 16, 3, "normal"), ("n128k32", 128, 32, 4, "normal"),
                                  ("dup", 64, 4, 5, "dup"), ("zero", 64, 4, 6, "zero")]:
        b = 3
        ori, nrm = O.make_synthetic_clouds(b, N, seed)
        g = torch.Generator().manual_seed(100 + seed)
        off = torch.randn(b, 3, N, generator=g) * 0.02
        if mode == "zero":
            off.zero_()
        adv = (ori + off).clone()
        if mode == "dup":   # duplicate points inside the adversarial cloud and onto originals
            adv[:, :, 1] = adv[:, :, 0]
            adv[:, :, 5] = ori[:, :, 9]
        adv.requires_grad_()
        kap_ori = RL._get_kappa_ori(ori, nrm, k)
        cd = RL.chamfer_loss(adv, ori)
        pcd = RL.pseudo_chamfer_loss(adv, ori)
        hd = RL.hausdorff_loss(adv, ori)
        l2 = RL.norm_l2_loss(adv, ori)
        kap_adv, nrm_adv = RL._get_kappa_adv(adv, ori, nrm, k)
        curv = RL.curvature_loss(adv, ori, kap_adv, kap_ori)
        grads = {}
        for name, val in [("cd", cd), ("pcd", pcd), ("hd", hd), ("l2", l2), ("curv", curv)]:
            (gr,) = torch.autograd.grad(val.sum(), adv, retain_graph=True)
            grads[name] = gr
        pre = "ops/%s/" % tag
        out[pre + "ori"], out[pre + "nrm"], out[pre + "adv"] = t2n(ori), t2n(nrm), t2n(adv)
        out[pre + "k"] = np.int64(k)
        for name, val in [("kappa_ori", kap_ori), ("cd", cd), ("pcd", pcd), ("hd", hd), ("l2", l2),
                          ("kappa_adv", kap_adv), ("normal_adv", nrm_adv), ("curv", curv)]:
            out[pre + name] = t2n(val)
        for name, gr in grads.items():
            out[pre + "g_" + name] = t2n(gr)
        cases.append(tag)
    out["ops/cases"] = np.array(cases)

    # ---------------------------------------------------------------- (2) PointNet
    sd = O.make_pointnet_state_dict(40, seed=0)
    net = RefPointNet(40)
    net.load_state_dict(sd)
    net.eval()
    out["pn/sd_checksum"] = np.float64(sum(float(v.double().abs().sum()) for v in sd.values()))
    for tag, b, N, seed in [("n64", 4, 64, 11), ("n256", 3, 256, 12), ("n1024", 2, 1024, 13)]:
        pc, _ = O.make_synthetic_clouds(b, N, seed)
        pc = pc.clone().requires_grad_()
        logits = net(pc)
        w = torch.randn(b, 40, generator=torch.Generator().manual_seed(seed))
        (gpc,) = torch.autograd.grad((logits * w).sum(), pc)
        with torch.no_grad():
            single = torch.cat([net(pc[i:i + 1]) for i in range(b)])
        pre = "pn/%s/" % tag
        out[pre + "pc"], out[pre + "logits"], out[pre + "w"], out[pre + "g_pc"] = map(t2n, (pc, logits, w, gpc))
        out[pre + "logits_batch1"] = t2n(single)
    out["pn/cases"] = np.array(["n64", "n256", "n1024"])

    # ---------------------------------------------------------------- (3) _forward_step
    fs_cases = []
    for tag, kw, targeted in [("ce_untarget", dict(cls_loss_type="CE", attack_label="Untarget"), False),
                              ("ce_target", dict(cls_loss_type="CE", attack_label="All"), True),
                              ("margin_target", dict(cls_loss_type="Margin", attack_label="All", confidence=0.5), True),
                              ("margin_untarget", dict(cls_loss_type="Margin", attack_label="Untarget"), False),
                              ("l2_nohd", dict(dis_loss_type="L2", hd_loss_weight=0.0, curv_loss_weight=0.0), False),
                              ("pcd", dict(is_cd_single_side=True), False)]:
        cfg = ref_cfg(curv_loss_knn=8, **kw)
        b, N = 4, 128
        ori, nrm = O.make_synthetic_clouds(b, N, 21)
        g = torch.Generator().manual_seed(22)
        x = (ori + torch.randn(b, 3, N, generator=g) * 0.01).requires_grad_()
        with torch.no_grad():
            gt = net(ori).argmax(1)
        target = (gt + 3) % 40 if targeted else gt
        kap = RL._get_kappa_ori(ori, nrm, cfg.curv_loss_knn) if cfg.curv_loss_weight != 0 else None
        sc = torch.tensor([10.0, 5.0, 20.0, 2.5])
        r = RA._forward_step(net, ori, x, nrm, kap, target, sc, cfg, targeted)
        logits, normal_curr, loss, loss_n, cls_loss, dis, hd, curv, constrain, info = r
        (gx,) = torch.autograd.grad(loss, x)
        pre = "fs/%s/" % tag
        out[pre + "ori"], out[pre + "nrm"], out[pre + "x"] = t2n(ori), t2n(nrm), t2n(x)
        out[pre + "gt"], out[pre + "target"], out[pre + "scale_const"] = t2n(gt), t2n(target), t2n(sc)
        for name, val in [("logits", logits), ("loss", loss), ("loss_n", loss_n), ("cls_loss", cls_loss),
                          ("dis_loss", dis), ("hd_loss", hd), ("curv_loss", curv), ("constrain", constrain),
                          ("g_x", gx)]:
            out[pre + name] = t2n(val).astype(np.float32) if not torch.is_tensor(val) else t2n(val)
        fs_cases.append(tag)
    out["fs/cases"] = np.array(fs_cases)

    # ---------------------------------------------------------------- (4)+(5) attack()
    def run_attack(tag, b, N, cfg, seed, targeted, trace_grads=True):
        ori, nrm = O.make_synthetic_clouds(b, N, seed)
        with torch.no_grad():
            gt = net(ori).argmax(1)
        tgt = (gt + 5) % 40
        g = torch.Generator().manual_seed(seed + 1000)
        inits = [torch.randn(b, 3, N, generator=g) * 1e-3 for _ in range(cfg.binary_max_steps)]
        it = iter(inits)

        def fake_normal_(t, mean=0.0, std=1.0):
            with torch.no_grad():
                t.copy_(next(it))
            return t

        real_normal_ = nn.init.normal_
        nn.init.normal_ = fake_normal_
        tr = dict(x=[], loss_n=[], constrain=[], logits=[])
        real_fs = RA._forward_step

        def fs_spy(net_, pc_ori, x, *a, **k):
            r = real_fs(net_, pc_ori, x, *a, **k)
            tr["x"].append(x.detach().clone())
            tr["loss_n"].append(r[3].detach().clone())
            tr["constrain"].append(r[8].detach().clone() if torch.is_tensor(r[8]) else torch.zeros(b))
            tr["logits"].append(r[0].detach().clone())
            return r

        RA._forward_step = fs_spy
        # the DataLoader layout: pc [bs,l,N,3], normal [bs,l,N,3], gt [bs,l], target [bs,l]
        data = [ori.permute(0, 2, 1).unsqueeze(1).contiguous(), nrm.permute(0, 2, 1).unsqueeze(1).contiguous(),
                gt.view(b, 1)]
        if targeted:
            data.append(tgt.view(b, 1))
        so = sys.stdout
        sys.stdout = io.StringIO()
        try:
            best, target, succ, best_step, all_loss = RA.attack(net, data, cfg, 0, 1, None)
        finally:
            sys.stdout = so
            nn.init.normal_ = real_normal_
            RA._forward_step = real_fs
        pre = "atk/%s/" % tag
        out[pre + "ori"], out[pre + "nrm"], out[pre + "gt"], out[pre + "tgt"] = map(t2n, (ori, nrm, gt, tgt))
        out[pre + "inits"] = np.stack([t2n(t) for t in inits])
        out[pre + "best_attack"], out[pre + "target"] = t2n(best), t2n(target)
        out[pre + "success"] = np.asarray(succ)
        out[pre + "best_step"] = np.asarray(best_step, dtype=np.int64)
        out[pre + "all_loss"] = np.asarray(all_loss, dtype=np.float32)
        out[pre + "tr_x"] = np.stack([t2n(t) for t in tr["x"]])
        out[pre + "tr_loss_n"] = np.stack([t2n(t) for t in tr["loss_n"]])
        out[pre + "tr_constrain"] = np.stack([t2n(t) for t in tr["constrain"]])
        out[pre + "tr_logits"] = np.stack([t2n(t) for t in tr["logits"]])
        out[pre + "cfg"] = np.array(repr(sorted(vars(cfg).items())))
        return tag

    atk = []
    for tag, (kw, targeted, bsz, seed) in ATK_CASES.items():
        atk.append(run_attack(tag, bsz, 64, ref_cfg(**kw), seed, targeted))
    out["atk/cases"] = np.array(atk)

    # ---------------------------------------------------------------- Adam pin (torch.optim.Adam)
    g = torch.Generator().manual_seed(77)
    p = (torch.randn(2, 3, 16, generator=g) * 1e-3).requires_grad_()
    opt = torch.optim.Adam([p], lr=0.01)
    grads, ps = [], [t2n(p).copy()]
    for t in range(6):
        gr = torch.randn(2, 3, 16, generator=g) * (10.0 ** (-t))
        p.grad = gr.clone()
        opt.step()
        grads.append(t2n(gr))
        ps.append(t2n(p).copy())
    out["adam/grads"], out["adam/params"] = np.stack(grads), np.stack(ps)

    # ---------------------------------------------------------------- PointNet++ SSG (config 4)
    from oracle import pointnet2_oracle as P2
    from PointNetPP_ssg import PointNet2ClassificationSSG as RefSSG      # reference Model/PointNetPP_ssg.py
    sd2 = P2.make_pn2_state_dict(0)
    net2 = RefSSG(use_xyz=True, use_normal=False)
    net2.load_state_dict(sd2)
    net2.eval()
    out["pn2/sd_checksum"] = np.float64(sum(float(v.double().abs().sum()) for v in sd2.values()))
    for tag, b, N, seed in [("n1024", 2, 1024, 41), ("n700", 1, 700, 42)]:
        pc, _ = O.make_synthetic_clouds(b, N, seed)
        pc = pc.clone()
        pc[:, :, 5] = 0.001            # a point FPS must skip (|p|^2 <= 1e-3)
        pc[:, :, 9] = pc[:, :, 8]      # a duplicate point (FPS / ball-query ties)
        pc.requires_grad_()
        logits = net2(pc)
        wgt = torch.randn(b, 40, generator=torch.Generator().manual_seed(seed))
        (gpc,) = torch.autograd.grad((logits * wgt).sum(), pc)
        pre = "pn2/%s/" % tag
        out[pre + "pc"], out[pre + "logits"], out[pre + "w"], out[pre + "g_pc"] = map(t2n, (pc, logits, wgt, gpc))
    out["pn2/cases"] = np.array(["n1024", "n700"])

    # ---------------------------------------------------------------- CLI flags + dataset expansion
    import json
    import re
    import tempfile
    src = open(os.path.join(REF, "main_attack.py")).read()
    flags = []
    for m in re.finditer(r"parser\.add_argument\((.*?)\)\s*$", src, re.M):
        body = m.group(1)
        names = re.findall(r"'(-{1,2}[A-Za-z_0-9]+)'", body.split("default")[0] if "default" in body else body)
        d = re.search(r"default=([^,\)]+)", body)
        flags.append([names, d.group(1).strip() if d else None, "store_true" in body])
    out["cli/flags_json"] = np.array(json.dumps(flags))
    from Provider.modelnet10_instance250 import ModelNet40 as RefDataset
    from geoa3_amd.data import TEN_LABEL_INDEXES, write_synthetic_mat
    with tempfile.TemporaryDirectory() as td:
        mat = write_synthetic_mat(os.path.join(td, "d.mat"), [TEN_LABEL_INDEXES[i // 25] for i in range(250)], 32, 9)
        for lab in ("All", "Untarget", "chair"):
            ds = RefDataset(data_mat_file=mat, attack_label=lab)
            out["ds/%s/len" % lab] = np.int64(len(ds))
            out["ds/%s/start" % lab] = np.int64(ds.start_index)
            for idx in (0, 7):
                item = ds[idx]
                for j, t in enumerate(item):
                    out["ds/%s/%d/%d" % (lab, idx, j)] = t2n(t)

    path = os.path.join(HERE, "geoa3_golden.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), len(out), "arrays")


if __name__ == "__main__":
    main()
