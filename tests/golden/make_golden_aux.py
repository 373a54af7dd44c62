#!/usr/bin/env python3
"""Generate tests/golden/geoa3_golden_aux.npz from the REFERENCE's own Python (build container only): the
dense-cloud / defence / measurement helpers of SURVEY.md 8f-3 and 8f-4.

Run:  python tests/golden/make_golden_aux.py        (needs /root/reference; CPU only)

Same shims as make_golden.py, plus: torch.symeig (removed from torch 2) answered by torch.linalg.eigh, and the
reference's random draws (torch.randint / torch.randn / torch.randperm / nn.init.normal_) replaced by recorded
sequences so that the draws can be stored next to the outputs.  Only inputs and outputs are stored.
"""
from __future__ import annotations

import io
import os
import runpy
import sys
import tempfile

sys.dont_write_bytecode = True
os.environ["PYTHONDONTWRITEBYTECODE"] = "1"

import numpy as np
import scipy.io as sio
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.abspath(os.path.join(HERE, "..", ".."))
sys.path.insert(0, REPO)

from oracle import geoa3_oracle as O  # noqa: E402
from tests.golden.make_golden import REF, install_shims, ref_cfg, t2n  # noqa: E402

# tag -> (cfg overrides, batch, dense N, seed); shared with tests/test_oracle_aux.py
AUX_ATK_CASES = {
    "subsample_vote": (dict(is_subsample_opt=True, npoint=64, eval_num=3, curv_loss_knn=4, binary_max_steps=2,
                            iter_max_steps=5, lr=0.002), 3, 160, 61),
    "subsample_vote1": (dict(is_subsample_opt=True, npoint=48, eval_num=1, curv_loss_knn=4, binary_max_steps=2,
                             iter_max_steps=4, lr=0.002, hd_loss_weight=0.0), 2, 96, 62),
    "partial_var": (dict(is_partial_var=True, knn_range=3, curv_loss_knn=4, binary_max_steps=2, iter_max_steps=53,
                         lr=0.01, npoint=64), 3, 64, 64),
    "partial_var_sgd": (dict(is_partial_var=True, knn_range=5, optim="sgd", lr=0.05, is_use_lr_scheduler=True,
                             curv_loss_knn=4, binary_max_steps=1, iter_max_steps=52, npoint=64), 2, 64, 65),
    "pre_jitter": (dict(is_pre_jitter_input=True, calculate_project_jitter_noise_iter=2, jitter_k=8,
                        jitter_sigma=0.01, jitter_clip=0.05, curv_loss_knn=4, binary_max_steps=1,
                        iter_max_steps=5, lr=0.002, npoint=64), 3, 64, 63),
}


class Draws:
    """Replaces torch.randint / torch.randn / torch.randperm by seeded draws that are recorded in call order."""

    def __init__(self, seed):
        self.g = torch.Generator().manual_seed(seed)
        self.randint_calls, self.randn_calls, self.randperm_calls = [], [], []
        self._real = (torch.randint, torch.randn, torch.randperm)

    def __enter__(self):
        real_randint, real_randn, real_randperm = self._real

        def randint(*a, **k):
            k.pop("generator", None)
            r = real_randint(*a, generator=self.g, **k)
            self.randint_calls.append(r.clone())
            return r

        def randn(*a, **k):
            k.pop("generator", None)
            r = real_randn(*a, generator=self.g, **k)
            self.randn_calls.append(r.clone())
            return r

        def randperm(*a, **k):
            k.pop("generator", None)
            r = real_randperm(*a, generator=self.g, **k)
            self.randperm_calls.append(r.clone())
            return r

        torch.randint, torch.randn, torch.randperm = randint, randn, randperm
        return self

    def __exit__(self, *exc):
        torch.randint, torch.randn, torch.randperm = self._real


def main():
    if not os.path.isdir(REF):
        sys.exit("needs /root/reference (build container only)")
    install_shims()
    torch.symeig = lambda A, eigenvectors=False, upper=True: torch.linalg.eigh(A)
    import utility as RU                           # reference Lib/utility.py
    from PointNet import PointNet as RefPointNet   # reference Model/PointNet.py
    from Attacker import geoA3_attack as RA        # reference Attacker/geoA3_attack.py
    import defense as RD                           # reference defense.py (main() is guarded)

    out = {}

    # ---------------------------------------------------------------- farthest_points_sample
    cases = []
    for tag, b, n, m, seed in [("n300m64", 3, 300, 64, 1), ("n1100m128", 2, 1100, 128, 2), ("dup", 2, 96, 96, 3)]:
        pc, _ = O.make_synthetic_clouds(b, n, 200 + seed)
        pc = pc.clone()
        if tag == "dup":
            pc[:, :, 7] = pc[:, :, 3]
            pc[:, :, 50] = pc[:, :, 49]
        with Draws(seed) as d:
            pts = RU.farthest_points_sample(pc, m)
        pre = "fps/%s/" % tag
        out[pre + "pc"], out[pre + "m"], out[pre + "start"], out[pre + "pts"] = \
            t2n(pc), np.int64(m), t2n(d.randint_calls[0].view(-1)), t2n(pts)
        cases.append(tag)
    out["fps/cases"] = np.array(cases)

    # ---------------------------------------------------------------- estimate_normal_via_ori_normal (b = 1)
    cases = []
    for tag, n_adv, n_ori, k, seed in [("k3", 200, 500, 3, 4), ("k5", 128, 128, 5, 5)]:
        ori, nrm = O.make_synthetic_clouds(1, n_ori, 210 + seed)
        g = torch.Generator().manual_seed(seed)
        pick = torch.randperm(n_ori, generator=g)[:n_adv]
        adv = ori[:, :, pick].clone()
        moved = torch.rand(n_adv, generator=g) < 0.6
        adv[:, :, moved] += torch.randn(1, 3, int(moved.sum()), generator=g) * 0.02
        est = RU.estimate_normal_via_ori_normal(adv, ori, nrm, k)
        pre = "nvo/%s/" % tag
        out[pre + "adv"], out[pre + "ori"], out[pre + "nrm"], out[pre + "k"], out[pre + "est"] = \
            t2n(adv), t2n(ori), t2n(nrm), np.int64(k), t2n(est)
        cases.append(tag)
    out["nvo/cases"] = np.array(cases)

    # ---------------------------------------------------------------- estimate_perpendicular
    cases = []
    for tag, b, n, k, seed in [("k16", 2, 200, 16, 6), ("k8", 3, 96, 8, 7)]:
        pc, _ = O.make_synthetic_clouds(b, n, 220 + seed)
        with Draws(seed) as d:
            noise = RU.estimate_perpendicular(pc, k, sigma=0.01, clip=0.012)
        pre = "perp/%s/" % tag
        out[pre + "pc"], out[pre + "k"], out[pre + "noise"] = t2n(pc), np.int64(k), t2n(noise)
        out[pre + "aux1"], out[pre + "aux2"] = t2n(d.randn_calls[0] * 0.01), t2n(d.randn_calls[1] * 0.01)
        out[pre + "clip"] = np.float32(0.012)
        cases.append(tag)
    out["perp/cases"] = np.array(cases)

    # ---------------------------------------------------------------- defense.py point removal
    cases = []
    for tag, n, dtype, drop, alpha, knn, seed in [("fix_k2", 256, "outliers_fixNum", 32, 1.1, 2, 8),
                                                  ("var_k2", 256, "outliers_variance", 0, 1.1, 2, 9),
                                                  ("var_k5", 300, "outliers_variance", 0, 0.5, 5, 10),
                                                  ("fix_k4", 200, "outliers_fixNum", 7, 1.1, 4, 11)]:
        pc, _ = O.make_synthetic_clouds(1, n, 230 + seed)
        g = torch.Generator().manual_seed(seed)
        far = torch.randperm(n, generator=g)[:12]
        pc = pc.clone()
        pc[:, :, far] += torch.randn(1, 3, 12, generator=g) * 0.08     # a few outliers
        kept, num = RD.point_removal_fn(pc, dtype, drop, alpha, knn)
        pre = "def/%s/" % tag
        out[pre + "pc"], out[pre + "kept"], out[pre + "num"] = t2n(pc), t2n(kept), np.int64(num)
        out[pre + "type"], out[pre + "drop"], out[pre + "alpha"], out[pre + "knn"] = \
            np.array(dtype), np.int64(drop), np.float64(alpha), np.int64(knn)
        cases.append(tag)
    out["def/cases"] = np.array(cases)
    pc, _ = O.make_synthetic_clouds(1, 128, 241)
    with Draws(12) as d:
        kept, num = RD.point_removal_fn(pc, "rand_drop", 20, 1.1, 2)
    out["def/rand/pc"], out["def/rand/kept"], out["def/rand/perm"], out["def/rand/drop"] = \
        t2n(pc), t2n(kept), t2n(d.randperm_calls[0]), np.int64(20)

    # ---------------------------------------------------------------- Measurement/compute_data_smoothness.py
    cases = []
    for tag, n, k, k2, seed in [("k16", 128, 16, 16, 13), ("k8k12", 160, 8, 12, 14)]:
        pc, _ = O.make_synthetic_clouds(1, n, 250 + seed)
        pc = pc[0] + torch.randn(3, n, generator=torch.Generator().manual_seed(seed)) * 0.01
        with tempfile.TemporaryDirectory() as td:
            os.mkdir(os.path.join(td, "Mat"))
            sio.savemat(os.path.join(td, "Mat", "adv_0.mat"), {"adversary_point_clouds": t2n(pc)})
            argv, so = sys.argv, sys.stdout
            sys.argv = ["compute_data_smoothness.py", "--datadir", td, "--k", str(k), "--k2", str(k2)]
            sys.stdout = io.StringIO()
            try:
                runpy.run_path(os.path.join(REF, "Measurement", "compute_data_smoothness.py"), run_name="__main__")
            finally:
                sys.argv, sys.stdout = argv, so
            val = sio.loadmat(os.path.join(td, "metric", "k%d.mat" % k))["smoothness"].reshape(-1)
            result_txt = open(os.path.join(td, "metric", "result.txt")).read()
        pre = "smooth/%s/" % tag
        out[pre + "pc"], out[pre + "k"], out[pre + "k2"], out[pre + "value"] = t2n(pc), np.int64(k), np.int64(k2), val
        out[pre + "result_txt"] = np.array(result_txt)
        cases.append(tag)
    out["smooth/cases"] = np.array(cases)

    # ---------------------------------------------------------------- attack(): dense-cloud and pre-jitter paths
    sd = O.make_pointnet_state_dict(40, seed=0)
    net = RefPointNet(40)
    net.load_state_dict(sd)
    net.eval()

    def run_attack(tag, kw, b, n, seed):
        cfg = ref_cfg(**kw)
        ori, nrm = O.make_synthetic_clouds(b, n, seed)
        with torch.no_grad():
            gt = net(ori[:, :, :cfg.npoint]).argmax(1)
        g = torch.Generator().manual_seed(seed + 1000)
        inits = [torch.randn(b, 3, n, generator=g) * 1e-3 for _ in range(cfg.binary_max_steps)]
        it = iter(inits)
        part_inits, part_points = [], []

        def fake_normal_(t, mean=0.0, std=1.0):
            with torch.no_grad():
                if cfg.is_partial_var:      # a fresh [b,3,knn_range] draw every 50 steps (geoA3_attack.py:246-247)
                    d = torch.randn(t.shape, generator=g) * std
                    part_inits.append(d.clone())
                    t.copy_(d)
                else:
                    t.copy_(next(it))
            return t

        real_np_randint = np.random.randint

        def fake_np_randint(*a, **k):
            v = int(torch.randint(0, a[0], (1,), generator=g).item())
            part_points.append(v)
            return v

        real_normal_, real_fs, real_perp = nn.init.normal_, RA._forward_step, RA.estimate_perpendicular
        nn.init.normal_ = fake_normal_
        tr = dict(x=[], loss_n=[], noise=[])

        def fs_spy(net_, pc_ori, x, *a, **k):
            r = real_fs(net_, pc_ori, x, *a, **k)
            tr["x"].append(x.detach().clone())
            tr["loss_n"].append(r[3].detach().clone())
            return r

        def perp_spy(*a, **k):
            r = real_perp(*a, **k)
            tr["noise"].append(r.detach().clone())
            return r

        RA._forward_step, RA.estimate_perpendicular = fs_spy, perp_spy
        np.random.randint = fake_np_randint
        data = [ori.permute(0, 2, 1).unsqueeze(1).contiguous(), nrm.permute(0, 2, 1).unsqueeze(1).contiguous(),
                gt.view(b, 1)]
        so = sys.stdout
        sys.stdout = io.StringIO()
        try:
            with Draws(seed) as d:
                best, target, succ, best_step, all_loss = RA.attack(net, data, cfg, 0, 1, None)
        finally:
            sys.stdout = so
            nn.init.normal_, RA._forward_step, RA.estimate_perpendicular = real_normal_, real_fs, real_perp
            np.random.randint = real_np_randint
        pre = "atk/%s/" % tag
        out[pre + "ori"], out[pre + "nrm"], out[pre + "gt"] = t2n(ori), t2n(nrm), t2n(gt)
        out[pre + "inits"] = np.stack([t2n(t) for t in inits])
        out[pre + "best_attack"], out[pre + "success"] = t2n(best), np.asarray(succ)
        out[pre + "best_step"] = np.asarray(best_step, dtype=np.int64)
        out[pre + "all_loss"] = np.asarray(all_loss, dtype=np.float32)
        out[pre + "tr_x"] = np.stack([t2n(t) for t in tr["x"]])
        out[pre + "tr_loss_n"] = np.stack([t2n(t) for t in tr["loss_n"]])
        if cfg.is_subsample_opt:   # per step: one [b,1] draw (the objective's sample), then b draws [eval_num,1]
            calls = d.randint_calls
            per = 1 + b
            assert len(calls) % per == 0
            out[pre + "sub_starts"] = np.stack([t2n(calls[i].view(-1)) for i in range(0, len(calls), per)])
            out[pre + "vote_starts"] = np.stack([np.stack([t2n(calls[i + 1 + k].view(-1)) for k in range(b)])
                                                 for i in range(0, len(calls), per)])
        if cfg.is_partial_var:
            out[pre + "part_inits"] = np.stack([t2n(t) for t in part_inits])
            out[pre + "part_points"] = np.asarray(part_points, dtype=np.int64)
        if cfg.is_pre_jitter_input:
            out[pre + "noise"] = np.stack([t2n(t) for t in tr["noise"]])
            out[pre + "aux"] = np.stack([t2n(t) * cfg.jitter_sigma for t in d.randn_calls])
        return tag

    atk = [run_attack(tag, kw, b, n, seed) for tag, (kw, b, n, seed) in AUX_ATK_CASES.items()]
    out["atk/cases"] = np.array(atk)

    # ---------------------------------------------------------------- flag sets of the two command lines
    import json
    import re
    for key, rel in [("cli/defense_flags_json", "defense.py"),
                     ("cli/smooth_flags_json", os.path.join("Measurement", "compute_data_smoothness.py"))]:
        src = open(os.path.join(REF, rel)).read()
        flags = []
        for m in re.finditer(r"parser\.add_argument\((.*?)\)\s*$", src, re.M):
            body = m.group(1)
            names = re.findall(r"'(-{1,2}[A-Za-z_0-9]+)'", body.split("default")[0] if "default" in body else body)
            d = re.search(r"default=([^,\)]+)", body)
            flags.append([names, d.group(1).strip() if d else None, "store_true" in body])
        out[key] = np.array(json.dumps(flags))

    path = os.path.join(HERE, "geoa3_golden_aux.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "%.1f KB" % (os.path.getsize(path) / 1024), len(out), "arrays")


if __name__ == "__main__":
    main()
