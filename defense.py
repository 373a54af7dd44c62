#!/usr/bin/env python3
"""Point-removal defences on the adversarial clouds main_attack.py wrote -- the reference's defense.py (flags,
console lines, defense_result.txt and the optional Defensed/*.obj records), with the statistics, the selection and
the victim's forward running in libgeoa3_hip.so (geoa3_amd/utility.py, geom_aux.hip) on ALL clouds of the
directory at once instead of one cloud per iteration (defense.py:85-99).

    python defense.py --datadir Exps/.../Mat --npoint 1024 --arch PointNet --defense_type outliers_fixNum --drop_num 128
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import torch


def build_parser():
    p = argparse.ArgumentParser(description="Point Cloud Defense")
    p.add_argument("--datadir", default="Data/modelnet40_1024_processed", type=str, metavar="DIR")
    p.add_argument("--npoint", default=1024, type=int)
    p.add_argument("-c", "--classes", default=40, type=int, metavar="N")
    p.add_argument("--arch", default="PointNet", type=str, metavar="ARCH")
    p.add_argument("--defense_type", default="outliers_fixNum", type=str,
                   help="[rand_drop, outliers_variance, outliers_fixNum]")
    p.add_argument("--outlier_knn", type=int, default=2)
    p.add_argument("--alpha", type=float, default=1.1)
    p.add_argument("--drop_num", type=int, default=128)
    p.add_argument("--is_record_all", action="store_true", default=False)
    p.add_argument("--is_record_wrong", action="store_true", default=False)
    p.add_argument("-j", "--num_workers", default=8, type=int, metavar="N")
    p.add_argument("--random_seed", default=0, type=int)
    p.add_argument("--print_freq", default=50, type=int)
    # ------------extensions (not in the reference)
    p.add_argument("--synthetic", action="store_true", default=False,
                   help="use the seeded calibrated random-init victim when Pretrained/... is missing")
    return p


def _record(path, cloud):
    with open(path, "w") as f:
        for m in range(cloud.shape[0]):
            f.write("v %f %f %f 0 0 0\n" % (cloud[m, 0], cloud[m, 1], cloud[m, 2]))


def main(cfg):
    from geoa3_amd import utility as U
    from geoa3_amd.data import AdvModelNet40, synthetic_state_dict
    from geoa3_amd.pointnet import PointNet

    assert cfg.datadir[-1] != "/"
    out_root = os.path.split(cfg.datadir)[0]
    if cfg.is_record_all or cfg.is_record_wrong:
        os.makedirs(os.path.join(out_root, "Defensed"), exist_ok=True)
    seed = cfg.random_seed if cfg.random_seed == 0 else int(time.time())      # defense.py:58-65
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    device = torch.device("cuda", 0)

    dataset = AdvModelNet40(cfg.datadir)
    model_path = os.path.join("Pretrained", cfg.arch, str(cfg.npoint), "model_best.pth.tar")
    if cfg.arch == "PointNet":
        net = PointNet(cfg.classes, npoint=cfg.npoint)
    elif cfg.arch == "PointNetPP":
        from geoa3_amd.pointnet2 import PointNet2ClassificationSSG
        net = PointNet2ClassificationSSG(use_xyz=True, use_normal=False)
    else:
        raise AssertionError("Not support such arch.")
    if os.path.isfile(model_path):
        net.load_state_dict(torch.load(model_path, map_location="cpu")["state_dict"])
        print("\nSuccessfully load pretrained-model from {}\n".format(model_path))
    elif cfg.synthetic and cfg.arch == "PointNet":
        net.load_state_dict(synthetic_state_dict(cfg.classes, seed=0, device=device))
    else:
        raise FileNotFoundError(model_path)
    net = net.to(device).eval()

    items = [dataset[i] for i in range(len(dataset))]
    total = len(items)
    gts = [int(np.asarray(it[1]).reshape(-1)[0]) for it in items]
    atks = [int(np.asarray(it[2]).reshape(-1)[0]) for it in items]
    # clouds of equal size go through the kernels together (the reference loops with batch size 1)
    groups = {}
    for i, it in enumerate(items):
        groups.setdefault(int(it[0].shape[1]), []).append(i)
    defended, dropped = [None] * total, [0] * total
    for n, members in groups.items():
        pc = torch.stack([items[i][0] for i in members]).to(device).contiguous()         # [g,3,n]
        if n > cfg.npoint:                                                                 # defense.py:94-96
            pc = U.farthest_points_sample(pc, cfg.npoint)
            n = cfg.npoint
        if cfg.defense_type == "rand_drop":
            for j, i in enumerate(members):
                defended[i], dropped[i] = U.random_drop_fn(pc[j:j + 1], cfg.drop_num)
        elif cfg.defense_type in ("outliers_variance", "outliers_fixNum"):
            idx, cnt = U.outlier_removal_indices(pc, cfg.defense_type, cfg.drop_num, cfg.alpha, cfg.outlier_knn)
            cnt = cnt.cpu().tolist()
            for j, i in enumerate(members):
                defended[i] = pc[j:j + 1, :, idx[j, :cnt[j]].long()].contiguous()
                dropped[i] = n - cnt[j]
        else:
            raise AssertionError("Wrong defense type!")
    # the victim on the defended clouds, again grouped by size
    preds = [0] * total
    by_size = {}
    for i, d in enumerate(defended):
        by_size.setdefault(int(d.shape[2]), []).append(i)
    with torch.no_grad():
        for n, members in by_size.items():
            out = net(torch.cat([defended[i] for i in members]).contiguous()).argmax(1).cpu().tolist()
            for i, p in zip(members, out):
                preds[i] = p

    num_defense_success = num_attack_still_success = 0
    num_drop_point = 0
    for i in range(total):                                                                # defense.py:103-128
        if gts[i] == atks[i]:
            defense_success, attack_still_success = 1, 0
        else:
            defense_success, attack_still_success = int(preds[i] == gts[i]), int(preds[i] == atks[i])
        num_defense_success += defense_success
        num_attack_still_success += attack_still_success
        num_drop_point += dropped[i]
        if cfg.is_record_all or (cfg.is_record_wrong and gts[i] != preds[i]):
            name = "Gt%d_record_%d_attack%d_defensedGT%d.obj" % (gts[i], i, atks[i], preds[i])
            _record(os.path.join(out_root, "Defensed", name), defended[i][0].t().cpu().numpy())
        if (i + 1) % cfg.print_freq == 0:
            cnt = i + 1
            print("[{0}/{1}]  attack success: {2:.2f} still attack success: {3:.2f} avg drop num: {4:.2f}".format(
                i + 1, total, (1 - num_defense_success / float(cnt)) * 100,
                num_attack_still_success / float(cnt) * 100, num_drop_point / float(cnt)))

    final_acc = num_defense_success / float(total) * 100
    final_attack_acc = num_attack_still_success / float(total) * 100
    avg_drop_point = num_drop_point / float(total)
    assert 100 - final_acc >= final_attack_acc, "Attack success must > or >= attack still success!"
    print("\nfinal attack success: {0:.2f}\n still attack success: {1:.2f}\n avg drop point: {2:.2f}".format(
        100 - final_acc, final_attack_acc, avg_drop_point))
    with open(os.path.join(out_root, "defense_result.txt"), "at") as f:                   # defense.py:139-148
        if cfg.defense_type == "rand_drop":
            f.write("[{0:.2f}%, {1:.2f}%, {2:.2f}n] random drop: drop_num {3}\n".format(
                final_acc, final_attack_acc, avg_drop_point, cfg.drop_num))
        elif cfg.defense_type == "outliers_variance":
            f.write("[{0:.2f}%, {1:.2f}%, {2:.2f}n] outlier alpha removal: k{3}, alpha{4}\n".format(
                final_acc, final_attack_acc, avg_drop_point, cfg.outlier_knn, cfg.alpha))
        else:
            f.write("[{0:.2f}%, {1:.2f}%, {2:.2f}n] outlier ramdom drop: drop_num {3}\n".format(
                final_acc, final_attack_acc, avg_drop_point, cfg.drop_num))
    print("\n Finished!")
    return final_acc, final_attack_acc, avg_drop_point


if __name__ == "__main__":
    cfg = build_parser().parse_args()
    print(cfg)
    main(cfg)
