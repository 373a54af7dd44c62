#!/usr/bin/env python3
"""Command line of the accelerated GeoA3 attack: the flag set, output-directory naming, seeding, per-batch loop
and `.mat` / `.obj` / attack_result.txt outputs of the reference's main_attack.py (main_attack.py:26-314 for the
harness, :317-384 for the 44 flags), driving geoa3_amd.attack instead of Attacker/geoA3_attack.

    python main_attack.py --attack GeoA3 --attack_label Untarget -b 250 [...]

Extensions (not in the reference): `--synthetic` writes a seeded synthetic `.mat` / uses calibrated random-init
weights when the data file / checkpoint do not exist (neither ships with the reference repository);
under `torchrun` (WORLD_SIZE > 1) every batch is sharded by instance over the ranks and rank 0 writes the files.
"""
from __future__ import annotations

import argparse
import os
import time

import numpy as np
import scipy.io as sio
import torch


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="Point Cloud Attacking")
    # ------------Model-----------------------
    p.add_argument("--id", type=int, default=0)
    p.add_argument("--arch", default="PointNet", type=str, metavar="ARCH")
    # ------------Dataset---------------------
    p.add_argument("--data_dir_file", default="Data/modelnet10_250instances1024_PointNet.mat", type=str)
    p.add_argument("--dense_data_dir_file", default=None, type=str)
    p.add_argument("-c", "--classes", default=40, type=int, metavar="N")
    p.add_argument("-b", "--batch_size", default=2, type=int, metavar="B")
    p.add_argument("--npoint", default=1024, type=int)
    # ------------Attack----------------------
    p.add_argument("--attack", default=None, type=str, help="GeoA3 | GeoA3_mesh")
    p.add_argument("--attack_label", default="All", type=str, help="[All; ...; Untarget]")
    p.add_argument("--binary_max_steps", type=int, default=10)
    p.add_argument("--initial_const", type=float, default=10)
    p.add_argument("--iter_max_steps", default=500, type=int, metavar="M")
    p.add_argument("--optim", default="adam", type=str, help="adam| sgd")
    p.add_argument("--lr", type=float, default=0.01)
    p.add_argument("--eval_num", type=int, default=1)
    p.add_argument("--cls_loss_type", default="CE", type=str, help="Margin | CE")
    p.add_argument("--confidence", type=float, default=0)
    p.add_argument("--dis_loss_type", default="CD", type=str, help="CD | L2 | None")
    p.add_argument("--dis_loss_weight", type=float, default=1.0)
    p.add_argument("--is_cd_single_side", action="store_true", default=False)
    p.add_argument("--hd_loss_weight", type=float, default=0.1)
    p.add_argument("--curv_loss_weight", type=float, default=1.0)
    p.add_argument("--curv_loss_knn", type=int, default=16)
    p.add_argument("--uniform_loss_weight", type=float, default=0.0)
    p.add_argument("--knn_smoothing_loss_weight", type=float, default=5.0)
    p.add_argument("--knn_smoothing_k", type=int, default=5)
    p.add_argument("--knn_threshold_coef", type=float, default=1.10)
    p.add_argument("--laplacian_loss_weight", type=float, default=0)
    p.add_argument("--edge_loss_weight", type=float, default=0)
    p.add_argument("--is_partial_var", dest="is_partial_var", action="store_true", default=False)
    p.add_argument("--knn_range", type=int, default=3)
    p.add_argument("--is_subsample_opt", dest="is_subsample_opt", action="store_true", default=False)
    p.add_argument("--is_use_lr_scheduler", dest="is_use_lr_scheduler", action="store_true", default=False)
    p.add_argument("--cc_linf", type=float, default=0.0)
    p.add_argument("--is_real_offset", action="store_true", default=False)
    p.add_argument("--is_pro_grad", action="store_true", default=False)
    p.add_argument("--is_pre_jitter_input", action="store_true", default=False)
    p.add_argument("--is_previous_jitter_input", action="store_true", default=False)
    p.add_argument("--calculate_project_jitter_noise_iter", default=50, type=int)
    p.add_argument("--jitter_k", type=int, default=16)
    p.add_argument("--jitter_sigma", type=float, default=0.01)
    p.add_argument("--jitter_clip", type=float, default=0.05)
    p.add_argument("--step_alpha", type=float, default=5)
    # ------------Recording settings----------
    p.add_argument("--is_record_converged_steps", action="store_true", default=False)
    p.add_argument("--is_record_loss", action="store_true", default=False)
    # ------------OS--------------------------
    p.add_argument("-j", "--num_workers", default=8, type=int, metavar="N")
    p.add_argument("--is_save_normal", action="store_true", default=False)
    p.add_argument("--is_debug", action="store_true", default=False)
    p.add_argument("--is_low_memory", action="store_true", default=False)
    # ------------extensions (not in the reference)
    p.add_argument("--synthetic", action="store_true", default=False,
                   help="fall back to seeded synthetic data / weights when the files are missing")
    p.add_argument("--out_root", default="Exps", type=str, help="root of the output tree (reference: Exps)")
    p.add_argument("--quiet", action="store_true", default=False)
    p.add_argument("--synthetic_npoint", type=int, default=0,
                   help="points per synthetic cloud when --synthetic writes the data file (default: --npoint)")
    p.add_argument("--synthetic_kind", default="ellipsoid", choices=["ellipsoid", "cad"],
                   help="what --synthetic writes: one ellipsoid per instance, or CAD-like clouds (boxes, tables on thin "
                        "legs, clusters of very different density, rods, exact duplicates)")
    return p


def saved_dir_name(cfg) -> str:
    """Output directory as a function of the flags (main_attack.py:36-82)."""
    root = os.path.join(cfg.out_root, cfg.arch + "_npoint" + str(cfg.npoint))
    if cfg.attack in ("GeoA3", "GeoA3_mesh"):
        parts = [str(cfg.attack), str(cfg.id), "BiStep" + str(cfg.binary_max_steps),
                 "IterStep" + str(cfg.iter_max_steps), "Opt" + cfg.optim, "Lr" + str(cfg.lr),
                 "Initcons" + str(cfg.initial_const), cfg.cls_loss_type,
                 str(cfg.dis_loss_type) + "Loss" + str(cfg.dis_loss_weight)]
        name = "_".join(parts)
        optional = [
            (cfg.hd_loss_weight != 0, "_HDLoss" + str(cfg.hd_loss_weight)),
            (cfg.curv_loss_weight != 0, "_CurLoss" + str(cfg.curv_loss_weight) + "_k" + str(cfg.curv_loss_knn)),
            (cfg.uniform_loss_weight != 0, "_UniLoss" + str(cfg.uniform_loss_weight)),
            (cfg.laplacian_loss_weight != 0, "_LapLoss" + str(cfg.laplacian_loss_weight)),
            (cfg.edge_loss_weight != 0, "_EdgeLoss" + str(cfg.edge_loss_weight)),
            (cfg.is_partial_var, "_PartOpt" + "_k" + str(cfg.knn_range)),
            (cfg.is_use_lr_scheduler, "_LRExp"),
            (cfg.is_pro_grad, "_ProGrad" + ("RO" if cfg.is_real_offset else "")),
            (cfg.cc_linf != 0, "_cclinf" + str(cfg.cc_linf)),
        ]
        for cond, suffix in optional:
            if cond:
                name += suffix
        if cfg.is_pre_jitter_input:
            name += "_PreJitter" + str(cfg.jitter_sigma) + "_" + str(cfg.jitter_clip)
            name += "_PreviousMethod" if cfg.is_previous_jitter_input else \
                "_estNormalVery" + str(cfg.calculate_project_jitter_noise_iter)
    else:
        assert cfg.attack is None
        name = "Evaluating_" + str(cfg.id)
    return os.path.join(root, cfg.attack_label, name)


def _write_outputs(saved_dir, name, cloud, gt, pred, est_normal=None):
    """One successful adversarial cloud: Mat/<name>.mat + PC/<name>.obj (main_attack.py:264-279)."""
    mat = {"adversary_point_clouds": cloud, "gt_label": gt, "attack_label": pred}
    if est_normal is not None:                                # --is_save_normal (main_attack.py:269-271)
        mat["est_normal"] = est_normal
    sio.savemat(os.path.join(saved_dir, "Mat", name + ".mat"), mat)
    with open(os.path.join(saved_dir, "PC", name + ".obj"), "w") as f:
        for m in range(cloud.shape[1]):
            f.write("v %f %f %f 0 0 0\n" % (cloud[0, m], cloud[1, m], cloud[2, m]))


class CountConvergeIter:
    """Count_converge_iter of the reference (Lib/utility.py:654-677): the best attack step of every instance, saved as
    Records/converge_iter.mat (+ a histogram when matplotlib is importable; the reference draws it with seaborn)."""

    def __init__(self, fsave):
        self.fsave = fsave
        os.makedirs(fsave, exist_ok=True)
        self.attack_step_list = []

    def record_converge_iter(self, attack_step_list):
        attack_step_list = list(attack_step_list)
        if -1 in attack_step_list:            # Lib/utility.py:662-663 drops the FIRST -1 only; kept as is
            attack_step_list.remove(-1)
        self.attack_step_list += attack_step_list

    def save_converge_iter(self):
        sio.savemat(os.path.join(self.fsave, "converge_iter.mat"), {"attack_step_list": self.attack_step_list})

    def plot_converge_iter_hist(self):
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:
            return
        if not self.attack_step_list:
            return
        fig = plt.figure()
        ax = fig.gca()
        ax.hist(self.attack_step_list, bins=np.histogram(np.hstack(self.attack_step_list), bins=50)[1])
        ax.set_xlabel("Converged iteration")
        ax.set_ylabel("Number of Samples")
        fig.savefig(os.path.join(self.fsave, "converge_iter.png"))
        plt.close(fig)


class CountLossIter:
    """Count_loss_iter of the reference (Lib/utility.py:680-714): the [steps, instances] loss history of the last
    binary step of every batch, saved as Records/loss_iter.mat (+ mean / std curve)."""

    def __init__(self, fsave):
        self.fsave = fsave
        os.makedirs(fsave, exist_ok=True)
        self.loss_numpy = None

    def record_loss_iter(self, loss_list):
        a = np.array(loss_list)
        self.loss_numpy = a if self.loss_numpy is None else np.concatenate((self.loss_numpy, a), axis=1)

    def save_loss_iter(self):
        sio.savemat(os.path.join(self.fsave, "loss_iter.mat"), {"loss": self.loss_numpy})

    def plot_loss_iter_hist(self):
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:
            return
        if self.loss_numpy is None:
            return
        x = np.arange(1, self.loss_numpy.shape[0] + 1)
        mean, std = self.loss_numpy.mean(1), self.loss_numpy.std(1)
        fig, ax = plt.subplots(1, 1)
        ax.plot(x, mean)
        ax.fill_between(x, mean - std, mean + std, alpha=0.2)
        ax.set_xlabel("Number of iteration")
        ax.set_ylabel("Magnitude of loss")
        fig.savefig(os.path.join(self.fsave, "loss_iter.png"))
        plt.close(fig)


def main(cfg):
    import torch.distributed as dist

    from geoa3_amd import attack as geoa3_attack
    from geoa3_amd.data import TEN_LABEL_INDEXES, ModelNet40, synthetic_state_dict, write_synthetic_mat
    from geoa3_amd.pointnet import PointNet
    from geoa3_amd.utility import estimate_normal_via_ori_normal, farthest_points_sample

    if cfg.attack == "GeoA3_mesh":
        raise AssertionError("Not uploaded yet.")          # as the reference (main_attack.py:27-28)
    if cfg.arch not in ("PointNet", "PointNetPP"):
        raise AssertionError("Not support such arch.")
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = max(torch.cuda.device_count(), 1)
    local_rank = local_rank % ndev                          # several ranks may share a GPU in a functional test
    if world > 1 and not dist.is_initialized():
        torch.cuda.set_device(local_rank)
        backend = os.environ.get("GEOA3_DIST_BACKEND", "nccl")   # "gloo": functional test of the sharded path on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    device = torch.device("cuda", local_rank if world > 1 else 0)
    targeted = cfg.attack_label != "Untarget"
    say = (lambda *a: None) if (cfg.quiet or rank != 0) else print

    say("=>Creating dir")
    saved_dir = saved_dir_name(cfg)
    if rank == 0:
        for sub in ("PC", "Mat", "Records"):
            os.makedirs(os.path.join(saved_dir, sub), exist_ok=True)
    say("==>Successfully created {}".format(saved_dir))

    seed = 0 if cfg.id == 0 else int(time.time())          # main_attack.py:98-104
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)

    if not os.path.isfile(cfg.data_dir_file) and cfg.synthetic:
        if rank == 0:
            labels = [TEN_LABEL_INDEXES[i // 25] for i in range(250)]
            write_synthetic_mat(cfg.data_dir_file, labels, cfg.synthetic_npoint or cfg.npoint, seed=0, kind=cfg.synthetic_kind)
        if world > 1:
            dist.barrier()
    dataset = ModelNet40(data_mat_file=cfg.data_dir_file, attack_label=cfg.attack_label, resample_num=-1)
    loader = torch.utils.data.DataLoader(dataset, batch_size=cfg.batch_size, shuffle=False, drop_last=False,
                                         num_workers=0, pin_memory=True)
    dense_iter = None
    if cfg.is_save_normal and cfg.dense_data_dir_file is not None:   # main_attack.py:124-130
        dense = ModelNet40(data_mat_file=cfg.dense_data_dir_file, attack_label=cfg.attack_label, resample_num=-1)
        dense_iter = iter(torch.utils.data.DataLoader(dense, batch_size=cfg.batch_size, shuffle=False,
                                                      drop_last=False, num_workers=0, pin_memory=True))
    elif cfg.is_save_normal:
        raise AssertionError("--is_save_normal needs --dense_data_dir_file (main_attack.py:124,246)")

    say("=>Loading model")
    model_path = os.path.join("Pretrained", cfg.arch, str(cfg.npoint), "model_best.pth.tar")
    if cfg.arch == "PointNet":
        net = PointNet(cfg.classes, npoint=cfg.npoint)
    else:   # main_attack.py:139-140: the SSG classifier, 40 classes
        from geoa3_amd.pointnet2 import PointNet2ClassificationSSG
        net = PointNet2ClassificationSSG(use_xyz=True, use_normal=False)
    if os.path.isfile(model_path):
        net.load_state_dict(torch.load(model_path, map_location="cpu")["state_dict"])
        say("==>Successfully load pretrained-model from {}".format(model_path))
    elif cfg.synthetic and cfg.arch == "PointNetPP":
        with torch.no_grad():   # default-initialised modules (seeded above) with non-trivial BatchNorm statistics
            for mod in net.modules():
                if isinstance(mod, (torch.nn.BatchNorm1d, torch.nn.BatchNorm2d)):
                    mod.running_mean.normal_(0, 0.1)
                    mod.running_var.uniform_(0.5, 1.5)
        say("==>No checkpoint at {}: seeded random-init weights (--synthetic)".format(model_path))
    elif cfg.synthetic:
        net.load_state_dict(synthetic_state_dict(cfg.classes, seed=0, device=device))
        say("==>No checkpoint at {}: calibrated random-init weights (--synthetic)".format(model_path))
    else:
        raise FileNotFoundError(model_path)
    net = net.to(device).eval()

    if cfg.synthetic and not os.path.isfile(model_path):
        # the reference data file only holds correctly classified shapes (gen_data_mat.py:255-260): mirror that
        # ... for the targeted modes among the ten ModelNet10 ids the data file can hold
        # (Provider/modelnet10_instance250.py:10): the `All` / class-name expansion builds "the other nine" from them
        with torch.no_grad():
            logits = net(dataset.data.to(device))
            if cfg.attack_label in ("Untarget", "Random"):
                pred = logits.argmax(1).cpu().numpy()
            else:
                ten = torch.as_tensor(TEN_LABEL_INDEXES, device=device)
                pred = ten[logits[:, ten].argmax(1)].cpu().numpy()
        dataset.label = pred.reshape(-1, 1).astype(np.int64)

    cci = CountConvergeIter(os.path.join(saved_dir, "Records")) if cfg.is_record_converged_steps else None
    cli = CountLossIter(os.path.join(saved_dir, "Records")) if cfg.is_record_loss else None
    acc_sum, acc_cnt = 0.0, 0                                # Average_meter of main_attack.py:155,222-224
    num_attack_classes = 9 if cfg.attack_label not in ("Untarget", "Random") else 1
    num_attack_success, cnt_ins, cnt_all = 0, dataset.start_index, 0
    runner_cache = {}
    t_attack = 0.0
    for i, data in enumerate(loader):
        pc, gt_labels = data[0], data[2]
        bs, l = pc.size(0), pc.size(1)
        b = bs * l
        gt_target = gt_labels.view(-1).to(device)
        if cfg.attack is None:                               # clean accuracy only (main_attack.py:213-225)
            with torch.no_grad():
                x = pc.permute(0, 1, 3, 2).reshape(b, 3, -1).to(device).contiguous()
                acc_sum += (net(x).argmax(1) == gt_target).float().sum().item() * 100.0
                acc_cnt += b
            say("Prec@1 {:.3f}".format(acc_sum / acc_cnt))   # running average over the batches so far, as the reference
            continue
        if cfg.attack != "GeoA3":
            raise AssertionError("Wrong type of attack.")
        t0 = time.perf_counter()
        if world > 1:
            out = geoa3_attack.attack_sharded(net, data, cfg, i, len(loader), saved_dir, verbose=not cfg.quiet)
        else:
            out = geoa3_attack.attack(net, data, cfg, i, len(loader), saved_dir, verbose=not cfg.quiet,
                                      runner_cache=runner_cache)
        adv_pc, targeted_label, success, best_attack_step, loss = out
        torch.cuda.synchronize()
        t_attack += time.perf_counter() - t0
        if cci is not None:                                  # main_attack.py:236-239
            cci.record_converge_iter(best_attack_step)
        if cli is not None:
            cli.record_loss_iter(loss)
        saved_normal = None
        if dense_iter is not None:                           # main_attack.py:196-210, 241-247
            from geoa3_amd.attack import unpack_input
            dense_data = next(dense_iter)
            dense_point, dense_normal, _, _ = unpack_input(dense_data, False)
            saved_normal = estimate_normal_via_ori_normal(adv_pc.contiguous(), dense_point.to(device).contiguous(),
                                                          dense_normal.to(device).contiguous(), k=3).cpu().numpy()
        with torch.no_grad():                                # re-evaluation (main_attack.py:249-261)
            eval_points = adv_pc.contiguous()
            if eval_points.size(2) > cfg.npoint:
                eval_points = farthest_points_sample(eval_points, cfg.npoint)
            test_pred = net(eval_points).argmax(1)
        saved_pc = adv_pc.cpu().numpy()
        if rank == 0:
            for k in range(b):
                if bool(success[k]):
                    num_attack_success += 1
                    name = "adv_" + str(cnt_ins + k // num_attack_classes) + "_gt" + str(gt_target[k].item()) + \
                           "_attack" + str(test_pred[k].item()) + "_expect" + str(targeted_label[k].item())
                    _write_outputs(saved_dir, name, saved_pc[k], gt_target[k].item(), test_pred[k].item(),
                                   None if saved_normal is None else saved_normal[k])
        cnt_ins += bs
        cnt_all += b

    if rank == 0:                                            # main_attack.py:298-303
        if cci is not None:
            cci.save_converge_iter()
            cci.plot_converge_iter_hist()
        if cli is not None and cli.loss_numpy is not None:
            cli.save_loss_iter()
            cli.plot_loss_iter_hist()
    if cfg.attack == "GeoA3" and rank == 0:
        line = "attack success: {0:.2f}\n".format(num_attack_success / float(max(cnt_all, 1)) * 100)
        say(line)
        with open(os.path.join(saved_dir, "attack_result.txt"), "at") as f:
            f.write(line)
        say("saved_dir: {0}".format(saved_dir))
        iters = cfg.binary_max_steps * cfg.iter_max_steps * len(loader)
        print("attack time: {:.2f} s, {:.1f} inner iterations/s".format(t_attack, iters / max(t_attack, 1e-9)))
    say("Finish!")
    if world > 1:
        dist.destroy_process_group()
    return saved_dir


if __name__ == "__main__":
    cfg = build_parser().parse_args()
    print(cfg, "\n")
    main(cfg)
