"""GPU: the command line end to end on synthetic data (reference-compatible outputs)."""
import glob
import os

import numpy as np
import pytest
from scipy.io import loadmat

pytestmark = pytest.mark.gpu


def test_main_attack_outputs(tmp_path, monkeypatch):
    import main_attack
    monkeypatch.chdir(tmp_path)
    args = ["--attack", "GeoA3", "--attack_label", "Untarget", "-b", "125", "--npoint", "128", "--synthetic",
            "--data_dir_file", str(tmp_path / "Data" / "syn128.mat"), "--binary_max_steps", "2", "--iter_max_steps",
            "6", "--lr", "0.005", "--curv_loss_knn", "8", "--quiet"]
    cfg = main_attack.build_parser().parse_args(args)
    saved_dir = main_attack.main(cfg)
    assert saved_dir == os.path.join("Exps", "PointNet_npoint128", "Untarget",
                                     "GeoA3_0_BiStep2_IterStep6_Optadam_Lr0.005_Initcons10_CE_CDLoss1.0_HDLoss0.1"
                                     "_CurLoss1.0_k8")
    res = open(os.path.join(saved_dir, "attack_result.txt")).read()
    assert res.startswith("attack success: ")
    rate = float(res.split(":")[1])
    mats = sorted(glob.glob(os.path.join(saved_dir, "Mat", "adv_*.mat")))
    objs = sorted(glob.glob(os.path.join(saved_dir, "PC", "adv_*.obj")))
    assert len(mats) == len(objs) == round(rate * 250 / 100.0) and len(mats) > 0
    m = loadmat(mats[0])   # schema consumed by Provider/defense_modelnet10_instance250.py:25-31
    assert m["adversary_point_clouds"].shape == (3, 128) and m["adversary_point_clouds"].dtype == np.float32
    assert "gt_label" in m and "attack_label" in m
    name = os.path.basename(mats[0])[:-4].split("_")
    assert name[0] == "adv" and name[2].startswith("gt") and name[3].startswith("attack") and name[4].startswith("expect")
    assert int(name[2][2:]) == int(m["gt_label"].item()) and int(name[3][6:]) == int(m["attack_label"].item())
    first = open(objs[0]).readline().split()
    assert first[0] == "v" and len(first) == 7 and first[4:] == ["0", "0", "0"]


def _attack_run(tmp_path, extra, npoint=128):
    import main_attack
    args = ["--attack", "GeoA3", "--attack_label", "Untarget", "-b", "125", "--npoint", str(npoint), "--synthetic",
            "--data_dir_file", str(tmp_path / "Data" / "syn.mat"), "--binary_max_steps", "2", "--iter_max_steps",
            "6", "--lr", "0.005", "--curv_loss_knn", "8", "--quiet"] + extra
    return main_attack.main(main_attack.build_parser().parse_args(args))


def test_dense_cloud_cli_with_saved_normals(tmp_path, monkeypatch):
    """--is_subsample_opt on 320-point clouds with a 96-point victim, eval_num vote, --is_save_normal from the
    dense file (main_attack.py:124-130,241-247,269-271)."""
    monkeypatch.chdir(tmp_path)
    saved_dir = _attack_run(tmp_path, ["--synthetic_npoint", "320", "--is_subsample_opt", "--eval_num", "2",
                                       "--is_save_normal", "--dense_data_dir_file",
                                       str(tmp_path / "Data" / "syn.mat")], npoint=96)
    mats = sorted(glob.glob(os.path.join(saved_dir, "Mat", "adv_*.mat")))
    assert len(mats) > 0
    m = loadmat(mats[0])
    assert m["adversary_point_clouds"].shape == (3, 320) and m["est_normal"].shape == (3, 320)
    nrm = np.linalg.norm(m["est_normal"], axis=0)
    np.testing.assert_allclose(nrm, 1.0, atol=1e-4)


@pytest.mark.parametrize("defense_type,extra", [("outliers_fixNum", ["--drop_num", "16"]),
                                                ("outliers_variance", ["--alpha", "1.1"]),
                                                ("rand_drop", ["--drop_num", "16"])])
def test_defense_cli_matches_oracle(tmp_path, monkeypatch, defense_type, extra):
    import torch
    import defense
    from oracle import aux_oracle as A
    from oracle import geoa3_oracle as O
    from geoa3_amd.data import synthetic_state_dict
    monkeypatch.chdir(tmp_path)
    saved_dir = _attack_run(tmp_path, [])
    datadir = os.path.join(saved_dir, "Mat")
    cfg = defense.build_parser().parse_args(["--datadir", datadir, "--npoint", "128", "--defense_type", defense_type,
                                             "--synthetic", "--is_record_wrong", "--print_freq", "1000"] + extra)
    final_acc, final_attack_acc, avg_drop = defense.main(cfg)
    line = open(os.path.join(saved_dir, "defense_result.txt")).read().strip()
    assert line.startswith("[%.2f%%, %.2f%%, %.2fn] " % (final_acc, final_attack_acc, avg_drop))
    if defense_type == "rand_drop":
        assert avg_drop == 16 and line.endswith("random drop: drop_num 16")
        return
    # the same numbers from the oracle, one cloud at a time like the reference
    sd = {k: v.cpu() for k, v in synthetic_state_dict(40, seed=0, device=torch.device("cuda")).items()}
    files = os.listdir(datadir)
    ok = still = drop = 0
    near_tie = 0
    for f in files:
        m = loadmat(os.path.join(datadir, f))
        pc = torch.from_numpy(m["adversary_point_clouds"]).unsqueeze(0)
        kept, num, _ = A.outlier_removal(pc, defense_type, cfg.drop_num, cfg.alpha, cfg.outlier_knn)
        logits = O.pointnet_forward(sd, kept)[0]
        top2 = logits.topk(2).values
        near_tie += int(top2[0] - top2[1] < 1e-3)
        pred, gt, atk = int(logits.argmax()), int(m["gt_label"].item()), int(m["attack_label"].item())
        if gt == atk:
            ok += 1
        else:
            ok += int(pred == gt)
            still += int(pred == atk)
        drop += num
    assert abs(avg_drop - drop / float(len(files))) < 1e-9
    assert abs(final_acc - ok / float(len(files)) * 100) <= near_tie * 100.0 / len(files) + 1e-9
    assert abs(final_attack_acc - still / float(len(files)) * 100) <= near_tie * 100.0 / len(files) + 1e-9
    wrong = glob.glob(os.path.join(saved_dir, "Defensed", "Gt*_record_*_attack*_defensedGT*.obj"))
    assert len(wrong) == round((100 - final_acc) / 100.0 * len(files))


def test_smoothness_cli_matches_oracle(tmp_path, monkeypatch):
    import importlib.util
    import torch
    from oracle import aux_oracle as A
    monkeypatch.chdir(tmp_path)
    saved_dir = _attack_run(tmp_path, [])
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    spec = importlib.util.spec_from_file_location("compute_data_smoothness",
                                                  os.path.join(root, "Measurement", "compute_data_smoothness.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    cfg = mod.build_parser().parse_args(["--datadir", saved_dir, "--k", "8", "--k2", "12"])
    vals = mod.main(cfg).numpy()
    files = os.listdir(os.path.join(saved_dir, "Mat"))
    assert vals.shape == (len(files),)
    for i in (0, len(files) // 2, len(files) - 1):
        pc = torch.from_numpy(loadmat(os.path.join(saved_dir, "Mat", files[i]))["adversary_point_clouds"])
        np.testing.assert_allclose(vals[i], float(A.smoothness(pc.t().contiguous(), 8, 12)), rtol=1e-4)
    saved = loadmat(os.path.join(saved_dir, "metric", "k8.mat"))["smoothness"].reshape(-1)
    np.testing.assert_array_equal(saved, vals)
    txt = open(os.path.join(saved_dir, "metric", "result.txt")).read()
    assert txt == "k: 8, avg: {0:.4f}, min: {1:.4f}, max: {2:.4f}\n".format(vals.mean(), vals.min(), vals.max())


def test_main_attack_targeted_all_with_records(tmp_path, monkeypatch, capsys):
    """`--attack_label All` (the CLI default): 9 targets per instance from the other ModelNet10 labels
    (Provider/modelnet10_instance250.py:66-72), file names numbered `cnt_ins + k // 9` (main_attack.py:262-263), and
    the two recorders of main_attack.py:150-153,236-239,298-303."""
    import main_attack
    from geoa3_amd.data import TEN_LABEL_INDEXES
    monkeypatch.chdir(tmp_path)
    args = ["--attack", "GeoA3", "--attack_label", "All", "-b", "5", "--npoint", "128", "--synthetic",
            "--data_dir_file", str(tmp_path / "Data" / "syn128.mat"), "--binary_max_steps", "2", "--iter_max_steps",
            "8", "--lr", "0.01", "--curv_loss_knn", "8", "--quiet", "--is_record_converged_steps", "--is_record_loss"]
    cfg = main_attack.build_parser().parse_args(args)
    # 250 instances x 9 targets is a long run for a test: keep the first 10 instances of the data file
    from geoa3_amd import data as D
    orig_init = D.ModelNet40.__init__

    def short_init(self, *a, **k):
        orig_init(self, *a, **k)
        self.data, self.normal, self.label = self.data[:10], self.normal[:10], self.label[:10]

    monkeypatch.setattr(D.ModelNet40, "__init__", short_init)
    saved_dir = main_attack.main(cfg)
    assert os.path.join("PointNet_npoint128", "All", "GeoA3_0_BiStep2_IterStep8") in saved_dir
    mats = sorted(glob.glob(os.path.join(saved_dir, "Mat", "adv_*.mat")))
    res = float(open(os.path.join(saved_dir, "attack_result.txt")).read().split(":")[1])
    assert len(mats) == round(res * 90 / 100.0)
    seen = set()
    for f in mats:
        name = os.path.basename(f)[:-4].split("_")
        ins, gt, exp = int(name[1]), int(name[2][2:]), int(name[4][6:])
        assert 0 <= ins < 10 and gt in TEN_LABEL_INDEXES and exp in TEN_LABEL_INDEXES and exp != gt
        assert (ins, exp) not in seen      # one file per (instance, target)
        seen.add((ins, exp))
    conv = loadmat(os.path.join(saved_dir, "Records", "converge_iter.mat"))["attack_step_list"].reshape(-1)
    loss = loadmat(os.path.join(saved_dir, "Records", "loss_iter.mat"))["loss"]
    assert loss.shape == (8, 90) and np.isfinite(loss).all()
    # every batch of 45 attacks contributes its best steps minus the first -1 (Lib/utility.py:662-663)
    assert 90 - 2 <= conv.size <= 90 and conv.max() < 8


def test_clean_accuracy_is_a_running_average(tmp_path, monkeypatch, capsys):
    """`--attack None`-style evaluation (main_attack.py:213-225): the printed Prec@1 is the Average_meter over the
    batches so far, so the LAST line is the dataset accuracy (100 on the synthetic file, whose labels are the
    net's own predictions)."""
    import main_attack
    monkeypatch.chdir(tmp_path)
    args = ["--attack_label", "Untarget", "-b", "100", "--npoint", "128", "--synthetic",
            "--data_dir_file", str(tmp_path / "Data" / "syn128.mat")]
    cfg = main_attack.build_parser().parse_args(args)
    assert cfg.attack is None
    main_attack.main(cfg)
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.startswith("Prec@1")]
    assert len(lines) == 3 and lines[-1] == "Prec@1 100.000"
