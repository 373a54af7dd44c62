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
    assert int(name[2][2:]) == int(m["gt_label"]) and int(name[3][6:]) == int(m["attack_label"])
    first = open(objs[0]).readline().split()
    assert first[0] == "v" and len(first) == 7 and first[4:] == ["0", "0", "0"]
