"""CPU: the command line and the `.mat` loader against what the reference defines."""
import ast
import json
import os

import numpy as np
import torch

import main_attack
from geoa3_amd.data import TEN_LABEL_INDEXES, ModelNet40, write_synthetic_mat


def test_flag_set_and_defaults_match_reference(golden):
    ref = json.loads(str(golden["cli/flags_json"]))
    assert len(ref) == 49
    parser = main_attack.build_parser()
    actions = {}
    for a in parser._actions:
        for o in a.option_strings:
            actions[o] = a
    for names, default, store_true in ref:
        for n in names:
            assert n in actions, n
        act = actions[names[-1]]
        if store_true:
            assert act.default is False and act.nargs == 0
        else:
            want = None if default == "None" else ast.literal_eval(default)
            assert act.default == want and type(act.default) is type(want), (names, act.default, want)


def test_saved_dir_naming():
    p = main_attack.build_parser()
    cfg = p.parse_args(["--attack", "GeoA3"])
    # the example of the reference's default flags
    assert main_attack.saved_dir_name(cfg) == os.path.join(
        "Exps", "PointNet_npoint1024", "All",
        "GeoA3_0_BiStep10_IterStep500_Optadam_Lr0.01_Initcons10_CE_CDLoss1.0_HDLoss0.1_CurLoss1.0_k16")
    cfg = p.parse_args(["--attack", "GeoA3", "--attack_label", "Untarget", "--hd_loss_weight", "0", "--cc_linf", "0.1",
                        "--is_use_lr_scheduler", "--is_pro_grad", "--is_real_offset", "--initial_const", "5",
                        "--curv_loss_weight", "0", "--dis_loss_type", "L2", "--optim", "sgd"])
    assert main_attack.saved_dir_name(cfg).endswith(
        "Untarget/GeoA3_0_BiStep10_IterStep500_Optsgd_Lr0.01_Initcons5.0_CE_L2Loss1.0_LRExp_ProGradRO_cclinf0.1")
    assert main_attack.saved_dir_name(p.parse_args([])).endswith("All/Evaluating_0")


def test_dataset_matches_reference_expansion(golden, tmp_path):
    mat = write_synthetic_mat(str(tmp_path / "d.mat"), [TEN_LABEL_INDEXES[i // 25] for i in range(250)], 32, 9)
    for lab in ("All", "Untarget", "chair"):
        ds = ModelNet40(data_mat_file=mat, attack_label=lab)
        assert len(ds) == int(golden["ds/%s/len" % lab]) and ds.start_index == int(golden["ds/%s/start" % lab])
        for idx in (0, 7):
            item = ds[idx]
            j = 0
            while "ds/%s/%d/%d" % (lab, idx, j) in golden:
                ref = golden["ds/%s/%d/%d" % (lab, idx, j)]
                got = item[j].numpy()
                assert got.shape == ref.shape and got.dtype == ref.dtype, (lab, idx, j, got.dtype, ref.dtype)
                np.testing.assert_array_equal(got, ref)
                j += 1
            assert j == len(item)
    # the DataLoader collation the harness relies on: [bs, l, N, 3]
    dl = torch.utils.data.DataLoader(ModelNet40(mat, "Untarget"), batch_size=4)
    b = next(iter(dl))
    assert b[0].shape == (4, 1, 32, 3) and b[2].shape == (4, 1)


def _check_flags(parser, ref):
    actions = {}
    for a in parser._actions:
        for o in a.option_strings:
            actions[o] = a
    for names, default, store_true in ref:
        for n in names:
            assert n in actions, n
        act = actions[names[-1]]
        if store_true:
            assert act.default is False and act.nargs == 0
        else:
            want = None if default == "None" else ast.literal_eval(default)
            assert act.default == want and type(act.default) is type(want), (names, act.default, want)


def test_defense_and_smoothness_flag_sets_match_reference():
    import importlib.util
    import numpy as np
    aux = np.load(os.path.join(os.path.dirname(__file__), "golden", "geoa3_golden_aux.npz"))
    import defense
    ref = json.loads(str(aux["cli/defense_flags_json"]))
    assert len(ref) == 13
    _check_flags(defense.build_parser(), ref)
    root = os.path.join(os.path.dirname(__file__), "..")
    spec = importlib.util.spec_from_file_location("compute_data_smoothness",
                                                  os.path.join(root, "Measurement", "compute_data_smoothness.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    ref = json.loads(str(aux["cli/smooth_flags_json"]))
    assert len(ref) == 5
    _check_flags(mod.build_parser(), ref)
