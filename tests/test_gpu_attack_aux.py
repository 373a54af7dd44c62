"""The dense-cloud (--is_subsample_opt + eval_num vote) and --is_pre_jitter_input branches of the device-resident
attack loop against the reference's own runs stored in tests/golden/geoa3_golden_aux.npz (the reference's random
draws are replayed through the runner's hooks)."""
import os

import numpy as np
import pytest
import torch

from oracle import geoa3_oracle as O
from tests.golden.make_golden_aux import AUX_ATK_CASES
from tests.test_gpu_attack import _loader_batch
from tests.test_oracle_aux import perp_terms_close
from tests.test_oracle_golden import _traj_close

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def aux():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "geoa3_golden_aux.npz"))


@pytest.fixture(scope="module")
def net():
    from geoa3_amd.pointnet import PointNet
    n = PointNet(40)
    n.load_state_dict(O.make_pointnet_state_dict(40, seed=0))
    return n.cuda().eval()


def _run_case(net, aux, tag, hooks):
    from geoa3_amd.attack import AttackRunner, unpack_input
    kw, b, n, seed = AUX_ATK_CASES[tag]
    pre = "atk/%s/" % tag
    cfg = O.AttackCfg(**kw)
    ori, nrm, gt = T(aux[pre + "ori"]), T(aux[pre + "nrm"]), T(aux[pre + "gt"])
    pc, nm, g, t = unpack_input(_loader_batch(ori, nrm, gt, None, False), False)
    r = AttackRunner(net, b, n, cfg, torch.device("cuda"))
    r.setup(pc, nm, g, t)
    xs = []

    def on_step(search_step, step):
        buf = "x_eval" if r.jitter else ("x_cur" if r.sub else "x")
        xs.append(r.t[buf].cpu().clone())

    if not r.sub and not r.jitter and not r.partial:
        raise AssertionError("case does not exercise the new paths")
    inits = [T(i).cuda() for i in aux[pre + "inits"]]
    r.run(inits, on_step=on_step, hooks=hooks)
    return cfg, r, r.results(), torch.stack(xs).numpy()


@pytest.mark.parametrize("tag", list(AUX_ATK_CASES))
def test_attack_matches_reference_run(net, aux, tag):
    kw, b, n, seed = AUX_ATK_CASES[tag]
    pre = "atk/%s/" % tag
    iters = kw["iter_max_steps"]
    hooks = {}
    if kw.get("is_subsample_opt"):
        hooks["sub_starts"] = lambda s, step: T(aux[pre + "sub_starts"][s * iters + step])
        hooks["vote_starts"] = lambda s, step: T(aux[pre + "vote_starts"][s * iters + step])
    if kw.get("is_partial_var"):
        per = (iters + 49) // 50
        hooks["partial_points"] = lambda s, step: int(aux[pre + "part_points"][s * per + step // 50])
        hooks["partial_inits"] = lambda s, step: T(aux[pre + "part_inits"][s * per + step // 50]).cuda()
    if kw.get("is_pre_jitter_input"):   # eigenvector signs are implementation-defined: replay the reference's noise
        every = kw["calculate_project_jitter_noise_iter"]
        hooks["jitter_noise"] = lambda s, step, x: T(aux[pre + "noise"][(s * iters + step) // every]).cuda()
    cfg, r, (best, tgt, succ, best_step, all_loss), xs = _run_case(net, aux, tag, hooks)
    assert xs.shape == aux[pre + "tr_x"].shape
    if kw.get("is_partial_var"):
        # the x buffer holds the NEXT step's iterate after a step (except before a re-draw, whose update is dropped):
        # compare it with the iterate the reference evaluated one step later
        keep = [j for j in range(len(xs) - 1) if (j % iters) + 1 < iters and ((j % iters) + 1) % 50 != 0]
        want = aux[pre + "tr_x"][[j + 1 for j in keep]]
        # SGD at lr 0.05 on a 64-point cloud: one 1-NN / arg-max tie broken the other way (round 5: the objective kernels are
        # compiled without packed FP32 and contract their multiply-adds differently; at step 27 of this case 0.8 % of the
        # coordinates leave the reference by more than 2e-5, 3.9 % by step 50) and the iterates part for good.  The update
        # rule and the moving set are pinned by the steps before: tight there, the loose bound and 95 % over the whole run.
        _traj_close(xs[keep][:20], want[:20], loose=2 * cfg.lr * cfg.iter_max_steps)
        _traj_close(xs[keep], want, frac=0.95, loose=2 * cfg.lr * cfg.iter_max_steps)
        moved = np.abs(aux[pre + "tr_x"][1] - aux[pre + "ori"]).max(axis=(0, 1)) > 0
        assert moved.sum() <= kw["knn_range"] * b       # only the chosen neighbourhood moves
    else:
        _traj_close(xs, aux[pre + "tr_x"], loose=2 * cfg.lr * cfg.iter_max_steps)
    got_loss, want_loss = np.asarray(all_loss, dtype=np.float32), aux[pre + "all_loss"]
    if tag == "partial_var_sgd":
        # (the run that parts from the reference at step 27, above: its losses [step][instance] to 5e-4 up to there, to 5e-3
        # -- they differ by up to 1.8e-3 -- behind it)
        np.testing.assert_allclose(got_loss[:25], want_loss[:25], rtol=5e-4, atol=5e-5)
        np.testing.assert_allclose(got_loss, want_loss, rtol=5e-3, atol=5e-5)
    else:
        np.testing.assert_allclose(got_loss, want_loss, rtol=5e-4, atol=5e-5)
    assert np.array_equal(np.asarray(succ), aux[pre + "success"])
    assert list(best_step) == aux[pre + "best_step"].tolist()
    ok = aux[pre + "success"]
    np.testing.assert_allclose(best.cpu().numpy()[ok], aux[pre + "best_attack"][ok], atol=2e-5)
    assert best.shape[2] == n          # the FULL cloud is what is recorded (geoA3_attack.py:305)


def test_pre_jitter_with_device_frames(net, aux):
    """Same run with the jitter produced by the HIP estimate_perpendicular from the reference's randn draws: the
    first jitter (computed on identical iterates) is the reference's up to eigenvector signs."""
    tag = "pre_jitter"
    kw, b, n, seed = AUX_ATK_CASES[tag]
    pre = "atk/%s/" % tag
    iters, every = kw["iter_max_steps"], kw["calculate_project_jitter_noise_iter"]
    seen = []

    def jitter_aux(s, step):
        j = (s * iters + step) // every
        return T(aux[pre + "aux"][2 * j]).cuda(), T(aux[pre + "aux"][2 * j + 1]).cuda()

    cfg, r, res, xs = _run_case(net, aux, tag, {"jitter_aux": jitter_aux})
    x0 = T(aux[pre + "ori"]) + T(aux[pre + "inits"][0])
    noise0 = T(xs[0]) - x0
    perp_terms_close(noise0, x0, kw["jitter_k"], T(aux[pre + "aux"][0]), T(aux[pre + "aux"][1]), kw["jitter_clip"],
                     tol=1e-4)
    assert np.isfinite(np.asarray(res[4])).all()


def test_subsample_default_draws_run(net):
    """Without hooks the start indices come from torch.randint on the device, as in the reference."""
    from geoa3_amd.attack import attack
    cfg = O.AttackCfg(is_subsample_opt=True, npoint=128, eval_num=2, curv_loss_knn=8, binary_max_steps=1,
                      iter_max_steps=4, lr=0.002)
    ori, nrm = O.make_synthetic_clouds(3, 500, seed=77)
    gt = torch.tensor([1, 2, 3])
    best, target, succ, best_step, all_loss = attack(net, _loader_batch(ori, nrm, gt, None, False), cfg, 0, 1,
                                                     verbose=False)
    assert best.shape == (3, 3, 500) and len(all_loss) == 4 and np.isfinite(np.asarray(all_loss)).all()
