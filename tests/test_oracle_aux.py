"""Pins oracle/aux_oracle.py (and the dense-cloud / pre-jitter branches of oracle.geoa3_oracle.attack) against
outputs of the reference's own functions stored in tests/golden/geoa3_golden_aux.npz (generator:
tests/golden/make_golden_aux.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import aux_oracle as A
from oracle import geoa3_oracle as O
from tests.golden.make_golden_aux import AUX_ATK_CASES
from tests.test_oracle_golden import _traj_close

T = torch.from_numpy


@pytest.fixture(scope="module")
def aux():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "geoa3_golden_aux.npz"))


def test_farthest_points_sample(aux):
    for tag in aux["fps/cases"]:
        pre = "fps/%s/" % tag
        pts, idx = A.farthest_points_sample(T(aux[pre + "pc"]), int(aux[pre + "m"]), T(aux[pre + "start"]))
        assert np.array_equal(pts.numpy(), aux[pre + "pts"]), tag
        assert idx[:, 0].tolist() == aux[pre + "start"].tolist()


def test_estimate_normal_via_ori_normal(aux):
    for tag in aux["nvo/cases"]:
        pre = "nvo/%s/" % tag
        est = A.estimate_normal_via_ori_normal(T(aux[pre + "adv"]), T(aux[pre + "ori"]), T(aux[pre + "nrm"]),
                                               int(aux[pre + "k"]))
        np.testing.assert_allclose(est.numpy(), aux[pre + "est"], rtol=1e-6, atol=1e-7)
        moved = (np.abs(aux[pre + "est"]).sum() > 0)
        assert moved


def perp_terms_close(noise, pc, k, aux1, aux2, clip, tol=3e-5):
    """noise [b,3,n] from an implementation under test; checks it is clamp(+-v_a*aux_1) + clamp(+-v_b*aux_2) for
    {v_a, v_b} = the two leading eigenvectors (either order) of the float64 neighbourhood covariance."""
    cov = A.local_covariance(pc.double(), k)
    w, v = torch.linalg.eigh(cov)
    v1, v2 = v[..., 2].permute(0, 2, 1), v[..., 1].permute(0, 2, 1)       # [b,3,n]
    best = None
    for a, bvec in ((v1, v2), (v2, v1)):
        for s1 in (1.0, -1.0):
            for s2 in (1.0, -1.0):
                cand = torch.clamp(s1 * a * aux1.unsqueeze(1).double(), -clip, clip) + \
                    torch.clamp(s2 * bvec * aux2.unsqueeze(1).double(), -clip, clip)
                err = (cand - noise.double()).abs().amax(dim=1)            # [b,n]
                best = err if best is None else torch.minimum(best, err)
    gap_ok = ((w[..., 2] - w[..., 1]) > 1e-2 * w[..., 2]) & ((w[..., 1] - w[..., 0]) > 1e-2 * w[..., 2])
    assert gap_ok.float().mean() > 0.8
    assert best[gap_ok].max().item() < tol, best[gap_ok].max().item()
    assert noise.abs().max().item() <= 2 * clip + 1e-7


def test_estimate_perpendicular(aux):
    for tag in aux["perp/cases"]:
        pre = "perp/%s/" % tag
        pc, k, clip = T(aux[pre + "pc"]), int(aux[pre + "k"]), float(aux[pre + "clip"])
        a1, a2 = T(aux[pre + "aux1"]), T(aux[pre + "aux2"])
        perp_terms_close(T(aux[pre + "noise"]), pc, k, a1, a2, clip)       # the reference's own output
        noise, _, _ = A.estimate_perpendicular(pc, k, a1, a2, clip)
        perp_terms_close(noise, pc, k, a1, a2, clip)                       # the oracle


def test_defense_point_removal(aux):
    for tag in aux["def/cases"]:
        pre = "def/%s/" % tag
        kept, num, idx = A.outlier_removal(T(aux[pre + "pc"]), str(aux[pre + "type"]), int(aux[pre + "drop"]),
                                           float(aux[pre + "alpha"]), int(aux[pre + "knn"]))
        assert num == int(aux[pre + "num"]) and num > 0, tag
        assert np.array_equal(kept.numpy(), aux[pre + "kept"]), tag
    kept, num = A.random_drop(T(aux["def/rand/pc"]), int(aux["def/rand/drop"]), T(aux["def/rand/perm"]))
    assert np.array_equal(kept.numpy(), aux["def/rand/kept"]) and num == 20


def test_smoothness(aux):
    for tag in aux["smooth/cases"]:
        pre = "smooth/%s/" % tag
        s = A.smoothness(T(aux[pre + "pc"]).t().contiguous(), int(aux[pre + "k"]), int(aux[pre + "k2"]))
        np.testing.assert_allclose(float(s), float(aux[pre + "value"][0]), rtol=2e-5)


def _oracle_attack_case(aux, tag, jitter_from_golden=True):
    kw, b, n, seed = AUX_ATK_CASES[tag]
    pre = "atk/%s/" % tag
    cfg = O.AttackCfg(**kw)
    sd = O.make_pointnet_state_dict(40, seed=0)
    net = lambda x: O.pointnet_forward(sd, x)
    ori, nrm, gt = T(aux[pre + "ori"]), T(aux[pre + "nrm"]), T(aux[pre + "gt"])
    inits = [T(i) for i in aux[pre + "inits"]]
    iters = cfg.iter_max_steps
    sub = (lambda s, step: T(aux[pre + "sub_starts"][s * iters + step])) if cfg.is_subsample_opt else None
    vote = (lambda s, step: T(aux[pre + "vote_starts"][s * iters + step])) if cfg.is_subsample_opt else None
    jit = None
    if cfg.is_pre_jitter_input:
        every = cfg.calculate_project_jitter_noise_iter

        def jit(s, step, x):
            j = (s * iters + step) // every
            if jitter_from_golden:
                return T(aux[pre + "noise"][j])
            return A.estimate_perpendicular(x, cfg.jitter_k, T(aux[pre + "aux"][2 * j]), T(aux[pre + "aux"][2 * j + 1]),
                                            cfg.jitter_clip)[0]
    pp = pi = None
    if cfg.is_partial_var:   # one (point, init) draw per 50 steps, in run order
        per = (iters + 49) // 50
        pp = lambda s, step: int(aux[pre + "part_points"][s * per + step // 50])
        pi = lambda s, step: T(aux[pre + "part_inits"][s * per + step // 50])
    return cfg, (net, ori, nrm, gt, None, cfg, inits), dict(sub_starts=sub, vote_starts=vote, jitter_noise=jit,
                                                            partial_points=pp, partial_inits=pi)


@pytest.mark.parametrize("tag", list(AUX_ATK_CASES))
def test_attack_dense_and_jitter_paths(aux, tag):
    cfg, args, kw = _oracle_attack_case(aux, tag)
    pre = "atk/%s/" % tag
    xs = []
    real_fs = O.forward_step

    def spy(net, pc_ori, x, *a, **k):
        xs.append(x.detach().clone())
        return real_fs(net, pc_ori, x, *a, **k)

    O.forward_step = spy
    try:
        best, tgt, succ, best_step, all_loss = O.attack(*args, **kw)
    finally:
        O.forward_step = real_fs
    xs = torch.stack(xs).numpy()
    assert xs.shape == aux[pre + "tr_x"].shape
    _traj_close(xs, aux[pre + "tr_x"], loose=2 * cfg.lr * cfg.iter_max_steps)
    assert np.array_equal(np.asarray(succ), aux[pre + "success"])
    assert list(best_step) == aux[pre + "best_step"].tolist()
    np.testing.assert_allclose(np.asarray(all_loss, dtype=np.float32), aux[pre + "all_loss"], rtol=2e-4, atol=2e-5)
    ok = aux[pre + "success"]
    np.testing.assert_allclose(best.numpy()[ok], aux[pre + "best_attack"][ok], atol=2e-5)


def test_pre_jitter_noise_from_the_oracle(aux):
    """The same run with the jitter produced by the oracle's estimate_perpendicular from the stored draws
    (CPU LAPACK on both sides, so signs and order agree here)."""
    cfg, args, kw = _oracle_attack_case(aux, "pre_jitter", jitter_from_golden=False)
    _, _, succ, best_step, all_loss = O.attack(*args, **kw)
    np.testing.assert_allclose(np.asarray(all_loss, dtype=np.float32), aux["atk/pre_jitter/all_loss"], rtol=5e-4,
                               atol=5e-5)
