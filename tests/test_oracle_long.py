"""The CPU oracle against the long-horizon reference runs (tests/golden/geoa3_golden_long.npz): the first iterations of
the first binary step, where the oracle and the reference still share one trajectory (both are torch-CPU fp32; the whole
runs take minutes and are only replayed on the GPU, statistically: tests/test_gpu_longrun.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import geoa3_oracle as O
from tests.golden.make_golden_long import LONG_CASES, adversarial, oracle_net, sd_checksum, victim_state_dict

T = torch.from_numpy
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("tag,steps", [("n256_b8_hard", 24), ("n256_b8_tgt", 24), ("n1024_b8_hard", 4),
                                       ("n1024_b4_margin", 6), ("pn2_n1024_b4_tgt", 3), ("n256_b8_fail", 10),
                                       ("pn2_n1024_b8_tgt", 2)])
def test_oracle_follows_reference_prefix(tag, steps):
    g = np.load(os.path.join(REPO, "tests", "golden", "geoa3_golden_long.npz"), allow_pickle=False)
    case = LONG_CASES[tag]
    cfg = O.AttackCfg(**dict(case["cfg"], binary_max_steps=1, iter_max_steps=steps))
    pre = "long/%s/" % tag
    sd = victim_state_dict(case)
    chk = sd_checksum(sd)
    assert abs(chk - float(g[pre + "sd_checksum"])) < 1e-6 * chk
    net = oracle_net(case, sd)
    ori, nrm, gt, tgt = T(g[pre + "ori"]), T(g[pre + "nrm"]), T(g[pre + "gt"]), T(g[pre + "tgt"])
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    with torch.no_grad():   # the stored labels / targets follow from the victim and the clouds
        clean = net(ori)
    np.testing.assert_allclose(clean.numpy(), g[pre + "clean_logits"], rtol=1e-4, atol=2e-4)
    assert (clean.argmax(1) == gt).all()
    targeted = case["target_rank"] > 0
    if targeted:
        assert (T(g[pre + "clean_logits"]).argsort(1, descending=True)[:, case["target_rank"]] == tgt).all()
    tr = {}
    O.attack(net, ori, nrm, gt, tgt if targeted else None, cfg, [T(g[pre + "inits"][0])], trace=tr)
    loss_n = torch.stack(tr["loss_n"]).numpy()
    con = torch.stack(tr["constrain"]).numpy()
    np.testing.assert_allclose(loss_n, g[pre + "tr_loss_n"][0, :steps], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(con, g[pre + "tr_constrain"][0, :steps], rtol=2e-4, atol=1e-7)
    assert (np.asarray(tr["labels"]) == g[pre + "tr_pred"][0, :steps]).mean() >= 0.98


def test_long_fixture_bookkeeping_is_consistent():
    """success / best_constrain stored by the generator follow from the stored traces by the reference's rule
    (geoA3_attack.py:301-310: the iterate of step s is ranked with the constrain loss of step s-1, strict '<'), and
    every case is a HARD one: the reference finds its best iterate tens of steps into a binary step."""
    g = np.load(os.path.join(REPO, "tests", "golden", "geoa3_golden_long.npz"), allow_pickle=False)
    assert sorted(g["long/cases"]) == sorted(LONG_CASES)
    for tag in g["long/cases"]:
        pre = "long/%s/" % tag
        targeted = LONG_CASES[str(tag)]["target_rank"] > 0
        pred, con, gt, tgt = g[pre + "tr_pred"], g[pre + "tr_constrain"], g[pre + "gt"], g[pre + "tgt"]
        S, Tn, b = pred.shape
        best = np.full(b, 1e10, np.float32)
        step = np.full(b, -1)
        for s in range(S):
            for t in range(1, Tn):
                ok = adversarial(pred[s, t], gt, tgt, targeted) & (con[s, t - 1] < best)
                best = np.where(ok, con[s, t - 1], best)
                step = np.where(ok, t, step)
        assert ((best < 1e10) == g[pre + "success"]).all()
        np.testing.assert_array_equal(best, g[pre + "best_constrain"])
        assert (step == g[pre + "best_step"]).mean() >= 0.75   # (batch-1 vs batched forward may differ on a near-tie)
        found = g[pre + "best_step"][g[pre + "success"]]
        assert found.size and np.median(found) > 10, (tag, g[pre + "best_step"])


def test_failing_case_holds_instances_the_reference_never_breaks():
    """`n256_b8_fail`: the reference leaves at least two of the eight instances un-attacked after 3 x 100 steps (never
    adversarial at any step: best_step -1, the all-ones placeholder of geoA3_attack.py:225-227 as best_attack), breaks at
    least two others, and the binary search lowers every constant on each step (:374-386 through the `output_label` quirk)."""
    g = np.load(os.path.join(REPO, "tests", "golden", "geoa3_golden_long.npz"), allow_pickle=False)
    pre = "long/n256_b8_fail/"
    succ = g[pre + "success"]
    assert (~succ).sum() >= 2 and succ.sum() >= 2
    pred, gt, tgt = g[pre + "tr_pred"], g[pre + "gt"], g[pre + "tgt"]
    never = ~adversarial(pred[:, 1:], gt, tgt, True).any((0, 1))
    assert (never == ~succ).all()
    assert (g[pre + "best_step"][~succ] == -1).all()
    assert (g[pre + "best_attack"][~succ] == 1.0).all() and not (g[pre + "best_attack"][succ] == 1.0).all()
    scale = g[pre + "tr_scale"][:, 0, :]
    assert (scale[1:] < scale[:-1]).all()
