"""The CPU oracle against the long-horizon reference runs (tests/golden/geoa3_golden_long.npz): the first iterations of
the first binary step, where the oracle and the reference still share one trajectory (both are torch-CPU fp32; the whole
runs take minutes and are only replayed on the GPU, statistically: tests/test_gpu_longrun.py)."""
import os

import numpy as np
import pytest
import torch

from oracle import geoa3_oracle as O
from tests.golden.make_golden_long import LONG_CASES

T = torch.from_numpy
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.parametrize("tag,steps", [("n256_b8", 24), ("n256_b8_hard", 24), ("n1024_b4", 6)])
def test_oracle_follows_reference_prefix(tag, steps):
    g = np.load(os.path.join(REPO, "tests", "golden", "geoa3_golden_long.npz"), allow_pickle=False)
    kw, b, n, _ = LONG_CASES[tag]
    cfg = O.AttackCfg(**dict(kw, binary_max_steps=1, iter_max_steps=steps))
    pre = "long/%s/" % tag
    sd = O.make_pointnet_state_dict(40, seed=0)
    chk = sum(float(v.double().abs().sum()) for v in sd.values())
    assert abs(chk - float(g["long/sd_checksum"])) < 1e-6 * chk
    net = lambda x: O.pointnet_forward(sd, x)
    ori, nrm, gt = T(g[pre + "ori"]), T(g[pre + "nrm"]), T(g[pre + "gt"])
    tr = {}
    torch.set_num_threads(min(8, os.cpu_count() or 1))
    O.attack(net, ori, nrm, gt, None, cfg, [T(g[pre + "inits"][0])], trace=tr)
    loss_n = torch.stack(tr["loss_n"]).numpy()
    con = torch.stack(tr["constrain"]).numpy()
    np.testing.assert_allclose(loss_n, g[pre + "tr_loss_n"][0, :steps], rtol=2e-4, atol=2e-5)
    np.testing.assert_allclose(con, g[pre + "tr_constrain"][0, :steps], rtol=2e-4, atol=1e-7)
    assert (np.asarray(tr["labels"]) == g[pre + "tr_pred"][0, :steps]).mean() >= 0.98


def test_long_fixture_bookkeeping_is_consistent():
    """success / best_constrain stored by the generator follow from the stored traces by the reference's rule
    (geoA3_attack.py:301-310: the iterate of step s is ranked with the constrain loss of step s-1, strict '<')."""
    g = np.load(os.path.join(REPO, "tests", "golden", "geoa3_golden_long.npz"), allow_pickle=False)
    for tag in g["long/cases"]:
        pre = "long/%s/" % tag
        pred, con, gt = g[pre + "tr_pred"], g[pre + "tr_constrain"], g[pre + "gt"]
        S, Tn, b = pred.shape
        best = np.full(b, 1e10, np.float32)
        step = np.full(b, -1)
        for s in range(S):
            for t in range(1, Tn):
                ok = (pred[s, t] != gt) & (con[s, t - 1] < best)
                best = np.where(ok, con[s, t - 1], best)
                step = np.where(ok, t, step)
        assert ((best < 1e10) == g[pre + "success"]).all()
        np.testing.assert_array_equal(best, g[pre + "best_constrain"])
        assert (step == g[pre + "best_step"]).mean() >= 0.75   # (batch-1 vs batched forward may differ on a near-tie)
