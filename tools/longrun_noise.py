"""Run-to-run spread of the long-horizon window statistics (tests/test_gpu_longrun.py bar (5)): the deterministic loop
with the start perturbed by a few 1e-7 (relative), and the atomics loop repeated -- both against the reference fixture."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from oracle import geoa3_oracle as O
from tests.golden.make_golden_long import LONG_CASES
from tests.test_gpu_longrun import _net
from geoa3_amd.attack import AttackRunner

T = torch.from_numpy
g = np.load("tests/golden/geoa3_golden_long.npz")


def run(tag, mode, det, eps):
    kw, b, n, _ = LONG_CASES[tag]
    pre = "long/%s/" % tag
    cfg = O.AttackCfg(**dict(kw, binary_max_steps=1))
    cfg.deterministic = det
    Tn = cfg.iter_max_steps
    r = AttackRunner(_net(mode), b, n, cfg, torch.device("cuda"))
    ori, nrm, gt = T(g[pre + "ori"]), T(g[pre + "nrm"]), T(g[pre + "gt"])
    r.setup(ori, nrm, gt, gt)
    con, hist = [], []

    def on_step(s, step):
        con.append(r.geo_out["constrain"].clone())
        if step == Tn - 1:
            hist.append(r.t["loss_hist"].cpu().numpy().copy())
    r.run([(T(g[pre + "inits"][0]) * (1 + eps)).cuda()], on_step=on_step)
    ln = hist[0].mean(1)
    cn = torch.stack(con).cpu().numpy().mean(1)
    rl, rc = g[pre + "tr_loss_n"][0].mean(1), g[pre + "tr_constrain"][0].mean(1)
    dl = max(abs(ln[w:w + 50].mean() - rl[w:w + 50].mean()) / abs(rl[w:w + 50].mean()) for w in range(0, Tn, 50))
    dc = max(abs(cn[w:w + 50].mean() - rc[w:w + 50].mean()) / abs(rc[w:w + 50].mean()) for w in range(0, Tn, 50))
    return dl, dc


for tag in ["n256_b8", "n1024_b4", "n256_b8_hard"]:
    for mode, det, epss in [("f16x2", True, [k * 1e-7 for k in range(-6, 7)]), ("f32", True, [k * 1e-7 for k in range(-6, 7)]),
                            ("f16x2", False, [0.0] * 13)]:
        res = np.array([run(tag, mode, det, e) for e in epss])
        print("%-13s %-5s det=%d  loss_n window dev: median %.4f max %.4f | constrain: median %.4f max %.4f" %
              (tag, mode, det, np.median(res[:, 0]), res[:, 0].max(), np.median(res[:, 1]), res[:, 1].max()), flush=True)
