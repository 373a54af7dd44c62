"""Run-to-run spread of the long-horizon statistics (tests/test_gpu_longrun.py bars): the deterministic loop with the
start perturbed by a few 1e-7 (relative), and the atomics loop repeated -- first binary step, against the reference fixture.

    python tools/longrun_noise.py [tag ...]            (GPU)
    python tools/longrun_noise.py --oracle tag [n]     (CPU oracle under the same perturbation: the reference side's chaos)
"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from oracle import geoa3_oracle as O
from tests.golden.make_golden_long import LONG_CASES, adversarial, oracle_net

T = torch.from_numpy
g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "geoa3_golden_long.npz"))


def stats(tag, loss_n, con, pred):
    """loss_n / con / pred [T,b] of the first binary step -> (max window deviation of loss_n, of constrain, |adv fraction
    difference|) against the reference."""
    case = LONG_CASES[tag]
    pre = "long/%s/" % tag
    Tn = loss_n.shape[0]
    rl, rc = g[pre + "tr_loss_n"][0].mean(1), g[pre + "tr_constrain"][0].mean(1)
    ln, cn = loss_n.mean(1), con.mean(1)
    dl = max(abs(ln[w:w + 50].mean() - rl[w:w + 50].mean()) / abs(rl[w:w + 50].mean()) for w in range(0, Tn, 50))
    dc = max(abs(cn[w:w + 50].mean() - rc[w:w + 50].mean()) / abs(rc[w:w + 50].mean()) for w in range(0, Tn, 50))
    tg = case["target_rank"] > 0
    fa = adversarial(pred, g[pre + "gt"], g[pre + "tgt"], tg).mean()
    fr = adversarial(g[pre + "tr_pred"][0], g[pre + "gt"], g[pre + "tgt"], tg).mean()
    return dl, dc, abs(fa - fr)


def run_gpu(tag, mode, det, eps):
    from tests.test_gpu_longrun import _net
    from geoa3_amd.attack import AttackRunner
    case = LONG_CASES[tag]
    pre = "long/%s/" % tag
    cfg = O.AttackCfg(**dict(case["cfg"], binary_max_steps=1))
    cfg.deterministic = det
    Tn = cfg.iter_max_steps
    r = AttackRunner(_net(mode, case), case["b"], case["n"], cfg, torch.device("cuda"))
    r.setup(T(g[pre + "ori"]), T(g[pre + "nrm"]), T(g[pre + "gt"]), T(g[pre + "tgt"]))
    con, hist, lab = [], [], []

    def on_step(s, step):
        con.append(r.geo_out["constrain"].clone())
        lab.append(r.t["label"].clone())
        if step == Tn - 1:
            hist.append(r.t["loss_hist"].cpu().numpy().copy())
    r.run([(T(g[pre + "inits"][0]) * (1 + eps)).cuda()], on_step=on_step)
    return stats(tag, hist[0], torch.stack(con).cpu().numpy(), torch.stack(lab).cpu().numpy())


def run_oracle(tag, eps):
    case = LONG_CASES[tag]
    pre = "long/%s/" % tag
    cfg = O.AttackCfg(**dict(case["cfg"], binary_max_steps=1))
    tr = {}
    tg = case["target_rank"] > 0
    O.attack(oracle_net(case), T(g[pre + "ori"]), T(g[pre + "nrm"]), T(g[pre + "gt"]), T(g[pre + "tgt"]) if tg else None, cfg,
             [T(g[pre + "inits"][0]) * (1 + eps)], trace=tr)
    return stats(tag, torch.stack(tr["loss_n"]).numpy(), torch.stack(tr["constrain"]).numpy(), np.asarray(tr["labels"]))


if __name__ == "__main__":
    if "--oracle" in sys.argv:
        tag = sys.argv[sys.argv.index("--oracle") + 1]
        n = int(sys.argv[sys.argv.index("--oracle") + 2]) if len(sys.argv) > sys.argv.index("--oracle") + 2 else 4
        torch.set_num_threads(8)
        res = np.array([run_oracle(tag, k * 1e-7) for k in range(1, n + 1)])
        print("%-18s oracle (CPU), start x (1 + k 1e-7), %d runs: loss_n window dev median %.4f max %.4f | constrain median %.4f max "
              "%.4f | adv fraction diff max %.4f" % (tag, n, np.median(res[:, 0]), res[:, 0].max(), np.median(res[:, 1]),
                                                     res[:, 1].max(), res[:, 2].max()), flush=True)
        sys.exit(0)
    tags = [a for a in sys.argv[1:] if a in LONG_CASES] or list(LONG_CASES)
    for tag in tags:
        pn = LONG_CASES[tag]["arch"] == "PointNet"
        for mode, det, epss in ([("f16x2", True, [k * 1e-7 for k in range(-6, 7)]), ("f32", True, [k * 1e-7 for k in range(-6, 7)]),
                                 ("f16x2", False, [0.0] * 13)] if pn else
                                [("native", True, [k * 1e-7 for k in range(-6, 7)]), ("native", False, [0.0] * 13)]):
            res = np.array([run_gpu(tag, mode, det, e) for e in epss])
            print("%-18s %-6s det=%d  loss_n window dev: median %.4f max %.4f | constrain: median %.4f max %.4f | adv fraction "
                  "diff: median %.4f max %.4f" % (tag, mode, det, np.median(res[:, 0]), res[:, 0].max(), np.median(res[:, 1]),
                                                  res[:, 1].max(), np.median(res[:, 2]), res[:, 2].max()), flush=True)
