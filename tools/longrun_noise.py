"""Run-to-run spread of the long-horizon statistics (tests/test_gpu_longrun.py bars): the deterministic loop with the
start perturbed by a few 1e-7 (relative), and the atomics loop repeated -- first binary step, against the reference fixture.

    python tools/longrun_noise.py [tag ...]            (GPU)
    python tools/longrun_noise.py --oracle tag [n]     (CPU oracle under the same perturbation: the reference side's chaos)
    python tools/longrun_noise.py --full tag ...       (GPU: the WHOLE run of the test, every binary step, 13 times per mode)
"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from oracle import geoa3_oracle as O
from tests.golden.make_golden_long import LONG_CASES, adversarial, oracle_net

T = torch.from_numpy
_FULL = []
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
    if "--full" in sys.argv:
        for tag in [a for a in sys.argv[1:] if a in LONG_CASES]:
            for det in (True, False):
                _FULL.append((tag, det))
    elif "--oracle" in sys.argv:
        tag = sys.argv[sys.argv.index("--oracle") + 1]
        n = int(sys.argv[sys.argv.index("--oracle") + 2]) if len(sys.argv) > sys.argv.index("--oracle") + 2 else 4
        torch.set_num_threads(8)
        res = np.array([run_oracle(tag, k * 1e-7) for k in range(1, n + 1)])
        print("%-18s oracle (CPU), start x (1 + k 1e-7), %d runs: loss_n window dev median %.4f max %.4f | constrain median %.4f max "
              "%.4f | adv fraction diff max %.4f" % (tag, n, np.median(res[:, 0]), res[:, 0].max(), np.median(res[:, 1]),
                                                     res[:, 1].max(), res[:, 2].max()), flush=True)
        sys.exit(0)
    tags = ([] if "--full" in sys.argv else [a for a in sys.argv[1:] if a in LONG_CASES] or list(LONG_CASES))
    for tag in tags:
        pn = LONG_CASES[tag]["arch"] == "PointNet"
        for mode, det, epss in ([("f16x2", True, [k * 1e-7 for k in range(-6, 7)]), ("f32", True, [k * 1e-7 for k in range(-6, 7)]),
                                 ("f16x2", False, [0.0] * 13)] if pn else
                                [("native", True, [k * 1e-7 for k in range(-6, 7)]), ("native", False, [0.0] * 13)]):
            res = np.array([run_gpu(tag, mode, det, e) for e in epss])
            print("%-18s %-6s det=%d  loss_n window dev: median %.4f max %.4f | constrain: median %.4f max %.4f | adv fraction "
                  "diff: median %.4f max %.4f" % (tag, mode, det, np.median(res[:, 0]), res[:, 0].max(), np.median(res[:, 1]),
                                                  res[:, 1].max(), np.median(res[:, 2]), res[:, 2].max()), flush=True)


def full_run_window_spread(tag, deterministic, repeats=13):
    """The WHOLE long run (every binary step, the reference's last labels replayed: exactly what
    tests/test_gpu_longrun.py::test_long_run_statistics_match_reference executes) `repeats` times -- the start perturbed by
    k 1e-7 in the deterministic mode, plain repeats with the float atomics -- and per run the largest 50-step window deviation
    of loss_n / of the constrain loss from the reference over ALL binary steps, and the largest adversarial-fraction
    difference: the quantities the test's `window` and `adv` bars hold."""
    from tests.test_gpu_longrun import _net, _run
    case = LONG_CASES[tag]
    pre = "long/%s/" % tag
    targeted = case["target_rank"] > 0
    out_rows = []
    for k in range(repeats):
        cfg = O.AttackCfg(**case["cfg"])
        eps = (k - repeats // 2) * 1e-7 if deterministic else 0.0
        gg = dict(g)
        gg[pre + "inits"] = g[pre + "inits"] * (1 + eps)
        out = _run(_net("native" if case["arch"] != "PointNet" else "f16x2", case), cfg, gg, pre, deterministic, replay_last=True)
        S, Tn = cfg.binary_max_steps, cfg.iter_max_steps
        ref_scale = g[pre + "tr_scale"][:, 0, :]
        dl = dc = 0.0
        for s in range(S):
            sel = np.isclose(out["scale"][s], ref_scale[s], rtol=1e-6)
            if not sel.any():
                continue
            rl, gl = g[pre + "tr_loss_n"][s][:, sel].mean(1), out["loss_n"][s][:, sel].mean(1)
            rc, gc = g[pre + "tr_constrain"][s][:, sel].mean(1), out["con"][s][:, sel].mean(1)
            for w in range(0, Tn, 50):
                dl = max(dl, abs(gl[w:w + 50].mean() - rl[w:w + 50].mean()) / abs(rl[w:w + 50].mean()))
                dc = max(dc, abs(gc[w:w + 50].mean() - rc[w:w + 50].mean()) / abs(rc[w:w + 50].mean()))
        fa_ref = adversarial(g[pre + "tr_pred"], g[pre + "gt"], g[pre + "tgt"], targeted).mean((1, 2))
        fa_got = adversarial(out["pred"], g[pre + "gt"], g[pre + "tgt"], targeted).mean((1, 2))
        out_rows.append((dl, dc, float(np.abs(fa_got - fa_ref).max())))
    r = np.array(out_rows)
    print("%-18s det=%d  WHOLE run, %d runs: loss_n window dev median %.4f max %.4f | constrain median %.4f max %.4f | adv "
          "fraction diff median %.4f max %.4f" % (tag, deterministic, repeats, np.median(r[:, 0]), r[:, 0].max(),
                                                  np.median(r[:, 1]), r[:, 1].max(), np.median(r[:, 2]), r[:, 2].max()), flush=True)


if __name__ == "__main__":
    for _tag, _det in _FULL:
        full_run_window_spread(_tag, _det)
