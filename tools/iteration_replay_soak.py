"""One attack iteration replayed N times from the SAME device state; after every replay every tensor the iteration
writes (logits, losses, neighbour tables, both gradient parts, Adam state, the iterate, and every buffer of the victim's
workspace) is compared bit for bit with the first replay.  A kernel whose result depends on timing shows up by the name
of the buffer it writes (NOTEBOOK 5a).  Used by tests/test_gpu_replay.py; stand-alone:

    python tools/iteration_replay_soak.py [--iters 3000] [--b 250] [--n 1024] [--k 16] [--arch PointNet|PointNetPP]
                                          [--presteps 20] [--mode f16x2|f32]

Prints one JSON line: {"iters": .., "differing_replays": .., "by_buffer": {name: replays that differ}}.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, REPO)

import torch  # noqa: E402


def workspace_regions(lib, b, n, classes, ws):
    """[(name, uint8 view)] of the PointNet workspace (geoa3_debug_pointnet_workspace_layout)."""
    names = (C.c_char_p * 64)()
    offs = (C.c_int64 * 64)()
    cnt = lib.geoa3_debug_pointnet_workspace_layout(b, n, classes, names, offs, 64)
    order = sorted(range(cnt), key=lambda i: offs[i])
    out = []
    for j, i in enumerate(order):
        end = offs[order[j + 1]] if j + 1 < cnt else ws.numel()
        out.append((names[i].decode(), ws[offs[i]:end]))
    return out


def replay(runner, iters, step_no, extra=None, progress=None):
    """Replays runner.step(step_no, 0) `iters` times from the runner's present state.  Returns (replays that differ from
    the first, {buffer name: count})."""
    r = runner
    tensors = {}
    for k, v in r.t.items():
        if isinstance(v, (list, tuple)):
            for i, x in enumerate(v):
                tensors["%s[%d]" % (k, i)] = x
        elif torch.is_tensor(v):
            tensors[k] = v
    for k, v in r.geo_out.items():
        tensors.setdefault("geo_out." + k, v)
    # aliases (t["proj_d"] is t["d_ao"], geo_out["grad"] is t["g_geo"]): keep one name per storage
    seen, uniq = set(), {}
    for k, v in tensors.items():
        key = (v.data_ptr(), v.numel())
        if key not in seen:
            seen.add(key)
            uniq[k] = v
    tensors = uniq
    start = {k: v.clone() for k, v in tensors.items()}
    attrs = {a: getattr(r, a) for a in ("knn_cur", "knn_seeded", "nn1_seeded", "_ws_fwd_shape") if hasattr(r, a)}
    regions = []
    if getattr(r, "native", False) and not getattr(r, "ssg", False):
        regions = workspace_regions(r.lib, r.b, r.ne, r.classes, r.ws)
    elif getattr(r, "native", False):
        regions = [("pn2_workspace", r.ws)]
    ref, ref_ws = None, None
    by, differing = {}, 0
    for it in range(iters):
        for k, v in tensors.items():
            v.copy_(start[k])
        for a, val in attrs.items():
            setattr(r, a, val)
        r.step(step_no, 0)
        torch.cuda.synchronize()
        if ref is None:
            ref = {k: v.clone() for k, v in tensors.items()}
            ref_ws = r.ws.clone() if regions else None
            continue
        bad = [k for k, v in tensors.items() if not torch.equal(v, ref[k])]
        if regions and not torch.equal(r.ws, ref_ws):
            off = 0
            for name, view in regions:
                o = view.data_ptr() - r.ws.data_ptr()
                if not torch.equal(view, ref_ws[o:o + view.numel()]):
                    bad.append("ws." + name)
        if bad:
            differing += 1
            for k in bad:
                by[k] = by.get(k, 0) + 1
        if progress and (it + 1) % progress == 0:
            print("replay %d: %d differ so far %s" % (it + 1, differing, by), file=sys.stderr, flush=True)
    return differing, by


def make_runner(b, n, k, arch="PointNet", mode=None, presteps=20, seed=2024, data="ellipsoid"):
    import bench
    from geoa3_amd.attack import AttackRunner
    from geoa3_amd.data import SYNTHETIC_GENERATORS, synthetic_state_dict
    from geoa3_amd.pointnet import PointNet
    dev = torch.device("cuda")
    ori, nrm = SYNTHETIC_GENERATORS[data](b, n, seed=seed)   # "cad": long reverse-list rows, full grid cells, duplicates
    ori, nrm = ori.to(dev), nrm.to(dev)
    if arch == "PointNet":
        net = PointNet(40)
        net.load_state_dict(synthetic_state_dict(40, seed=0))
        net = net.to(dev).eval()
        if mode:
            net.wide_mode = mode
    else:
        from geoa3_amd.pointnet2 import PointNet2ClassificationSSG
        torch.manual_seed(0)
        net = PointNet2ClassificationSSG(use_xyz=True, use_normal=False).to(dev).eval()
    with torch.no_grad():
        gt = net(ori).argmax(1)
    cfg = bench.cfg_full_geoa3(presteps + 8, n, k)
    r = AttackRunner(net, b, n, cfg, dev)
    r.setup(ori, nrm, gt, gt)
    init = (torch.randn(b, 3, n, generator=torch.Generator().manual_seed(11)) * 1e-3).to(dev)
    r.begin_search_step(init)
    for s in range(presteps):
        r.step(s, 0)
    torch.cuda.synchronize()
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=3000)
    ap.add_argument("--b", type=int, default=250)
    ap.add_argument("--n", type=int, default=1024)
    ap.add_argument("--k", type=int, default=16)
    ap.add_argument("--arch", default="PointNet")
    ap.add_argument("--mode", default=None)
    ap.add_argument("--presteps", type=int, default=20)
    ap.add_argument("--data", default="ellipsoid", choices=["ellipsoid", "cad"])
    a = ap.parse_args()
    r = make_runner(a.b, a.n, a.k, a.arch, a.mode, a.presteps, data=a.data)
    differing, by = replay(r, a.iters, a.presteps, progress=1000)
    print(json.dumps({"iters": a.iters, "b": a.b, "n": a.n, "k": a.k, "arch": a.arch, "mode": a.mode,
                      "lib": os.environ.get("GEOA3_LIB_PATH", "product"), "differing_replays": differing,
                      "by_buffer": by}))


if __name__ == "__main__":
    main()
