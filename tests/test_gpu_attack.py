"""The device-resident attack loop (geoa3_amd.attack) against the reference trajectories stored in the golden
fixtures (produced by the reference's own attack()) and against the oracle on a fresh case."""
import os

import numpy as np
import pytest
import torch

from oracle import geoa3_oracle as O
from tests.golden.make_golden import ATK_CASES
from tests.test_oracle_golden import _traj_close

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def net():
    from geoa3_amd.pointnet import PointNet
    n = PointNet(40)
    n.load_state_dict(O.make_pointnet_state_dict(40, seed=0))
    return n.cuda().eval()


def _loader_batch(ori, nrm, gt, tgt, targeted):
    b = ori.shape[0]
    data = [ori.permute(0, 2, 1).unsqueeze(1).contiguous(), nrm.permute(0, 2, 1).unsqueeze(1).contiguous(),
            gt.view(b, 1)]
    if targeted:
        data.append(tgt.view(b, 1))
    return data


def _run(net, cfg, ori, nrm, gt, tgt, targeted, inits, **kw):
    from geoa3_amd.attack import AttackRunner, unpack_input
    pc, nm, g, t = unpack_input(_loader_batch(ori, nrm, gt, tgt, targeted), targeted)
    r = AttackRunner(net, pc.shape[0], pc.shape[2], cfg, torch.device("cuda"), kw.get("global_batch"))
    r.setup(pc, nm, g, t)
    xs, labels = [], []

    def grab(search_step, step):
        labels.append(r.t["label"].cpu().tolist())

    # iterate of step s = ori + offset BEFORE the update of step s: record at begin and after each step
    orig_begin = r.begin_search_step

    def begin(init):
        orig_begin(init)
        xs.append(r.t["x"].cpu().clone())

    r.begin_search_step = begin

    def on_step(search_step, step):
        grab(search_step, step)
        if step < r.iters - 1:
            xs.append(r.t["x"].cpu().clone())

    r.run([i.cuda() for i in inits], on_step=on_step, sync_last_label=kw.get("sync_last_label"))
    return r, r.results(), torch.stack(xs).numpy(), np.asarray(labels)


@pytest.mark.parametrize("tag", list(ATK_CASES))
def test_attack_matches_reference_trajectory(net, golden, tag):
    kw, targeted, _, _ = ATK_CASES[tag]
    cfg = O.AttackCfg(**kw)
    pre = "atk/%s/" % tag
    ori, nrm, gt, tgt = (T(golden[pre + n]) for n in ("ori", "nrm", "gt", "tgt"))
    inits = [T(a) for a in golden[pre + "inits"]]
    r, (best, target, succ, best_step, all_loss), xs, labels = _run(net, cfg, ori, nrm, gt, tgt, targeted, inits)
    ref_x = golden[pre + "tr_x"]
    # stated fp32 tolerance for short trajectories: >= 99.5 % of coordinates within 2e-5, all within 2e-3
    # (Adam turns a 1e-9 difference on a near-zero gradient into a +-lr step; SURVEY 7 'hard parts')
    # an Adam step is at most ~lr per coordinate, so a sign flip on a near-zero gradient can drift 2*lr per step
    loose = max(2e-3, 2.0 * cfg.lr * cfg.iter_max_steps)
    _traj_close(xs, ref_x, loose=loose)
    ref_logits = golden[pre + "tr_logits"]
    ref_labels = ref_logits.argmax(-1)
    # a label may differ only where the reference's own top-2 logits are closer than the fp32 forward tolerance
    # (3e-4, test_gpu_pointnet.py); instances touched by such a near-tie are excluded from the exact checks
    top2 = np.sort(ref_logits, axis=-1)[..., -2:]
    near_tie = (top2[..., 1] - top2[..., 0]) < 3e-4
    assert (labels == ref_labels)[~near_tie].all()
    clean = ~((labels != ref_labels).any(axis=0))
    assert clean.mean() >= 0.5
    assert (np.asarray(succ) == golden[pre + "success"])[clean].all()
    assert (np.asarray(best_step) == golden[pre + "best_step"])[clean].all()
    _traj_close(best.cpu().numpy()[clean], golden[pre + "best_attack"][clean], loose=loose)
    assert (target.cpu().numpy() == golden[pre + "target"]).all()
    np.testing.assert_allclose(np.asarray(all_loss, dtype=np.float32), golden[pre + "all_loss"], rtol=2e-3, atol=2e-4)


def test_attack_api_matches_oracle_b16_n256(net):
    """The public attack() entry point on a case the fixtures do not hold (b=16, N=256, k=16, 2x12 steps)."""
    from geoa3_amd.attack import attack
    cfg = O.AttackCfg(binary_max_steps=2, iter_max_steps=12, lr=0.002, initial_const=50.0, curv_loss_knn=16)
    sd = O.make_pointnet_state_dict(40, seed=0)
    onet = lambda x: O.pointnet_forward(sd, x)
    ori, nrm = O.make_synthetic_clouds(16, 256, seed=77)
    with torch.no_grad():
        gt = onet(ori).argmax(1)
    g = torch.Generator().manual_seed(78)
    inits = [torch.randn(16, 3, 256, generator=g) * 1e-3 for _ in range(2)]
    tr = {}
    ob, ot, osucc, ostep, oloss = O.attack(onet, ori, nrm, gt, None, cfg, inits, trace=tr)
    best, target, succ, best_step, all_loss = attack(net, _loader_batch(ori, nrm, gt, None, False), cfg, 0, 1,
                                                     init_offsets=[i.cuda() for i in inits], verbose=False)
    assert best.is_cuda and best.shape == (16, 3, 256) and target.dtype == torch.int64
    assert isinstance(best_step, list) and len(all_loss) == 12 and len(all_loss[0]) == 16
    agree = (np.asarray(succ) == osucc)
    assert agree.mean() >= 0.9, agree          # success flags (a label flip can land one step apart)
    np.testing.assert_allclose(np.asarray(all_loss, dtype=np.float32), np.asarray(oloss, dtype=np.float32),
                               rtol=5e-3, atol=5e-4)
    both = agree & osucc
    if both.any():
        _traj_close(best.cpu().numpy()[both], ob.numpy()[both], tight=5e-5, frac=0.99, loose=5e-3)


def test_shard_invariance_on_device(net, golden):
    """Two half-batches with the global divisor and the global last label == the full batch."""
    kw, targeted, _, _ = ATK_CASES["untarget_mixed"]
    cfg = O.AttackCfg(**kw)
    pre = "atk/untarget_mixed/"
    ori, nrm, gt, tgt = (T(golden[pre + n]) for n in ("ori", "nrm", "gt", "tgt"))
    inits = [T(a) for a in golden[pre + "inits"]]
    rf, full, _, labels = _run(net, cfg, ori, nrm, gt, tgt, targeted, inits)
    iters = cfg.iter_max_steps
    last = [int(labels[(s + 1) * iters - 1][-1]) for s in range(cfg.binary_max_steps)]
    assert rf.deterministic    # the default: the objective's gradient is summed in a fixed order (geoa3_geo_args)
    for rep in range(10):      # ... so this holds on every repetition, not just when the atomics happen to line up
        for lo, hi in [(0, 3), (3, 6)]:
            state = {"s": 0}

            def sync(tensor):
                tensor.fill_(last[state["s"]])
                state["s"] += 1

            _, part, _, _ = _run(net, cfg, ori[lo:hi], nrm[lo:hi], gt[lo:hi], tgt[lo:hi], targeted,
                                 [i[lo:hi] for i in inits], global_batch=6, sync_last_label=sync)
            assert torch.equal(part[0].cpu(), full[0][lo:hi].cpu())          # bit-identical shards
            assert (part[2] == full[2][lo:hi]).all() and list(part[3]) == list(full[3][lo:hi])


@pytest.mark.timeout(1500)
def test_config5_shape_n4096_k32(net):
    """BASELINE configs[4]: N = 4096 points, curv_loss_knn = 32 (cell-grid K-NN with K = 33, the fixed-point objective kernel),
    32 iterations against the oracle with the short-trajectory bars: iterates (>= 98 % of the 786 K coordinates within 5e-5
    after 32 steps -- observed 98.98 % --, all within 2 lr per step), labels, per-step losses, success flags."""
    steps, lr = 32, 0.002
    cfg = O.AttackCfg(binary_max_steps=1, iter_max_steps=steps, lr=lr, curv_loss_knn=32, initial_const=200.0)
    sd = O.make_pointnet_state_dict(40, seed=0)
    onet = lambda x: O.pointnet_forward(sd, x)
    ori, nrm = O.make_synthetic_clouds(2, 4096, seed=91)
    torch.set_num_threads(min(32, os.cpu_count() or 1))    # (torch's intra-op pool stops scaling long before 256 threads)
    with torch.no_grad():
        gt = onet(ori).argmax(1)
    inits = [torch.randn(2, 3, 4096, generator=torch.Generator().manual_seed(92)) * 1e-3]
    tr = {}
    _, _, osucc, _, oloss = O.attack(onet, ori, nrm, gt, None, cfg, inits, trace=tr)
    r, (best, target, succ, best_step, all_loss), xs, labels = _run(net, cfg, ori, nrm, gt, gt, False, inits)
    assert r.geo_scratch is not None                       # the two-kernel objective of the big clouds is the one in use
    ref_x = torch.stack([ori + o for o in tr["offsets"]]).numpy()
    _traj_close(xs, ref_x, tight=5e-5, frac=0.98, loose=2.0 * lr * steps)
    assert (labels == np.asarray(tr["labels"])).mean() >= 0.95
    np.testing.assert_allclose(np.asarray(all_loss, dtype=np.float32)[:8], np.asarray(oloss, dtype=np.float32)[:8],
                               rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(np.asarray(all_loss, dtype=np.float32), np.asarray(oloss, dtype=np.float32), rtol=2e-2, atol=2e-3)
    assert (np.asarray(succ) == osucc).all()


def test_pointnetpp_attack_matches_oracle():
    """BASELINE configs[3] shape: PointNet++ SSG victim (HIP FPS / ball query / grouping, torch MLPs) driven through
    the same device-resident loop; compared with the oracle loop over the oracle SSG forward."""
    from geoa3_amd.attack import attack
    from geoa3_amd.pointnet2 import PointNet2ClassificationSSG
    from oracle import pointnet2_oracle as P2
    sd = P2.make_pn2_state_dict(0)
    net2 = PointNet2ClassificationSSG(use_xyz=True, use_normal=False)
    net2.load_state_dict(sd)
    net2 = net2.cuda().eval()
    onet = lambda x: P2.pointnet2_ssg_forward(sd, x)
    ori, nrm = O.make_synthetic_clouds(3, 640, seed=55)
    with torch.no_grad():
        gt = onet(ori).argmax(1)
    cfg = O.AttackCfg(binary_max_steps=2, iter_max_steps=4, lr=0.002, curv_loss_knn=8, initial_const=20.0)
    inits = [torch.randn(3, 3, 640, generator=torch.Generator().manual_seed(56 + i)) * 1e-3 for i in range(2)]
    ob, ot, osucc, ostep, oloss = O.attack(onet, ori, nrm, gt, None, cfg, inits)
    best, target, succ, best_step, all_loss = attack(net2, _loader_batch(ori, nrm, gt, None, False), cfg, 0, 1,
                                                     init_offsets=[i.cuda() for i in inits], verbose=False)
    np.testing.assert_allclose(np.asarray(all_loss, dtype=np.float32), np.asarray(oloss, dtype=np.float32),
                               rtol=5e-3, atol=2e-3)
    assert (np.asarray(succ) == osucc).mean() >= 0.66


def test_late_join_equals_early_join(net, golden, monkeypatch):
    """The split head (geoa3_attack_head_classify + _finish, geometry joined after the victim's backward: the small-shard
    schedule) and the single head (geometry joined before it) give the same bits."""
    kw, targeted, _, _ = ATK_CASES["target_full"]
    cfg = O.AttackCfg(**kw)
    pre = "atk/target_full/"
    ori, nrm, gt, tgt = (T(golden[pre + n]) for n in ("ori", "nrm", "gt", "tgt"))
    inits = [T(a) for a in golden[pre + "inits"]]
    res = {}
    for mode in ("1", "0"):
        cfg.late_join = mode == "1"
        r, out, xs, labels = _run(net, cfg, ori, nrm, gt, tgt, targeted, inits)
        assert r.late_join == (mode == "1")
        res[mode] = (out, xs, labels, r.t["loss_hist"].cpu().clone())
    assert np.array_equal(res["1"][1], res["0"][1]) and np.array_equal(res["1"][2], res["0"][2])
    assert torch.equal(res["1"][0][0], res["0"][0][0]) and torch.equal(res["1"][3], res["0"][3])
    assert (res["1"][0][2] == res["0"][0][2]).all() and list(res["1"][0][3]) == list(res["0"][0][3])
