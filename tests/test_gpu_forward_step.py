"""Op-level golden tests of the loop's head and optimiser kernels on the HIP path:

* the six `fs/*` fixtures are `_forward_step` tuples of the reference's own Python (Attacker/geoA3_attack.py:100-180:
  CE / Margin x targeted / untargeted incl. Margin-targeted with confidence 0.5, L2 distance, single-sided CD), with a
  per-instance `scale_const` vector: logits, cls_loss, the distance losses, loss_n and d loss / d x through
  geoa3_pointnet_forward -> geometry kernels -> geoa3_attack_head_vote -> geoa3_pointnet_backward;
* the `adam/*` fixture is a six-step torch.optim.Adam trace (gradients spanning six decades) replayed through
  geoa3_attack_update (2 ulp of the iterate + 3e-7 of the step).
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from oracle import geoa3_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy

FS_CASES = [
    ("ce_untarget", dict(cls_loss_type="CE", attack_label="Untarget"), False),
    ("ce_target", dict(cls_loss_type="CE", attack_label="All"), True),
    ("margin_target", dict(cls_loss_type="Margin", attack_label="All", confidence=0.5), True),
    ("margin_untarget", dict(cls_loss_type="Margin", attack_label="Untarget"), False),
    ("l2_nohd", dict(dis_loss_type="L2", hd_loss_weight=0.0, curv_loss_weight=0.0), False),
    ("pcd", dict(is_cd_single_side=True), False)]


@pytest.fixture(scope="module")
def net():
    from geoa3_amd.pointnet import PointNet
    n = PointNet(40)
    n.load_state_dict(O.make_pointnet_state_dict(40, seed=0))
    return n.cuda().eval()


@pytest.mark.parametrize("tag,kw,targeted", FS_CASES)
def test_forward_step_golden(net, golden, tag, kw, targeted):
    from geoa3_amd.attack import AttackRunner
    from tests.test_gpu_pointnet import assert_grad_close
    cfg = O.AttackCfg(curv_loss_knn=8, **kw)
    pre = "fs/%s/" % tag
    ori, nrm, x = (T(golden[pre + n]).cuda() for n in ("ori", "nrm", "x"))
    gt, target = T(golden[pre + "gt"]), T(golden[pre + "target"])
    b, _, n = ori.shape
    r = AttackRunner(net, b, n, cfg, torch.device("cuda"))
    r.setup(ori, nrm, gt, target if targeted else gt)
    r.begin_search_step(torch.zeros_like(ori))
    sc = T(golden[pre + "scale_const"]).cuda()
    assert len(set(sc.tolist())) > 1, "the fixture is meant to carry a per-instance scale_const vector"
    r.t["scale_const"].copy_(sc)
    r.t["x"].copy_(x)                      # the reference's iterate, bit for bit
    g_cls, g_geo = r.objective(0, 0)
    torch.cuda.synchronize()
    ref = lambda name: np.asarray(golden[pre + name])
    # logits: the PointNet bar (tests/test_gpu_pointnet.py); everything computed from them inherits it
    np.testing.assert_allclose(r.t["logits"].cpu().numpy(), ref("logits"), rtol=1e-4, atol=3e-4)
    np.testing.assert_allclose(r.t["cls_loss"].cpu().numpy(), ref("cls_loss"), rtol=1e-4, atol=3e-4)
    np.testing.assert_allclose(r.geo_out["dis_loss"].cpu().numpy(), ref("dis_loss"), rtol=2e-5, atol=1e-7)
    if cfg.hd_loss_weight != 0:
        np.testing.assert_allclose(r.geo_out["hd_loss"].cpu().numpy(), ref("hd_loss"), rtol=2e-5, atol=1e-7)
    if cfg.curv_loss_weight != 0:
        np.testing.assert_allclose(r.geo_out["curv_loss"].cpu().numpy(), ref("curv_loss"), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(r.geo_out["constrain"].cpu().numpy(), ref("constrain"), rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(r.t["loss_n"].cpu().numpy(), ref("loss_n"), rtol=1e-4, atol=3e-4)
    np.testing.assert_allclose(float(r.t["loss_n"].mean()), float(ref("loss")), rtol=1e-4, atol=3e-4)
    # d mean(loss_n) / d x = g_cls (carries 1/b) + scale_const / b * d constrain / d x  (geoA3_attack.py:176-178)
    g = g_cls + (sc / b).view(b, 1, 1) * g_geo
    assert_grad_close(g.cpu().numpy(), ref("g_x"))
    # the label the success check reads (geoA3_attack.py:298) is the arg-max of the same logits
    assert r.t["label"].cpu().tolist() == ref("logits").argmax(1).tolist()


def test_adam_trace_golden(golden):
    """geoa3_attack_update against torch.optim.Adam's own iterates (geoA3_attack.py:269-275,326-328)."""
    from geoa3_amd import _lib
    lib = _lib.load()
    ps, gs = golden["adam/params"], golden["adam/grads"]
    steps, b, _, n = gs.shape
    dev = torch.device("cuda")
    offset = T(ps[0]).to(dev).contiguous()
    m, v, x = torch.zeros_like(offset), torch.zeros_like(offset), torch.zeros_like(offset)
    ori = torch.zeros_like(offset)
    sc = torch.ones(b, device=dev)
    st = _lib.AttackState(B=b, N=n, classes=40, inv_global_batch=1.0 / b, scale_const=sc.data_ptr())
    s = torch.cuda.current_stream().cuda_stream
    mags = np.abs(gs[gs != 0])
    assert mags.max() / mags.min() > 1e5, "the trace is meant to span decades of gradient magnitude"
    for t in range(steps):
        g = T(gs[t]).to(dev).contiguous()
        offset.copy_(T(ps[t]))              # every step starts from the reference's iterate (m, v carry on)
        tt = t + 1
        _lib.check(lib.geoa3_attack_update(C.byref(st), g.data_ptr(), None, ori.data_ptr(), offset.data_ptr(),
                                           m.data_ptr(), v.data_ptr(), x.data_ptr(), 0, 0.01 / (1.0 - 0.9 ** tt),
                                           math.sqrt(1.0 - 0.999 ** tt), 0.0, s), "attack_update")
        got, want = offset.cpu().numpy(), ps[t + 1]
        # the reference's iterate within 2 ulp of itself + 3e-7 of the step taken: torch's CPU kernels (lerp via fmadd,
        # addcmul, addcdiv as (value * m) / denom) and hipcc's contraction of the same expressions round m, v and the
        # ~1e-2 step differently in their last bit
        ulp = np.spacing(np.abs(want).astype(np.float32)).astype(np.float64)
        bound = 2 * ulp + 3e-7 * np.abs(want.astype(np.float64) - ps[t].astype(np.float64))
        err = np.abs(got.astype(np.float64) - want.astype(np.float64))
        assert (err <= bound).all(), (t, float((err / bound).max()))
        assert torch.equal(x, offset)       # x = ori + offset with ori = 0
