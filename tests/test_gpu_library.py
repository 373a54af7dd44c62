"""The registered custom ops (geoa3_amd/library.py) under torch.compile: the reference's loss composition and the victim
trace with fullgraph=True (no graph break) and give the eager results, values and gradients."""
import numpy as np
import pytest
import torch

from oracle import geoa3_oracle as O

pytestmark = pytest.mark.gpu


def _inputs(b=3, n=128, seed=5):
    ori, nrm = O.make_synthetic_clouds(b, n, seed)
    g = torch.Generator().manual_seed(seed + 1)
    adv = ori + 0.01 * torch.randn(b, 3, n, generator=g)
    return adv.cuda(), ori.cuda(), nrm.cuda()


def test_compiled_losses_equal_eager_and_have_no_graph_break():
    from geoa3_amd import loss_utils as L
    adv0, ori, nrm = _inputs()
    kap_ori = L._get_kappa_ori(ori, nrm, 8)

    def constrain(adv):
        cd, hd = L.chamfer_loss(adv, ori), L.hausdorff_loss(adv, ori)
        ka, _ = L._get_kappa_adv(adv, ori, nrm, 8)
        return cd + 0.1 * hd + L.curvature_loss(adv, ori, ka, kap_ori, 8)

    res = []
    for fn in (constrain, torch.compile(constrain, fullgraph=True, backend="aot_eager")):
        adv = adv0.clone().requires_grad_()
        val = fn(adv)
        (g,) = torch.autograd.grad(val.sum(), adv)
        res.append((val.detach().cpu().numpy(), g.cpu().numpy()))
    np.testing.assert_array_equal(res[0][0], res[1][0])
    np.testing.assert_array_equal(res[0][1], res[1][1])
    # ... and they are the reference's values (oracle)
    a, o, n_ = adv0.cpu(), ori.cpu(), nrm.cpu()
    ka, _ = O.get_kappa_adv(a, o, n_, 8)
    want = O.chamfer_loss(a, o) + 0.1 * O.hausdorff_loss(a, o) + O.curvature_loss(a, o, ka, O.get_kappa_ori(o, n_, 8))
    np.testing.assert_allclose(res[1][0], want.numpy(), rtol=2e-5, atol=1e-7)


def test_chamfer_loss_compiles_with_inductor_default_backend():
    """torch.compile(chamfer_loss) with the default backend: the custom op is opaque to the compiler, the graph is whole."""
    from geoa3_amd import loss_utils as L
    adv0, ori, _ = _inputs(2, 64, 9)
    explain = torch._dynamo.explain(L.chamfer_loss)(adv0, ori)
    assert explain.graph_break_count == 0, explain.break_reasons
    got = torch.compile(L.chamfer_loss, fullgraph=True)(adv0, ori)
    assert torch.equal(got, L.chamfer_loss(adv0, ori))


def test_knn_points_operator_traces_and_differentiates():
    from geoa3_amd import ops
    adv, ori, _ = _inputs(2, 96, 11)
    p1 = adv.permute(0, 2, 1).contiguous().requires_grad_()
    p2 = ori.permute(0, 2, 1).contiguous().requires_grad_()

    def f(a, b):
        return ops.knn_points(a, b, K=4).dists.sum()

    g_e = torch.autograd.grad(f(p1, p2), (p1, p2))
    g_c = torch.autograd.grad(torch.compile(f, fullgraph=True, backend="aot_eager")(p1, p2), (p1, p2))
    assert torch.equal(g_e[0], g_c[0])
    # (dp2 is a scatter_add of torch: float atomics, free order)
    np.testing.assert_allclose(g_e[1].cpu().numpy(), g_c[1].cpu().numpy(), rtol=1e-5, atol=1e-6)
    # pytorch3d's formula against autograd through the dense form
    a, b = p1.detach().cpu().requires_grad_(), p2.detach().cpu().requires_grad_()
    d, _ = O.knn_points(a, b, 4)
    ga, gb = torch.autograd.grad(d.sum(), (a, b))
    np.testing.assert_allclose(g_e[0].cpu().numpy(), ga.numpy(), rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(g_e[1].cpu().numpy(), gb.numpy(), rtol=1e-5, atol=1e-6)


def test_pointnet_module_under_compile():
    from geoa3_amd.pointnet import PointNet
    net = PointNet(40)
    net.load_state_dict(O.make_pointnet_state_dict(40, seed=0))
    net = net.cuda().eval()
    x0, _, _ = _inputs(3, 256, 21)
    w = torch.randn(3, 40, generator=torch.Generator().manual_seed(3)).cuda()
    out = []
    for fn in (net, torch.compile(net, fullgraph=True, backend="aot_eager")):
        x = x0.clone().requires_grad_()
        lg = fn(x)
        (g,) = torch.autograd.grad((lg * w).sum(), x)
        out.append((lg.detach(), g))
    assert torch.equal(out[0][0], out[1][0]) and torch.equal(out[0][1], out[1][1])


def test_custom_op_backward_after_an_interleaved_eager_forward():
    """geoa3::pointnet_forward(x), then an eager net(y) (which overwrites the workspace), then the custom op's backward:
    the backward must notice that the workspace no longer holds x's activations and recompute them."""
    from geoa3_amd import library
    from geoa3_amd.pointnet import PointNet
    net = PointNet(40)
    net.load_state_dict(O.make_pointnet_state_dict(40, seed=0))
    net = net.cuda().eval()
    x0, _, _ = _inputs(3, 256, 21)
    y, _, _ = _inputs(3, 256, 22)
    w = torch.randn(3, 40, generator=torch.Generator().manual_seed(3)).cuda()
    x = x0.clone().requires_grad_()
    (want,) = torch.autograd.grad((net(x) * w).sum(), x)
    x = x0.clone().requires_grad_()
    lg = torch.ops.geoa3.pointnet_forward(x, library.net_handle(net))
    with torch.no_grad():
        net(y)
    (got,) = torch.autograd.grad((lg * w).sum(), x)
    assert torch.equal(got, want)
