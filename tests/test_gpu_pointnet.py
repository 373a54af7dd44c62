"""HIP PointNet forward / input-gradient against the golden fixtures (reference outputs) and the oracle."""
import numpy as np
import pytest
import torch

from oracle import geoa3_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module", params=["f32", "f16x2"])
def net(request):
    """Both arithmetic modes of the 1024-wide layers (fp32 MFMA; split-fp16 operands with fp32 accumulation) must
    meet the same bars."""
    from geoa3_amd.pointnet import PointNet
    n = PointNet(40)
    n.load_state_dict(O.make_pointnet_state_dict(40, seed=0))
    n.wide_mode = request.param
    return n.cuda().eval()


def assert_grad_close(got, ref):
    """Input gradient: rtol 2e-3 / atol 2e-4 of the largest entry.  A channel whose two best points are closer than
    the summation-order noise (~1e-7 relative) may take its arg-max at the other point -- the gradient of that one
    channel then lands on another point; at most 0.05 % of the entries may differ for that reason (each such entry
    still within 1 % of the largest gradient)."""
    scale = np.abs(ref).max()
    bad = np.abs(got - ref) > 2e-3 * np.abs(ref) + 2e-4 * scale
    assert bad.mean() <= 5e-4, "%d of %d gradient entries differ" % (bad.sum(), bad.size)
    assert np.abs(got - ref).max() <= 1e-2 * scale


@pytest.mark.parametrize("tag", ["n64", "n256", "n1024"])
def test_forward_backward_golden(net, golden, tag):
    pre = "pn/%s/" % tag
    x = T(golden[pre + "pc"]).cuda().requires_grad_()
    logits = net(x)
    # fp32 tolerance: the MFMA fmaf chain sums in a different order than the CPU reference
    np.testing.assert_allclose(logits.detach().cpu().numpy(), golden[pre + "logits"], rtol=1e-4, atol=3e-4)
    (logits * T(golden[pre + "w"]).cuda()).sum().backward()
    assert_grad_close(x.grad.cpu().numpy(), golden[pre + "g_pc"])


@pytest.mark.parametrize("B,N", [(5, 200), (3, 1000), (33, 128), (2, 2048)])
def test_forward_backward_oracle_ragged(net, B, N):
    """N not a multiple of the 128-point tiles, B not a multiple of the 32-row FC tiles."""
    sd = O.make_pointnet_state_dict(40, seed=0)
    pc, _ = O.make_synthetic_clouds(B, N, seed=B * 1000 + N)
    xc = pc.clone().requires_grad_()
    lo = O.pointnet_forward(sd, xc)
    w = torch.randn(B, 40, generator=torch.Generator().manual_seed(1))
    (lo * w).sum().backward()
    xg = pc.cuda().requires_grad_()
    lg = net(xg)
    np.testing.assert_allclose(lg.detach().cpu().numpy(), lo.detach().numpy(), rtol=1e-4, atol=3e-4)
    (lg * w.cuda()).sum().backward()
    assert_grad_close(xg.grad.cpu().numpy(), xc.grad.numpy())


def test_batch_independence(net):
    """Row k of a batched forward == the batch-1 forward of cloud k, bit for bit: what allows the b separate
    batch-1 success-check forwards of geoA3_attack.py:297 to be read off the one batched forward."""
    pc, _ = O.make_synthetic_clouds(6, 256, seed=5)
    x = pc.cuda()
    with torch.no_grad():
        full = net(x).clone()
        for k in range(6):
            assert torch.equal(net(x[k:k + 1].contiguous())[0], full[k])


@pytest.mark.parametrize("scale", [1e-3, 1.0, 300.0])
def test_wide_split_operand_range(scale):
    """f16x2 mode carries every fp32 operand of the 1024-wide layers as two fp16 values (hi, 2^11 * lo): activations
    of very different magnitude going into conv5 / the T-Nets' conv3 (here through the BatchNorm scale of the layer in
    front of them) must keep fp32-level agreement with the oracle; beyond fp16's range (65504) the result must be
    loud (NaN), never silently wrong."""
    from geoa3_amd.pointnet import PointNet
    sd = O.make_pointnet_state_dict(40, seed=2)
    for name in ("bn4", "input_transform.bn2", "feature_transform.bn2"):
        sd[name + ".weight"] = sd[name + ".weight"] * scale
        sd[name + ".bias"] = sd[name + ".bias"] * scale
    pc, _ = O.make_synthetic_clouds(4, 300, seed=8)
    lo = O.pointnet_forward(sd, pc)
    outs = {}
    for mode in ("f32", "f16x2"):
        n = PointNet(40)
        n.load_state_dict(sd)
        n.wide_mode = mode
        with torch.no_grad():
            outs[mode] = n.cuda().eval()(pc.cuda()).cpu()
        np.testing.assert_allclose(outs[mode].numpy(), lo.numpy(), rtol=1e-4, atol=1e-4 * float(lo.abs().max()))
    # out of range: the hi part overflows to inf and the logits are NaN
    for name in ("bn4",):
        sd[name + ".weight"] = sd[name + ".weight"] * 1e6
    n = PointNet(40)
    n.load_state_dict(sd)
    n.wide_mode = "f16x2"
    with torch.no_grad():
        assert not torch.isfinite(n.cuda().eval()(pc.cuda())).all()
