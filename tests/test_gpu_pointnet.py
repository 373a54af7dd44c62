"""HIP PointNet forward / input-gradient against the golden fixtures (reference outputs) and the oracle."""
import numpy as np
import pytest
import torch

from oracle import geoa3_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def net():
    from geoa3_amd.pointnet import PointNet
    n = PointNet(40)
    n.load_state_dict(O.make_pointnet_state_dict(40, seed=0))
    return n.cuda().eval()


@pytest.mark.parametrize("tag", ["n64", "n256", "n1024"])
def test_forward_backward_golden(net, golden, tag):
    pre = "pn/%s/" % tag
    x = T(golden[pre + "pc"]).cuda().requires_grad_()
    logits = net(x)
    # fp32 tolerance: the MFMA fmaf chain sums in a different order than the CPU reference
    np.testing.assert_allclose(logits.detach().cpu().numpy(), golden[pre + "logits"], rtol=1e-4, atol=3e-4)
    (logits * T(golden[pre + "w"]).cuda()).sum().backward()
    ref = golden[pre + "g_pc"]
    np.testing.assert_allclose(x.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max())


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
    ref = xc.grad.numpy()
    np.testing.assert_allclose(xg.grad.cpu().numpy(), ref, rtol=2e-3, atol=2e-4 * np.abs(ref).max())


def test_batch_independence(net):
    """Row k of a batched forward == the batch-1 forward of cloud k, bit for bit: what allows the b separate
    batch-1 success-check forwards of geoA3_attack.py:297 to be read off the one batched forward."""
    pc, _ = O.make_synthetic_clouds(6, 256, seed=5)
    x = pc.cuda()
    with torch.no_grad():
        full = net(x).clone()
        for k in range(6):
            assert torch.equal(net(x[k:k + 1].contiguous())[0], full[k])
