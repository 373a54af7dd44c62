"""CPU: the host-side BN folding / packing and the hand-derived sparse input-gradient, evaluated with plain
torch ops (tests/_packed_eval.py), against the oracle forward + autograd and the golden fixtures."""
import numpy as np
import pytest
import torch

from geoa3_amd.pointnet import PointNet, pack_pointnet
from oracle import geoa3_oracle as O
from tests import _packed_eval as PE

T = torch.from_numpy


@pytest.mark.parametrize("tag", ["n64", "n256"])
def test_packed_forward_backward_matches_golden(golden, tag):
    sd = O.make_pointnet_state_dict(40, seed=0)
    p = pack_pointnet(sd)
    pre = "pn/%s/" % tag
    x = T(golden[pre + "pc"])
    logits, S = PE.forward(p, x)
    np.testing.assert_allclose(logits.numpy(), golden[pre + "logits"], rtol=1e-4, atol=2e-4)
    dx = PE.backward(p, x, S, T(golden[pre + "w"]))
    ref = golden[pre + "g_pc"]
    np.testing.assert_allclose(dx.numpy(), ref, rtol=1e-3, atol=1e-4 * np.abs(ref).max())


def test_module_state_dict_layout_matches_reference():
    """Same entry names and shapes as the reference module; the oracle state_dict was loaded
    strictly into the reference's own PointNet when the golden fixtures were generated."""
    sd = O.make_pointnet_state_dict(40, seed=0)
    net = PointNet(40)
    own = net.state_dict()
    assert set(own) == set(sd) and len(own) == 125
    for k in sd:
        assert tuple(own[k].shape) == tuple(sd[k].shape), k
    net.load_state_dict(sd)            # strict
    net.eval()
    with pytest.raises(Exception):     # product path has no CPU forward
        net(torch.zeros(1, 3, 16))
