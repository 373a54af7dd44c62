"""The matrix-core filter of the exact 1-NN search (csrc/geom_filter.h: approximate distances of all pairs on the matrix
core, exact evaluation of the pairs under the seed's radius) and the policies around it -- filter, in-kernel sweep, grid
walk, the shipped choice -- against the all-pairs kernel, bit for bit (distances AND indices, both directions, with exact /
junk / no priors, in place): CAD-like clouds, unequal and odd sizes, clouds beyond the grid's 4096 points, queries far from
the cloud, shifted / tiny / huge coordinates, duplicates and exact ties, degenerate clouds, NaN / inf coordinates.
Reference: pytorch3d.ops.knn_points(K=1) at Lib/loss_utils.py:32-33,48,70,92 (index of the first minimum)."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def _tool():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "nn1_filter_check.py")
    spec = importlib.util.spec_from_file_location("nn1_filter_check", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_every_policy_returns_the_all_pairs_bits():
    assert torch.cuda.is_available(), "needs the MI355X"
    assert _tool().check(verbose=False) == 0


def test_filter_agrees_with_the_oracle_beyond_4096_points():
    from geoa3_amd import ops
    from geoa3_amd.data import synthetic_clouds
    from oracle import geoa3_oracle as O
    ori, _ = synthetic_clouds(1, 5000, seed=11)
    adv = ori + 0.05 * torch.randn(1, 3, 5000, generator=torch.Generator().manual_seed(2))
    got = ops.nn1_pair(adv.cuda(), ori.cuda(), method="grid")
    d, i = O.knn_points(adv.permute(0, 2, 1), ori.permute(0, 2, 1), 1)
    assert torch.equal(got[1].cpu().long(), i[:, :, 0]) and torch.equal(got[0].cpu(), d[:, :, 0])
    d, i = O.knn_points(ori.permute(0, 2, 1), adv.permute(0, 2, 1), 1)
    assert torch.equal(got[3].cpu().long(), i[:, :, 0]) and torch.equal(got[2].cpu(), d[:, :, 0])
