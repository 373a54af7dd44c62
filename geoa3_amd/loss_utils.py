"""Module-level mirror of the reference's Lib/loss_utils.py: the same function names and
[b,3,n] -> [b] signatures, differentiable w.r.t. the adversarial cloud, evaluated by the HIP library
(geoa3_nn1_pair / geoa3_knn / geoa3_kappa / geoa3_geo_loss_grad).  The device-resident attack loop
(geoa3_amd/attack.py) calls the fused kernel directly; this module is for callers that keep the reference's
_forward_step composition (Attacker/geoA3_attack.py:131-166) and swap only the import.

    Lib/loss_utils.py:25-26   norm_l2_loss          Lib/loss_utils.py:52-62   _get_kappa_ori
    Lib/loss_utils.py:28-35   chamfer_loss          Lib/loss_utils.py:64-82   _get_kappa_adv
    Lib/loss_utils.py:37-43   pseudo_chamfer_loss   Lib/loss_utils.py:84-97   curvature_loss
    Lib/loss_utils.py:45-50   hausdorff_loss
"""
from __future__ import annotations

import torch

from . import library, ops  # noqa: F401  (library registers the geoa3:: custom ops)

Tensor = torch.Tensor


def _planar(x: Tensor) -> Tensor:
    return x.detach().contiguous().float()


# The losses are the registered custom ops of geoa3_amd/library.py (schema + fake kernel + autograd formula): the same
# kernels as before, and traceable -- `torch.compile(chamfer_loss, fullgraph=True)` has no graph break.
def norm_l2_loss(adv_pc: Tensor, ori_pc: Tensor) -> Tensor:
    return torch.ops.geoa3.point_loss(adv_pc, ori_pc, 3)[0]


def chamfer_loss(adv_pc: Tensor, ori_pc: Tensor) -> Tensor:
    return torch.ops.geoa3.point_loss(adv_pc, ori_pc, 0)[0]


def pseudo_chamfer_loss(adv_pc: Tensor, ori_pc: Tensor) -> Tensor:
    return torch.ops.geoa3.point_loss(adv_pc, ori_pc, 1)[0]


def hausdorff_loss(adv_pc: Tensor, ori_pc: Tensor) -> Tensor:
    return torch.ops.geoa3.point_loss(adv_pc, ori_pc, 2)[0]


def _get_kappa_ori(pc: Tensor, normal: Tensor, k: int = 2) -> Tensor:
    _, idx = torch.ops.geoa3.knn(pc, pc, k + 1)
    return torch.ops.geoa3.kappa(pc, normal, idx, None)


def _get_kappa_adv(adv_pc: Tensor, ori_pc: Tensor, ori_normal: Tensor, k: int = 2):
    """-> (kappa_adv [b,n] (differentiable w.r.t. adv_pc), normal [b,3,n])"""
    kap, normal, _, _ = torch.ops.geoa3.kappa_adv(adv_pc, ori_pc, ori_normal, int(k))
    return kap, normal.detach()     # (no gradient flows through the nearest-point index, Lib/loss_utils.py:70-72)


def curvature_loss(adv_pc: Tensor, ori_pc: Tensor, adv_kappa: Tensor, ori_kappa: Tensor, k: int = 2) -> Tensor:
    _, i_ao, _, _ = torch.ops.geoa3.nn1_pair(adv_pc, ori_pc, False)
    onenn_ori_kappa = torch.gather(ori_kappa, 1, i_ao.long())
    return ((adv_kappa - onenn_ori_kappa) ** 2).mean(-1)
