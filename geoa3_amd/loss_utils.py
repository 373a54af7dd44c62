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

from . import ops

Tensor = torch.Tensor


def _planar(x: Tensor) -> Tensor:
    return x.detach().contiguous().float()


class _PointLoss(torch.autograd.Function):
    """loss[b] (one of CD / pseudo-CD / HD / L2) with d loss[b] / d adv from the fused kernel."""

    @staticmethod
    def forward(ctx, adv_pc, ori_pc, kind):
        adv, ori = _planar(adv_pc), _planar(ori_pc)
        kw = dict(dis_type=0, w_dis=0.0, w_hd=0.0)
        if kind in ("cd", "pcd", "hd"):
            d_ao, i_ao, d_oa, i_oa = ops.nn1_pair(adv, ori, both=(kind == "cd"))
            kw.update(d_ao=d_ao, i_ao=i_ao, d_oa=d_oa, i_oa=i_oa)
        if kind == "cd":
            kw.update(dis_type=1, w_dis=1.0)
        elif kind == "pcd":
            kw.update(dis_type=1, w_dis=1.0, single_side=True)
        elif kind == "hd":
            kw.update(w_hd=1.0)
        else:
            kw.update(dis_type=2, w_dis=1.0)
        out = ops.geo_loss_grad(adv, ori, **kw)
        ctx.save_for_backward(out["grad"])
        return out["hd_loss"].clone() if kind == "hd" else out["dis_loss"].clone()

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return grad * g.view(-1, 1, 1), None, None


def norm_l2_loss(adv_pc: Tensor, ori_pc: Tensor) -> Tensor:
    return _PointLoss.apply(adv_pc, ori_pc, "l2")


def chamfer_loss(adv_pc: Tensor, ori_pc: Tensor) -> Tensor:
    return _PointLoss.apply(adv_pc, ori_pc, "cd")


def pseudo_chamfer_loss(adv_pc: Tensor, ori_pc: Tensor) -> Tensor:
    return _PointLoss.apply(adv_pc, ori_pc, "pcd")


def hausdorff_loss(adv_pc: Tensor, ori_pc: Tensor) -> Tensor:
    return _PointLoss.apply(adv_pc, ori_pc, "hd")


def _get_kappa_ori(pc: Tensor, normal: Tensor, k: int = 2) -> Tensor:
    pc, normal = _planar(pc), _planar(normal)
    _, idx = ops.knn_planar(pc, pc, k + 1)
    return ops.kappa(pc, normal, idx)


class _KappaAdv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, adv_pc, ori_pc, ori_normal, k):
        adv, ori, nrm = _planar(adv_pc), _planar(ori_pc), _planar(ori_normal)
        _, i_ao, _, _ = ops.nn1_pair(adv, ori, both=False)
        _, knn_adv = ops.knn_planar(adv, adv, k + 1)
        kap = ops.kappa(adv, nrm, knn_adv, i_ao)
        b, _, n = adv.shape
        normal = torch.gather(nrm, 2, i_ao.long().unsqueeze(1).expand(b, 3, n)).contiguous()
        ctx.save_for_backward(adv, ori, nrm, i_ao, knn_adv)
        ctx.k = k
        ctx.mark_non_differentiable(normal)
        return kap, normal

    @staticmethod
    def backward(ctx, gk, _gn):
        adv, ori, nrm, i_ao, knn_adv = ctx.saved_tensors
        out = ops.geo_loss_grad(adv, ori, normal_ori=nrm, i_ao=i_ao, knn_adv=knn_adv, dkappa=gk.contiguous().float(),
                                k=ctx.k, dis_type=0, w_dis=0.0)
        return out["grad"], None, None, None


def _get_kappa_adv(adv_pc: Tensor, ori_pc: Tensor, ori_normal: Tensor, k: int = 2):
    """-> (kappa_adv [b,n] (differentiable w.r.t. adv_pc), normal [b,3,n])"""
    return _KappaAdv.apply(adv_pc, ori_pc, ori_normal, k)


def curvature_loss(adv_pc: Tensor, ori_pc: Tensor, adv_kappa: Tensor, ori_kappa: Tensor, k: int = 2) -> Tensor:
    _, i_ao, _, _ = ops.nn1_pair(_planar(adv_pc), _planar(ori_pc), both=False)
    onenn_ori_kappa = torch.gather(ori_kappa, 1, i_ao.long())
    return ((adv_kappa - onenn_ori_kappa) ** 2).mean(-1)
