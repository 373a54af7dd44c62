"""Device mirrors of the reference's Lib/utility.py helpers on the dense-cloud / defence / measurement paths
(SURVEY.md 8f-3, 8f-4), same names and argument meaning, running in libgeoa3_hip.so (geom_aux.hip).  The
reference's random draws stay the caller's: every function takes them as optional tensors and otherwise draws
them with torch's device generator, exactly where the reference calls torch.randint / randn / randperm.
No CPU path: CPU tensors are rejected."""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check
from .ops import _p, _stream, knn_planar

Tensor = torch.Tensor


def _compare(output, target, gt, targeted):
    """Lib/utility.py:151-155."""
    return output == target if targeted else output != gt


def fps_indices(pc: Tensor, num_points: int, start: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """pc [b,3,n] -> (points [b,3,m], idx int32 [b,m]); start [b]: first index (default: torch.randint as
    Lib/utility.py:179)."""
    b, _, n = pc.shape
    if start is None:
        start = torch.randint(n, (b,), device=pc.device)
    start = start.to(device=pc.device, dtype=torch.int32).contiguous()
    pc = pc.contiguous()
    idx = torch.empty(b, num_points, device=pc.device, dtype=torch.int32)
    pts = torch.empty(b, 3, num_points, device=pc.device, dtype=torch.float32)
    check(_lib.load().geoa3_fps_sample(_p(pc, torch.float32), b, n, num_points, _p(start), _p(idx), _p(pts),
                                       _stream()), "geoa3_fps_sample")
    return pts, idx


class _GatherPlanar(torch.autograd.Function):
    """x [b,3,n], idx int32 [b,m] -> x[:, :, idx]; backward scatter-adds (torch.gather's autograd, utility.py:185)."""

    @staticmethod
    def forward(ctx, x, idx):
        b, c, n = x.shape
        m = idx.shape[1]
        out = torch.empty(b, c, m, device=x.device, dtype=torch.float32)
        check(_lib.load().geoa3_pn2_gather_points(_p(x.contiguous(), torch.float32), _p(idx, torch.int32), b, c, n, m,
                                                  _p(out), _stream()), "geoa3_pn2_gather_points")
        ctx.save_for_backward(idx)
        ctx.n = n
        return out

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        b, c, m = g.shape
        gx = torch.empty(b, c, ctx.n, device=g.device, dtype=torch.float32)
        check(_lib.load().geoa3_pn2_gather_points_grad(_p(g.contiguous(), torch.float32), _p(idx), b, c, ctx.n, m,
                                                       _p(gx), _stream()), "geoa3_pn2_gather_points_grad")
        return gx, None


def farthest_points_sample(obj_points: Tensor, num_points: int, start: Optional[Tensor] = None) -> Tensor:
    """Lib/utility.py:175-187: [b,3,n] -> [b,3,num_points], differentiable in the points like the reference's
    torch.gather."""
    assert obj_points.size(1) == 3
    pts, idx = fps_indices(obj_points.detach(), num_points, start)
    if obj_points.requires_grad:
        return _GatherPlanar.apply(obj_points, idx)
    return pts


def estimate_normal_via_ori_normal(pc_adv: Tensor, pc_ori: Tensor, normal_ori: Tensor, k: int) -> Tensor:
    """Lib/utility.py:91-108, batched per instance: [b,3,n], [b,3,N], [b,3,N] -> [b,3,n]."""
    b, _, n = pc_adv.shape
    N = pc_ori.shape[2]
    d, i = knn_planar(pc_adv.contiguous(), pc_ori.contiguous(), k)
    out = torch.empty(b, 3, n, device=pc_adv.device, dtype=torch.float32)
    check(_lib.load().geoa3_knn_normal(_p(d), _p(i), _p(normal_ori.contiguous(), torch.float32), b, n, N, k, _p(out),
                                       _stream()), "geoa3_knn_normal")
    return out


def local_frames(pc: Tensor, k: int, knn_idx: Optional[Tensor] = None):
    """Eigen-decomposition of the covariance of the k nearest neighbours of every point: (evals [b,3,n] ascending,
    evecs [b,3(e),3(xyz),n], knn_idx int32 [b,n,k+1])."""
    b, _, n = pc.shape
    pc = pc.contiguous()
    if knn_idx is None:
        _, knn_idx = knn_planar(pc, pc, k + 1)
    evals = torch.empty(b, 3, n, device=pc.device, dtype=torch.float32)
    evecs = torch.empty(b, 3, 3, n, device=pc.device, dtype=torch.float32)
    check(_lib.load().geoa3_local_frames(_p(pc, torch.float32), _p(knn_idx, torch.int32), b, n, k + 1, _p(evals),
                                         _p(evecs), _stream()), "geoa3_local_frames")
    return evals, evecs, knn_idx


def estimate_perpendicular(pc: Tensor, k: int, sigma: float = 0.01, clip: float = 0.05,
                           aux: Optional[Tuple[Tensor, Tensor]] = None) -> Tensor:
    """Lib/utility.py:116-149: jitter in the tangent plane of every point, [b,3,n].  aux = (aux1, aux2) [b,n]:
    the reference's sigma * randn draws (default: drawn here)."""
    b, _, n = pc.shape
    with torch.no_grad():
        _, evecs, _ = local_frames(pc.detach(), k)
        if aux is None:
            aux = (sigma * torch.randn(b, n, device=pc.device), sigma * torch.randn(b, n, device=pc.device))
        out = torch.empty(b, 3, n, device=pc.device, dtype=torch.float32)
        check(_lib.load().geoa3_perp_jitter(_p(evecs), _p(aux[0].contiguous(), torch.float32),
                                            _p(aux[1].contiguous(), torch.float32), b, n, float(clip), _p(out),
                                            _stream()), "geoa3_perp_jitter")
    return out


def smoothness(pc: Tensor, k: int = 16, k2: int = 16) -> Tensor:
    """Measurement/compute_data_smoothness.py:37-67 for a batch of clouds [b,3,n] -> [b]."""
    b, _, n = pc.shape
    pc = pc.contiguous()
    _, idx = knn_planar(pc, pc, max(k, k2) + 1)
    idx2 = idx[:, :, : k2 + 1].contiguous()
    _, evecs, _ = local_frames(pc, k2, idx2)
    idx1 = idx[:, :, : k + 1].contiguous()
    out = torch.empty(b, device=pc.device, dtype=torch.float32)
    check(_lib.load().geoa3_smoothness(_p(pc, torch.float32), _p(idx1), _p(evecs), b, n, k + 1, None, _p(out),
                                       _stream()), "geoa3_smoothness")
    return out


# ------------------------------------------------------------------ defense.py:18-45
def sor_statistic(pc: Tensor, outlier_knn: int) -> Tensor:
    """defense.py:27-28: [b,3,n] -> mean distance to the outlier_knn nearest neighbours [b,n]."""
    b, _, n = pc.shape
    dis = torch.empty(b, n, device=pc.device, dtype=torch.float32)
    check(_lib.load().geoa3_sor_statistic(_p(pc.contiguous(), torch.float32), b, n, int(outlier_knn), _p(dis),
                                          _stream()), "geoa3_sor_statistic")
    return dis


def outlier_removal_indices(pc: Tensor, defense_type: str, drop_num: int, alpha: float, outlier_knn: int):
    """Kept indices of every cloud: (idx int32 [b,n] ascending, -1 padded; count int32 [b])."""
    b, _, n = pc.shape
    mode = {"outliers_fixNum": 0, "outliers_variance": 1}.get(defense_type)
    if mode is None:
        raise AssertionError("Wrong defense type!")
    dis = sor_statistic(pc, outlier_knn)
    idx = torch.empty(b, n, device=pc.device, dtype=torch.int32)
    cnt = torch.empty(b, device=pc.device, dtype=torch.int32)
    check(_lib.load().geoa3_sor_select(_p(dis), b, n, mode, int(drop_num), float(alpha), _p(idx), _p(cnt), None,
                                       _stream()), "geoa3_sor_select")
    return idx, cnt


def outlier_removal_fn(pc: Tensor, defense_type: str, drop_num: int, alpha: float, outlier_knn: int):
    """defense.py:26-45 for one cloud [1,3,n] -> ([1,3,n'], number dropped)."""
    idx, cnt = outlier_removal_indices(pc, defense_type, drop_num, alpha, outlier_knn)
    kept = int(cnt[0].item())
    sel = idx[0, :kept].long()
    return pc[:, :, sel].contiguous(), pc.size(2) - kept


def random_drop_fn(pc: Tensor, drop_num: int, perm: Optional[Tensor] = None):
    """defense.py:18-23; perm: the torch.randperm(n) draw."""
    n = pc.size(2)
    if perm is None:
        perm = torch.randperm(n, device=pc.device)
    idx = torch.sort(perm.to(pc.device)[drop_num:].long())[0]
    return pc[:, :, idx].contiguous(), drop_num


def point_removal_fn(pc, defense_type, drop_num, alpha, outlier_knn, perm=None):
    """defense.py:47-55."""
    if defense_type == "rand_drop":
        return random_drop_fn(pc, drop_num, perm)
    if defense_type in ("outliers_variance", "outliers_fixNum"):
        return outlier_removal_fn(pc, defense_type, drop_num, alpha, outlier_knn)
    raise AssertionError("Wrong defense type!")
