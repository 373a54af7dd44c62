"""CPU restatement of the reference's dense-cloud / defence / measurement helpers.

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
the product path (geoa3_amd/ fails loudly when libgeoa3_hip.so is missing).

Each function restates one reference function (file:line under Gorilla-Lab-SCUT/GeoA3) with its random draws
turned into explicit inputs; pinned against the reference's own outputs by tests/golden/make_golden_aux.py ->
tests/golden/geoa3_golden_aux.npz (tests/test_oracle_aux.py).
"""
from __future__ import annotations

import numpy as np
import torch

from . import geoa3_oracle as O


# ------------------------------------------------------------------ Lib/utility.py:175-187
def farthest_points_sample(pc: torch.Tensor, m: int, start: torch.Tensor):
    """pc [b,3,n], start [b] (the reference draws it with torch.randint, utility.py:179) -> (points [b,3,m],
    idx [b,m] int64).  m-1 rounds of dists = min(dists, |p - p_last|), next = argmax (first maximal index).
    The distance is sqrt(fl(fl(dx^2+dy^2)+dz^2)): the order torch.norm(dim=1) reduces three rows in."""
    b, _, n = pc.shape
    sel = start.view(b, 1).long()
    dists = torch.full((b, n), float("inf"), dtype=pc.dtype)
    for _ in range(m - 1):
        last = torch.gather(pc, 2, sel[:, -1].view(b, 1, 1).expand(b, 3, 1))
        d = pc - last
        dd = d * d
        dists = torch.min(dists, torch.sqrt((dd[:, 0] + dd[:, 1]) + dd[:, 2]))
        sel = torch.cat([sel, torch.argmax(dists, dim=1, keepdim=True)], dim=1)
    return torch.gather(pc, 2, sel.unsqueeze(1).expand(b, 3, m)), sel


# ------------------------------------------------------------------ Lib/utility.py:91-108
def estimate_normal_via_ori_normal(pc_adv, pc_ori, normal_ori, k):
    """Normal of each adversarial point := normal of its nearest original point when the point did not move
    (squared distance < 1e-6), otherwise the normalised mean of the normals of its k nearest original points.
    Evaluated per instance (the reference's broadcast at utility.py:99 is only valid for b == 1, which is how
    main_attack.py:244-246 calls it)."""
    p1, p2 = pc_adv.permute(0, 2, 1), pc_ori.permute(0, 2, 1)
    dists, idx = O.knn_points(p1, p2, k)
    normal_pts = O.knn_gather(normal_ori.permute(0, 2, 1), idx).permute(0, 3, 1, 2)   # [b,3,n,k]
    avg = normal_pts.mean(dim=-1)
    avg = avg / (avg.norm(dim=1, keepdim=True) + 1e-12)
    cond = (dists[:, :, 0] < 1e-6).unsqueeze(1).expand_as(avg)
    return torch.where(cond, normal_pts[:, :, :, 0], avg)


# ------------------------------------------------------------------ Lib/utility.py:116-149
def local_covariance(pc: torch.Tensor, k: int):
    """[b,n,3,3] covariance (factor 1/(k-1)) of the k nearest neighbours (self excluded) of every point."""
    p = pc.permute(0, 2, 1)
    _, idx = O.knn_points(p, p, k + 1)
    nn_pts = O.knn_gather(p, idx)[:, :, 1:, :]                       # [b,n,k,3]
    c = nn_pts - nn_pts.mean(dim=2, keepdim=True)
    return torch.matmul(c.transpose(2, 3), c) * (1.0 / (k - 1))


def estimate_perpendicular(pc, k, aux1, aux2, clip=0.05):
    """aux1, aux2 [b,n]: the reference's sigma * randn draws (utility.py:146-147).  Returns the jitter
    clamp(v1*aux1) + clamp(v2*aux2) [b,3,n] with v1/v2 the eigenvectors of the largest / middle eigenvalue of
    the neighbourhood covariance.  Eigenvector signs (and, in the reference, which of the two is v1:
    topk(sorted=False), utility.py:134) are implementation-defined; callers compare modulo those."""
    cov = local_covariance(pc, k)
    w, v = torch.linalg.eigh(cov)                                    # ascending
    v1 = v[..., 2].permute(0, 2, 1)                                  # [b,3,n]
    v2 = v[..., 1].permute(0, 2, 1)
    return (torch.clamp(v1 * aux1.unsqueeze(1), -clip, clip) + torch.clamp(v2 * aux2.unsqueeze(1), -clip, clip),
            w, v)


# ------------------------------------------------------------------ defense.py:18-45
def sor_statistic(pc: torch.Tensor, outlier_knn: int):
    """defense.py:27-28: mean distance to the outlier_knn nearest neighbours, with the reference's +1e-10 inside
    the difference.  pc [b,3,n] -> [b,n]."""
    dis = (pc.unsqueeze(2) - pc.unsqueeze(3) + 1e-10).pow(2).sum(dim=1).sqrt()
    return dis.topk(outlier_knn + 1, dim=2, largest=False, sorted=True)[0][:, :, 1:].contiguous().mean(dim=-1)


def outlier_removal(pc, defense_type, drop_num, alpha, outlier_knn):
    """defense.py:26-45 for ONE cloud pc [1,3,n] -> (kept cloud [1,3,n'], number dropped, kept indices)."""
    dis = sor_statistic(pc, outlier_knn)
    n = pc.size(2)
    if defense_type == "outliers_variance":
        keep = dis < (dis.mean(-1) + alpha * dis.std(-1)).unsqueeze(-1)
        idx = torch.nonzero(keep[0]).view(-1)
    elif defense_type == "outliers_fixNum":
        idx = torch.sort(dis.topk(n - drop_num, dim=1, largest=False, sorted=True)[1].view(-1))[0]
    else:
        raise AssertionError("Wrong defense type!")
    return pc[:, :, idx].contiguous(), n - idx.numel(), idx


def random_drop(pc, drop_num, perm):
    """defense.py:18-23 with the torch.randperm draw as an input."""
    idx = torch.sort(perm[drop_num:].long())[0]
    return pc[:, :, idx].contiguous(), drop_num


# ------------------------------------------------------------------ Measurement/compute_data_smoothness.py:37-67
def smoothness(pc_n3: torch.Tensor, k: int, k2: int):
    """pc [n,3] -> scalar: max_i mean_{q in kNN_k(p_i)} |<q - p_i, n_i>| with n_i the smallest-eigenvalue
    eigenvector of the covariance of the k2 nearest neighbours (float64 np.cov / eig as the reference)."""
    n = pc_n3.size(0)
    dis = ((pc_n3.unsqueeze(1) - pc_n3.unsqueeze(0)) ** 2).sum(2)
    idx2 = dis.topk(k2 + 1, dim=-1, largest=False, sorted=True)[1][:, 1:]
    pts = pc_n3[idx2] - pc_n3.unsqueeze(1)                           # [n,k2,3]
    normal = torch.empty(n, 3)
    pts_np = pts.numpy()
    for j in range(n):
        C = np.cov(pts_np[j].T)
        v, t = np.linalg.eigh(C)
        normal[j] = torch.from_numpy(t[:, 0].astype(np.float32))
    idx = dis.topk(k + 1, dim=-1, largest=False, sorted=True)[1][:, 1:]
    pts = pc_n3[idx] - pc_n3.unsqueeze(1)
    return torch.abs((pts * normal.unsqueeze(1)).sum(2)).mean(1).max()


# ------------------------------------------------------------------ geoA3_attack.py:283-310 (eval_num vote)
def vote(labels: torch.Tensor, target: int, gt: int, targeted: bool, eval_num: int):
    """labels [eval_num] -> (success, output_label): geoA3_attack.py:294-295 (majority of _compare, torch.mode)."""
    ok = (labels == target) if targeted else (labels != gt)
    return bool(ok.sum() > 0.5 * eval_num), int(labels.mode().values.item())
