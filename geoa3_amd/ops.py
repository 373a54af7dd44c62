"""Tensor-level entry points over the C ABI (include/geoa3_hip.h): device memory and streams are
PyTorch-ROCm's, the arithmetic is the HIP library's.  No CPU fallback: CPU tensors are rejected.

Planar layout: clouds are fp32 [B,3,N] contiguous (the reference's logical layout)."""
from __future__ import annotations

import collections
import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import GeoArgs, check

Tensor = torch.Tensor


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor], dtype=None) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.Geoa3Error("geoa3_amd ops need device (cuda/HIP) tensors; there is no CPU path")
    if not t.is_contiguous():
        raise _lib.Geoa3Error("tensor must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise _lib.Geoa3Error("expected dtype %s, got %s" % (dtype, t.dtype))
    return t.data_ptr()


def nn1_pair(a: Tensor, r: Tensor, both: bool = True, method: str = "brute", prior=None, policy=None):
    """a [B,3,Na], r [B,3,Nr] -> (d_ar [B,Na], i_ar int32 [B,Na], d_ra [B,Nr] | None, i_ra | None).
    method: "brute" (all pairs, geoa3_nn1_pair), "grid" (geoa3_grid_nn1_pair: the pruned searches -- matrix-core filter /
    grid walk -- same bits; prior = (i_ar, i_ra) of a previous call seeds its radii; policy = (brute_frac, filter) of
    geoa3_debug_grid_nn1_pair for tests / tools) or "auto" (the pruned search whenever both clouds hold >= 32 points)."""
    B, _, Na = a.shape
    Nr = r.shape[2]
    d_ar = torch.empty(B, Na, device=a.device, dtype=torch.float32)
    i_ar = torch.empty(B, Na, device=a.device, dtype=torch.int32)
    d_ra = i_ra = None
    if both:
        d_ra = torch.empty(B, Nr, device=a.device, dtype=torch.float32)
        i_ra = torch.empty(B, Nr, device=a.device, dtype=torch.int32)
    if method == "auto":
        method = "grid" if min(Na, Nr) >= 32 else "brute"
    if method == "grid":
        p_ar, p_ra = prior if prior is not None else (None, None)
        if policy is not None:
            check(_lib.load().geoa3_debug_grid_nn1_pair(_p(a, torch.float32), _p(r, torch.float32), B, Na, Nr,
                                                        _p(p_ar, torch.int32) if p_ar is not None else None,
                                                        _p(p_ra, torch.int32) if p_ra is not None and both else None,
                                                        _p(d_ar), _p(i_ar), _p(d_ra), _p(i_ra), float(policy[0]),
                                                        int(policy[1]), _stream()), "geoa3_debug_grid_nn1_pair")
            return d_ar, i_ar, d_ra, i_ra
        check(_lib.load().geoa3_grid_nn1_pair(_p(a, torch.float32), _p(r, torch.float32), B, Na, Nr,
                                              _p(p_ar, torch.int32) if p_ar is not None else None,
                                              _p(p_ra, torch.int32) if p_ra is not None and both else None,
                                              _p(d_ar), _p(i_ar), _p(d_ra), _p(i_ra), _stream()), "geoa3_grid_nn1_pair")
    else:
        check(_lib.load().geoa3_nn1_pair(_p(a, torch.float32), _p(r, torch.float32), B, Na, Nr, _p(d_ar), _p(i_ar),
                                         _p(d_ra), _p(i_ra), _stream()), "geoa3_nn1_pair")
    return d_ar, i_ar, d_ra, i_ra


def knn_planar(q: Tensor, r: Tensor, K: int, prior: Optional[Tensor] = None, out=None) -> Tuple[Tensor, Tensor]:
    """q [B,3,Nq], r [B,3,Nr] -> (dists [B,Nq,K] ascending, idx int32 [B,Nq,K])."""
    B, _, Nq = q.shape
    Nr = r.shape[2]
    if out is None:
        d = torch.empty(B, Nq, K, device=q.device, dtype=torch.float32)
        i = torch.empty(B, Nq, K, device=q.device, dtype=torch.int32)
    else:
        d, i = out
    check(_lib.load().geoa3_knn(_p(q, torch.float32), _p(r, torch.float32), B, Nq, Nr, K, _p(prior, torch.int32),
                                _p(d), _p(i), _stream()), "geoa3_knn")
    return d, i


def knn_self_scratch(B: int, N: int, device) -> Tensor:
    """Scratch buffer of geoa3_knn_self for [B,3,N] clouds."""
    return torch.empty(int(_lib.load().geoa3_knn_self_scratch_bytes(B, N)), dtype=torch.uint8, device=device)


def knn_self_planar(pc: Tensor, K: int, prior: Optional[Tensor] = None, scratch: Optional[Tensor] = None,
                    out=None, method: int = 0) -> Tuple[Tensor, Tensor]:
    """== knn_planar(pc, pc, K, prior) bit for bit; pruned when prior and scratch are given (method 1: slab along the
    longest axis, 2: cell grid with one wavefront per query, 0: by (K, N))."""
    B, _, N = pc.shape
    if out is None:
        d = torch.empty(B, N, K, device=pc.device, dtype=torch.float32)
        i = torch.empty(B, N, K, device=pc.device, dtype=torch.int32)
    else:
        d, i = out
    check(_lib.load().geoa3_knn_self(_p(pc, torch.float32), B, N, K, _p(prior, torch.int32), _p(d), _p(i),
                                     _p(scratch), int(method), _stream()), "geoa3_knn_self")
    return d, i


def kappa(pc: Tensor, normal: Tensor, knn_idx: Tensor, nn_idx: Optional[Tensor] = None) -> Tensor:
    """pc/normal [B,3,N], knn_idx int32 [B,N,k+1] -> kappa [B,N] (Lib/loss_utils.py:52-62)."""
    B, _, N = pc.shape
    k = knn_idx.shape[2] - 1
    out = torch.empty(B, N, device=pc.device, dtype=torch.float32)
    check(_lib.load().geoa3_kappa(_p(pc, torch.float32), _p(normal, torch.float32), _p(knn_idx, torch.int32),
                                  _p(nn_idx, torch.int32), B, N, int(normal.shape[2]), k, _p(out), _stream()),
          "geoa3_kappa")
    return out


def geo_scratch(B: int, N: int, device) -> Tensor:
    """geoa3_geo_args.scratch for B clouds of N points (16 bytes per point)."""
    return torch.empty(B * N * 16, dtype=torch.uint8, device=device)


def geo_loss_grad(adv: Tensor, ori: Tensor, *, normal_ori=None, kappa_ori=None, d_ao=None, i_ao=None, d_oa=None,
                  i_oa=None, knn_adv=None, dkappa=None, k: int = 0, dis_type: int = 1, single_side: bool = False,
                  w_dis: float = 1.0, w_hd: float = 0.0, w_curv: float = 0.0, want_grad: bool = True,
                  want_kappa: bool = False, out: Optional[dict] = None, deterministic: bool = True,
                  scratch: Optional[Tensor] = None) -> dict:
    """The fused geometric objective (Attacker/geoA3_attack.py:131-166) and d constrain / d adv.
    deterministic (default): every point's gradient is summed by its owner in a fixed order -- bit-for-bit reproducible
    and independent of the rest of the batch; False: LDS float atomics (free summation order)."""
    B, _, N = adv.shape
    dev = adv.device
    o = out if out is not None else {}
    for name in ("dis_loss", "hd_loss", "curv_loss", "constrain"):
        if name not in o:
            o[name] = torch.empty(B, device=dev, dtype=torch.float32)
    if want_grad and "grad" not in o:
        o["grad"] = torch.empty(B, 3, N, device=dev, dtype=torch.float32)
    if want_kappa and "kappa_adv" not in o:
        o["kappa_adv"] = torch.empty(B, N, device=dev, dtype=torch.float32)
    if scratch is None and (deterministic or not want_grad) and (1024 < N or k > 32) and N <= 4096 and knn_adv is not None:
        scratch = geo_scratch(B, N, dev)      # (callers in a loop hand over their own: AttackRunner)
    elif scratch is False:                    # tests: the one-workgroup kernel
        scratch = None
    a = GeoArgs(adv=_p(adv, torch.float32), ori=_p(ori, torch.float32), normal_ori=_p(normal_ori),
                kappa_ori=_p(kappa_ori), d_ao=_p(d_ao), i_ao=_p(i_ao, torch.int32) if i_ao is not None else None,
                d_oa=_p(d_oa), i_oa=_p(i_oa, torch.int32) if i_oa is not None else None,
                knn_adv=_p(knn_adv, torch.int32) if knn_adv is not None else None, dkappa=_p(dkappa),
                B=B, N=N, k=k, Nr=int(ori.shape[2]), dis_type=dis_type, single_side=int(single_side), w_dis=w_dis, w_hd=w_hd,
                w_curv=w_curv, dis_loss=_p(o["dis_loss"]), hd_loss=_p(o["hd_loss"]), curv_loss=_p(o["curv_loss"]),
                constrain=_p(o["constrain"]), kappa_adv=_p(o.get("kappa_adv")) if want_kappa else None,
                grad=_p(o["grad"]) if want_grad else None, deterministic=int(bool(deterministic)),
                scratch=_p(scratch))
    check(_lib.load().geoa3_geo_loss_grad(C.byref(a), _stream()), "geoa3_geo_loss_grad")
    return o


# ---------------------------------------------------------------------------------------------
# Operator-level mirror of pytorch3d.ops: [b,n,3] point-major arguments, int64 idx,
# differentiable through `dists`.
# ---------------------------------------------------------------------------------------------
_KNN = collections.namedtuple("KNN", ["dists", "idx", "knn"])


def knn_points(p1: Tensor, p2: Tensor, K: int = 1, **_ignored):
    """pytorch3d.ops.knn_points(p1 [b,n1,3], p2 [b,n2,3], K) -> KNN(dists, idx, knn=None)."""
    from . import library  # noqa: F401  (registers geoa3::knn_points with its autograd formula)
    d, idx = torch.ops.geoa3.knn_points(p1, p2, int(K))
    return _KNN(d, idx, None)


def knn_gather(x: Tensor, idx: Tensor) -> Tensor:
    """pytorch3d.ops.knn_gather(x [b,m,u], idx [b,l,k]) -> [b,l,k,u] (pure data movement)."""
    b, m, u = x.shape
    _, l, k = idx.shape
    return torch.gather(x, 1, idx.reshape(b, l * k, 1).expand(b, l * k, u)).view(b, l, k, u)
