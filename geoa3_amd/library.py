"""`torch.library` registration of the operator boundary (north_star "PyTorch-ROCm custom ops").

Every op of the `geoa3::` namespace has a schema, a CUDA(HIP) implementation that calls the C ABI through
`geoa3_amd.ops` (ctypes: the transport stays what it is), a fake (meta) kernel so that FakeTensor tracing --
`torch.compile`, `torch.export`, `make_fx` -- sees shapes and dtypes without running anything, and, where the reference
differentiates through it, a registered autograd formula.  `geoa3_amd.ops.knn_points` and the loss functions of
`geoa3_amd.loss_utils` are built on these ops, so a caller that keeps the reference's `_forward_step` composition
(Attacker/geoA3_attack.py:131-166) can put it under `torch.compile(fullgraph=True)` without a graph break.

    geoa3::nn1_pair        knn_points(K=1) both directions            Lib/loss_utils.py:32-33,41,48,70,92
    geoa3::knn             knn_points(K)                              Lib/loss_utils.py:57,77
    geoa3::knn_points      the pytorch3d operator itself ([b,n,3] arguments, int64 idx, differentiable dists)
    geoa3::kappa           _get_kappa_ori / the forward of _get_kappa_adv   Lib/loss_utils.py:52-82
    geoa3::geo_loss_grad   CD / HD / L2 / curvature values and d constrain / d adv   Attacker/geoA3_attack.py:131-166
    geoa3::point_loss      chamfer / pseudo-chamfer / hausdorff / l2 with autograd    Lib/loss_utils.py:25-50
    geoa3::kappa_adv       _get_kappa_adv with autograd                               Lib/loss_utils.py:64-82
    geoa3::pointnet_forward / _backward   PointNet.forward (eval) and its input gradient   Model/PointNet.py:132-160
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
from torch.library import custom_op

from . import ops

Tensor = torch.Tensor
_f32 = torch.float32
_i32 = torch.int32


def _planar(x: Tensor) -> Tensor:
    return x.detach().contiguous().float()


# ------------------------------------------------------------------------------------------------ searches
@custom_op("geoa3::nn1_pair", mutates_args=(), device_types="cuda")
def nn1_pair(a: Tensor, r: Tensor, both: bool) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """a [B,3,Na], r [B,3,Nr] planar -> d_ar [B,Na], i_ar int32, d_ra [B,Nr], i_ra (empty [B,0] when not both)."""
    d_ar, i_ar, d_ra, i_ra = ops.nn1_pair(_planar(a), _planar(r), both=both, method="auto")
    if not both:
        d_ra, i_ra = a.new_empty(a.shape[0], 0, dtype=_f32), a.new_empty(a.shape[0], 0, dtype=_i32)
    return d_ar, i_ar, d_ra, i_ra


@nn1_pair.register_fake
def _(a, r, both):
    B, Na, Nr = a.shape[0], a.shape[2], (r.shape[2] if both else 0)
    return (a.new_empty(B, Na, dtype=_f32), a.new_empty(B, Na, dtype=_i32), a.new_empty(B, Nr, dtype=_f32),
            a.new_empty(B, Nr, dtype=_i32))


@custom_op("geoa3::knn", mutates_args=(), device_types="cuda")
def knn(q: Tensor, r: Tensor, K: int) -> Tuple[Tensor, Tensor]:
    """q [B,3,Nq], r [B,3,Nr] planar -> dists [B,Nq,K] ascending, idx int32 [B,Nq,K]."""
    return ops.knn_planar(_planar(q), _planar(r), K)


@knn.register_fake
def _(q, r, K):
    B, Nq = q.shape[0], q.shape[2]
    return q.new_empty(B, Nq, K, dtype=_f32), q.new_empty(B, Nq, K, dtype=_i32)


@custom_op("geoa3::kappa", mutates_args=(), device_types="cuda")
def kappa(pc: Tensor, normal: Tensor, knn_idx: Tensor, nn_idx: Optional[Tensor]) -> Tensor:
    return ops.kappa(_planar(pc), _planar(normal), knn_idx.contiguous(), None if nn_idx is None else nn_idx.contiguous())


@kappa.register_fake
def _(pc, normal, knn_idx, nn_idx):
    return pc.new_empty(pc.shape[0], pc.shape[2], dtype=_f32)


# ------------------------------------------------------------------------------------------------ the fused objective
@custom_op("geoa3::geo_loss_grad", mutates_args=(), device_types="cuda")
def geo_loss_grad(adv: Tensor, ori: Tensor, normal_ori: Optional[Tensor], kappa_ori: Optional[Tensor],
                  d_ao: Optional[Tensor], i_ao: Optional[Tensor], d_oa: Optional[Tensor], i_oa: Optional[Tensor],
                  knn_adv: Optional[Tensor], dkappa: Optional[Tensor], k: int, dis_type: int, single_side: bool,
                  w_dis: float, w_hd: float, w_curv: float, deterministic: bool
                  ) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """-> (dis_loss [B], hd_loss [B], curv_loss [B], constrain [B], d constrain / d adv [B,3,N])."""
    c = lambda t: None if t is None else t.contiguous()
    o = ops.geo_loss_grad(_planar(adv), _planar(ori), normal_ori=c(normal_ori), kappa_ori=c(kappa_ori), d_ao=c(d_ao),
                          i_ao=c(i_ao), d_oa=c(d_oa), i_oa=c(i_oa), knn_adv=c(knn_adv), dkappa=c(dkappa), k=k,
                          dis_type=dis_type, single_side=single_side, w_dis=w_dis, w_hd=w_hd, w_curv=w_curv,
                          deterministic=deterministic)
    return o["dis_loss"], o["hd_loss"], o["curv_loss"], o["constrain"], o["grad"]


@geo_loss_grad.register_fake
def _(adv, ori, normal_ori, kappa_ori, d_ao, i_ao, d_oa, i_oa, knn_adv, dkappa, k, dis_type, single_side, w_dis, w_hd,
      w_curv, deterministic):
    B = adv.shape[0]
    v = lambda: adv.new_empty(B, dtype=_f32)
    return v(), v(), v(), v(), adv.new_empty(B, 3, adv.shape[2], dtype=_f32)


# ------------------------------------------------------------------------------------------------ pytorch3d.ops.knn_points
@custom_op("geoa3::knn_points", mutates_args=(), device_types="cuda")
def knn_points_op(p1: Tensor, p2: Tensor, K: int) -> Tuple[Tensor, Tensor]:
    """p1 [b,n1,3], p2 [b,n2,3] -> dists [b,n1,K] (squared, ascending), idx int64 [b,n1,K]."""
    q = p1.detach().permute(0, 2, 1).contiguous().float()
    r = p2.detach().permute(0, 2, 1).contiguous().float()
    if K == 1:
        d, i, _, _ = ops.nn1_pair(q, r, both=False, method="auto")
        d, i = d.unsqueeze(-1), i.unsqueeze(-1)
    else:
        d, i = ops.knn_planar(q, r, K)
    return d, i.long()


@knn_points_op.register_fake
def _(p1, p2, K):
    b, n1 = p1.shape[0], p1.shape[1]
    return p1.new_empty(b, n1, K, dtype=_f32), p1.new_empty(b, n1, K, dtype=torch.int64)


def _knn_points_setup(ctx, inputs, output):
    p1, p2, _K = inputs
    ctx.save_for_backward(p1, p2, output[1])


def _knn_points_backward(ctx, gd, _gi):
    # pytorch3d's knn backward: dp1 += 2 g (p1 - p2[idx]);  dp2[idx] -= the same (scatter-add)
    p1, p2, idx = ctx.saved_tensors
    b, n1, K = idx.shape
    nb = ops.knn_gather(p2, idx)                         # [b,n1,K,3]
    diff = 2.0 * gd.unsqueeze(-1) * (p1.unsqueeze(2) - nb)
    g1 = diff.sum(2)
    g2 = torch.zeros_like(p2).scatter_add(1, idx.reshape(b, n1 * K, 1).expand(b, n1 * K, 3), -diff.reshape(b, n1 * K, 3))
    return g1, g2, None


knn_points_op.register_autograd(_knn_points_backward, setup_context=_knn_points_setup)

# ------------------------------------------------------------------------------------------------ Lib/loss_utils.py
_KINDS = {"cd": 0, "pcd": 1, "hd": 2, "l2": 3}


@custom_op("geoa3::point_loss", mutates_args=(), device_types="cuda")
def point_loss(adv_pc: Tensor, ori_pc: Tensor, kind: int) -> Tuple[Tensor, Tensor]:
    """One of chamfer (0) / pseudo-chamfer (1) / hausdorff (2) / l2 (3): -> (loss [b], d loss / d adv [b,3,n])."""
    adv, ori = _planar(adv_pc), _planar(ori_pc)
    kw = dict(dis_type=0, w_dis=0.0, w_hd=0.0)
    if kind in (0, 1, 2):
        d_ao, i_ao, d_oa, i_oa = ops.nn1_pair(adv, ori, both=(kind == 0), method="auto")
        kw.update(d_ao=d_ao, i_ao=i_ao, d_oa=d_oa, i_oa=i_oa)
    if kind == 0:
        kw.update(dis_type=1, w_dis=1.0)
    elif kind == 1:
        kw.update(dis_type=1, w_dis=1.0, single_side=True)
    elif kind == 2:
        kw.update(w_hd=1.0)
    else:
        kw.update(dis_type=2, w_dis=1.0)
    out = ops.geo_loss_grad(adv, ori, **kw)
    return (out["hd_loss"] if kind == 2 else out["dis_loss"]).clone(), out["grad"]


@point_loss.register_fake
def _(adv_pc, ori_pc, kind):
    return adv_pc.new_empty(adv_pc.shape[0], dtype=_f32), adv_pc.new_empty(adv_pc.shape, dtype=_f32)


def _point_loss_setup(ctx, inputs, output):
    ctx.save_for_backward(output[1])


def _point_loss_backward(ctx, g, _gg):
    (grad,) = ctx.saved_tensors
    return grad * g.view(-1, 1, 1), None, None


point_loss.register_autograd(_point_loss_backward, setup_context=_point_loss_setup)


@custom_op("geoa3::kappa_adv", mutates_args=(), device_types="cuda")
def kappa_adv(adv_pc: Tensor, ori_pc: Tensor, ori_normal: Tensor, k: int) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """_get_kappa_adv: -> (kappa_adv [b,n], normal [b,3,n], i_ao int32 [b,n], knn_adv int32 [b,n,k+1])."""
    adv, ori, nrm = _planar(adv_pc), _planar(ori_pc), _planar(ori_normal)
    _, i_ao, _, _ = ops.nn1_pair(adv, ori, both=False, method="auto")
    _, knn_adv = ops.knn_planar(adv, adv, k + 1)
    kap = ops.kappa(adv, nrm, knn_adv, i_ao)
    b, _, n = adv.shape
    normal = torch.gather(nrm, 2, i_ao.long().unsqueeze(1).expand(b, 3, n)).contiguous()
    return kap, normal, i_ao, knn_adv


@kappa_adv.register_fake
def _(adv_pc, ori_pc, ori_normal, k):
    b, _, n = adv_pc.shape
    return (adv_pc.new_empty(b, n, dtype=_f32), adv_pc.new_empty(b, 3, n, dtype=_f32), adv_pc.new_empty(b, n, dtype=_i32),
            adv_pc.new_empty(b, n, k + 1, dtype=_i32))


def _kappa_adv_setup(ctx, inputs, output):
    adv_pc, ori_pc, ori_normal, k = inputs
    ctx.save_for_backward(adv_pc, ori_pc, ori_normal, output[2], output[3])
    ctx.k = k


def _kappa_adv_backward(ctx, gk, _gn, _gi, _gt):
    adv, ori, nrm, i_ao, knn_adv = ctx.saved_tensors
    out = torch.ops.geoa3.geo_loss_grad(adv, ori, nrm, None, None, i_ao, None, None, knn_adv, gk.contiguous().float(),
                                        ctx.k, 0, False, 0.0, 0.0, 0.0, True)
    return out[4], None, None, None


kappa_adv.register_autograd(_kappa_adv_backward, setup_context=_kappa_adv_setup)

# ------------------------------------------------------------------------------------------------ the victim
import weakref

_NETS = weakref.WeakValueDictionary()   # handle -> geoa3_amd.pointnet.PointNet (custom ops take tensors and scalars only)


import itertools

_NEXT_HANDLE = itertools.count(1)


def register_net(net) -> int:
    """A handle for `net` (a geoa3_amd.pointnet.PointNet in eval mode) to pass to geoa3::pointnet_forward.  Handles
    come from a counter (never reused); a copy of a module (copy.deepcopy / pickle) registers itself again in
    PointNet.__setstate__."""
    h = next(_NEXT_HANDLE)
    _NETS[h] = net
    return h


def net_handle(net) -> int:
    """The handle under which `net` itself is registered (re-registered if the stored one belongs to another module).
    Eager callers; a traced forward reads net._handle, which PointNet.__setstate__ keeps its own for copies."""
    h = getattr(net, "_handle", None)
    if h is None or _NETS.get(h) is not net:
        h = net._handle = register_net(net)
    return h


@custom_op("geoa3::pointnet_forward", mutates_args=(), device_types="cuda")
def pointnet_forward(x: Tensor, handle: int) -> Tensor:
    """Eval-mode PointNet.forward on x [b,3,n] -> logits [b,classes] (the library's forward; activations stay in the
    net's workspace for a backward of the SAME input)."""
    from .pointnet import pointnet_forward_raw
    return pointnet_forward_raw(_NETS[handle], x)


@pointnet_forward.register_fake
def _(x, handle):
    return x.new_empty(x.shape[0], int(_NETS[handle].classes), dtype=_f32)


@custom_op("geoa3::pointnet_backward", mutates_args=(), device_types="cuda")
def pointnet_backward(x: Tensor, dlogits: Tensor, handle: int) -> Tensor:
    """d (logits . dlogits) / d x [b,3,n]: recomputes the forward of x into the workspace, then the input gradient."""
    from .pointnet import pointnet_backward_raw
    return pointnet_backward_raw(_NETS[handle], x, dlogits)


@pointnet_backward.register_fake
def _(x, dlogits, handle):
    return x.new_empty(x.shape, dtype=_f32)


def _pn_setup(ctx, inputs, output):
    x, handle = inputs
    ctx.save_for_backward(x)
    ctx.handle = handle


def _pn_backward(ctx, g):
    (x,) = ctx.saved_tensors
    return torch.ops.geoa3.pointnet_backward(x, g.contiguous(), ctx.handle), None


pointnet_forward.register_autograd(_pn_backward, setup_context=_pn_setup)
