"""CHECKER (test infrastructure, not product code): the PointNet++ SSG classifier composed LAYER BY LAYER from torch modules
and the single HIP operators -- what geoa3_amd.pointnet2 shipped as its "module path" until round 6.  The product package
has ONE backend (the native classifier, geoa3_pn2ssg_forward / _backward); this composition stays here so that the tests
can hold the native path, the fused level-1 kernel, the fused convolution tails and the pre-transformed level against an
independent evaluation on the same weights (torch.matmul = a library GEMM: allowed in a checker, not in the product).

    module_forward(net, pointcloud, fuse_level1=True, fuse_tail=True, pretransform=True) -> logits   (differentiable)
    ModulePathNet(net, ...)   an nn.Module around it: a generic victim for attack() (driven through torch autograd)
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from geoa3_amd import _lib
from geoa3_amd._lib import check
from geoa3_amd.pointnet2 import (QueryAndGroup, _chk, _fold_triples, _frozen, _s, ball_query, furthest_point_sample,
                                 gather_operation)

Tensor = torch.Tensor


class _BiasRelu(torch.autograd.Function):
    """y = relu(z + shift[c]) in place on the GEMM output (one pass over the [B,C,M*S] tensor)."""

    @staticmethod
    def forward(ctx, z, shift):
        B, C, L = z.shape
        check(_lib.load().geoa3_pn2_bias_relu(z.data_ptr(), shift.data_ptr(), B, C, L, _s()), "bias_relu")
        ctx.mark_dirty(z)
        ctx.save_for_backward(z)
        return z

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        g = g.contiguous()
        check(_lib.load().geoa3_pn2_relu_grad(y.data_ptr(), g.data_ptr(), g.data_ptr(), g.numel(), _s()), "relu_grad")
        return g, None


class _BiasReluMax(torch.autograd.Function):
    """out[b,c,m] = max_s relu(z[b,c,m,s] + shift[c]) without writing the activated tensor."""

    @staticmethod
    def forward(ctx, z, shift, M, S):
        B, C, _ = z.shape
        out = torch.empty(B, C, M, device=z.device, dtype=torch.float32)
        arg = torch.empty(B, C, M, device=z.device, dtype=torch.int32)
        check(_lib.load().geoa3_pn2_bias_relu_max(z.data_ptr(), shift.data_ptr(), B, C, M, S, out.data_ptr(),
                                                  arg.data_ptr(), _s()), "bias_relu_max")
        ctx.save_for_backward(out, arg)
        ctx.S = S
        return out

    @staticmethod
    def backward(ctx, g):
        out, arg = ctx.saved_tensors
        B, C, M = out.shape
        dz = torch.empty(B, C, M * ctx.S, device=out.device, dtype=torch.float32)
        check(_lib.load().geoa3_pn2_bias_relu_max_grad(g.contiguous().data_ptr(), out.data_ptr(), arg.data_ptr(), B, C, M,
                                                       ctx.S, dz.data_ptr(), _s()), "bias_relu_max_grad")
        return dz, None, None, None


def _conv1x1(x: Tensor, w: Tensor, bias: Optional[Tensor], gate: Optional[Tensor], relu: bool) -> Tensor:
    """geoa3_conv1x1: [B,K,L] -> [B,Co,L] with the epilogue fused (csrc/pointnet_conv_split.hip)."""
    B, K, L = x.shape
    Co = w.shape[0]
    y = torch.empty(B, Co, L, device=x.device, dtype=torch.float32)
    check(_lib.load().geoa3_conv1x1(x.data_ptr(), w.data_ptr(), bias.data_ptr() if bias is not None else None,
                                    gate.data_ptr() if gate is not None else None, y.data_ptr(), B, L, K, Co,
                                    1 if relu else 0, _s()), "geoa3_conv1x1")
    return y


class _SharedTail(torch.autograd.Function):
    """Layers 2.. of a shared MLP with frozen weights, ending in the max over the S samples, on the HIP 1x1-convolution
    operator: relu(W h + shift) is ONE kernel per layer (no separate bias/relu pass), the last layer's bias + relu +
    max is the existing tail kernel, and in backward every input-gradient product carries the relu gate of the layer
    below in its epilogue (no relu_grad pass).  Returns [B,C,M]; gradient w.r.t. the input activation only."""

    @staticmethod
    def forward(ctx, h, M, S, *wb):
        ws, bs = wb[0::2], wb[1::2]
        acts = [h]
        for w, b in zip(ws[:-1], bs[:-1]):
            acts.append(_conv1x1(acts[-1], w, b, None, True))
        z = _conv1x1(acts[-1], ws[-1], None, None, False)
        B, C, _ = z.shape
        out = torch.empty(B, C, M, device=z.device, dtype=torch.float32)
        arg = torch.empty(B, C, M, device=z.device, dtype=torch.int32)
        check(_lib.load().geoa3_pn2_bias_relu_max(z.data_ptr(), bs[-1].data_ptr(), B, C, M, S, out.data_ptr(),
                                                  arg.data_ptr(), _s()), "bias_relu_max")
        ctx.save_for_backward(out, arg, *acts[1:])
        ctx.wts = [w.t().contiguous() for w in ws]
        ctx.S = S
        return out

    @staticmethod
    def backward(ctx, g):
        out, arg, *acts = ctx.saved_tensors          # acts: the relu'd outputs of layers 2 .. last-1
        B, C, M = out.shape
        dz = torch.empty(B, C, M * ctx.S, device=out.device, dtype=torch.float32)
        check(_lib.load().geoa3_pn2_bias_relu_max_grad(g.contiguous().data_ptr(), out.data_ptr(), arg.data_ptr(), B, C, M,
                                                       ctx.S, dz.data_ptr(), _s()), "bias_relu_max_grad")
        for i in range(len(ctx.wts) - 1, -1, -1):   # d/d(input of layer i), gated by that input's relu when it is ours
            dz = _conv1x1(dz, ctx.wts[i], None, acts[i - 1] if i > 0 else None, False)
        return (dz, None, None) + (None,) * (2 * len(ctx.wts))


class _PretransformedSA(torch.autograd.Function):
    """A whole set-abstraction level with frozen weights after its first layer has been applied to the un-grouped
    points (PointnetSAModuleMSG._pretransformed_level): gather + shift + relu in one pass, the remaining layers on the
    fused convolution operator, the max over the samples; in backward the last input-gradient convolution carries the
    first layer's relu gate, so what is left is the row sums (d shift) and the scatter-add of the gather (d r).
    forward(r [B,C,N], idx [B,M,S] int32, shift [B,C,M], *(W, shift) of layers 2..) -> [B,C_out,M]."""

    @staticmethod
    def forward(ctx, r, idx, shift, *wb):
        lib = _lib.load()
        B, C, N = r.shape
        M, S = idx.shape[1], idx.shape[2]
        h = torch.empty(B, C, M * S, device=r.device, dtype=torch.float32)
        check(lib.geoa3_pn2_group_shift_relu(r.data_ptr(), idx.data_ptr(), shift.data_ptr(), B, C, N, M, S,
                                             h.data_ptr(), _s()), "group_shift_relu")
        ws, bs = wb[0::2], wb[1::2]
        acts = [h]
        for w, b in zip(ws[:-1], bs[:-1]):
            acts.append(_conv1x1(acts[-1], w, b, None, True))
        Co, Kl = ws[-1].shape
        out = torch.empty(B, Co, M, device=r.device, dtype=torch.float32)
        arg = torch.empty(B, Co, M, device=r.device, dtype=torch.int32)
        ctx.pooled = S == 64 and Kl == 128          # the last layer + max over the samples as ONE kernel
        if ctx.pooled:
            check(lib.geoa3_conv1x1_max64(acts[-1].data_ptr(), ws[-1].data_ptr(), bs[-1].data_ptr(), out.data_ptr(),
                                          arg.data_ptr(), B, M * S, Kl, Co, _s()), "conv1x1_max64")
        else:
            z = _conv1x1(acts[-1], ws[-1], None, None, False)
            check(lib.geoa3_pn2_bias_relu_max(z.data_ptr(), bs[-1].data_ptr(), B, Co, M, S, out.data_ptr(),
                                              arg.data_ptr(), _s()), "bias_relu_max")
        ctx.save_for_backward(out, arg, idx, *acts)
        ctx.wts = [w.t().contiguous() for w in ws]
        ctx.dims = (N, M, S)
        return out

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        out, arg, idx, *acts = ctx.saved_tensors      # acts[0]: the gathered first layer, acts[i]: output of layer i + 1
        N, M, S = ctx.dims
        B, Co, _ = out.shape
        last = len(ctx.wts) - 1
        if ctx.pooled and Co == 256:   # the pooled layer's sparse gradient is formed inside the convolution
            gz = (g * (out > 0)).transpose(1, 2).contiguous()          # centre-major [B, M, Co]
            argt = arg.transpose(1, 2).contiguous()
            ci = ctx.wts[last].shape[0]
            dz = torch.empty(B, ci, M * S, device=out.device, dtype=torch.float32)
            check(lib.geoa3_conv1x1_onehot64(gz.data_ptr(), argt.data_ptr(), ctx.wts[last].data_ptr(), acts[last].data_ptr(),
                                             dz.data_ptr(), B, M * S, Co, ci, _s()), "conv1x1_onehot64")
            last -= 1
        else:
            dz = torch.empty(B, Co, M * S, device=out.device, dtype=torch.float32)
            check(lib.geoa3_pn2_bias_relu_max_grad(g.contiguous().data_ptr(), out.data_ptr(), arg.data_ptr(), B, Co, M, S,
                                                   dz.data_ptr(), _s()), "bias_relu_max_grad")
        for i in range(last, -1, -1):                 # every product gated by the relu of the layer below
            dz = _conv1x1(dz, ctx.wts[i], None, acts[i], False)
        C = dz.shape[1]
        dshift = torch.empty(B, C, M, device=dz.device, dtype=torch.float32)
        dr = torch.empty(B, C, N, device=dz.device, dtype=torch.float32)
        rc = lib.geoa3_pn2_group_points_grad_sums(dz.data_ptr(), idx.data_ptr(), B, C, N, M, S, dr.data_ptr(),
                                                  dshift.data_ptr(), _s())
        if rc == _lib.ENOSUPPORT:   # other ball sizes: two passes
            check(lib.geoa3_pn2_shift_relu_grad(None, dz.data_ptr(), None, dshift.data_ptr(), B * C * M, S, _s()),
                  "shift_relu_grad")
            rc = lib.geoa3_pn2_group_points_grad(dz.data_ptr(), idx.data_ptr(), B, C, N, M, S, dr.data_ptr(), _s())
        check(rc, "group_points_grad")
        return (dr, None, dshift) + (None,) * (2 * len(ctx.wts))


def _tail_eligible(folded, first: int) -> bool:
    """Layers first.. can run on geoa3_conv1x1 in both directions: K and Co in {64,128,256} / multiples of 64."""
    return all(w.shape[1] in (64, 128, 256) and w.shape[0] in (64, 128, 256) for w, _ in folded[first:])


def run_shared_mlp(mlp: nn.Sequential, x: Tensor, fuse_max: bool = False, fuse_tail: bool = True) -> Tensor:
    """Apply a build_shared_mlp() stack to x [B,C,M,S] (eval mode).  Each Conv2d 1x1 + BatchNorm2d + ReLU triple is
    ONE channel GEMM (hipBLASLt; BatchNorm's running-statistics scale folded into the weights) followed by ONE
    in-place HIP pass relu(z + shift); with fuse_max the last triple's tail also takes the max over the S samples
    (== F.max_pool2d over nsample, pointnet2_modules.py:66-70) and returns [B,C,M] without materialising its
    activation.  Falls back to the plain module stack in training mode."""
    layers = list(mlp)
    triples = []
    ok = len(layers) % 3 == 0
    for i in range(0, len(layers) - 2, 3):
        conv, bn, act = layers[i], layers[i + 1], layers[i + 2]
        ok = ok and isinstance(conv, nn.Conv2d) and conv.kernel_size == (1, 1) and conv.bias is None and \
            isinstance(bn, nn.BatchNorm2d) and not bn.training and isinstance(act, nn.ReLU)
        triples.append((conv, bn))
    if not ok or not x.is_cuda:
        y = mlp(x)
        return F.max_pool2d(y, kernel_size=[1, y.size(3)]).squeeze(-1) if fuse_max else y
    B, _, M, S = x.shape
    h = x.reshape(B, x.shape[1], M * S)
    # frozen weights (the attack's victim) + the max at the end: the first layer (K = Ci + 3, not a multiple of 64)
    # stays a GEMM + tail pass, the remaining layers run on the fused HIP operator
    if (fuse_max and fuse_tail and len(triples) >= 2 and _frozen(mlp)):
        folded = _fold_triples(mlp)
        if folded is not None and _tail_eligible(folded, 1):
            (w0, b0) = folded[0]
            h = _BiasRelu.apply(torch.matmul(w0, h), b0)
            flat = [t for wb in folded[1:] for t in wb]
            return _SharedTail.apply(h.contiguous(), M, S, *flat)
    for n, (conv, bn) in enumerate(triples):
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        shift = (bn.bias - bn.running_mean * scale).contiguous()
        w = conv.weight.view(conv.out_channels, conv.in_channels) * scale.view(-1, 1)
        z = torch.matmul(w, h)                                   # [B,Co,M*S]
        if fuse_max and n == len(triples) - 1:
            return _BiasReluMax.apply(z, shift, M, S)
        h = _BiasRelu.apply(z, shift)
    return h.view(B, -1, M, S)



class _SA1Fused(torch.autograd.Function):
    """geoa3_pn2_sa1_forward / _backward: grouped xyz -> MLP 3->64->64->128 -> max over 64 samples, one wavefront per
    centroid, no activation in memory (pointnet2_sa.hip).  Differentiable in xyz and new_xyz (input-gradient only:
    the attack never needs weight gradients)."""

    @staticmethod
    def forward(ctx, xyz, new_xyz, idx, w1, b1, w2, b2, w3, b3):
        B, N, _ = xyz.shape
        M = new_xyz.shape[1]
        xyz, new_xyz = _chk(xyz.contiguous(), torch.float32), _chk(new_xyz.contiguous(), torch.float32)
        out = torch.empty(B, M, 128, device=xyz.device, dtype=torch.float32)    # centroid-major rows
        arg = torch.empty(B, M, 128, device=xyz.device, dtype=torch.uint8)
        ws = _lib.Sa1Weights(*[t.data_ptr() for t in (w1, b1, w2, b2, w3, b3)])
        check(_lib.load().geoa3_pn2_sa1_forward(xyz.data_ptr(), new_xyz.data_ptr(), _chk(idx, torch.int32).data_ptr(),
                                                ws, B, N, M, out.data_ptr(), arg.data_ptr(), _s()), "sa1_forward")
        ctx.save_for_backward(xyz, new_xyz, idx, out, arg, w1, b1, w2, b2, w3, b3)
        return out.transpose(1, 2).contiguous()                                  # [B,128,M] as the reference

    @staticmethod
    def backward(ctx, g):
        xyz, new_xyz, idx, out, arg, w1, b1, w2, b2, w3, b3 = ctx.saved_tensors
        B, N, _ = xyz.shape
        M = new_xyz.shape[1]
        gx = torch.empty_like(xyz)
        gn = torch.empty_like(new_xyz)
        ws = _lib.Sa1Weights(*[t.data_ptr() for t in (w1, b1, w2, b2, w3, b3)])
        scratch = torch.empty(B, M, 64, 3, device=xyz.device, dtype=torch.float32)   # owner-ordered scatter (deterministic)
        check(_lib.load().geoa3_pn2_sa1_backward(xyz.data_ptr(), new_xyz.data_ptr(), idx.data_ptr(), ws, B, N, M,
                                                 out.data_ptr(), arg.data_ptr(),
                                                 g.transpose(1, 2).contiguous().data_ptr(),
                                                 gx.data_ptr(), gn.data_ptr(), scratch.data_ptr(), _s()), "sa1_backward")
        return gx, gn, None, None, None, None, None, None, None



def _pretransformed_level(grouper, mlp, xyz, new_xyz, features, pretransform=True, fuse_tail=True):
    """A level with features (SA_modules[1] of the SSG classifier), frozen weights: its first layer is linear in
    the gathered inputs, W [xyz_j - c_m ; f_j] = (W_x xyz + W_f f)_j - (W_x c)_m, so it is applied to the N
    un-grouped points once (two small GEMMs) and the RESULT is gathered; the grouped [B, C+3, npoint, nsample]
    tensor, the cat and the K = C + 3 GEMM over npoint * nsample columns never exist.  The remaining layers run on
    the fused convolution operator (_SharedTail).  None when the level has another shape."""
    if not (pretransform and fuse_tail and features is not None and isinstance(grouper, QueryAndGroup) and
            grouper.use_xyz and xyz.is_cuda and not mlp.training and
            _frozen(mlp)):
        return None
    folded = _fold_triples(mlp)
    if folded is None or len(folded) < 2 or not _tail_eligible(folded, 1):
        return None
    (w0, b0) = folded[0]                                       # [Co, 3 + C]: xyz columns first (QueryAndGroup's cat)
    idx = ball_query(grouper.radius, grouper.nsample, xyz, new_xyz)
    wx, wf = w0[:, :3].contiguous(), w0[:, 3:].contiguous()
    r = torch.matmul(wf, features) + torch.matmul(wx, xyz.transpose(1, 2))          # [B, Co, N]
    shift = b0.view(1, -1, 1) - torch.matmul(wx, new_xyz.transpose(1, 2))           # [B, Co, npoint]
    flat = [t for wb in folded[1:] for t in wb]
    return _PretransformedSA.apply(r.contiguous(), idx, shift.contiguous(), *flat)

def _fused_level1(grouper, mlp, xyz, new_xyz, features, fuse_level1=True):
    """The xyz-only 3->64->64->128 level with 64 samples per ball (SA_modules[0] of the SSG classifier) as ONE
    kernel per direction; None when this level has another shape, is training, or keeps weight gradients."""
    if not (fuse_level1 and features is None and isinstance(grouper, QueryAndGroup) and grouper.use_xyz and
            grouper.nsample == 64 and xyz.is_cuda and _frozen(mlp)):
        return None
    folded = _fold_triples(mlp)
    if folded is None or [tuple(w.shape) for w, _ in folded] != [(64, 3), (64, 64), (128, 64)]:
        return None
    idx = ball_query(grouper.radius, grouper.nsample, xyz, new_xyz)
    (w1, b1), (w2, b2), (w3, b3) = folded
    return _SA1Fused.apply(xyz, new_xyz, idx, w1, b1, w2, b2, w3, b3)



def sa_module_forward(module, xyz: Tensor, features: Optional[Tensor], fuse_level1=True, fuse_tail=True, pretransform=True):
    """_PointnetSAModuleBase.forward (pointnet2_modules.py:29-74) over a geoa3_amd PointnetSAModuleMSG's groupers / MLPs."""
    new_xyz = None
    if module.npoint is not None:
        centres = furthest_point_sample(xyz, module.npoint)
        new_xyz = gather_operation(xyz.transpose(1, 2).contiguous(), centres).transpose(1, 2).contiguous()
    outs = []
    for grouper, mlp in zip(module.groupers, module.mlps):
        fused = _fused_level1(grouper, mlp, xyz, new_xyz, features, fuse_level1)
        if fused is None:
            fused = _pretransformed_level(grouper, mlp, xyz, new_xyz, features, pretransform, fuse_tail)
        if fused is not None:
            outs.append(fused)
            continue
        outs.append(run_shared_mlp(mlp, grouper(xyz, new_xyz, features), fuse_max=True, fuse_tail=fuse_tail))
    return new_xyz, torch.cat(outs, dim=1)


def module_forward(net, pointcloud: Tensor, fuse_level1=True, fuse_tail=True, pretransform=True) -> Tensor:
    """PointNet2ClassificationSSG.forward (Model/PointNetPP_ssg.py:106-124), layer by layer."""
    pc = pointcloud.transpose(2, 1)
    xyz = pc[..., 0:3].contiguous()
    features = pc[..., 3:].transpose(1, 2).contiguous() if pc.size(-1) > 3 else None
    for module in net.SA_modules:
        xyz, features = sa_module_forward(module, xyz, features, fuse_level1, fuse_tail, pretransform)
    return net.fc_layer(features.squeeze(-1))


class ModulePathNet(nn.Module):
    """A victim that evaluates `net`'s weights through the layer-by-layer composition: to attack() a generic nn.Module."""

    def __init__(self, net, **switches):
        super().__init__()
        self.net, self.switches = net, switches

    def forward(self, pointcloud: Tensor) -> Tensor:
        return module_forward(self.net, pointcloud, **self.switches)
