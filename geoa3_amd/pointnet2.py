"""PointNet++ SSG victim (BASELINE configs[3]) on the HIP set-abstraction operators.

Mirrors, for the classifier path only, the three layers of the reference's vendored package
`Model/pointnet2_ops_lib/pointnet2_ops`:
  * `ext`  -- the attribute surface of the native module `_ext` (_ext-src/src/bindings.cpp:6-19), each function a
    thin call into libgeoa3_hip.so (geoa3_pn2_*), same argument order, dtypes and fresh-output convention;
  * the autograd Functions / groupers of pointnet2_utils.py:34-101,194-379;
  * the set-abstraction modules of pointnet2_modules.py:9-146 and `PointNet2ClassificationSSG`
    (Model/PointNetPP_ssg.py:51-124) with the reference's state_dict keys (SA_modules.{0,1,2}.mlps.0.{0..8}.*,
    fc_layer.{0,1,3,4,7}.*), so `load_state_dict(torch.load(...)['state_dict'])` works (main_attack.py:139-145).

ONE execution path: the whole classifier, forward and input gradient, is one call each into libgeoa3_hip.so
(`geoa3_pn2ssg_forward / _backward`, csrc/pointnet2_net.hip): FPS, ball query, the fused levels, the split-fp16 1x1
convolutions, the pooled layers and the FC head are hand-written HIP kernels; no torch operator and no library GEMM runs
in between.  It serves what the attack needs -- an eval-mode victim, xyz-only input, d logits / d input -- and refuses
everything else loudly (training-mode statistics, weight gradients, extra feature channels: the reference's training is
outside SURVEY 8).  The operator boundary (`ext`, the autograd Functions, the groupers) is usable on its own.
`ext_contract` / GEOA3_PN2_CONTRACT=1: the sampler's and the ball queries' squared distances as nvcc's default
contraction forms them; default: un-fused, as the CPU oracle.
GPU tensors only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional

import torch
import torch.nn as nn

from . import _lib
from ._lib import check

Tensor = torch.Tensor


def _s() -> int:
    return torch.cuda.current_stream().cuda_stream


def _chk(t: Tensor, dtype) -> Tensor:
    if not t.is_cuda:
        raise _lib.Geoa3Error("CPU not supported")          # the reference asserts the same (ball_query.cpp:27-29)
    if t.dtype != dtype:
        raise _lib.Geoa3Error("expected %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise _lib.Geoa3Error("tensor must be contiguous")  # utils.h:5-25 CHECK_CONTIGUOUS
    return t


def _frozen(mlp) -> bool:
    """True when no weight gradient can be asked of `mlp` in this call (its parameters are frozen, or autograd is off):
    the fused HIP paths only form the input gradient.  Under torch.no_grad() the answer does not depend on the
    parameters' flags, so evaluation before and after an attack runs the same kernels."""
    return not torch.is_grad_enabled() or not any(p.requires_grad for p in mlp.parameters())


def ext_contract_default() -> bool:
    """GEOA3_PN2_CONTRACT=1: the `_ext` distances contracted (fmaf(dz, dz, fmaf(dy, dy, dx * dx))), as nvcc -O3 with its
    default -fmad=true most likely compiled sampling_gpu.cu:100,103-104 / ball_query_gpu.cu:31-32; default un-fused."""
    return os.environ.get("GEOA3_PN2_CONTRACT", "0") == "1"


def _ext_flags(contract: Optional[bool]) -> int:
    return _lib.PN2_CONTRACT if (ext_contract_default() if contract is None else contract) else 0


class _Ext:
    """`pointnet2_ops._ext` replacement: same six entry points, backed by the C ABI (`contract`: see
    ext_contract_default; None = the environment's choice)."""

    @staticmethod
    def furthest_point_sampling(xyz: Tensor, nsamples: int, contract: Optional[bool] = None) -> Tensor:
        _chk(xyz, torch.float32)
        B, N, _ = xyz.shape
        out = torch.zeros(B, nsamples, device=xyz.device, dtype=torch.int32)
        check(_lib.load().geoa3_pn2_furthest_point_sampling_ex(xyz.data_ptr(), B, N, nsamples, None, out.data_ptr(),
                                                               _ext_flags(contract), _s()), "furthest_point_sampling")
        return out

    @staticmethod
    def gather_points(points: Tensor, idx: Tensor) -> Tensor:
        _chk(points, torch.float32), _chk(idx, torch.int32)
        B, C, N = points.shape
        M = idx.shape[1]
        out = torch.empty(B, C, M, device=points.device, dtype=torch.float32)
        check(_lib.load().geoa3_pn2_gather_points(points.data_ptr(), idx.data_ptr(), B, C, N, M, out.data_ptr(), _s()),
              "gather_points")
        return out

    @staticmethod
    def gather_points_grad(grad_out: Tensor, idx: Tensor, n: int) -> Tensor:
        _chk(grad_out, torch.float32), _chk(idx, torch.int32)
        B, C, M = grad_out.shape
        out = torch.empty(B, C, n, device=grad_out.device, dtype=torch.float32)
        check(_lib.load().geoa3_pn2_gather_points_grad(grad_out.data_ptr(), idx.data_ptr(), B, C, n, M,
                                                       out.data_ptr(), _s()), "gather_points_grad")
        return out

    @staticmethod
    def ball_query(new_xyz: Tensor, xyz: Tensor, radius: float, nsample: int, contract: Optional[bool] = None) -> Tensor:
        _chk(new_xyz, torch.float32), _chk(xyz, torch.float32)
        B, M, _ = new_xyz.shape
        N = xyz.shape[1]
        out = torch.empty(B, M, nsample, device=xyz.device, dtype=torch.int32)
        check(_lib.load().geoa3_pn2_ball_query_ex(new_xyz.data_ptr(), xyz.data_ptr(), B, N, M, float(radius), nsample,
                                                  out.data_ptr(), _ext_flags(contract), _s()), "ball_query")
        return out

    @staticmethod
    def group_points(points: Tensor, idx: Tensor) -> Tensor:
        _chk(points, torch.float32), _chk(idx, torch.int32)
        B, C, N = points.shape
        _, M, S = idx.shape
        out = torch.empty(B, C, M, S, device=points.device, dtype=torch.float32)
        check(_lib.load().geoa3_pn2_group_points(points.data_ptr(), idx.data_ptr(), B, C, N, M, S, out.data_ptr(),
                                                 _s()), "group_points")
        return out

    @staticmethod
    def group_points_grad(grad_out: Tensor, idx: Tensor, n: int) -> Tensor:
        _chk(grad_out, torch.float32), _chk(idx, torch.int32)
        B, C, M, S = grad_out.shape
        out = torch.empty(B, C, n, device=grad_out.device, dtype=torch.float32)
        check(_lib.load().geoa3_pn2_group_points_grad(grad_out.data_ptr(), idx.data_ptr(), B, C, n, M, S,
                                                      out.data_ptr(), _s()), "group_points_grad")
        return out


ext = _Ext()


# ------------------------------------------------------------------ pointnet2_utils.py:34-101,194-276
class _FPS(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xyz, npoint):
        out = ext.furthest_point_sampling(xyz, npoint)
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, g):
        return None, None


class _Gather(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n = features.size(2)
        return ext.gather_points(features, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return ext.gather_points_grad(g.contiguous(), idx, ctx.n), None


class _Group(torch.autograd.Function):
    @staticmethod
    def forward(ctx, features, idx):
        ctx.save_for_backward(idx)
        ctx.n = features.size(2)
        return ext.group_points(features, idx)

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        return ext.group_points_grad(g.contiguous(), idx, ctx.n), None


class _BallQuery(torch.autograd.Function):
    @staticmethod
    def forward(ctx, radius, nsample, xyz, new_xyz):
        out = ext.ball_query(new_xyz, xyz, radius, nsample)   # note the argument order of the native call
        ctx.mark_non_differentiable(out)
        return out

    @staticmethod
    def backward(ctx, g):
        return None, None, None, None


furthest_point_sample = _FPS.apply
gather_operation = _Gather.apply
grouping_operation = _Group.apply
ball_query = _BallQuery.apply


class QueryAndGroup(nn.Module):
    """pointnet2_utils.py:279-333: ball query, group xyz (re-centred on the ball centre) and features."""

    def __init__(self, radius: float, nsample: int, use_xyz: bool = True):
        super().__init__()
        self.radius, self.nsample, self.use_xyz = radius, nsample, use_xyz

    def forward(self, xyz: Tensor, new_xyz: Tensor, features: Optional[Tensor] = None) -> Tensor:
        idx = ball_query(self.radius, self.nsample, xyz, new_xyz)
        grouped_xyz = grouping_operation(xyz.transpose(1, 2).contiguous(), idx)
        grouped_xyz = grouped_xyz - new_xyz.transpose(1, 2).unsqueeze(-1)
        if features is None:
            assert self.use_xyz, "Cannot have not features and not use xyz as a feature!"
            return grouped_xyz
        grouped = grouping_operation(features, idx)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped


class GroupAll(nn.Module):
    """pointnet2_utils.py:336-379: one group holding every point."""

    def __init__(self, use_xyz: bool = True):
        super().__init__()
        self.use_xyz = use_xyz

    def forward(self, xyz: Tensor, new_xyz: Optional[Tensor], features: Optional[Tensor] = None) -> Tensor:
        grouped_xyz = xyz.transpose(1, 2).unsqueeze(2)
        if features is None:
            return grouped_xyz
        grouped = features.unsqueeze(2)
        return torch.cat([grouped_xyz, grouped], dim=1) if self.use_xyz else grouped


# ------------------------------------------------------------------ pointnet2_modules.py:9-146
def build_shared_mlp(mlp_spec: List[int], bn: bool = True) -> nn.Sequential:
    layers: List[nn.Module] = []
    for cin, cout in zip(mlp_spec[:-1], mlp_spec[1:]):
        layers.append(nn.Conv2d(cin, cout, kernel_size=1, bias=not bn))
        if bn:
            layers.append(nn.BatchNorm2d(cout))
        layers.append(nn.ReLU(True))
    return nn.Sequential(*layers)


def _fold_triples(mlp: nn.Sequential):
    """[(W [Co,Ci] with the BatchNorm scale folded in, shift [Co])] of an eval-mode build_shared_mlp stack, or None
    when the stack is not Conv2d 1x1 (no bias) + eval BatchNorm2d + ReLU triples."""
    layers = list(mlp)
    if len(layers) % 3 != 0:
        return None
    out = []
    for i in range(0, len(layers), 3):
        conv, bn, act = layers[i], layers[i + 1], layers[i + 2]
        if not (isinstance(conv, nn.Conv2d) and conv.kernel_size == (1, 1) and conv.bias is None and
                isinstance(bn, nn.BatchNorm2d) and not bn.training and isinstance(act, nn.ReLU)):
            return None
        scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
        out.append(((conv.weight.view(conv.out_channels, conv.in_channels) * scale.view(-1, 1)).float().contiguous(),
                    (bn.bias - bn.running_mean * scale).float().contiguous()))
    return out


class PointnetSAModuleMSG(nn.Module):
    """pointnet2_modules.py:77-126: holds the level's groupers and shared MLPs under the reference's parameter names
    (`mlps.0.{0,1,3,4,6,7}.*`), so checkpoints load unchanged.  The levels are EXECUTED by the native classifier
    (geoa3_pn2ssg_forward / _backward through PointNet2ClassificationSSG.forward); the layer-by-layer composition of
    torch modules is not part of this package (tests/_pn2_module_path.py keeps one as a checker)."""

    def __init__(self, npoint, radii, nsamples, mlps, bn=True, use_xyz=True):
        super().__init__()
        assert len(radii) == len(nsamples) == len(mlps)
        self.npoint = npoint
        self.groupers = nn.ModuleList()
        self.mlps = nn.ModuleList()
        for radius, nsample, spec in zip(radii, nsamples, mlps):
            self.groupers.append(QueryAndGroup(radius, nsample, use_xyz=use_xyz) if npoint is not None
                                 else GroupAll(use_xyz))
            spec = list(spec)
            if use_xyz:
                spec[0] += 3
            self.mlps.append(build_shared_mlp(spec, bn))

    def forward(self, xyz: Tensor, features: Optional[Tensor]):
        raise _lib.Geoa3Error("a set-abstraction level does not run on its own in geoa3_amd: the SSG classifier is one native "
                              "forward / input-gradient (PointNet2ClassificationSSG.forward, eval mode)")


class PointnetSAModule(PointnetSAModuleMSG):
    def __init__(self, mlp, npoint=None, radius=None, nsample=None, bn=True, use_xyz=True):
        super().__init__(mlps=[mlp], npoint=npoint, radii=[radius], nsamples=[nsample], bn=bn, use_xyz=use_xyz)


def _fold_linear_bn(lin: nn.Linear, bn: Optional[nn.BatchNorm1d]):
    """Linear (+ eval BatchNorm1d) -> (W', shift) in float64 -> float32."""
    w = lin.weight.detach().double()
    b = lin.bias.detach().double() if lin.bias is not None else torch.zeros(w.shape[0], dtype=torch.float64,
                                                                              device=w.device)
    if bn is not None:
        scale = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
        w = w * scale.view(-1, 1)
        b = (b - bn.running_mean.detach().double()) * scale + bn.bias.detach().double()
    return w.float().contiguous(), b.float().contiguous()


class PackedSSG:
    """Folded device weights of the SSG classifier + the ctypes struct of geoa3_pn2ssg_forward / _backward."""

    def __init__(self, net: "PointNet2ClassificationSSG", device):
        self._keep = []

        def dev(t: Tensor) -> int:
            d = t.detach().to(device=device, dtype=torch.float32).contiguous()
            self._keep.append(d)
            return d.data_ptr()

        folded = [_fold_triples(m.mlps[0]) for m in net.SA_modules]
        if any(f is None for f in folded):
            raise _lib.Geoa3Error("the native SSG path needs eval-mode Conv2d 1x1 (no bias) + BatchNorm2d + ReLU stacks")
        shapes = [[tuple(w.shape) for w, _ in f] for f in folded]
        if shapes != [[(64, 3), (64, 64), (128, 64)], [(128, 131), (128, 128), (256, 128)],
                      [(256, 259), (512, 256), (1024, 512)]]:
            raise _lib.Geoa3Error("the native SSG path is built for the classifier of Model/PointNetPP_ssg.py:58-82, "
                                  "got layer shapes %s" % (shapes,))
        (w11, b11), (w12, b12), (w13, b13) = folded[0]
        f = {}
        for lvl, fl in ((2, folded[1]), (3, folded[2])):
            (w0, b0), (w1, b1), (w2, b2) = fl
            f["sa%d_wx" % lvl], f["sa%d_wf" % lvl], f["sa%d_b0" % lvl] = w0[:, :3], w0[:, 3:], b0
            f["sa%d_wft" % lvl] = w0[:, 3:].t()
            f["sa%d_w1" % lvl], f["sa%d_b1" % lvl], f["sa%d_w1t" % lvl] = w1, b1, w1.t()
            f["sa%d_w2" % lvl], f["sa%d_b2" % lvl], f["sa%d_w2t" % lvl] = w2, b2, w2.t()
        fc = net.fc_layer
        for i, (lin, bn) in enumerate(((fc[0], fc[1]), (fc[3], fc[4]), (fc[7], None)), start=1):
            w, b = _fold_linear_bn(lin, bn)
            f["f%d" % i], f["fb%d" % i], f["f%dt" % i] = w, b, w.t()
        self.classes = int(fc[7].out_features)
        sa1 = _lib.Sa1Weights(*[dev(t) for t in (w11, b11, w12, b12, w13, b13)])
        self.struct = _lib.Pn2SsgWeights(classes=self.classes, sa1=sa1, images=None, **{k: dev(v) for k, v in f.items()})
        self.struct.flags = _ext_flags(getattr(net, "ext_contract", None))
        # the level-2 / level-3 matrices once more, as the split-fp16 fragment images the matrix-core loops read
        lib = _lib.load()
        images = torch.empty(int(lib.geoa3_pn2ssg_images_bytes()), dtype=torch.uint8, device=device)
        check(lib.geoa3_pn2ssg_pack_images(C.byref(self.struct), images.data_ptr(), _s()), "geoa3_pn2ssg_pack_images")
        self._keep.append(images)
        self.struct.images = images.data_ptr()
        # a side queue (one stream + two events, owned by this object): level 2's sampling / ball query run beside level 1's MLP
        # (csrc/pointnet2_net.hip; GEOA3_PN2_SIDE=0: one stream, e.g. for per-kernel traces)
        self._side = None
        if os.environ.get("GEOA3_PN2_SIDE", "1") != "0":
            with torch.cuda.device(device):
                self._side = lib.geoa3_side_queue_create()
            if not self._side:
                raise _lib.Geoa3Error("geoa3_side_queue_create failed")
            self.struct.side = self._side

    def __del__(self):
        side, self._side = getattr(self, "_side", None), None
        if side:
            try:
                _lib.load().geoa3_side_queue_destroy(side)
            except Exception:   # interpreter shutdown: the library or the runtime may be gone
                pass


class _SSGFn(torch.autograd.Function):
    """logits = geoa3_pn2ssg_forward(x); d logits -> d x = geoa3_pn2ssg_backward (same workspace)."""

    @staticmethod
    def forward(ctx, x: Tensor, packed: PackedSSG, ws_cache: dict):
        x = x.contiguous().float()
        B, _, N = x.shape
        lib = _lib.load()
        nbytes = lib.geoa3_pn2ssg_workspace_bytes(B, N)
        if nbytes < 0:
            raise _lib.Geoa3Error("geoa3_pn2ssg: unsupported cloud size (B=%d, N=%d)" % (B, N))
        ws = ws_cache.get("ws")
        if ws is None or ws.numel() < nbytes or ws.device != x.device:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            ws_cache["ws"] = ws
        logits = torch.empty(B, packed.classes, device=x.device, dtype=torch.float32)
        check(lib.geoa3_pn2ssg_forward(C.byref(packed.struct), x.data_ptr(), B, N, logits.data_ptr(), ws.data_ptr(), _s()),
              "geoa3_pn2ssg_forward")
        ctx.packed, ctx.ws, ctx.ws_cache = packed, ws, ws_cache
        ctx.ws_version = ws_cache["version"] = ws_cache.get("version", 0) + 1
        ctx.save_for_backward(x)
        return logits

    @staticmethod
    def backward(ctx, g: Tensor):
        (x,) = ctx.saved_tensors
        if ctx.ws_cache.get("version") != ctx.ws_version:
            raise _lib.Geoa3Error("the SSG workspace was overwritten by a later forward before backward ran")
        B, _, N = x.shape
        dx = torch.empty_like(x)
        check(_lib.load().geoa3_pn2ssg_backward(C.byref(ctx.packed.struct), x.data_ptr(),
                                                g.contiguous().float().data_ptr(), B, N, dx.data_ptr(),
                                                ctx.ws.data_ptr(), _s()), "geoa3_pn2ssg_backward")
        return dx, None, None


class PointNet2ClassificationSSG(nn.Module):
    """Model/PointNetPP_ssg.py:51-124 (40 output classes hard-coded as in the reference, main_attack.py:140)."""

    def __init__(self, use_xyz: bool = True, use_normal: bool = False, ext_contract: Optional[bool] = None):
        super().__init__()
        self.use_xyz, self.use_normal = use_xyz, use_normal
        self.ext_contract = ext_contract   # None: GEOA3_PN2_CONTRACT decides (ext_contract_default)
        c0 = 3 if use_normal else 0
        self.SA_modules = nn.ModuleList([
            PointnetSAModule(npoint=512, radius=0.2, nsample=64, mlp=[c0, 64, 64, 128], use_xyz=use_xyz),
            PointnetSAModule(npoint=128, radius=0.4, nsample=64, mlp=[128, 128, 128, 256], use_xyz=use_xyz),
            PointnetSAModule(mlp=[256, 256, 512, 1024], use_xyz=use_xyz),
        ])
        self.fc_layer = nn.Sequential(
            nn.Linear(1024, 512, bias=False), nn.BatchNorm1d(512), nn.ReLU(True),
            nn.Linear(512, 256, bias=False), nn.BatchNorm1d(256), nn.ReLU(True),
            nn.Dropout(0.5), nn.Linear(256, 40))

    def __getstate__(self):
        """copy.deepcopy / pickle: the packed weights (ctypes struct, device images, the side queue's stream) and the
        workspace belong to THIS object and are rebuilt by the copy's first forward."""
        state = dict(self.__dict__)
        for name in ("_packed", "_packed_key", "_ws_cache"):
            state.pop(name, None)
        return state

    def _weights_key(self, device):
        return (str(device), _ext_flags(self.ext_contract)) + tuple(
            (p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def packed(self, device) -> PackedSSG:
        key = self._weights_key(device)
        if getattr(self, "_packed", None) is None or key != self._packed_key:
            self._packed, self._packed_key = PackedSSG(self, device), key
        return self._packed

    MIN_POINTS = 32   # (a sampler block needs points to choose from; the reference samples 512 centroids of any cloud)

    def native_eligible(self, pointcloud: Tensor) -> bool:
        """What the native kernels serve: an eval-mode xyz-only victim whose weights take no gradient (they form the input
        gradient only and fold the BatchNorm running statistics) on a device tensor [B,3,N]."""
        return (pointcloud.is_cuda and not self.training and pointcloud.dim() == 3 and pointcloud.size(1) == 3 and
                pointcloud.size(2) >= self.MIN_POINTS and not self.use_normal and self.use_xyz and _frozen(self))

    def forward(self, pointcloud: Tensor) -> Tensor:
        """pointcloud [B, 3, N] (the attack's layout) -> logits [B, 40]"""
        if not self.native_eligible(pointcloud):
            raise _lib.Geoa3Error(
                "PointNet2ClassificationSSG runs as the attack's victim only: eval() mode, a CUDA tensor [B,3,N>=%d] (xyz "
                "only, use_normal=False), and no weight gradient (torch.no_grad() or requires_grad_(False) on the "
                "parameters) -- got training=%s, shape=%s, device=%s, use_normal=%s, weights frozen=%s.  Training and "
                "feature-channel inputs are outside this package (SURVEY 8)."
                % (self.MIN_POINTS, self.training, tuple(pointcloud.shape), pointcloud.device, self.use_normal, _frozen(self)))
        if not hasattr(self, "_ws_cache"):
            self._ws_cache = {}
        return _SSGFn.apply(pointcloud, self.packed(pointcloud.device), self._ws_cache)
