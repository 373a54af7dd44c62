"""Module-level boundary: a PointNet nn.Module with the reference's constructor signature
and state_dict layout (Model/PointNet.py:56-179: 110 entries), whose eval-mode forward and input
gradient run in the HIP library (geoa3_pointnet_forward / geoa3_pointnet_backward).

BatchNorm (eval: an affine map from the running statistics) is folded into the preceding conv / linear
layer on the host, in float64, once per weight version; the packed buffers are what the C ABI consumes.
There is no CPU forward: a CPU input raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import _lib
from ._lib import PointNetWeights, TnetWeights, check

Tensor = torch.Tensor


# --------------------------------------------------------------------------------------------
# host-side weight folding / packing (pure tensor algebra, runs on any device)
# --------------------------------------------------------------------------------------------
def _fold(w: Tensor, b: Tensor, sd: Dict[str, Tensor], bn: Optional[str], eps: float):
    """(W, bias) of `bn(conv(x))` as one affine map.  w: [Co, ...]."""
    w64, b64 = w.double(), b.double()
    if bn is None:
        return w64, b64
    g, beta = sd[bn + ".weight"].double(), sd[bn + ".bias"].double()
    mean, var = sd[bn + ".running_mean"].double(), sd[bn + ".running_var"].double()
    s = g / torch.sqrt(var + eps)
    return w64 * s.view(-1, *([1] * (w64.dim() - 1))), (b64 - mean) * s + beta


def pack_wide_fragments(w: Tensor, taps: int) -> Tensor:
    """[Co, taps*128] (k = tap*128 + ci) -> MFMA A-fragment order consumed by wide_max_kernel:
    out[((T*taps + tap)*16 + j)*64 + lane, i] = w[32T + (lane & 31), tap*128 + 8j + 4(lane >> 5) + i]."""
    co = w.shape[0]
    v = w.reshape(co // 32, 32, taps, 16, 2, 4)          # T, r, tap, j, h, i
    return v.permute(0, 2, 3, 4, 1, 5).reshape(-1, 4).contiguous()   # T, tap, j, (h, r) = lane, i


def pack_wide_split(w: Tensor, taps: int):
    """[Co, taps*128] fp32 -> (fragments, unscale) for csrc/pointnet_wide_split.hip: v = w * 2^e (max |v| in
    [2^13, 2^14)), hi = rn16(v), lo = rn16(v - hi) -- |v - hi - lo| <= 2^-24 |v| while lo is a normal fp16, i.e. for
    every weight above 2^-16 of the largest -- as an fp16 tensor
    [T = Co/32][s = K/16][piece][lane = 32h + r][j]  =  piece(w[32T + r][16s + 8h + j]);  unscale = 2^-e."""
    w = w.float()
    co, K = w.shape
    amax = float(w.abs().max())
    e = 13 - int(torch.frexp(torch.tensor(amax)).exponent) + 1 if amax > 0 else 0   # amax * 2^e in [2^13, 2^14)
    v = torch.ldexp(w, torch.tensor(e))
    hi = v.half()
    lo = (v - hi.float()).half()
    frag = torch.stack((hi, lo), 0).reshape(2, co // 32, 32, K // 16, 2, 8)       # p, T, r, s, h, j
    frag = frag.permute(1, 3, 0, 4, 2, 5).contiguous().reshape(-1, 8)              # T, s, p, (h, r), j
    return frag, float(2.0 ** -e)


def pack_wide_split16(w: Tensor):
    """pack_wide_split for the 16x16x32 MFMA shape (csrc/pointnet_wide16.hip): the same scaled hi / lo pieces as
    [T = Co/16][s = K/32][piece][lane = 16q + r][j] = piece(w[16T + r][32s + 8q + j])."""
    w = w.float()
    co, K = w.shape
    amax = float(w.abs().max())
    e = 13 - int(torch.frexp(torch.tensor(amax)).exponent) + 1 if amax > 0 else 0
    v = torch.ldexp(w, torch.tensor(e))
    hi = v.half()
    lo = (v - hi.float()).half()
    frag = torch.stack((hi, lo), 0).reshape(2, co // 16, 16, K // 32, 4, 8)       # p, T, r, s, q, j
    return frag.permute(1, 3, 0, 4, 2, 5).contiguous().reshape(-1, 8), float(2.0 ** -e)   # T, s, p, (q, r), j


def wide_shape(layer: str = "conv5") -> int:
    """MFMA shape of a split 1024-wide layer: 16 = v_mfma_f32_16x16x32_f16 (pointnet_wide16.hip), 32 =
    v_mfma_f32_32x32x16_f16 (pointnet_wide_split.hip).  Measured on MI355X: conv5 (K = 384 per group) 0.495 -> 0.444 ms on
    the 16x16x32 shape, the T-Nets' conv3 (K = 128 per group: a third of the MFMAs between two epilogues) 0.182 -> 0.292 ms
    -- so conv5 takes 16, the T-Nets 32.  (GEOA3_WIDE_SHAPE_TNET / GEOA3_WIDE_SHAPE_CONV5 = 16 | 32: A/B runs of tools/.)"""
    env = os.environ.get("GEOA3_WIDE_SHAPE_CONV5" if layer == "conv5" else "GEOA3_WIDE_SHAPE_TNET")
    if env in ("16", "32"):
        return int(env)
    return 16 if layer == "conv5" else 32


def pack_tnet(sd: Dict[str, Tensor], prefix: str, K: int) -> Dict[str, Tensor]:
    eps = 1e-3  # transform_net.eps, Model/PointNet.py:59
    sub = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    out: Dict[str, Tensor] = {}
    w1, out["b1"] = _fold(sub["conv1.weight"], sub["conv1.bias"], sub, "bn1", eps)
    w2, out["b2"] = _fold(sub["conv2.weight"], sub["conv2.bias"], sub, "bn2", eps)
    w3, out["b3"] = _fold(sub["conv3.weight"], sub["conv3.bias"], sub, "bn3", eps)
    out["w1"], out["w2"], out["w3"] = w1.squeeze(-1), w2.squeeze(-1), w3.squeeze(-1)
    out["w2t"] = out["w2"].t()
    out["w3p"] = pack_wide_fragments(out["w3"], 1)
    out["f1"], out["fb1"] = _fold(sub["fc1.weight"], sub["fc1.bias"], sub, "bn4", eps)
    out["f2"], out["fb2"] = _fold(sub["fc2.weight"], sub["fc2.bias"], sub, "bn5", eps)
    out["f3"], out["fb3"] = _fold(sub["fc3.weight"], sub["fc3.bias"], sub, None, eps)
    for n in ("f1", "f2", "f3"):
        out[n + "t"] = out[n].t()
    out = {k: v.float().contiguous() for k, v in out.items()}
    out["w3h"], out["w3h_unscale"] = pack_wide_split(out["w3"], 1)
    out["w2h"], out["w2h_unscale"] = pack_wide_split(out["w2"], 1)
    out["w3h16"], _ = pack_wide_split16(out["w3"])
    out["w2th"], out["w2th_unscale"] = pack_wide_split(out["w2t"], 1)
    return out


def pack_pointnet(sd: Dict[str, Tensor]) -> Dict[str, object]:
    """state_dict (reference layout) -> {'t3': {...}, 't64': {...}, 'w1': ..., ...} fp32 contiguous."""
    eps = 1e-3  # PointNet.eps for bn1..bn5 (Model/PointNet.py:100); bn6/bn7 use the default 1e-5 (:119,122)
    out: Dict[str, object] = {"t3": pack_tnet(sd, "input_transform.", 3),
                              "t64": pack_tnet(sd, "feature_transform.", 64)}
    t: Dict[str, Tensor] = {}
    for i, name in enumerate(["conv1", "conv2", "conv3", "conv4"], 1):
        w, t["b%d" % i] = _fold(sd[name + ".weight"], sd[name + ".bias"], sd, "bn%d" % i, eps)
        t["w%d" % i] = w.squeeze(-1)
    w5, t["b5"] = _fold(sd["conv5.weight"], sd["conv5.bias"], sd, "bn5", eps)   # [1024,128,3]
    t["w5"] = w5.permute(0, 2, 1).reshape(w5.shape[0], -1)                         # k = tap*128 + ci
    t["w5p"] = pack_wide_fragments(t["w5"], 3)
    t["w4t"], t["w3t"], t["w2t"] = t["w4"].t(), t["w3"].t(), t["w2"].t()
    t["f1"], t["fb1"] = _fold(sd["fc1.weight"], sd["fc1.bias"], sd, "bn6", 1e-5)
    t["f2"], t["fb2"] = _fold(sd["fc2.weight"], sd["fc2.bias"], sd, "bn7", 1e-5)
    t["f3"], t["fb3"] = _fold(sd["fc3.weight"], sd["fc3.bias"], sd, None, eps)
    for n in ("f1", "f2", "f3"):
        t[n + "t"] = t[n].t()
    out.update({k: v.float().contiguous() for k, v in t.items()})
    out["w5h"], out["w5h_unscale"] = pack_wide_split(out["w5"], 3)
    out["w4h"], out["w4h_unscale"] = pack_wide_split(out["w4"], 1)
    out["w5h16"], _ = pack_wide_split16(out["w5"])
    out["w4th"], out["w4th_unscale"] = pack_wide_split(out["w4t"], 1)
    return out


WIDE_MODES = ("f32", "f16x2")


def default_wide_mode() -> str:
    """Arithmetic of the network's convolutions (the three 1024-wide layers, the 64/128-wide layers in both directions,
    the Gram product of the backward).  'f16x2' (default): every fp32 operand carried as two fp16 values on the f16
    matrix pipe, fp32 accumulation (csrc/pointnet_wide_split.hip, pointnet_conv_split.hip, pointnet_gram.hip; error
    against float64 no larger than the fp32 MFMA kernels', iteration 1.8x faster).  'f32': fp32 MFMA throughout (exact
    fmaf chains).  GEOA3_WIDE_MODE overrides."""
    mode = os.environ.get("GEOA3_WIDE_MODE", "f16x2")
    if mode not in WIDE_MODES:
        raise ValueError("GEOA3_WIDE_MODE must be one of %s" % (WIDE_MODES,))
    return mode


def fuse_front() -> bool:
    """The 64 -> 128 layers in front of the three 1024-wide layers evaluated inside the wide kernels' staging pass (the
    library supports it: geoa3_tnet_weights.w2h / geoa3_pointnet_weights.w4h).  Same results to rounding, but measured
    SLOWER on MI355X (conv5 0.50 -> 0.71 ms against 52-60 us per convolution saved): never handed over by this host."""
    return False


# A/B switches of the library (geoa3_pointnet_weights.flags; the library itself reads no environment): bit 0 / bit 1 select
# the unfused sparse backward / the one-layer-per-launch trunk, bit 3 (8) the sparse backward that builds its hit lists per
# workgroup instead of reading the forward's (same bits either way, tests/test_gpu_pointnet.py)
AB_FLAGS = 0


def ab_flags() -> int:
    return int(AB_FLAGS)


class PackedPointNet:
    """Device copies of the packed weights + the ctypes struct handed to the library."""

    def __init__(self, sd: Dict[str, Tensor], device: torch.device, wide_mode: Optional[str] = None):
        packed = pack_pointnet({k: v.detach().cpu() for k, v in sd.items()})
        self.classes = int(packed["f3"].shape[0])
        self.wide_mode = wide_mode or default_wide_mode()
        split = self.wide_mode == "f16x2"
        self._keep = []

        def dev(t) -> Optional[int]:
            if not isinstance(t, Tensor):
                return t                       # the float scales
            d = t.to(device).contiguous()
            self._keep.append(d)
            return d.data_ptr()

        def pick(p: Dict[str, object], name: str):
            if name in ("w3h", "w5h", "w2h", "w4h") and (not split or (name in ("w2h", "w4h") and not fuse_front())):
                return None
            if name in ("w2th", "w4th") and not split:
                return None
            if name == "w5h16" and (not split or wide_shape("conv5") != 16):
                return None
            if name == "w3h16" and (not split or wide_shape("tnet") != 16):
                return None
            return dev(p[name])

        def tnet(p: Dict[str, Tensor], K: int) -> TnetWeights:
            return TnetWeights(K=K, **{f[0]: pick(p, f[0]) for f in TnetWeights._fields_ if f[0] != "K"})

        fields = {f[0]: pick(packed, f[0]) for f in PointNetWeights._fields_
                  if f[0] not in ("classes", "t3", "t64", "flags")}
        fields["flags"] = ab_flags()
        self.struct = PointNetWeights(classes=self.classes, t3=tnet(packed["t3"], 3), t64=tnet(packed["t64"], 64),
                                      **fields)


class _PointNetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, packed: PackedPointNet, ws_cache: dict):
        if not x.is_cuda:
            raise _lib.Geoa3Error("geoa3_amd.PointNet runs on the GPU only (no CPU path)")
        x = x.contiguous().float()
        B, _, N = x.shape
        lib = _lib.load()
        nbytes = lib.geoa3_pointnet_workspace_bytes(B, N, packed.classes)
        ws = ws_cache.get("ws")
        if ws is None or ws.numel() < nbytes or ws.device != x.device:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
            ws_cache["ws"] = ws
        logits = torch.empty(B, packed.classes, device=x.device, dtype=torch.float32)
        s = torch.cuda.current_stream().cuda_stream
        packed.struct.flags = ab_flags()
        check(lib.geoa3_pointnet_forward(C.byref(packed.struct), x.data_ptr(), B, N, logits.data_ptr(),
                                         ws.data_ptr(), s), "geoa3_pointnet_forward")
        ctx.packed, ctx.ws = packed, ws
        ctx.ws_version = ws_cache["version"] = ws_cache.get("version", 0) + 1
        ws_cache.pop("holds", None)     # the custom-op backward must not take these activations for its own x
        ctx.ws_cache = ws_cache
        ctx.save_for_backward(x)
        return logits

    @staticmethod
    def backward(ctx, g: Tensor):
        (x,) = ctx.saved_tensors
        if ctx.ws_cache.get("version") != ctx.ws_version:
            raise _lib.Geoa3Error("PointNet workspace was overwritten by a later forward before backward ran")
        B, _, N = x.shape
        dx = torch.empty_like(x)
        s = torch.cuda.current_stream().cuda_stream
        check(_lib.load().geoa3_pointnet_backward(C.byref(ctx.packed.struct), x.data_ptr(),
                                                  g.contiguous().float().data_ptr(), B, N, dx.data_ptr(),
                                                  ctx.ws.data_ptr(), s), "geoa3_pointnet_backward")
        return dx, None, None


class _TransformNet(nn.Module):
    """Parameter container with the layout of transform_net (Model/PointNet.py:56-94)."""

    def __init__(self, K: int = 3):
        super().__init__()
        self.K = K
        self.conv1, self.conv2, self.conv3 = nn.Conv1d(K, 64, 1), nn.Conv1d(64, 128, 1), nn.Conv1d(128, 1024, 1)
        self.fc1, self.fc2, self.fc3 = nn.Linear(1024, 512), nn.Linear(512, 256), nn.Linear(256, K * K)
        self.bn1, self.bn2, self.bn3 = (nn.BatchNorm1d(c, eps=1e-3) for c in (64, 128, 1024))
        self.bn4, self.bn5 = nn.BatchNorm1d(512, eps=1e-3), nn.BatchNorm1d(256, eps=1e-3)
        with torch.no_grad():  # Model/PointNet.py:89-94
            for m in (self.conv1, self.conv2, self.conv3, self.fc1, self.fc2):
                nn.init.xavier_uniform_(m.weight)
                m.bias.zero_()
            self.fc3.weight.zero_()
            self.fc3.bias.copy_(torch.eye(K).view(-1))


class PointNet(nn.Module):
    """Drop-in for Model.PointNet.PointNet(classes, return_idx=False, npoint=1024) in EVAL mode.
    `net.load_state_dict(torch.load(path)['state_dict'])` works unchanged (main_attack.py:144-145)."""

    def __init__(self, classes: int, return_idx: bool = False, npoint: int = 1024):
        super().__init__()
        if return_idx:
            raise NotImplementedError("return_idx is not on the attack path")
        self.num_class = classes
        self.input_transform = _TransformNet(3)
        self.feature_transform = _TransformNet(64)
        self.conv1, self.conv2, self.conv3 = nn.Conv1d(3, 64, 1), nn.Conv1d(64, 64, 1), nn.Conv1d(64, 64, 1)
        self.conv4, self.conv5 = nn.Conv1d(64, 128, 1), nn.Conv1d(128, 1024, 3, 1, 1)
        self.bn1, self.bn2, self.bn3 = (nn.BatchNorm1d(64, eps=1e-3) for _ in range(3))
        self.bn4, self.bn5 = nn.BatchNorm1d(128, eps=1e-3), nn.BatchNorm1d(1024, eps=1e-3)
        self.fc1, self.bn6 = nn.Linear(1024, 512), nn.BatchNorm1d(512)
        self.fc2, self.bn7 = nn.Linear(512, 256), nn.BatchNorm1d(256)
        self.fc3 = nn.Linear(256, classes)
        with torch.no_grad():  # Model/PointNet.py:162-164
            for m in (self.conv1, self.conv2, self.conv3, self.conv4, self.conv5, self.fc1, self.fc2, self.fc3):
                nn.init.xavier_uniform_(m.weight)
                m.bias.zero_()
        self._packed: Optional[PackedPointNet] = None
        self._packed_key = None
        self._ws_cache: dict = {}
        self.wide_mode: Optional[str] = None       # None = GEOA3_WIDE_MODE / 'f16x2'; see default_wide_mode()
        from . import library
        self._handle = library.register_net(self)  # the scalar the custom op geoa3::pointnet_forward takes for this module

    def __getstate__(self):
        """copy.deepcopy / pickle: the packed device weights (a ctypes structure of raw pointers: not picklable), the GPU
        workspace and the handle belong to THIS module and are not copied -- the copy rebuilds its own on first use."""
        state = dict(self.__dict__)
        for k in ("_packed", "_packed_key", "_ws_cache", "_handle"):
            state.pop(k, None)
        return state

    def __setstate__(self, state):
        """copy.deepcopy / pickle: the copy is a module of its own -- its own handle, packed weights and workspace."""
        super().__setstate__(state)
        from . import library
        self._packed, self._packed_key, self._ws_cache = None, None, {}
        self._handle = library.register_net(self)

    def _weights_key(self, device):
        return (str(device), self.wide_mode or default_wide_mode(), fuse_front(), wide_shape("conv5"), wide_shape("tnet")) + tuple((p.data_ptr(), p._version) for p in list(self.parameters()) + list(self.buffers()))

    def packed(self, device) -> PackedPointNet:
        key = self._weights_key(device)
        if self._packed is None or key != self._packed_key:
            self._packed = PackedPointNet(self.state_dict(), device, self.wide_mode)
            self._packed_key = key
        return self._packed

    def forward(self, pc: Tensor) -> Tensor:
        assert pc.size(1) == 3
        if self.training:
            raise NotImplementedError("only the eval-mode forward (the attack's victim) is implemented")
        if torch.compiler.is_compiling():   # traced: the registered custom op (geoa3_amd/library.py), same kernels
            return torch.ops.geoa3.pointnet_forward(pc, self._handle)
        return _PointNetFn.apply(pc, self.packed(pc.device), self._ws_cache)

    @property
    def classes(self) -> int:
        return self.num_class


def _ws_for(net: "PointNet", x: Tensor):
    lib = _lib.load()
    B, _, N = x.shape
    nbytes = lib.geoa3_pointnet_workspace_bytes(B, N, net.num_class)
    ws = net._ws_cache.get("ws")
    if ws is None or ws.numel() < nbytes or ws.device != x.device:
        ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        net._ws_cache["ws"] = ws
    return ws


def pointnet_forward_raw(net: "PointNet", x: Tensor) -> Tensor:
    """The library's forward on x [b,3,n] -> logits (implementation of the custom op geoa3::pointnet_forward)."""
    x = x.detach().contiguous().float()
    packed = net.packed(x.device)
    B, _, N = x.shape
    ws = _ws_for(net, x)
    logits = torch.empty(B, packed.classes, device=x.device, dtype=torch.float32)
    packed.struct.flags = ab_flags()
    check(_lib.load().geoa3_pointnet_forward(C.byref(packed.struct), x.data_ptr(), B, N, logits.data_ptr(), ws.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream), "geoa3_pointnet_forward")
    net._ws_cache["version"] = net._ws_cache.get("version", 0) + 1
    net._ws_cache["holds"] = (x.data_ptr(), x._version, tuple(x.shape))
    return logits


def pointnet_backward_raw(net: "PointNet", x: Tensor, dlogits: Tensor) -> Tensor:
    """d (logits . dlogits) / dx (implementation of geoa3::pointnet_backward): the workspace must hold the forward of
    this x; it is recomputed when a later forward has overwritten it (a traced graph may reorder calls)."""
    x = x.detach().contiguous().float()
    if net._ws_cache.get("holds") != (x.data_ptr(), x._version, tuple(x.shape)):
        pointnet_forward_raw(net, x)
    packed = net.packed(x.device)
    B, _, N = x.shape
    dx = torch.empty_like(x)
    check(_lib.load().geoa3_pointnet_backward(C.byref(packed.struct), x.data_ptr(), dlogits.contiguous().float().data_ptr(),
                                              B, N, dx.data_ptr(), net._ws_cache["ws"].data_ptr(),
                                              torch.cuda.current_stream().cuda_stream), "geoa3_pointnet_backward")
    return dx
