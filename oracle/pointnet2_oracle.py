"""CPU oracle for the PointNet++ SSG path (config 4)  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Restates, in numpy / torch-CPU, the six native functions of the reference's vendored CUDA extension that the SSG
classifier uses, following the .cu sources line by line (all under
Model/pointnet2_ops_lib/pointnet2_ops/_ext-src/src/):

    furthest_point_sampling   sampling_gpu.cu:69-173 (+ sampling.cpp:66-87, cuda_utils.h:13-19)
    gather_points / _grad     sampling_gpu.cu:8-57
    ball_query                ball_query_gpu.cu:9-44
    group_points / _grad      group_points_gpu.cu:8-75

The extension is CUDA-only (every host wrapper asserts "CPU not supported", e.g. ball_query.cpp:27-29) and cannot be
built in this image (no nvcc, no CUDA runtime): there is no oracle/_ref for it.  These restatements are what
tests/golden/make_golden.py registers as `pointnet2_ops._ext` BEFORE importing the reference's Python
(pointnet2_utils.py / pointnet2_modules.py / PointNetPP_ssg.py run unchanged on top of them), so the golden fixtures
pin the reference's Python composition; the native arithmetic itself is pinned to the .cu semantics only
("parity unpinned" for the nvcc contraction choice in the distance expressions: evaluated un-fused by default).
`contract=True` evaluates them as nvcc -O3 most likely compiled them (setup.py:32 passes no -fmad flag, the default is
-fmad=true): dx*dx + dy*dy + dz*dz -> fmaf(dz, dz, fmaf(dy, dy, dx*dx)) (sampling_gpu.cu:100,103-104,
ball_query_gpu.cu:31-32), with the fused multiply-add emulated exactly (_fmaf32).
"""
from __future__ import annotations

import numpy as np
import torch

Tensor = torch.Tensor


def _sq3(dx, dy, dz):
    f = np.float32
    return f(f(f(dx * dx) + f(dy * dy)) + f(dz * dz))


def _fmaf32(a, b, c):
    """fl32(a * b + c) with ONE rounding, element-wise on float32 arrays.  The product of two float32 values is exact in
    float64; the sum with c is rounded to float64 first, which could double-round only if that float64 lay within one
    float64 ulp of a float32 rounding boundary: those elements (none in practice) are redone in exact rational
    arithmetic."""
    a, b, c = (np.asarray(v, dtype=np.float32) for v in (a, b, c))
    a, b, c = np.broadcast_arrays(a, b, c)
    s = a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)
    out = s.astype(np.float32)
    # a float32 boundary is a float64 whose low 29 mantissa bits are 1000...0: flag anything within 2 float64 ulps of one
    bits = s.view(np.int64) if s.flags.c_contiguous else np.ascontiguousarray(s).view(np.int64)
    low = bits & ((1 << 29) - 1)
    risky = np.abs(low - (1 << 28)) <= 2
    if risky.any():
        from fractions import Fraction
        flat_out = out.reshape(-1).copy()
        fa, fb, fc = a.reshape(-1), b.reshape(-1), c.reshape(-1)
        for i in np.nonzero(risky.reshape(-1))[0]:
            if not np.isfinite(s.reshape(-1)[i]):
                continue
            exact = Fraction(float(fa[i])) * Fraction(float(fb[i])) + Fraction(float(fc[i]))
            lo = np.float32(float(exact))                      # float(Fraction) rounds correctly to float64 ...
            cands = [lo, np.nextafter(lo, np.float32(np.inf)), np.nextafter(lo, np.float32(-np.inf))]
            best = min(cands, key=lambda v: (abs(Fraction(float(v)) - exact), int(np.float32(v).view(np.int32)) & 1))
            flat_out[i] = best                                  # ... and the nearest float32 (ties to even) is picked exactly
        out = flat_out.reshape(out.shape)
    return out


def _sq3_arrays(dx, dy, dz, contract: bool):
    """the squared distance of float32 arrays: un-fused (default) or as fmaf(dz, dz, fmaf(dy, dy, dx * dx))."""
    f = np.float32
    if contract:
        return _fmaf32(dz, dz, _fmaf32(dy, dy, (dx * dx).astype(f)))
    s = ((dx * dx).astype(f) + (dy * dy).astype(f)).astype(f)
    return (s + (dz * dz).astype(f)).astype(f)


def opt_n_threads(work_size: int) -> int:
    """cuda_utils.h:13-19: clamp(2^floor(log2(work_size)), 1, 512)."""
    p = 1
    while p * 2 <= work_size:
        p *= 2
    return max(min(p, 512), 1)


def furthest_point_sampling(xyz: Tensor, npoint: int, contract: bool = False) -> Tensor:
    """xyz [B,N,3] f32 -> idx [B,npoint] int32."""
    P = xyz.detach().cpu().numpy().astype(np.float32)
    B, N, _ = P.shape
    T = opt_n_threads(N)
    out = np.zeros((B, npoint), dtype=np.int32)
    kk = np.arange(N)
    tt = kk % T
    for b in range(B):
        p = P[b]
        mag = _sq3_arrays(p[:, 0], p[:, 1], p[:, 2], contract)
        use = ~(mag <= np.float32(1e-3))
        temp = np.full(N, 1e10, dtype=np.float32)
        old = 0
        for j in range(1, npoint):
            d = p - p[old]
            dd = _sq3_arrays(d[:, 0], d[:, 1], d[:, 2], contract)
            temp = np.where(use, np.minimum(dd, temp), temp)
            if not use.any():
                old = 0
            else:
                v = np.where(use, temp, np.float32(-2))
                best = v.max()
                cand = np.nonzero(v == best)[0]
                # tie rule of the block reduction: lowest owning thread (k mod T), then lowest k
                order = np.lexsort((kk[cand], tt[cand]))
                old = int(cand[order[0]])
            out[b, j] = old
    return torch.from_numpy(out)


def gather_points(points: Tensor, idx: Tensor) -> Tensor:
    """points [B,C,N], idx [B,M] -> [B,C,M]"""
    B, C, N = points.shape
    return torch.gather(points, 2, idx.long().unsqueeze(1).expand(B, C, idx.shape[1])).contiguous()


def gather_points_grad(grad_out: Tensor, idx: Tensor, n: int) -> Tensor:
    B, C, M = grad_out.shape
    g = torch.zeros(B, C, n, dtype=grad_out.dtype)
    g.scatter_add_(2, idx.long().unsqueeze(1).expand(B, C, M), grad_out)
    return g


def ball_query(new_xyz: Tensor, xyz: Tensor, radius: float, nsample: int, contract: bool = False) -> Tensor:
    """new_xyz [B,M,3], xyz [B,N,3] -> idx [B,M,nsample] int32 (note the extension's argument order)."""
    C, P = new_xyz.detach().cpu().numpy().astype(np.float32), xyz.detach().cpu().numpy().astype(np.float32)
    B, M, _ = C.shape
    r2 = np.float32(np.float32(radius) * np.float32(radius))
    out = np.zeros((B, M, nsample), dtype=np.int32)
    for b in range(B):
        d = C[b][:, None, :] - P[b][None, :, :]                                   # [M,N,3] f32
        d2 = _sq3_arrays(d[..., 0], d[..., 1], d[..., 2], contract)
        inside = d2 < r2
        for j in range(M):
            hits = np.nonzero(inside[j])[0][:nsample]
            if len(hits):
                out[b, j, :] = hits[0]
                out[b, j, :len(hits)] = hits
    return torch.from_numpy(out)


def group_points(points: Tensor, idx: Tensor) -> Tensor:
    """points [B,C,N], idx [B,M,S] -> [B,C,M,S]"""
    B, C, N = points.shape
    _, M, S = idx.shape
    flat = idx.long().reshape(B, 1, M * S).expand(B, C, M * S)
    return torch.gather(points, 2, flat).reshape(B, C, M, S).clone()   # a fresh tensor, as the extension returns


def group_points_grad(grad_out: Tensor, idx: Tensor, n: int) -> Tensor:
    B, C, M, S = grad_out.shape
    g = torch.zeros(B, C, n, dtype=grad_out.dtype)
    g.scatter_add_(2, idx.long().reshape(B, 1, M * S).expand(B, C, M * S), grad_out.reshape(B, C, M * S))
    return g


class ExtModule:
    """Object with the attribute surface of `pointnet2_ops._ext` (bindings.cpp:6-19)."""
    furthest_point_sampling = staticmethod(furthest_point_sampling)
    gather_points = staticmethod(gather_points)
    gather_points_grad = staticmethod(gather_points_grad)
    ball_query = staticmethod(ball_query)
    group_points = staticmethod(group_points)
    group_points_grad = staticmethod(group_points_grad)


def make_pn2_state_dict(seed: int = 0):
    """Seeded synthetic weights in the state_dict layout of PointNet2ClassificationSSG(use_xyz=True,
    use_normal=False) (68 entries), with randomised BatchNorm statistics."""
    import math
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def bn(prefix, c):
        sd[prefix + ".weight"] = torch.rand(c, generator=g) + 0.5
        sd[prefix + ".bias"] = torch.randn(c, generator=g) * 0.1
        sd[prefix + ".running_mean"] = torch.randn(c, generator=g) * 0.1
        sd[prefix + ".running_var"] = torch.rand(c, generator=g) + 0.5
        sd[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    def w(shape):
        fan_in = shape[1]
        return (torch.rand(shape, generator=g) * 2 - 1) * math.sqrt(6.0 / fan_in)   # he-uniform: keeps activations O(1)

    for i, spec in enumerate([[3, 64, 64, 128], [131, 128, 128, 256], [259, 256, 512, 1024]]):
        for j, (ci, co) in enumerate(zip(spec[:-1], spec[1:])):
            sd["SA_modules.%d.mlps.0.%d.weight" % (i, 3 * j)] = w((co, ci, 1, 1))
            bn("SA_modules.%d.mlps.0.%d" % (i, 3 * j + 1), co)
    sd["fc_layer.0.weight"] = w((512, 1024))
    bn("fc_layer.1", 512)
    sd["fc_layer.3.weight"] = w((256, 512))
    bn("fc_layer.4", 256)
    sd["fc_layer.7.weight"] = w((40, 256))
    sd["fc_layer.7.bias"] = torch.randn(40, generator=g) * 0.05
    return sd


def calibrate_pn2_state_dict(sd, n_clouds: int = 16, npoint: int = 1024, seed: int = 12345):
    """The last layer rescaled on a fixed set of synthetic clouds so that the logits are centred per class with unit
    spread across instances (as geoa3_oracle.make_pointnet_state_dict does for PointNet): the raw random-init SSG net
    maps every cloud to the same class with a margin no attack step changes, which would leave the long-run success
    bookkeeping untested (tests/golden/make_golden_long.py)."""
    from . import geoa3_oracle as O
    calib, _ = O.make_synthetic_clouds(n_clouds, npoint, seed=seed)
    sd = dict(sd)
    with torch.no_grad():
        lg = pointnet2_ssg_forward(sd, calib)
        gain = 1.0 / lg.std(0).mean().clamp(min=1e-6)
        sd["fc_layer.7.bias"] = (sd["fc_layer.7.bias"] - lg.mean(0)) * gain
        sd["fc_layer.7.weight"] = sd["fc_layer.7.weight"] * gain
    return sd


def pointnet2_ssg_forward(sd, pc: Tensor) -> Tensor:
    """Eval-mode PointNet2ClassificationSSG.forward (Model/PointNetPP_ssg.py:106-124 over
    pointnet2_modules.py:29-74 and pointnet2_utils.py:296-333,349-379), functional over a state_dict.
    pc [b,3,N] -> logits [b,40].  Differentiable w.r.t. pc (torch.gather carries the scatter-add backward)."""
    import torch.nn.functional as F

    def bn(x, prefix):
        return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"], sd[prefix + ".weight"],
                            sd[prefix + ".bias"], False, 0.0, 1e-5)

    def mlp(x, i):
        for j in range(3):
            x = F.relu(bn(F.conv2d(x, sd["SA_modules.%d.mlps.0.%d.weight" % (i, 3 * j)]),
                          "SA_modules.%d.mlps.0.%d" % (i, 3 * j + 1)))
        return x

    xyz = pc.transpose(2, 1).contiguous()                    # [b,N,3]
    features = None
    for i, (npoint, radius, nsample) in enumerate([(512, 0.2, 64), (128, 0.4, 64), (None, None, None)]):
        if npoint is not None:
            centres = furthest_point_sampling(xyz.detach(), npoint)
            new_xyz = gather_points(xyz.transpose(1, 2).contiguous(), centres).transpose(1, 2).contiguous()
            idx = ball_query(new_xyz.detach(), xyz.detach(), radius, nsample)
            grouped_xyz = group_points(xyz.transpose(1, 2).contiguous(), idx) - new_xyz.transpose(1, 2).unsqueeze(-1)
            new = grouped_xyz if features is None else torch.cat([grouped_xyz, group_points(features, idx)], dim=1)
        else:
            new_xyz = None
            new = torch.cat([xyz.transpose(1, 2).unsqueeze(2), features.unsqueeze(2)], dim=1)
        new = mlp(new, i)
        features = F.max_pool2d(new, kernel_size=[1, new.size(3)]).squeeze(-1)
        xyz = new_xyz
    f = features.squeeze(-1)
    f = F.relu(bn(F.linear(f, sd["fc_layer.0.weight"]), "fc_layer.1"))
    f = F.relu(bn(F.linear(f, sd["fc_layer.3.weight"]), "fc_layer.4"))
    return F.linear(f, sd["fc_layer.7.weight"], sd["fc_layer.7.bias"])
