"""Input side of the hot path: the 250-instance `.mat` file and its per-item expansion, as the
reference's Provider/modelnet10_instance250.ModelNet40 defines them (schema written by
Provider/gen_data_mat.py:304: data [M,3,N] f32, normal [M,3,N] f32, label [M,1]).  Plus a seeded synthetic
generator for the same schema (the real file is hosted outside the reference repository)."""
from __future__ import annotations

import os
from random import choice
from typing import List

import numpy as np
import torch
from scipy.io import loadmat, savemat

# the ten ModelNet10 classes as ModelNet40 label ids / names (Provider/modelnet10_instance250.py:10-11)
TEN_LABEL_INDEXES = [17, 9, 36, 20, 3, 16, 34, 38, 23, 15]
TEN_LABEL_NAMES = ["airplane", "bed", "bookshelf", "bottle", "chair", "monitor", "sofa", "table", "toilet", "vase"]


class ModelNet40(torch.utils.data.Dataset):
    """Same constructor, fields (`start_index`, `data`, `normal`, `label`) and item layout as the reference class.
    item = [pcs [l,N,3], normals [l,N,3], gt_labels [l] (, target_labels [l])] with l = 9 for `All` / a class
    name (the nine other ModelNet10 labels as targets) and l = 1 for `Untarget` / `Random`."""

    def __init__(self, data_mat_file="../Data/modelnet10_250instances_1024.mat", attack_label="All", resample_num=-1,
                 is_half_forward=False):
        if not os.path.isfile(data_mat_file):
            raise AssertionError("No exists .mat file!")
        if resample_num > 0:
            raise NotImplementedError("FPS re-sampling of the input file (resample_num > 0) is not on the hot path")
        mat = loadmat(data_mat_file)
        data, normal, label = torch.FloatTensor(mat["data"]), torch.FloatTensor(mat["normal"]), mat["label"]
        self.attack_label, self.is_half_forward = attack_label, is_half_forward
        if attack_label in TEN_LABEL_NAMES:            # 25 instances of one class
            k = TEN_LABEL_NAMES.index(attack_label)
            self.start_index, sl = k * 25, slice(k * 25, (k + 1) * 25)
        elif attack_label in ("All", "Untarget", "Random"):
            self.start_index, sl = 0, slice(None)
        else:
            raise AssertionError("unknown attack_label %r" % attack_label)
        self.data, self.normal, self.label = data[sl], normal[sl], label[sl]

    def __len__(self):
        return self.data.size(0)

    def __getitem__(self, index):
        label = self.label[index]
        pc = self.data[index].contiguous().t()
        nm = self.normal[index].contiguous().t()
        if self.attack_label in TEN_LABEL_NAMES or self.attack_label == "All":
            targets = torch.tensor([i for i in TEN_LABEL_INDEXES if label != i], dtype=torch.long)
            assert targets.size(0) == 9
            gts = torch.as_tensor(np.asarray(label)).long().expand_as(targets)
            pcs, nms = pc.unsqueeze(0).expand(9, -1, -1), nm.unsqueeze(0).expand(9, -1, -1)
            if self.is_half_forward:
                return [[pcs[:4], nms[:4], gts[:4], targets[:4]], [pcs[4:], nms[4:], gts[4:], targets[4:]]]
            return [pcs, nms, gts, targets]
        gts = torch.as_tensor(np.asarray(label)).long().view(-1)
        item = [pc.unsqueeze(0), nm.unsqueeze(0), gts]
        if self.attack_label == "Random":
            item.append(torch.tensor([choice([i for i in range(40) if i != int(gts.item())])], dtype=torch.long))
        return item


class AdvModelNet40(torch.utils.data.Dataset):
    """The directory of adversarial `.mat` files main_attack.py writes (Mat/adv_*.mat), as the reference's
    Provider/defense_modelnet10_instance250.ModelNet40 reads it: item = [pc [3,N] f32, gt_label, attack_label]."""

    def __init__(self, advdatadir):
        self.advdatadir = advdatadir
        self.filename = os.listdir(advdatadir)

    def __len__(self):
        return len(self.filename)

    def __getitem__(self, index):
        data = loadmat(os.path.join(self.advdatadir, self.filename[index]))
        return [torch.FloatTensor(data["adversary_point_clouds"]), data["gt_label"], data["attack_label"]]


def synthetic_clouds(M: int, N: int, seed: int = 0):
    """Seeded stand-in for the ModelNet instances: N points on a random ellipsoid (semi-axes U(0.3,1)),
    analytic unit normals, centred and scaled to unit max radius as Provider/gen_data_mat.py:153-157.
    -> (data [M,3,N], normal [M,3,N]) float32 tensors."""
    g = torch.Generator().manual_seed(seed)
    axes = torch.rand(M, 3, 1, generator=g) * 0.7 + 0.3
    u = torch.randn(M, 3, N, generator=g)
    u = u / u.norm(dim=1, keepdim=True).clamp(min=1e-12)
    pts = u * axes
    nrm = u / axes
    nrm = nrm / nrm.norm(dim=1, keepdim=True).clamp(min=1e-12)
    pts = pts - pts.mean(dim=2, keepdim=True)
    pts = pts / pts.norm(dim=1).max(dim=1)[0].view(M, 1, 1)
    return pts.contiguous().float(), nrm.contiguous().float()


def _box_surface(g, n, ext, centre=(0.0, 0.0, 0.0)):
    """n points uniform by area on the six faces of an axis-aligned box of half-extents ext, outward normals."""
    ext = torch.as_tensor(ext, dtype=torch.float32)
    area = torch.stack([ext[1] * ext[2], ext[0] * ext[2], ext[0] * ext[1]]).repeat_interleave(2)    # +x -x +y -y +z -z
    face = torch.multinomial(area / area.sum(), n, replacement=True, generator=g)
    axis, sign = face // 2, 1.0 - 2.0 * (face % 2).float()
    p = (torch.rand(n, 3, generator=g) * 2 - 1) * ext
    p[torch.arange(n), axis] = sign * ext[axis]
    nrm = torch.zeros(n, 3)
    nrm[torch.arange(n), axis] = sign
    return p + torch.as_tensor(centre, dtype=torch.float32), nrm


def _ellipsoid_surface(g, n, axes, centre=(0.0, 0.0, 0.0)):
    axes = torch.as_tensor(axes, dtype=torch.float32)
    u = torch.randn(n, 3, generator=g)
    u = u / u.norm(dim=1, keepdim=True).clamp(min=1e-12)
    nrm = u / axes
    return u * axes + torch.as_tensor(centre, dtype=torch.float32), nrm / nrm.norm(dim=1, keepdim=True).clamp(min=1e-12)


def _cylinder_surface(g, n, radius, half_len, centre=(0.0, 0.0, 0.0)):
    """side + two caps of a cylinder along z, by area."""
    a_side, a_cap = 2 * np.pi * radius * 2 * half_len, np.pi * radius * radius
    part = torch.multinomial(torch.tensor([a_side, a_cap, a_cap], dtype=torch.float32), n, replacement=True, generator=g)
    th = torch.rand(n, generator=g) * (2 * np.pi)
    r = torch.where(part == 0, torch.full((n,), float(radius)), radius * torch.rand(n, generator=g).sqrt())
    z = torch.where(part == 0, (torch.rand(n, generator=g) * 2 - 1) * half_len,
                    torch.where(part == 1, torch.full((n,), float(half_len)), torch.full((n,), -float(half_len))))
    p = torch.stack([r * th.cos(), r * th.sin(), z], 1)
    nrm = torch.stack([th.cos(), th.sin(), torch.zeros(n)], 1)
    nrm[part == 1] = torch.tensor([0.0, 0.0, 1.0])
    nrm[part == 2] = torch.tensor([0.0, 0.0, -1.0])
    return p + torch.as_tensor(centre, dtype=torch.float32), nrm


CAD_KINDS = ("box", "table", "clusters", "rod", "ellipsoid2")


def synthetic_cad_clouds(M: int, N: int, seed: int = 0, duplicates: float = 0.05, kinds=CAD_KINDS):
    """Seeded clouds shaped like the reference's data -- ModelNet CAD models sampled on their surfaces
    (Provider/gen_data_mat.py:142-159,161-306): planar faces, thin parts, strongly non-uniform density, exact coordinate
    repeats -- where `synthetic_clouds` is the friendliest input the searches, FPS and ball queries can get (one
    ellipsoid, uniform).  Instance j is of kind kinds[j % len(kinds)] (default: all five in turn):
      box         six planar faces of a random box;
      table       a thin slab on four thin legs (40 % of the points on the legs);
      clusters    a large ellipsoid carrying 25 % of the points and a small one (1/8 of its size) carrying 75 %;
      rod         a thin cylinder (radius 3 % of its length) with caps;
      ellipsoid2  `synthetic_clouds`' ellipsoid with 10 % of the points on a second, smaller one.
    `duplicates` of the points are exact copies of other points (coordinates and normal).  Analytic unit normals; centred
    on the mean and scaled to unit maximum radius as gen_data_mat.py:153-157.  -> (data [M,3,N], normal [M,3,N]) float32."""
    g = torch.Generator().manual_seed(seed)
    data, normal = torch.empty(M, 3, N), torch.empty(M, 3, N)
    for j in range(M):
        kind = kinds[j % len(kinds)]
        r = torch.rand(8, generator=g)
        if kind == "box":
            p, nr = _box_surface(g, N, (0.3 + 0.7 * r[0], 0.3 + 0.7 * r[1], 0.1 + 0.9 * r[2]))
        elif kind == "table":
            w, d, h, t = 0.6 + 0.4 * r[0], 0.4 + 0.4 * r[1], 0.5 + 0.4 * r[2], 0.03 + 0.03 * r[3]
            n_leg = int(0.1 * N)
            parts = [_box_surface(g, N - 4 * n_leg, (w, d, t), (0.0, 0.0, float(h)))]
            for sx in (-1.0, 1.0):
                for sy in (-1.0, 1.0):
                    parts.append(_box_surface(g, n_leg, (t, t, h / 2), (float(sx * (w - 2 * t)), float(sy * (d - 2 * t)), float(h / 2))))
            p, nr = torch.cat([q[0] for q in parts]), torch.cat([q[1] for q in parts])
        elif kind == "clusters":
            n_big = N // 4
            a = 0.5 + 0.5 * r[:3]
            pb, nb = _ellipsoid_surface(g, n_big, a)
            ps, ns = _ellipsoid_surface(g, N - n_big, a / 8, (float(1.3 * a[0]), float(0.2 * r[3]), float(0.2 * r[4])))
            p, nr = torch.cat([pb, ps]), torch.cat([nb, ns])
        elif kind == "rod":
            p, nr = _cylinder_surface(g, N, 0.03 + 0.02 * float(r[0]), 1.0)
            rot = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
            p, nr = p @ rot.T, nr @ rot.T
        else:
            n2 = N // 10
            a = 0.3 + 0.7 * r[:3]
            p1, n1 = _ellipsoid_surface(g, N - n2, a)
            p2, n2_ = _ellipsoid_surface(g, n2, a * 0.3, (float(0.9 * a[0]), 0.0, float(0.5 * a[2])))
            p, nr = torch.cat([p1, p2]), torch.cat([n1, n2_])
        perm = torch.randperm(N, generator=g)
        p, nr = p[perm], nr[perm]
        n_dup = int(duplicates * N)
        if n_dup > 0:
            src = torch.randint(0, N - n_dup, (n_dup,), generator=g)
            p[N - n_dup:], nr[N - n_dup:] = p[src], nr[src]
        p = p - p.mean(0, keepdim=True)
        p = p / p.norm(dim=1).max()
        data[j], normal[j] = p.t(), nr.t()
    return data.contiguous().float(), normal.contiguous().float()


SYNTHETIC_GENERATORS = {"ellipsoid": synthetic_clouds, "cad": synthetic_cad_clouds}


def write_synthetic_mat(path: str, labels: List[int], N: int, seed: int = 0, kind: str = "ellipsoid") -> str:
    """Write a `.mat` in the reference schema with len(labels) synthetic instances (kind: SYNTHETIC_GENERATORS)."""
    data, normal = SYNTHETIC_GENERATORS[kind](len(labels), N, seed)
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    savemat(path, {"data": data.numpy(), "normal": normal.numpy(),
                   "label": np.asarray(labels, dtype=np.int64).reshape(-1, 1)})
    return path


def synthetic_state_dict(classes: int = 40, seed: int = 0, device=None):
    """Random-init PointNet weights in the reference state_dict layout for runs without the (missing) checkpoint:
    the reference initialiser (Model/PointNet.py:89-94,162-164) with randomised BatchNorm statistics so that BN
    folding matters, and the last layer calibrated ON THE GPU (HIP forward) so that logits are centred per class
    with unit spread over a fixed set of synthetic clouds (a raw random net sends every cloud to one class and
    no attack could ever succeed)."""
    import math

    from .pointnet import PointNet
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def xavier(shape):
        rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
        a = math.sqrt(6.0 / ((shape[0] + shape[1]) * rf))
        return (torch.rand(shape, generator=g) * 2 - 1) * a

    def layer(name, shape):
        sd[name + ".weight"] = xavier(shape)
        sd[name + ".bias"] = torch.randn(shape[0], generator=g) * 0.05

    def bn(name, c):
        sd[name + ".weight"] = torch.rand(c, generator=g) + 0.5
        sd[name + ".bias"] = torch.randn(c, generator=g) * 0.1
        sd[name + ".running_mean"] = torch.randn(c, generator=g) * 0.1
        sd[name + ".running_var"] = torch.rand(c, generator=g) + 0.5
        sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

    for prefix, K in (("input_transform.", 3), ("feature_transform.", 64)):
        layer(prefix + "conv1", (64, K, 1))
        layer(prefix + "conv2", (128, 64, 1))
        layer(prefix + "conv3", (1024, 128, 1))
        layer(prefix + "fc1", (512, 1024))
        layer(prefix + "fc2", (256, 512))
        sd[prefix + "fc3.weight"] = torch.randn(K * K, 256, generator=g) * (0.02 / K)
        sd[prefix + "fc3.bias"] = torch.eye(K).view(-1).clone()
        for i, c in enumerate([64, 128, 1024, 512, 256], 1):
            bn(prefix + "bn%d" % i, c)
    for name, shape in (("conv1", (64, 3, 1)), ("conv2", (64, 64, 1)), ("conv3", (64, 64, 1)),
                        ("conv4", (128, 64, 1)), ("conv5", (1024, 128, 3))):
        layer(name, shape)
    for i, c in enumerate([64, 64, 64, 128, 1024, 512, 256], 1):
        bn("bn%d" % i, c)
    layer("fc1", (512, 1024))
    layer("fc2", (256, 512))
    layer("fc3", (classes, 256))
    if device is None:
        device = torch.device("cuda")
    net = PointNet(classes)
    net.load_state_dict(sd)
    net = net.to(device).eval()
    calib, _ = synthetic_clouds(32, 256, seed=12345)
    with torch.no_grad():
        lg = net(calib.to(device)).cpu()
    gain = 1.0 / lg.std(0).mean().clamp(min=1e-6)
    sd["fc3.bias"] = (sd["fc3.bias"] - lg.mean(0)) * gain
    sd["fc3.weight"] = sd["fc3.weight"] * gain
    return sd
