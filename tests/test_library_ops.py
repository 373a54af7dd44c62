"""`torch.library` registration of the operator boundary (geoa3_amd/library.py).  Without a GPU: every op has a schema
and a fake (meta) kernel, so the reference's loss composition traces under FakeTensorMode with the right shapes and
dtypes and nothing running.  On the GPU (tests/test_gpu_library.py): torch.compile(fullgraph=True) == eager."""
import torch
from torch._subclasses import FakeTensorMode
from torch.fx.experimental.proxy_tensor import make_fx

from geoa3_amd import library, loss_utils as L, ops  # noqa: F401

OPS = ("nn1_pair", "knn", "kappa", "geo_loss_grad", "knn_points", "point_loss", "kappa_adv", "pointnet_forward",
       "pointnet_backward")


def test_ops_registered_with_schemas():
    for name in OPS:
        op = getattr(torch.ops.geoa3, name)
        schema = str(op.default._schema)
        assert schema.startswith("geoa3::" + name + "("), schema
    s = str(torch.ops.geoa3.geo_loss_grad.default._schema)
    assert "Tensor? normal_ori" in s and "bool deterministic" in s and s.count("Tensor") >= 15
    assert "-> (Tensor, Tensor)" in str(torch.ops.geoa3.knn_points.default._schema)


def test_forward_step_composition_traces_with_fake_tensors():
    """Attacker/geoA3_attack.py:131-166 as the reference composes it (CD + HD + curvature), through the mirrors of
    Lib/loss_utils.py: one FX graph, the geoa3:: ops as nodes, shapes from the fake kernels."""
    def constrain(adv, ori, nrm, kap_ori):
        cd, hd = L.chamfer_loss(adv, ori), L.hausdorff_loss(adv, ori)
        ka, n_adv = L._get_kappa_adv(adv, ori, nrm, 4)
        cv = L.curvature_loss(adv, ori, ka, kap_ori, 4)
        knn = ops.knn_points(adv.permute(0, 2, 1), ori.permute(0, 2, 1), K=3)
        return cd + 0.1 * hd + cv, n_adv, knn.dists, knn.idx, L._get_kappa_ori(ori, nrm, 4)

    with FakeTensorMode():
        adv, ori, nrm = (torch.empty(2, 3, 64, device="cuda") for _ in range(3))
        ko = torch.empty(2, 64, device="cuda")
        gm = make_fx(constrain, tracing_mode="real")(adv, ori, nrm, ko)
        out = constrain(adv, ori, nrm, ko)
    targets = [str(n.target) for n in gm.graph.nodes if n.op == "call_function" and "geoa3" in str(n.target)]
    for name in ("point_loss", "kappa_adv", "nn1_pair", "knn_points", "knn", "kappa"):
        assert any(name in t for t in targets), (name, targets)
    con, n_adv, d, idx, kori = out
    assert con.shape == (2,) and con.dtype == torch.float32 and con.device.type == "cuda"
    assert n_adv.shape == (2, 3, 64) and d.shape == (2, 64, 3) and idx.dtype == torch.int64 and kori.shape == (2, 64)


def test_fake_kernels_cover_optional_arguments():
    with FakeTensorMode():
        adv, ori = torch.empty(3, 3, 50, device="cuda"), torch.empty(3, 3, 80, device="cuda")
        d_ar, i_ar, d_ra, i_ra = torch.ops.geoa3.nn1_pair(adv, ori, True)
        assert d_ar.shape == (3, 50) and i_ar.dtype == torch.int32 and d_ra.shape == (3, 80) and i_ra.shape == (3, 80)
        assert torch.ops.geoa3.nn1_pair(adv, ori, False)[2].shape == (3, 0)
        out = torch.ops.geoa3.geo_loss_grad(adv, ori, None, None, d_ar, i_ar, d_ra, i_ra, None, None, 0, 1, False, 1.0,
                                            0.1, 0.0, True)
        assert [tuple(o.shape) for o in out] == [(3,), (3,), (3,), (3,), (3, 3, 50)]


def test_net_handles_are_per_module_and_survive_deepcopy():
    """A copied module gets its own handle, packed weights and workspace (PointNet.__setstate__), so the custom op
    resolves each module to ITS weights; handles come from a counter, never from id(), and are never reused."""
    import copy
    import gc
    from geoa3_amd.pointnet import PointNet
    a = PointNet(40)
    b = copy.deepcopy(a)
    assert b._handle != a._handle and library._NETS[a._handle] is a and library._NETS[b._handle] is b
    assert b._ws_cache is not a._ws_cache and library.net_handle(b) == b._handle
    hb, ha = b._handle, a._handle
    del a
    gc.collect()
    assert ha not in library._NETS and library.net_handle(b) == hb and PointNet(40)._handle not in (ha, hb)
    b._handle = 10 ** 9                       # a stale integer (e.g. restored by hand): net_handle() repairs it
    assert library.net_handle(b) != 10 ** 9 and library._NETS[b._handle] is b
    # a module that HAS run (packed device weights = a ctypes structure of raw pointers, a workspace cache) copies and pickles
    # too: those belong to the original and are left behind (__getstate__)
    import ctypes
    import pickle

    class _Ptrs(ctypes.Structure):
        _fields_ = [("p", ctypes.c_void_p)]

    class _FakePacked:
        struct = _Ptrs()

    b._packed, b._packed_key, b._ws_cache = _FakePacked(), ("k",), {"ws": object()}
    c = copy.deepcopy(b)
    d = pickle.loads(pickle.dumps(b))
    for m in (c, d):
        assert m._packed is None and m._ws_cache == {} and m._handle not in (b._handle,) and library._NETS[m._handle] is m
        assert all(torch.equal(x, y) for x, y in zip(m.state_dict().values(), b.state_dict().values()))
