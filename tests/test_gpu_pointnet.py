"""HIP PointNet forward / input-gradient against the golden fixtures (reference outputs) and the oracle."""
import numpy as np
import pytest
import torch

from oracle import geoa3_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module", params=["f32", "f16x2"])
def net(request):
    """Both arithmetic modes of the 1024-wide layers (fp32 MFMA; split-fp16 operands with fp32 accumulation) must
    meet the same bars."""
    from geoa3_amd.pointnet import PointNet
    n = PointNet(40)
    n.load_state_dict(O.make_pointnet_state_dict(40, seed=0))
    n.wide_mode = request.param
    return n.cuda().eval()


def assert_grad_close(got, ref):
    """Input gradient [B,3,N]: rtol 2e-3 / atol 2e-4 of the largest entry.  A channel of a max-pooled layer whose two
    best points are closer than the rounding noise of the dot products (~1e-7 relative: about 3e-5 of the 3 x 1024
    pooled channels of an instance) may take its arg-max at the other point; the whole gradient of that one channel
    then lands on other points (2 points x 3 coordinates, x 3 taps for conv5).  Budget: 18 entries per allowed flip,
    ceil(5e-5 * 3072 * B) flips."""
    scale = np.abs(ref).max()
    bad = np.abs(got - ref) > 2e-3 * np.abs(ref) + 2e-4 * scale
    flips = int(np.ceil(5e-5 * 3072 * got.shape[0]))
    assert bad.sum() <= 18 * flips, "%d of %d gradient entries differ" % (bad.sum(), bad.size)


@pytest.mark.parametrize("tag", ["n64", "n256", "n1024"])
def test_forward_backward_golden(net, golden, tag):
    pre = "pn/%s/" % tag
    x = T(golden[pre + "pc"]).cuda().requires_grad_()
    logits = net(x)
    # fp32 tolerance: the MFMA fmaf chain sums in a different order than the CPU reference
    np.testing.assert_allclose(logits.detach().cpu().numpy(), golden[pre + "logits"], rtol=1e-4, atol=3e-4)
    (logits * T(golden[pre + "w"]).cuda()).sum().backward()
    assert_grad_close(x.grad.cpu().numpy(), golden[pre + "g_pc"])


@pytest.mark.parametrize("B,N", [(5, 200), (3, 1000), (33, 128), (2, 2048)])
def test_forward_backward_oracle_ragged(net, B, N):
    """N not a multiple of the 128-point tiles, B not a multiple of the 32-row FC tiles."""
    sd = O.make_pointnet_state_dict(40, seed=0)
    pc, _ = O.make_synthetic_clouds(B, N, seed=B * 1000 + N)
    xc = pc.clone().requires_grad_()
    lo = O.pointnet_forward(sd, xc)
    w = torch.randn(B, 40, generator=torch.Generator().manual_seed(1))
    (lo * w).sum().backward()
    xg = pc.cuda().requires_grad_()
    lg = net(xg)
    np.testing.assert_allclose(lg.detach().cpu().numpy(), lo.detach().numpy(), rtol=1e-4, atol=3e-4)
    (lg * w.cuda()).sum().backward()
    assert_grad_close(xg.grad.cpu().numpy(), xc.grad.numpy())


def test_batch_independence(net):
    """Row k of a batched forward == the batch-1 forward of cloud k, bit for bit: what allows the b separate
    batch-1 success-check forwards of geoA3_attack.py:297 to be read off the one batched forward."""
    pc, _ = O.make_synthetic_clouds(6, 256, seed=5)
    x = pc.cuda()
    with torch.no_grad():
        full = net(x).clone()
        for k in range(6):
            assert torch.equal(net(x[k:k + 1].contiguous())[0], full[k])


@pytest.mark.parametrize("B,N", [(3, 256), (5, 1000), (2, 77), (9, 1024)])
def test_chain_kernels_same_bits_as_layer_by_layer(B, N, monkeypatch):
    """The chain kernels of the trunk (pointnet_conv_chain.hip: conv2 -> T-Net conv1 -> conv2, conv3 -> conv4, and the
    backward of the front) against the one-layer-per-launch path (geoa3_amd.pointnet.AB_FLAGS bit 1, read per call): logits and input
    gradient bit for bit, ragged N (partly dead wavefronts) included."""
    from geoa3_amd.pointnet import PointNet
    n = PointNet(40)
    n.load_state_dict(O.make_pointnet_state_dict(40, seed=0))
    n.wide_mode = "f16x2"
    n = n.cuda().eval()
    pc, _ = O.make_synthetic_clouds(B, N, seed=17 * B + N)
    w = torch.randn(B, 40, generator=torch.Generator().manual_seed(2)).cuda()
    out = {}
    from geoa3_amd import pointnet as PN
    for flag in ("0", "1"):
        monkeypatch.setattr(PN, "AB_FLAGS", 2 if flag == "0" else 0)
        x = pc.cuda().requires_grad_()
        lg = n(x)
        (lg * w).sum().backward()
        out[flag] = (lg.detach().clone(), x.grad.clone())
    assert torch.equal(out["0"][0], out["1"][0])
    assert torch.equal(out["0"][1], out["1"][1])


@pytest.mark.parametrize("B,N", [(3, 256), (5, 1000), (2, 77), (9, 1024), (2, 1500), (1, 4096), (2, 4100)])
def test_forward_built_hit_lists_same_bits(B, N, monkeypatch):
    """The sparse backward reading the hit lists the forward's finalize pass built once per instance
    (wide_finalize_hits_kernel: by column, (chunk, tap, channel) order inside a column) against every workgroup building
    its tile's lists itself (AB_FLAGS bit 3): logits and input gradient bit for bit -- also with a cloud whose points
    coincide (all channels' maxima on few columns: long lists), ragged N, and N > 4096 where the forward builds none."""
    from geoa3_amd.pointnet import PointNet
    from geoa3_amd import pointnet as PN
    n = PointNet(40)
    n.load_state_dict(O.make_pointnet_state_dict(40, seed=0))
    n.wide_mode = "f16x2"
    n = n.cuda().eval()
    pc, _ = O.make_synthetic_clouds(B, N, seed=3 * B + N)
    pc[0, :, N // 2:] = pc[0, :, :1]      # half of the first cloud is one point: ties -> the lowest index takes every maximum
    w = torch.randn(B, 40, generator=torch.Generator().manual_seed(2)).cuda()
    out = {}
    for flag in (8, 0):
        monkeypatch.setattr(PN, "AB_FLAGS", flag)
        x = pc.cuda().requires_grad_()
        lg = n(x)
        (lg * w).sum().backward()
        out[flag] = (lg.detach().clone(), x.grad.clone())
    assert torch.equal(out[0][0], out[8][0])
    assert torch.equal(out[0][1], out[8][1])
    assert torch.isfinite(out[0][1]).all() and float(out[0][1].abs().max()) > 0


def test_without_transposed_fragments_same_bits():
    """geoa3_tnet_weights.w2th / geoa3_pointnet_weights.w4th are optional (include/geoa3_hip.h): a caller that does not
    hand them over gets the sparse backward and the 128 -> 64 layer behind it as two kernels -- the same gradient, bit
    for bit."""
    from geoa3_amd.pointnet import PointNet
    n = PointNet(40)
    n.load_state_dict(O.make_pointnet_state_dict(40, seed=0))
    n.wide_mode = "f16x2"
    n = n.cuda().eval()
    pc, _ = O.make_synthetic_clouds(4, 700, seed=21)
    w = torch.randn(4, 40, generator=torch.Generator().manual_seed(5)).cuda()
    out = []
    st = n.packed(torch.device("cuda")).struct
    keep = (st.t3.w2th, st.t64.w2th, st.w4th)
    assert all(keep)
    try:
        for drop in (False, True):
            if drop:
                st.t3.w2th = st.t64.w2th = st.w4th = None
            x = pc.cuda().requires_grad_()
            lg = n(x)
            (lg * w).sum().backward()
            out.append((lg.detach().clone(), x.grad.clone()))
    finally:
        st.t3.w2th, st.t64.w2th, st.w4th = keep
    assert torch.equal(out[0][0], out[1][0])
    assert torch.equal(out[0][1], out[1][1])
    assert float(out[0][1].abs().max()) > 0


@pytest.mark.parametrize("scale", [1e-6, 1e-3, 1.0, 300.0, 1e6])
def test_wide_split_operand_range(scale):
    """f16x2 mode carries every fp32 operand of the 1024-wide layers as two fp16 values after a power-of-two scaling
    taken from the tile's own maximum: activations of very different magnitude going into conv5 / the T-Nets' conv3
    (here through the BatchNorm scale of the layer in front of them) -- far outside fp16's own range -- must keep
    fp32-level agreement with the oracle."""
    from geoa3_amd.pointnet import PointNet
    sd = O.make_pointnet_state_dict(40, seed=2)
    for name in ("bn4", "input_transform.bn2", "feature_transform.bn2"):
        sd[name + ".weight"] = sd[name + ".weight"] * scale
        sd[name + ".bias"] = sd[name + ".bias"] * scale
    pc, _ = O.make_synthetic_clouds(4, 300, seed=8)
    lo = O.pointnet_forward(sd, pc)
    for mode in ("f32", "f16x2"):
        n = PointNet(40)
        n.load_state_dict(sd)
        n.wide_mode = mode
        n = n.cuda().eval()
        with torch.no_grad():
            out = n(pc.cuda()).cpu()
            np.testing.assert_allclose(out.numpy(), lo.numpy(), rtol=1e-4, atol=1e-4 * float(lo.abs().max()))


@pytest.mark.parametrize("taps,N", [(3, 1024), (1, 1024), (3, 200), (1, 77), (3, 128)])
def test_wide_layer_both_modes_against_float64(taps, N):
    """One 1024-wide layer (conv + bias + relu + max over points, arg-max) through geoa3_debug_wide_fwd in both
    arithmetic modes against a float64 evaluation: values to fp32 rounding of a K = 128 * taps dot product; the
    arg-max is the float64 one wherever the float64 runner-up is not within rounding noise; a non-finite activation
    poisons that instance's features with NaN in f16x2 mode (loud, never silently dropped by the max)."""
    from geoa3_amd import _lib
    from geoa3_amd.pointnet import pack_wide_fragments, pack_wide_split
    lib = _lib.load()
    B = 5
    g = torch.Generator().manual_seed(taps * 1000 + N)
    X = (torch.randn(B, 128, N, generator=g) * torch.logspace(-3, 2, 128).view(1, 128, 1)).relu()
    X[1] *= 1e-9
    X[3] *= 1e7
    W = torch.randn(1024, taps * 128, generator=g) * 0.05
    bias = torch.randn(1024, generator=g)
    conv = torch.nn.functional.conv1d(X.double(), W.double().view(1024, taps, 128).permute(0, 2, 1), padding=taps // 2)
    ref, ref_arg = conv.max(dim=2)
    second = conv.scatter(2, ref_arg.unsqueeze(2), -float("inf")).max(dim=2).values
    ref_out = (ref + bias.double()).clamp_min(0)
    mag = torch.nn.functional.conv1d(X.double().abs(), W.double().abs().view(1024, taps, 128).permute(0, 2, 1),
                                     padding=taps // 2).max(dim=2).values          # bound on sum |a w|
    Wp, (Wh, uns) = pack_wide_fragments(W, taps).cuda(), pack_wide_split(W, taps)
    Wh, Xd, bd = Wh.cuda(), X.cuda(), bias.cuda()
    out = torch.empty(B, 1024, device="cuda")
    arg = torch.empty(B, 1024, device="cuda", dtype=torch.int32)
    keys = torch.empty(B, 1024, device="cuda", dtype=torch.int64)
    s = torch.cuda.current_stream().cuda_stream

    def run(split, x):
        _lib.check(lib.geoa3_debug_wide_fwd(x.data_ptr(), Wp.data_ptr(), Wh.data_ptr() if split else None, uns,
                                            bd.data_ptr(), out.data_ptr(), arg.data_ptr(), keys.data_ptr(), B, N, taps,
                                            0, None, s), "geoa3_debug_wide_fwd")
        return out.cpu().double(), arg.cpu().long()

    for split in (False, True):
        o, a = run(split, Xd)
        tol = 4e-6 * mag + 2e-7 * (ref.abs() + bias.double().abs())   # ~ sqrt(K) 2^-24 sum |a w|, + the bias add
        assert ((o - ref_out).abs() <= tol).all(), (split, float(((o - ref_out).abs() / tol).max()))
        clear = (ref - second) > 2 * tol
        assert clear.float().mean() > 0.7    # (instance 1: the bias dominates, nothing is clear there)
        assert torch.equal(a[clear], ref_arg[clear])
    bad = Xd.clone()
    bad[2, 5, N // 2] = float("inf")
    bad[4, 100, 0] = float("nan")
    o, _ = run(True, bad)
    assert torch.isnan(o[2]).all() and torch.isnan(o[4]).all() and torch.isfinite(o[[0, 1, 3]]).all()


@pytest.mark.parametrize("K,Co,N,gate", [(64, 64, 1024, False), (64, 128, 300, False), (128, 64, 1000, True),
                                        (64, 64, 77, True), (128, 128, 256, False)])
def test_narrow_convolution_both_modes_against_float64(K, Co, N, gate):
    """conv_cm64 (fp32 MFMA) and conv_cm64s (split-fp16 operands, running per-wave scale) through
    geoa3_debug_conv_cm against a float64 evaluation: the same bound for both.  Rows of very different magnitude
    (the running scale has to shrink from chunk to chunk and rescale its sums), an all-zero instance, ragged N."""
    from geoa3_amd import _lib
    lib = _lib.load()
    B = 4
    g = torch.Generator().manual_seed(K * 7 + Co + N)
    X = torch.randn(B, K, N, generator=g) * torch.logspace(-4, 3, K).view(1, K, 1)   # later chunks are larger
    X[2] = 0.0
    X[3] *= 1e-12
    W = torch.randn(Co, K, generator=g) * 0.1
    bias = torch.randn(Co, generator=g) * (X.abs().max() * 0.01)
    Z = torch.randn(B, Co, N, generator=g) if gate else None
    pre = torch.einsum("ok,bkn->bon", W.double(), X.double())
    ref = torch.relu(pre + bias.double().view(1, -1, 1))
    if gate:
        ref = ref * (Z > 0)
    mag = torch.einsum("ok,bkn->bon", W.double().abs(), X.double().abs())
    tol = 2e-6 * mag + 2e-7 * (pre.abs() + bias.double().abs().view(1, -1, 1)) + 1e-30
    Xd, Wd, bd = X.cuda(), W.cuda(), bias.cuda()
    Zd = Z.cuda() if gate else None
    s = torch.cuda.current_stream().cuda_stream
    for split in (0, 1):
        Y = torch.full((B, Co, N), float("nan"), device="cuda")
        _lib.check(lib.geoa3_debug_conv_cm(Xd.data_ptr(), Wd.data_ptr(), bd.data_ptr(),
                                           Zd.data_ptr() if gate else None, Y.data_ptr(), B, N, K, Co, 1, split, s),
                   "geoa3_debug_conv_cm")
        err = (Y.cpu().double() - ref).abs()
        assert (err <= tol).all(), (split, float((err / tol).max()))
