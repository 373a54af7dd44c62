"""HIP geometry kernels (through the C ABI) against the CPU oracle and the golden fixtures.
Runs on the GPU box: python -m pytest tests -m gpu"""
import numpy as np
import pytest
import torch

from oracle import geoa3_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy
CASES = ["n64k2", "n64k16", "n256k16", "n128k32", "dup", "zero"]


@pytest.fixture(scope="module")
def ops():
    from geoa3_amd import ops as _ops
    assert torch.cuda.is_available(), "needs the MI355X"
    return _ops


def dev(x):
    return x.contiguous().cuda()


def _rand_clouds(B, Na, Nr, seed):
    g = torch.Generator().manual_seed(seed)
    a = torch.randn(B, 3, Na, generator=g)
    r = torch.randn(B, 3, Nr, generator=g)
    return a, r


@pytest.mark.parametrize("B,Na,Nr", [(3, 64, 64), (2, 1000, 1000), (2, 1024, 1024), (2, 300, 777), (1, 2500, 2049),
                                     (1, 5, 3)])
def test_nn1_pair_bit_exact(ops, B, Na, Nr):
    a, r = _rand_clouds(B, Na, Nr, 7)
    a[:, :, 0] = r[:, :, min(2, Nr - 1)]          # an exact zero distance
    if Nr > 4:
        r[:, :, 4] = r[:, :, 1]                   # duplicate reference points: tie -> lower index
    d_ar, i_ar, d_ra, i_ra = ops.nn1_pair(dev(a), dev(r))
    od, oi = O.knn_points(a.permute(0, 2, 1), r.permute(0, 2, 1), 1)
    assert torch.equal(i_ar.cpu().long(), oi.squeeze(-1))
    assert torch.equal(d_ar.cpu(), od.squeeze(-1))     # bit-exact: same un-fused arithmetic
    od, oi = O.knn_points(r.permute(0, 2, 1), a.permute(0, 2, 1), 1)
    assert torch.equal(i_ra.cpu().long(), oi.squeeze(-1))
    assert torch.equal(d_ra.cpu(), od.squeeze(-1))


@pytest.mark.parametrize("tag", CASES)
def test_nn1_on_golden_clouds(ops, golden, tag):
    pre = "ops/%s/" % tag
    adv, ori = T(golden[pre + "adv"]), T(golden[pre + "ori"])
    d_ar, i_ar, d_ra, i_ra = ops.nn1_pair(dev(adv), dev(ori))
    od, oi = O.knn_points(adv.permute(0, 2, 1), ori.permute(0, 2, 1), 1)
    assert torch.equal(i_ar.cpu().long(), oi.squeeze(-1)) and torch.equal(d_ar.cpu(), od.squeeze(-1))


@pytest.mark.parametrize("B,Nq,Nr,K", [(2, 64, 64, 3), (2, 1024, 1024, 17), (1, 1024, 1024, 33), (2, 500, 300, 5),
                                       (1, 2100, 2100, 17), (1, 10, 6, 8), (1, 700, 700, 64)])
@pytest.mark.parametrize("prior_kind", ["none", "good", "stale", "duplicate"])
def test_knn_bit_exact(ops, B, Nq, Nr, K, prior_kind):
    q, r = _rand_clouds(B, Nq, Nr, 11)
    if Nq == Nr:
        q = r + 0.01 * q                         # self-like query (the curvature use case)
        q[:, :, 3] = q[:, :, 2]                   # duplicates inside the cloud
    od, oi = O.knn_points(q.permute(0, 2, 1), r.permute(0, 2, 1), min(K, Nr))
    prior = None
    if prior_kind != "none":
        if K > Nr:
            pytest.skip("prior needs K <= Nr")
        if prior_kind == "good":
            prior = oi.int()
        elif prior_kind == "stale":               # neighbours of a perturbed cloud
            _, pi = O.knn_points((q + 0.05 * torch.randn_like(q)).permute(0, 2, 1), r.permute(0, 2, 1), K)
            prior = pi.int()
        else:                                     # invalid prior: all the same index
            prior = torch.zeros(B, Nq, K, dtype=torch.int32)
        prior = dev(prior)
    d, i = ops.knn_planar(dev(q), dev(r), K, prior)
    kk = min(K, Nr)
    assert torch.equal(i.cpu().long()[:, :, :kk], oi)
    assert torch.equal(d.cpu()[:, :, :kk], od)
    if K > Nr:
        assert (i.cpu()[:, :, kk:] == -1).all() and torch.isinf(d.cpu()[:, :, kk:]).all()


@pytest.mark.parametrize("tag", CASES)
def test_kappa_ori_golden(ops, golden, tag):
    pre = "ops/%s/" % tag
    ori, nrm, k = T(golden[pre + "ori"]), T(golden[pre + "nrm"]), int(golden[pre + "k"])
    _, idx = ops.knn_planar(dev(ori), dev(ori), k + 1)
    kap = ops.kappa(dev(ori), dev(nrm), idx)
    np.testing.assert_allclose(kap.cpu().numpy(), golden[pre + "kappa_ori"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("N,k", [(2048, 16), (4096, 32), (3001, 20), (1024, 16)])
def test_kappa_ori_staged_in_lds_matches_the_gather_and_the_oracle(ops, N, k):
    """Clouds of 2048 points and more take kappa_kernel<true> (the instance's cloud staged in LDS, 1024 points per
    workgroup); the arithmetic is the one-point routine of the small-cloud kernel: the same bits as the oracle's
    expression evaluated per point (Lib/loss_utils.py:52-62), at the golden test's tolerance.  With a clean-to-cloud
    index (`nn_idx`: the normal of the nearest clean point) as _get_kappa_adv's callers pass it."""
    from geoa3_amd.data import synthetic_cad_clouds
    B = 3
    ori, nrm = synthetic_cad_clouds(B, N, seed=N + k)
    _, idx = ops.knn_planar(dev(ori), dev(ori), k + 1)
    kap = ops.kappa(dev(ori), dev(nrm), idx)
    want = O.get_kappa_ori(ori, nrm, k)
    np.testing.assert_allclose(kap.cpu().numpy(), want.numpy(), rtol=2e-5, atol=2e-6)
    perm = torch.stack([torch.randperm(N, generator=torch.Generator().manual_seed(b)) for b in range(B)]).int()
    nrm_p = torch.gather(nrm, 2, torch.argsort(perm.long(), 1).unsqueeze(1).expand(-1, 3, -1))   # nrm_p[:, :, perm[i]] = nrm[:, :, i]
    kap2 = ops.kappa(dev(ori), dev(nrm_p), idx, nn_idx=dev(perm))
    assert torch.equal(kap2, kap)


def _geo(ops, golden, tag, **kw):
    pre = "ops/%s/" % tag
    adv, ori, nrm = (dev(T(golden[pre + n])) for n in ("adv", "ori", "nrm"))
    k = int(golden[pre + "k"])
    d_ao, i_ao, d_oa, i_oa = ops.nn1_pair(adv, ori)
    _, knn_ori = ops.knn_planar(ori, ori, k + 1)
    kap_ori = ops.kappa(ori, nrm, knn_ori)
    _, knn_adv = ops.knn_planar(adv, adv, k + 1, knn_ori)
    return ops.geo_loss_grad(adv, ori, normal_ori=nrm, kappa_ori=kap_ori, d_ao=d_ao, i_ao=i_ao, d_oa=d_oa, i_oa=i_oa,
                             knn_adv=knn_adv, k=k, want_kappa=True, **kw), pre


@pytest.mark.parametrize("det", [True, False])
@pytest.mark.parametrize("tag", CASES)
def test_geo_losses_and_grads_golden(ops, golden, tag, det):
    # fp32 tolerance: 2e-5 relative on values, 1e-4 relative (+1e-7 abs) on gradients; the curvature gradient (sums of
    # k = 2..32 cancelling terms per point): 1e-4 with the owner-ordered accumulation, 2e-3 with the LDS float atomics
    def close(a, b, rtol, atol):
        np.testing.assert_allclose(a.cpu().numpy(), b, rtol=rtol, atol=atol)

    import functools
    _geo = functools.partial(globals()["_geo"], deterministic=det)
    curv_rtol = CURV_RTOL_DET if det else 2e-3

    o, pre = _geo(ops, golden, tag, dis_type=1, w_dis=1.0)
    close(o["dis_loss"], golden[pre + "cd"], 2e-5, 1e-7)
    close(o["grad"], golden[pre + "g_cd"], 1e-4, 1e-7)
    o, _ = _geo(ops, golden, tag, dis_type=1, single_side=True, w_dis=1.0)
    close(o["dis_loss"], golden[pre + "pcd"], 2e-5, 1e-7)
    close(o["grad"], golden[pre + "g_pcd"], 1e-4, 1e-7)
    o, _ = _geo(ops, golden, tag, dis_type=2, w_dis=1.0)
    close(o["dis_loss"], golden[pre + "l2"], 2e-5, 1e-7)
    close(o["grad"], golden[pre + "g_l2"], 1e-4, 1e-7)
    o, _ = _geo(ops, golden, tag, dis_type=0, w_hd=1.0)
    close(o["hd_loss"], golden[pre + "hd"], 2e-5, 1e-7)
    if tag != "zero":   # all distances are 0: the arg-max point is a tie and its gradient is 0 anyway
        close(o["grad"], golden[pre + "g_hd"], 1e-4, 1e-7)
    o, _ = _geo(ops, golden, tag, dis_type=0, w_curv=1.0)
    close(o["kappa_adv"], golden[pre + "kappa_adv"], 2e-5, 2e-6)
    close(o["curv_loss"], golden[pre + "curv"], 1e-4, 1e-8)
    close(o["grad"], golden[pre + "g_curv"], curv_rtol, 2e-6)
    # the combined objective: constrain and its gradient
    o, _ = _geo(ops, golden, tag, dis_type=1, w_dis=1.0, w_hd=0.1, w_curv=1.0)
    con = golden[pre + "cd"] + 0.1 * golden[pre + "hd"] + golden[pre + "curv"]
    close(o["constrain"], con, 5e-5, 1e-7)
    g = golden[pre + "g_cd"] + 0.1 * golden[pre + "g_hd"] + golden[pre + "g_curv"]
    if tag != "zero":
        close(o["grad"], g, curv_rtol, 2e-6)


CURV_RTOL_DET = 1e-4


@pytest.mark.parametrize("tag", ["n256k16", "n128k32", "dup"])
def test_deterministic_gradient_is_reproducible_and_batch_independent(ops, golden, tag):
    """geoa3_geo_args.deterministic: ten launches give the same bits, and an instance's gradient does not depend on
    which other instances share the batch (the reference's scatter-adds -- knn_gather / index backward,
    sampling_gpu.cu:42, group_points_gpu.cu:60 -- promise neither)."""
    kw = dict(dis_type=1, w_dis=1.0, w_hd=0.1, w_curv=1.0, deterministic=True)
    first, pre = _geo(ops, golden, tag, **kw)
    g0 = first["grad"].clone()
    for _ in range(9):
        o, _ = _geo(ops, golden, tag, **kw)
        assert torch.equal(o["grad"], g0)
    # the same clouds as rows 1.. of a larger batch whose row 0 is another cloud
    adv, ori, nrm = (dev(T(golden[pre + n])) for n in ("adv", "ori", "nrm"))
    k = int(golden[pre + "k"])
    adv2 = torch.cat([adv[-1:].flip(2), adv]).contiguous()
    ori2 = torch.cat([ori[-1:].flip(2), ori]).contiguous()
    nrm2 = torch.cat([nrm[-1:].flip(2), nrm]).contiguous()
    d_ao, i_ao, d_oa, i_oa = ops.nn1_pair(adv2, ori2)
    _, knn_ori = ops.knn_planar(ori2, ori2, k + 1)
    kap_ori = ops.kappa(ori2, nrm2, knn_ori)
    _, knn_adv = ops.knn_planar(adv2, adv2, k + 1, knn_ori)
    o2 = ops.geo_loss_grad(adv2, ori2, normal_ori=nrm2, kappa_ori=kap_ori, d_ao=d_ao, i_ao=i_ao, d_oa=d_oa, i_oa=i_oa,
                           knn_adv=knn_adv, k=k, **kw)
    assert torch.equal(o2["grad"][1:], g0)


def test_knn_points_operator_autograd(ops, golden):
    """pytorch3d-shaped operator: values, indices and gradients through dists."""
    pre = "ops/n64k16/"
    adv = T(golden[pre + "adv"]).permute(0, 2, 1).contiguous()
    ori = T(golden[pre + "ori"]).permute(0, 2, 1).contiguous()
    for K in (1, 5):
        a_c = adv.clone().requires_grad_()
        o_c = ori.clone().requires_grad_()
        od, oi = O.knn_points(a_c, o_c, K)
        w = torch.rand(od.shape, generator=torch.Generator().manual_seed(K))
        (od * w).sum().backward()
        a_g = adv.cuda().requires_grad_()
        o_g = ori.cuda().requires_grad_()
        r = ops.knn_points(a_g, o_g, K)
        assert r.idx.dtype == torch.int64 and torch.equal(r.idx.cpu(), oi)
        assert torch.equal(r.dists.detach().cpu(), od.detach())
        (r.dists * w.cuda()).sum().backward()
        np.testing.assert_allclose(a_g.grad.cpu().numpy(), a_c.grad.numpy(), rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(o_g.grad.cpu().numpy(), o_c.grad.numpy(), rtol=1e-5, atol=1e-7)
        g = ops.knn_gather(o_g.detach(), r.idx)
        assert torch.equal(g.cpu(), O.knn_gather(ori, oi))


def test_cpu_tensors_are_rejected(ops):
    from geoa3_amd._lib import Geoa3Error
    a, r = _rand_clouds(1, 8, 8, 0)
    with pytest.raises(Geoa3Error):
        ops.nn1_pair(a, r)


@pytest.mark.parametrize("tag", CASES)
def test_loss_utils_mirror_golden(golden, tag):
    """The reference's function names / signatures (Lib/loss_utils.py:25-97) through autograd."""
    from geoa3_amd import loss_utils as LU
    pre = "ops/%s/" % tag
    ori, nrm = dev(T(golden[pre + "ori"])), dev(T(golden[pre + "nrm"]))
    k = int(golden[pre + "k"])
    kap_ori = LU._get_kappa_ori(ori, nrm, k)
    np.testing.assert_allclose(kap_ori.cpu().numpy(), golden[pre + "kappa_ori"], rtol=2e-5, atol=2e-6)
    for name, fn in [("cd", LU.chamfer_loss), ("pcd", LU.pseudo_chamfer_loss), ("hd", LU.hausdorff_loss),
                     ("l2", LU.norm_l2_loss)]:
        adv = dev(T(golden[pre + "adv"])).requires_grad_()
        v = fn(adv, ori)
        np.testing.assert_allclose(v.detach().cpu().numpy(), golden[pre + name], rtol=2e-5, atol=1e-7)
        v.sum().backward()
        if not (tag == "zero" and name == "hd"):
            np.testing.assert_allclose(adv.grad.cpu().numpy(), golden[pre + "g_" + name], rtol=1e-4, atol=1e-7)
    adv = dev(T(golden[pre + "adv"])).requires_grad_()
    kap_adv, normal = LU._get_kappa_adv(adv, ori, nrm, k)
    np.testing.assert_allclose(kap_adv.detach().cpu().numpy(), golden[pre + "kappa_adv"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(normal.cpu().numpy(), golden[pre + "normal_adv"], rtol=0, atol=0)
    curv = LU.curvature_loss(adv, ori, kap_adv, kap_ori)
    np.testing.assert_allclose(curv.detach().cpu().numpy(), golden[pre + "curv"], rtol=1e-4, atol=1e-8)
    curv.sum().backward()
    np.testing.assert_allclose(adv.grad.cpu().numpy(), golden[pre + "g_curv"], rtol=2e-3, atol=2e-6)


@pytest.mark.parametrize("N,K,scale", [(1024, 17, 0.002), (1024, 17, 0.05), (1024, 17, 0.6), (700, 9, 0.02),
                                       (4096, 33, 0.01), (2048, 64, 0.01), (300, 17, 0.0), (40, 17, 0.05),
                                       (8192, 5, 0.01), (9000, 5, 0.01)])
def test_slab_pruned_self_knn_is_bit_identical_to_brute_force(ops, N, K, scale, monkeypatch):
    """geoa3_knn_self (slab pruning along the longest axis) against the all-pairs kernel and the oracle: good priors
    (the clean cloud's table), stale priors (a different cloud's table), degenerate priors (duplicates / out of range:
    the unpruned second pass) and no prior; duplicate points give exact ties."""
    B = 3
    ori, _ = O.make_synthetic_clouds(B, N, seed=N + K)
    g = torch.Generator().manual_seed(23)
    off = scale * torch.randn(B, 3, N, generator=g)
    off[:, :, : N // 2] *= 0.1
    adv = ori + off
    if N > 12:
        adv[:, :, 5] = adv[:, :, 4]
        adv[:, :, 9] = adv[:, :, 11]
    adv[1, 0] *= 0.05                                # a cloud whose longest axis is not x
    oriD, advD = dev(ori), dev(adv)
    kk = min(K, N)
    bd, bi = ops.knn_planar(advD, advD, kk)
    if N <= 4096:
        od, oi = O.knn_points(adv.permute(0, 2, 1), adv.permute(0, 2, 1), kk)
        assert torch.equal(bi.cpu().long(), oi) and torch.equal(bd.cpu(), od)
    scratch = ops.knn_self_scratch(B, N, advD.device)
    _, clean = ops.knn_planar(oriD, oriD, kk)
    stale = torch.roll(clean, 1, 0).contiguous()
    bad = clean.clone()
    bad[:, ::3, 1] = bad[:, ::3, 0]                  # duplicates: fewer than K distinct candidates within the radius
    bad[:, 1::7, 2] = N + 5                          # out of range: no radius
    for method in (1, 2, 0):                         # slab, cell grid (one wavefront per query), the library's choice
        for prior in (clean, stale, bad, None):
            d, i = ops.knn_self_planar(advD, kk, prior=prior, scratch=scratch, method=method)
            assert torch.equal(i, bi) and torch.equal(d, bd), (method, prior is None)
    if kk <= 40:
        # the position-list forms of the slab kernel (knn_slabp_kernel<40, false> / <56, true>), which the library takes
        # for large launches only
        for method in (3, 4):     # 3: the (distance, index)-list slab kernel, 4: the position-list kernel, whatever the size
            for prior in (clean, stale, bad):
                d, i = ops.knn_self_planar(advD, kk, prior=prior, scratch=scratch, method=method)
                assert torch.equal(i, bi) and torch.equal(d, bd), method
    d, i = ops.knn_self_planar(advD, kk, prior=clean, scratch=None)
    assert torch.equal(i, bi) and torch.equal(d, bd)
    # in place over the prior (the loop's double buffer may alias)
    tbl = clean.clone()
    d, i = ops.knn_self_planar(advD, kk, prior=tbl, scratch=scratch, out=(torch.empty_like(bd), tbl))
    assert torch.equal(tbl, bi)


@pytest.mark.parametrize("Na,Nr,k", [(64, 160, 4), (200, 1500, 16), (1024, 4096, 16)])
def test_loss_utils_unequal_cloud_sizes(Na, Nr, k):
    """The dense-cloud path (--is_subsample_opt, geoA3_attack.py:283-284): the adversarial cloud is an npoint-sample,
    the clean cloud / normals / kappa_ori keep all Nr points.  Values and gradients against the oracle's autograd."""
    from geoa3_amd import loss_utils as LU
    B = 3
    ori, nrm = O.make_synthetic_clouds(B, Nr, seed=Na + Nr)
    g = torch.Generator().manual_seed(5)
    pick = torch.stack([torch.randperm(Nr, generator=g)[:Na] for _ in range(B)])
    adv0 = torch.gather(ori, 2, pick.unsqueeze(1).expand(B, 3, Na)) + 0.01 * torch.randn(B, 3, Na, generator=g)
    kap_ori = O.get_kappa_ori(ori, nrm, k)
    want, gwant = {}, {}
    for name, fn in [("cd", O.chamfer_loss), ("pcd", O.pseudo_chamfer_loss), ("hd", O.hausdorff_loss)]:
        a = adv0.clone().requires_grad_()
        v = fn(a, ori)
        v.sum().backward()
        want[name], gwant[name] = v.detach(), a.grad
    a = adv0.clone().requires_grad_()
    kap_adv, _ = O.get_kappa_adv(a, ori, nrm, k)
    curv = O.curvature_loss(a, ori, kap_adv, kap_ori)
    curv.sum().backward()
    want["curv"], gwant["curv"] = curv.detach(), a.grad
    d_ori, d_nrm = dev(ori), dev(nrm)
    d_kap = LU._get_kappa_ori(d_ori, d_nrm, k)
    np.testing.assert_allclose(d_kap.cpu().numpy(), kap_ori.numpy(), rtol=2e-5, atol=2e-6)
    for name, fn in [("cd", LU.chamfer_loss), ("pcd", LU.pseudo_chamfer_loss), ("hd", LU.hausdorff_loss)]:
        a = dev(adv0).requires_grad_()
        v = fn(a, d_ori)
        np.testing.assert_allclose(v.detach().cpu().numpy(), want[name].numpy(), rtol=2e-5, atol=1e-8)
        v.sum().backward()
        np.testing.assert_allclose(a.grad.cpu().numpy(), gwant[name].numpy(), rtol=1e-4, atol=1e-8)
    a = dev(adv0).requires_grad_()
    ka, _ = LU._get_kappa_adv(a, d_ori, d_nrm, k)
    np.testing.assert_allclose(ka.detach().cpu().numpy(), kap_adv.detach().numpy(), rtol=2e-5, atol=2e-6)
    c = LU.curvature_loss(a, d_ori, ka, d_kap)
    np.testing.assert_allclose(c.detach().cpu().numpy(), want["curv"].numpy(), rtol=1e-4, atol=1e-8)
    c.sum().backward()
    np.testing.assert_allclose(a.grad.cpu().numpy(), gwant["curv"].numpy(), rtol=2e-3, atol=2e-6)


@pytest.mark.parametrize("B,Na,Nr,kind", [(3, 1024, 1024, "attack"), (2, 1024, 1024, "far"), (2, 300, 777, "attack"),
                                          (2, 64, 64, "dup"), (1, 4096, 4096, "attack"), (2, 1024, 4096, "attack"),
                                          (2, 1000, 1000, "line"), (2, 128, 128, "same"), (1, 2500, 2049, "far")])
def test_grid_nn1_is_bit_identical_to_brute_force(ops, B, Na, Nr, kind):
    """Uniform-grid search == all-pairs search, bit for bit: attack-like offsets, queries far outside the searched
    cloud's bounding box, duplicates / exact ties, degenerate clouds (collinear, all points equal)."""
    g = torch.Generator().manual_seed(Na * 7 + Nr)
    ori, _ = O.make_synthetic_clouds(B, max(Na, Nr), seed=Na + Nr)
    r = ori[:, :, :Nr].contiguous()
    a = ori[:, :, :Na].contiguous() if Na <= Nr else ori[:, :, :Na].contiguous()
    a = a + 0.03 * torch.randn(B, 3, Na, generator=g)
    if kind == "far":
        a[:, :, : Na // 2] += torch.tensor([3.0, -2.0, 0.5]).view(1, 3, 1)
        a[:, :, 5] = 1e3
    elif kind == "dup":
        a[:, :, 1] = a[:, :, 0]
        r[:, :, 7] = r[:, :, 3]
        a[:, :, 9] = r[:, :, 11]
    elif kind == "line":
        r[:, 1:, :] = 0.25
        a[:, 2, :] = 0.25
    elif kind == "same":
        r[:] = r[:, :, :1]
        a[:, :, ::2] = r[:, :, :1]
    want = ops.nn1_pair(dev(a), dev(r))
    got = ops.nn1_pair(dev(a), dev(r), method="grid")
    for w, x in zip(want, got):
        assert torch.equal(w, x)
    d_ao, i_ao = O.knn_points(a.permute(0, 2, 1), r.permute(0, 2, 1), 1)
    assert torch.equal(got[1].cpu().long(), i_ao[:, :, 0]) and torch.equal(got[0].cpu(), d_ao[:, :, 0])
    one = ops.nn1_pair(dev(a), dev(r), both=False, method="grid")
    assert one[2] is None and torch.equal(one[0], want[0]) and torch.equal(one[1], want[1])
    # any prior (exact answers, garbage, out-of-range indices) only changes the work, never the result
    for prior in ((want[1].clone(), want[3].clone()),
                  (torch.randint(0, Nr, (B, Na), generator=g).int().cuda(), torch.randint(0, Na, (B, Nr), generator=g).int().cuda()),
                  (torch.full((B, Na), 10 ** 6).int().cuda(), torch.full((B, Nr), -5).int().cuda())):
        again = ops.nn1_pair(dev(a), dev(r), method="grid", prior=prior)
        for w, x in zip(want, again):
            assert torch.equal(w, x)


def _objective_inputs(ops, adv, ori, nrm, k):
    advD, oriD, nrmD = dev(adv), dev(ori), dev(nrm)
    d_ao, i_ao, d_oa, i_oa = ops.nn1_pair(advD, oriD)
    _, knn_ori = ops.knn_planar(oriD, oriD, k + 1)
    kap = ops.kappa(oriD, nrmD, knn_ori)
    _, knn_adv = ops.knn_planar(advD, advD, k + 1)
    return dict(normal_ori=nrmD, kappa_ori=kap, d_ao=d_ao, i_ao=i_ao, d_oa=d_oa, i_oa=i_oa, knn_adv=knn_adv, k=k,
                dis_type=1, w_dis=1.0, w_hd=0.1, w_curv=1.0), advD, oriD


def _oracle_objective(adv, ori, nrm, k):
    a = adv.clone().requires_grad_()
    ka, _ = O.get_kappa_adv(a, ori, nrm, k)
    con = O.chamfer_loss(a, ori) + 0.1 * O.hausdorff_loss(a, ori) + O.curvature_loss(a, ori, ka, O.get_kappa_ori(ori, nrm, k))
    (g,) = torch.autograd.grad(con.sum(), a)
    return con.detach(), g


@pytest.mark.parametrize("N,k,ncoin", [(512, 8, 40), (700, 16, 40), (1024, 16, 120)])
def test_pair_parallel_objective_long_rows_are_bit_stable_on_a_full_chip(ops, N, k, ncoin):
    """NOTEBOOK 5a, third sighting: with the packed-FP32 instructions the SLP vectoriser formed in geo_fused_kernel, the
    gradient of the points that own 40-source rows differed from launch to launch -- in half of the launches at 40
    instances (one workgroup per CU on 40 CUs, four wavefronts per SIMD), almost never at the 3 instances the test above
    uses.  geom_loss.hip is compiled without them (geoa3_amd/build.py FILE_FLAGS / ISA_GUARDS); 200 launches, every bit."""
    B = 40
    ori, nrm = O.make_synthetic_clouds(B, N, seed=N + k)
    g = torch.Generator().manual_seed(N)
    adv = ori + 0.02 * torch.randn(B, 3, N, generator=g)
    adv[:, :, 100:100 + ncoin] = adv[:, :, 100:101]
    kw, advD, oriD = _objective_inputs(ops, adv, ori, nrm, k)
    out = ops.geo_loss_grad(advD, oriD, deterministic=True, **kw)
    grad, con = out["grad"].clone(), out["constrain"].clone()
    for rep in range(200):
        out = ops.geo_loss_grad(advD, oriD, deterministic=True, out=out, **kw)
        assert torch.equal(out["grad"], grad) and torch.equal(out["constrain"], con), rep


@pytest.mark.parametrize("N,k,kind", [(1024, 16, "coincident120"), (700, 16, "coincident40"), (512, 8, "coincident40"),
                                      (1024, 16, "plain"), (2048, 16, "plain")])
def test_pair_parallel_objective_special_paths(ops, N, k, kind):
    """The pair-parallel objective kernel (clouds of at most 1024 points) off its fast path: exactly coincident points
    all list the lowest-indexed of their twins (ties go to the lower index), which therefore collect 40 / 120 sources
    each -- rows of 33..64 sources (two sorted runs merged) and rows beyond the LDS capacity (summed by their owner from
    the table) -- through zero-length pairs (the clamped normalisation); 2048 points take the one-workgroup kernel.  Against the oracle's autograd, reproducible bit
    for bit, and independent of the batch (a batch of one is split over four owner-range workgroups, a large one is
    not)."""
    B = 3
    ori, nrm = O.make_synthetic_clouds(B, N, seed=N + k)
    g = torch.Generator().manual_seed(N)
    adv = ori + 0.02 * torch.randn(B, 3, N, generator=g)
    ncoin = int(kind[len("coincident"):]) if kind.startswith("coincident") else 0
    if ncoin:
        adv[:, :, 100:100 + ncoin] = adv[:, :, 100:101]
    kw, advD, oriD = _objective_inputs(ops, adv, ori, nrm, k)
    if ncoin:
        deg = torch.zeros(B, N, device="cuda").scatter_add_(1, kw["knn_adv"][:, :, 1:].reshape(B, -1).long(),
                                                          torch.ones(B, N * k, device="cuda"))
        assert deg.max().item() >= ncoin - 2       # the lowest-indexed twins are listed by all the others
    out = ops.geo_loss_grad(advD, oriD, deterministic=True, **kw)
    con, grad = out["constrain"].clone(), out["grad"].clone()
    want_con, want_g = _oracle_objective(adv, ori, nrm, k)
    np.testing.assert_allclose(con.cpu().numpy(), want_con.numpy(), rtol=5e-5, atol=1e-7)
    # (coincident points: the gradient of |v| / max(|v|, eps) at v = 0 is implementation defined in torch's autograd as
    # here -- compare away from them)
    keep = torch.ones(N, dtype=torch.bool)
    if ncoin:
        involved = (kw["knn_adv"].cpu()[:, :, :].unsqueeze(-1) == torch.arange(100, 100 + ncoin).view(1, 1, 1, -1)).any(-1).any(-1).any(0)
        keep &= ~involved
        keep[100:100 + ncoin] = False
    scale = want_g.abs().max().item()
    np.testing.assert_allclose(grad.cpu()[:, :, keep].numpy(), want_g[:, :, keep].numpy(), rtol=2e-4, atol=2e-6 * max(scale, 1.0))
    for _ in range(3):
        again = ops.geo_loss_grad(advD, oriD, deterministic=True, **kw)
        assert torch.equal(again["grad"], grad) and torch.equal(again["constrain"], con)
    # batch independence: instance 1 alone (B = 1: four owner-range workgroups) and inside a batch of 160 copies (one)
    one = {n: (v[1:2].contiguous() if torch.is_tensor(v) else v) for n, v in kw.items()}
    alone = ops.geo_loss_grad(advD[1:2].contiguous(), oriD[1:2].contiguous(), deterministic=True, **one)
    assert torch.equal(alone["grad"][0], grad[1]) and torch.equal(alone["constrain"][0], con[1])
    if N <= 1024:
        many = {n: (v[1:2].expand(160, *v.shape[1:]).contiguous() if torch.is_tensor(v) else v) for n, v in kw.items()}
        big = ops.geo_loss_grad(advD[1:2].expand(160, 3, N).contiguous(), oriD[1:2].expand(160, 3, N).contiguous(),
                                deterministic=True, **many)
        assert torch.equal(big["grad"][77], grad[1]) and torch.equal(big["constrain"][159], con[1])


@pytest.mark.parametrize("N,Nr,k,kind", [(1500, 1500, 16, "plain"), (2048, 2048, 32, "plain"), (4096, 4096, 32, "plain"),
                                         (4096, 4096, 32, "coincident130"), (2048, 2048, 16, "coincident40"),
                                         (1500, 3000, 16, "plain"), (1100, 1100, 40, "plain"), (1024, 1024, 40, "plain"),
                                         (600, 600, 48, "coincident40")])
def test_fixed_point_objective_for_big_clouds(ops, N, Nr, k, kind):
    """Clouds of 1025..4096 points (BASELINE configs[4]): geo_big_gather_kernel + geo_big_kernel (a 16-byte record per point in a
    scratch buffer, the gradient as 64-bit fixed-point sums in LDS).  Against the oracle's autograd and the one-workgroup
    kernel it replaces; 130 / 40 coincident points (zero-length pairs, hubs of the neighbour graph);
    a clean cloud larger than the sample (Nr > N); k = 40 / 48 on 64 lanes per centre, also for clouds of at most 1024
    points (where k > 32 takes this kernel instead of geo_fused_kernel<64> and its spills).  Reproducible bit for bit and
    independent of the batch (alone / inside a batch of 19 copies: other XCDs, other workgroup ids)."""
    B = 3
    ori, nrm = O.make_synthetic_clouds(B, Nr, seed=N + k)
    g = torch.Generator().manual_seed(N)
    adv = ori[:, :, :N] + 0.02 * torch.randn(B, 3, N, generator=g)
    ncoin = int(kind[len("coincident"):]) if kind.startswith("coincident") else 0
    if ncoin:
        adv[:, :, 100:100 + ncoin] = adv[:, :, 100:101]
    kw, advD, oriD = _objective_inputs(ops, adv, ori, nrm, k)
    out = ops.geo_loss_grad(advD, oriD, deterministic=True, want_kappa=True, **kw)
    con, grad, kap = out["constrain"].clone(), out["grad"].clone(), out["kappa_adv"].clone()
    old = ops.geo_loss_grad(advD, oriD, deterministic=True, want_kappa=True, scratch=False, **kw)
    np.testing.assert_allclose(con.cpu().numpy(), old["constrain"].cpu().numpy(), rtol=2e-5, atol=1e-8)
    np.testing.assert_allclose(kap.cpu().numpy(), old["kappa_adv"].cpu().numpy(), rtol=2e-5, atol=2e-6)
    for name in ("dis_loss", "hd_loss", "curv_loss"):
        np.testing.assert_allclose(out[name].cpu().numpy(), old[name].cpu().numpy(), rtol=2e-5, atol=1e-9)
    scale = max(old["grad"].abs().max().item(), 1.0)
    keep = torch.ones(N, dtype=torch.bool)
    if ncoin:   # (the gradient of |v| / max(|v|, eps) at v = 0 is implementation defined: compare away from the twins)
        involved = (kw["knn_adv"].cpu().unsqueeze(-1) == torch.arange(100, 100 + ncoin).view(1, 1, 1, -1)).any(-1).any(-1).any(0)
        keep &= ~involved
        keep[100:100 + ncoin] = False
    np.testing.assert_allclose(grad.cpu()[:, :, keep].numpy(), old["grad"].cpu()[:, :, keep].numpy(), rtol=2e-4, atol=2e-6 * scale)
    if Nr == N and N <= 2048:     # (the dense oracle at N = 4096 takes minutes; the old kernel is pinned to it at 2048)
        want_con, want_g = _oracle_objective(adv, ori, nrm, k)
        np.testing.assert_allclose(con.cpu().numpy(), want_con.numpy(), rtol=5e-5, atol=1e-7)
        np.testing.assert_allclose(grad.cpu()[:, :, keep].numpy(), want_g[:, :, keep].numpy(), rtol=2e-4,
                                   atol=2e-6 * max(want_g.abs().max().item(), 1.0))
    for _ in range(3):
        again = ops.geo_loss_grad(advD, oriD, deterministic=True, **kw)
        assert torch.equal(again["grad"], grad) and torch.equal(again["constrain"], con)
    one = {n: (v[1:2].contiguous() if torch.is_tensor(v) else v) for n, v in kw.items()}
    alone = ops.geo_loss_grad(advD[1:2].contiguous(), oriD[1:2].contiguous(), deterministic=True, **one)
    assert torch.equal(alone["grad"][0], grad[1]) and torch.equal(alone["constrain"][0], con[1])
    many = {n: (v[1:2].expand(19, *v.shape[1:]).contiguous() if torch.is_tensor(v) else v) for n, v in kw.items()}
    big = ops.geo_loss_grad(advD[1:2].expand(19, 3, N).contiguous(), oriD[1:2].expand(19, 3, Nr).contiguous(),
                            deterministic=True, **many)
    assert torch.equal(big["grad"][7], grad[1]) and torch.equal(big["grad"][18], grad[1]) and torch.equal(big["constrain"][18], con[1])
    # the operator-level vector-Jacobian product of _get_kappa_adv (dkappa given) takes the same kernels
    dk = torch.randn(B, N, generator=g).cuda()
    vjp = {n: v for n, v in kw.items()}
    vjp.update(w_dis=0.0, w_hd=0.0, w_curv=0.0, dis_type=0)
    a1 = ops.geo_loss_grad(advD, oriD, deterministic=True, dkappa=dk, **vjp)
    a0 = ops.geo_loss_grad(advD, oriD, deterministic=True, dkappa=dk, scratch=False, **vjp)
    np.testing.assert_allclose(a1["grad"].cpu()[:, :, keep].numpy(), a0["grad"].cpu()[:, :, keep].numpy(), rtol=2e-4,
                               atol=2e-6 * max(a0["grad"].abs().max().item(), 1.0))


@pytest.mark.parametrize("N,k,case", [(2048, 16, "dkappa1e4"), (4096, 32, "dkappa1e4"), (2048, 16, "near1e-6"),
                                      (4096, 32, "near1e-6"), (2048, 16, "weights"), (2048, 16, "all")])
def test_fixed_point_objective_outside_the_attack_loops_magnitudes(ops, N, k, case):
    """geo_big_kernel's 64-bit fixed-point sums take their scales from the instance (GeoFix: fine + coarse, no silent
    clamp): a caller-supplied dkappa ~ N(0, 1e4) (the operator-level VJP of _get_kappa_adv, Lib/loss_utils.py:64-82),
    pairs 1e-6 apart (NOT coincident: pair terms ~ 2 dk / r), w_dis = 1e-3 with w_curv = 1e3.  Against the one-workgroup
    kernel (fp32 ordered sums) at rtol 2e-4, against the oracle at N = 2048; bit-reproducible."""
    B = 2
    ori, nrm = O.make_synthetic_clouds(B, N, seed=N + 3)
    g = torch.Generator().manual_seed(N + len(case))
    adv = ori + 0.02 * torch.randn(B, 3, N, generator=g)
    if case in ("near1e-6", "all"):
        for j in range(40, 60):                                  # twenty pairs at ~1e-6
            adv[:, :, 200 + j] = adv[:, :, j] + 1e-6 * torch.randn(B, 3, generator=g)
    w_dis, w_curv = (1e-3, 1e3) if case in ("weights", "all") else (1.0, 1.0)
    kw, advD, oriD = _objective_inputs(ops, adv, ori, nrm, k)
    kw.update(w_dis=w_dis, w_curv=w_curv, w_hd=0.1)
    new = ops.geo_loss_grad(advD, oriD, deterministic=True, **kw)
    old = ops.geo_loss_grad(advD, oriD, deterministic=True, scratch=False, **kw)
    gn, go = new["grad"].cpu().numpy(), old["grad"].cpu().numpy()
    assert np.isfinite(gn).all()
    np.testing.assert_allclose(gn, go, rtol=2e-4, atol=2e-7 * np.abs(go).max())
    if N <= 2048:
        a = adv.clone().requires_grad_()
        ka, _ = O.get_kappa_adv(a, ori, nrm, k)
        con = w_dis * O.chamfer_loss(a, ori) + 0.1 * O.hausdorff_loss(a, ori) + w_curv * O.curvature_loss(a, ori, ka, O.get_kappa_ori(ori, nrm, k))
        (gw,) = torch.autograd.grad(con.sum(), a)
        np.testing.assert_allclose(gn, gw.numpy(), rtol=2e-4, atol=2e-6 * float(gw.abs().max()))
    assert torch.equal(ops.geo_loss_grad(advD, oriD, deterministic=True, **kw)["grad"], new["grad"])
    if case in ("dkappa1e4", "all"):
        dk = (1e2 * torch.randn(B, N, generator=g)).cuda() * 1e2     # ~ N(0, 1e4)
        vjp = dict(kw)
        vjp.update(w_dis=0.0, w_hd=0.0, w_curv=0.0, dis_type=0)
        a1 = ops.geo_loss_grad(advD, oriD, deterministic=True, dkappa=dk, **vjp)["grad"]
        a0 = ops.geo_loss_grad(advD, oriD, deterministic=True, dkappa=dk, scratch=False, **vjp)["grad"]
        assert torch.isfinite(a1).all()
        np.testing.assert_allclose(a1.cpu().numpy(), a0.cpu().numpy(), rtol=2e-4, atol=2e-7 * float(a0.abs().max()))
        assert torch.equal(ops.geo_loss_grad(advD, oriD, deterministic=True, dkappa=dk, **vjp)["grad"], a1)


def test_fixed_point_objective_is_loud_about_nan_and_huge_terms(ops):
    """A NaN coordinate, or a pair term beyond the coarse range (dkappa = 1e30 on a pair 1e-7 apart), poisons the gradient of
    the points it reaches with NaN instead of clamping silently (round 4: every term clamped to +-2^18, a NaN became
    -262144)."""
    B, N, k = 2, 2048, 16
    ori, nrm = O.make_synthetic_clouds(B, N, seed=5)
    g = torch.Generator().manual_seed(1)
    adv = ori + 0.02 * torch.randn(B, 3, N, generator=g)
    kw, advD, oriD = _objective_inputs(ops, adv, ori, nrm, k)
    bad = advD.clone()
    bad[0, 1, 77] = float("nan")
    got = ops.geo_loss_grad(bad, oriD, deterministic=True, **kw)["grad"]
    assert torch.isnan(got[0, :, 77]).all() and torch.isfinite(got[1]).all()
    adv2 = adv.clone()
    adv2[:, :, 301] = adv2[:, :, 300] + 1e-7
    kw2, adv2D, _ = _objective_inputs(ops, adv2, ori, nrm, k)
    dk = torch.ones(B, N).cuda()
    dk[0, 300] = 1e30
    vjp = dict(kw2)
    vjp.update(w_dis=0.0, w_hd=0.0, w_curv=0.0, dis_type=0)
    got = ops.geo_loss_grad(adv2D, oriD, deterministic=True, dkappa=dk, **vjp)["grad"]
    assert not torch.isfinite(got[0, :, 301]).all()        # (NaN, or the one-workgroup kernel's inf: never a finite clamp)
    assert torch.isfinite(got[1]).all()


@pytest.mark.parametrize("N,k,nblob", [(1024, 16, 700), (1024, 16, 120), (512, 8, 300), (700, 16, 60)])
def test_pair_parallel_objective_with_overflowing_reverse_lists(ops, N, k, nblob):
    """geo_fused_kernel's rows hold ~50 sources; a destination with more -- here `nblob` clean points of a tiny dense blob
    whose common nearest adversarial point is ONE point (what a CAD cloud with a small dense part looks like once the
    offsets exceed the blob's size), plus hubs of the K-NN graph -- goes through the fixed-point pool.  Values and gradients
    against the oracle's autograd (Lib/loss_utils.py:28-97), reproducible, batch-independent."""
    B = 3
    ori, nrm = O.make_synthetic_clouds(B, N, seed=N + nblob)
    g = torch.Generator().manual_seed(nblob)
    ori[:, :, :nblob] = ori[:, :, :1] * 1.5 + 1e-3 * torch.randn(B, 3, nblob, generator=g)     # the blob, off the surface
    adv = ori + 0.02 * torch.randn(B, 3, N, generator=g)
    adv[:, :, 1:nblob] += 0.5 * torch.randn(B, 3, nblob - 1, generator=g)                        # its points have left; point 0 stays
    kw, advD, oriD = _objective_inputs(ops, adv, ori, nrm, k)
    assert int(torch.bincount(kw["i_oa"][0].long().cpu(), minlength=N).max()) > 56             # a row beyond the capacity
    out = ops.geo_loss_grad(advD, oriD, deterministic=True, **kw)
    want_con, want_g = _oracle_objective(adv, ori, nrm, k)
    np.testing.assert_allclose(out["constrain"].cpu().numpy(), want_con.numpy(), rtol=5e-5, atol=1e-7)
    np.testing.assert_allclose(out["grad"].cpu().numpy(), want_g.numpy(), rtol=2e-4, atol=2e-6 * float(want_g.abs().max()))
    grad = out["grad"].clone()
    for _ in range(3):
        assert torch.equal(ops.geo_loss_grad(advD, oriD, deterministic=True, **kw)["grad"], grad)
    one = {n: (v[1:2].contiguous() if torch.is_tensor(v) else v) for n, v in kw.items()}
    alone = ops.geo_loss_grad(advD[1:2].contiguous(), oriD[1:2].contiguous(), deterministic=True, **one)
    assert torch.equal(alone["grad"][0], grad[1])


def test_pair_parallel_objective_more_overflowed_rows_than_pool_slots(ops):
    """More rows overflow in ONE workgroup than the fixed-point pool has slots (128): a dense clean cloud (16 x the
    adversarial cloud's points, the dense-cloud path's shape) whose nearest adversarial points are the 140 that stayed
    near the surface, ~117 clean sources each.  Round 5 wrote NaN into those rows; now every overflowed row of such a
    workgroup is summed by its owner's walk over the tables: finite, equal to the oracle's autograd
    (Lib/loss_utils.py:28-50), bit-reproducible and batch-independent."""
    B, N, Nr, stay = 2, 1024, 16384, 140
    g = torch.Generator().manual_seed(17)
    ori = torch.randn(B, 3, Nr, generator=g)
    ori = ori / ori.norm(dim=1, keepdim=True)
    adv = torch.randn(B, 3, N, generator=g)
    adv = adv / adv.norm(dim=1, keepdim=True)
    i = torch.arange(stay, dtype=torch.float64)                # the stayers on a Fibonacci lattice: equal shares of the clean cloud
    z = 1.0 - (2.0 * i + 1.0) / stay
    phi = i * (3.141592653589793 * (3.0 - 5.0 ** 0.5))
    fib = torch.stack([(1 - z * z).sqrt() * phi.cos(), (1 - z * z).sqrt() * phi.sin(), z]).float()
    adv[:, :, :stay] = fib.unsqueeze(0) + 1e-3 * torch.randn(B, 3, stay, generator=g)
    adv[:, 0, stay:] += 10.0                                   # the others have left: no clean point is nearest to them
    advD, oriD = dev(adv), dev(ori)
    d_ao, i_ao, d_oa, i_oa = ops.nn1_pair(advD, oriD)
    rows = torch.bincount(i_oa[0].long().cpu(), minlength=N)
    assert int((rows > 56).sum()) > 128 and int(rows[stay:].sum()) == 0      # > 128 overflowed rows (capacity 48), all in the first owner range
    kw = dict(d_ao=d_ao, i_ao=i_ao, d_oa=d_oa, i_oa=i_oa, k=0, dis_type=1, w_dis=1.0, w_hd=0.1, w_curv=0.0)
    out = ops.geo_loss_grad(advD, oriD, deterministic=True, **kw)
    a = adv.clone().requires_grad_()
    con = O.chamfer_loss(a, ori) + 0.1 * O.hausdorff_loss(a, ori)
    (want_g,) = torch.autograd.grad(con.sum(), a)
    grad = out["grad"].clone()
    assert torch.isfinite(grad).all()
    np.testing.assert_allclose(out["constrain"].cpu().numpy(), con.detach().numpy(), rtol=5e-5, atol=1e-7)
    np.testing.assert_allclose(grad.cpu().numpy(), want_g.numpy(), rtol=2e-4, atol=2e-6 * float(want_g.abs().max()))
    for _ in range(3):
        assert torch.equal(ops.geo_loss_grad(advD, oriD, deterministic=True, **kw)["grad"], grad)
    one = {n: (v[1:2].contiguous() if torch.is_tensor(v) else v) for n, v in kw.items()}
    alone = ops.geo_loss_grad(advD[1:2].contiguous(), oriD[1:2].contiguous(), deterministic=True, **one)
    assert torch.equal(alone["grad"][0], grad[1])
