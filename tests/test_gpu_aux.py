"""HIP helpers of the dense-cloud / defence / measurement paths (geoa3_amd/utility.py over geom_aux.hip) against
the reference's own outputs (tests/golden/geoa3_golden_aux.npz) and against the oracle on fresh inputs."""
import os

import numpy as np
import pytest
import torch

from oracle import aux_oracle as A
from oracle import geoa3_oracle as O
from tests.test_oracle_aux import perp_terms_close

pytestmark = pytest.mark.gpu
T = torch.from_numpy


@pytest.fixture(scope="module")
def aux():
    return np.load(os.path.join(os.path.dirname(__file__), "golden", "geoa3_golden_aux.npz"))


def test_farthest_points_sample_golden(aux):
    from geoa3_amd import utility as U
    for tag in aux["fps/cases"]:
        pre = "fps/%s/" % tag
        pts, idx = U.fps_indices(T(aux[pre + "pc"]).cuda(), int(aux[pre + "m"]), T(aux[pre + "start"]).cuda())
        assert np.array_equal(pts.cpu().numpy(), aux[pre + "pts"]), tag       # bit-exact selection


@pytest.mark.parametrize("b,n,m,seed", [(3, 1024, 512, 1), (2, 5000, 300, 2), (2, 1500, 1500, 3), (1, 16384, 64, 4),
                                         (4, 37, 20, 5)])
def test_farthest_points_sample_oracle(b, n, m, seed):
    from geoa3_amd import utility as U
    pc, _ = O.make_synthetic_clouds(b, n, 300 + seed)
    start = torch.randint(n, (b,), generator=torch.Generator().manual_seed(seed))
    want_pts, want_idx = A.farthest_points_sample(pc, m, start)
    pts, idx = U.fps_indices(pc.cuda(), m, start.cuda())
    assert torch.equal(idx.cpu().long(), want_idx)
    assert torch.equal(pts.cpu(), want_pts)


def test_farthest_points_sample_gradient():
    from geoa3_amd import utility as U
    pc, _ = O.make_synthetic_clouds(2, 200, 310)
    start = torch.tensor([5, 17])
    x = pc.clone().requires_grad_()
    pts, sel = A.farthest_points_sample(x.detach(), 40, start)
    w = torch.randn(2, 3, 40, generator=torch.Generator().manual_seed(1))
    (torch.gather(x, 2, sel.unsqueeze(1).expand(2, 3, 40)) * w).sum().backward()
    xg = pc.clone().cuda().requires_grad_()
    out = U.farthest_points_sample(xg, 40, start.cuda())
    (out * w.cuda()).sum().backward()
    assert torch.equal(out.detach().cpu(), pts)
    assert torch.equal(xg.grad.cpu(), x.grad)


def test_estimate_normal_via_ori_normal(aux):
    from geoa3_amd import utility as U
    for tag in aux["nvo/cases"]:
        pre = "nvo/%s/" % tag
        est = U.estimate_normal_via_ori_normal(T(aux[pre + "adv"]).cuda(), T(aux[pre + "ori"]).cuda(),
                                               T(aux[pre + "nrm"]).cuda(), int(aux[pre + "k"]))
        np.testing.assert_allclose(est.cpu().numpy(), aux[pre + "est"], rtol=2e-6, atol=2e-7)
    # batched == per instance (the reference's broadcast only allows b = 1)
    ori, nrm = O.make_synthetic_clouds(3, 400, 320)
    adv = ori[:, :, :150] + torch.randn(3, 3, 150, generator=torch.Generator().manual_seed(2)) * 0.02
    adv[:, :, :20] = ori[:, :, :20]
    got = U.estimate_normal_via_ori_normal(adv.cuda(), ori.cuda(), nrm.cuda(), 3).cpu()
    for i in range(3):
        want = A.estimate_normal_via_ori_normal(adv[i:i + 1], ori[i:i + 1], nrm[i:i + 1], 3)
        np.testing.assert_allclose(got[i:i + 1].numpy(), want.numpy(), rtol=2e-6, atol=2e-7)


def test_local_frames_are_eigenpairs():
    from geoa3_amd import utility as U
    pc, _ = O.make_synthetic_clouds(3, 500, 330)
    k = 12
    evals, evecs, idx = U.local_frames(pc.cuda(), k)
    cov = A.local_covariance(pc, k).double()                                  # [b,n,3,3]
    w = evals.cpu().permute(0, 2, 1).double()                                 # [b,n,3]
    v = evecs.cpu().permute(0, 3, 2, 1).double()                              # [b,n,xyz,e]
    want_w = torch.linalg.eigvalsh(cov)
    scale = want_w[..., 2:3]
    assert ((w - want_w).abs() / scale).max().item() < 5e-5
    resid = torch.matmul(cov, v) - v * w.unsqueeze(2)
    assert (resid.abs().amax(dim=(2, 3)) / scale[..., 0]).max().item() < 5e-5
    eye = torch.matmul(v.transpose(2, 3), v)
    assert (eye - torch.eye(3, dtype=torch.float64)).abs().max().item() < 1e-5
    lead = v.abs().argmax(dim=2, keepdim=True)                                # sign convention
    assert (torch.gather(v, 2, lead) > 0).all()


def test_estimate_perpendicular(aux):
    from geoa3_amd import utility as U
    for tag in aux["perp/cases"]:
        pre = "perp/%s/" % tag
        pc, k, clip = T(aux[pre + "pc"]), int(aux[pre + "k"]), float(aux[pre + "clip"])
        a1, a2 = T(aux[pre + "aux1"]), T(aux[pre + "aux2"])
        noise = U.estimate_perpendicular(pc.cuda(), k, clip=clip, aux=(a1.cuda(), a2.cuda())).cpu()
        perp_terms_close(noise, pc, k, a1, a2, clip)
    pc, _ = O.make_synthetic_clouds(2, 300, 340)
    noise = U.estimate_perpendicular(pc.cuda(), 16, sigma=0.01, clip=0.05)     # own draws: bounded, tangent
    assert noise.shape == (2, 3, 300) and noise.abs().max().item() <= 0.1
    assert 1e-3 < noise.std().item() < 2e-2


def test_defense_point_removal(aux):
    from geoa3_amd import utility as U
    for tag in aux["def/cases"]:
        pre = "def/%s/" % tag
        kept, num = U.point_removal_fn(T(aux[pre + "pc"]).cuda(), str(aux[pre + "type"]), int(aux[pre + "drop"]),
                                       float(aux[pre + "alpha"]), int(aux[pre + "knn"]))
        assert num == int(aux[pre + "num"]), tag
        assert np.array_equal(kept.cpu().numpy(), aux[pre + "kept"]), tag
    kept, num = U.point_removal_fn(T(aux["def/rand/pc"]).cuda(), "rand_drop", int(aux["def/rand/drop"]), 1.1, 2,
                                   perm=T(aux["def/rand/perm"]).cuda())
    assert np.array_equal(kept.cpu().numpy(), aux["def/rand/kept"]) and num == 20


@pytest.mark.parametrize("dtype,drop,alpha,knn", [("outliers_fixNum", 128, 1.1, 2), ("outliers_variance", 0, 1.1, 2),
                                                   ("outliers_variance", 0, 0.3, 8), ("outliers_fixNum", 1, 1.1, 16)])
def test_defense_batched_matches_oracle(dtype, drop, alpha, knn):
    from geoa3_amd import utility as U
    pc, _ = O.make_synthetic_clouds(6, 1024, 350)
    pc = pc + torch.randn(6, 3, 1024, generator=torch.Generator().manual_seed(3)) * 0.01
    dis = U.sor_statistic(pc.cuda(), knn).cpu()
    np.testing.assert_allclose(dis.numpy(), A.sor_statistic(pc, knn).numpy(), rtol=2e-6)
    idx, cnt = U.outlier_removal_indices(pc.cuda(), dtype, drop, alpha, knn)
    idx, cnt = idx.cpu(), cnt.cpu()
    for i in range(6):
        _, num, want = A.outlier_removal(pc[i:i + 1], dtype, drop, alpha, knn)
        assert int(cnt[i]) == 1024 - num
        assert idx[i, : int(cnt[i])].tolist() == want.tolist()
        assert (idx[i, int(cnt[i]):] == -1).all()


def test_smoothness(aux):
    from geoa3_amd import utility as U
    for tag in aux["smooth/cases"]:
        pre = "smooth/%s/" % tag
        s = U.smoothness(T(aux[pre + "pc"]).unsqueeze(0).cuda(), int(aux[pre + "k"]), int(aux[pre + "k2"]))
        np.testing.assert_allclose(float(s[0]), float(aux[pre + "value"][0]), rtol=5e-5)
    pc, _ = O.make_synthetic_clouds(4, 1024, 360)
    pc = pc + torch.randn(4, 3, 1024, generator=torch.Generator().manual_seed(4)) * 0.005
    got = U.smoothness(pc.cuda(), 16, 16).cpu()
    for i in range(4):
        np.testing.assert_allclose(float(got[i]), float(A.smoothness(pc[i].t().contiguous(), 16, 16)), rtol=5e-5)
