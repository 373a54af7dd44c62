"""Size-independent properties of the HIP path at BASELINE.json's full size (B=250, N=1024, k=16), where the dense
CPU oracle is too slow to be the checker."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
B, N, K = 250, 1024, 17


@pytest.fixture(scope="module")
def clouds():
    from geoa3_amd.data import synthetic_clouds
    ori, nrm = synthetic_clouds(B, N, seed=2024)
    g = torch.Generator().manual_seed(1)
    adv = ori + 0.01 * torch.randn(B, 3, N, generator=g)
    return ori.cuda(), nrm.cuda(), adv.cuda().contiguous()


def test_knn_properties_full_size(clouds):
    from geoa3_amd import ops
    ori, nrm, adv = clouds
    d, i = ops.knn_planar(adv, adv, K)
    # ascending, self first at distance 0, indices valid and distinct per query
    assert (d[:, :, 1:] >= d[:, :, :-1]).all() and (d[:, :, 0] == 0).all()
    assert (i[:, :, 0] == torch.arange(N, device="cuda", dtype=torch.int32).view(1, N)).all()
    assert (i >= 0).all() and (i < N).all()
    srt = i.sort(dim=2)[0]
    assert (srt[:, :, 1:] != srt[:, :, :-1]).all()
    # the reported distances are the distances of the reported indices (bit-exact, same arithmetic)
    nb = torch.gather(adv.unsqueeze(2).expand(B, 3, N, N), 3, i.long().unsqueeze(1).expand(B, 3, N, K))
    diff = adv.unsqueeze(3) - nb
    dd = (diff[:, 0] * diff[:, 0] + diff[:, 1] * diff[:, 1]) + diff[:, 2] * diff[:, 2]
    assert torch.equal(dd, d)
    # idempotence: the result used as prior, and no prior at all, give the identical table
    d2, i2 = ops.knn_planar(adv, adv, K, i)
    d3, i3 = ops.knn_planar(adv, adv, K, torch.zeros_like(i))        # an invalid (non-distinct) prior
    assert torch.equal(i2, i) and torch.equal(d2, d) and torch.equal(i3, i)
    # K = 1 of the general kernel == the dedicated 1-NN kernel, in both directions
    d_ao, i_ao, d_oa, i_oa = ops.nn1_pair(adv, ori)
    k1d, k1i = ops.knn_planar(adv, ori, 1)
    assert torch.equal(k1i[:, :, 0], i_ao) and torch.equal(k1d[:, :, 0], d_ao)
    k1d, k1i = ops.knn_planar(ori, adv, 1)
    assert torch.equal(k1i[:, :, 0], i_oa) and torch.equal(k1d[:, :, 0], d_oa)
    # nothing is closer than the reported neighbour (checked on a random sample of pairs)
    g = torch.Generator(device="cuda").manual_seed(3)
    j = torch.randint(0, N, (B, N), device="cuda", generator=g)
    pj = torch.gather(ori, 2, j.unsqueeze(1).expand(B, 3, N))
    dj = ((adv - pj) ** 2).sum(1)
    assert (dj >= d_ao * (1 - 1e-6)).all()


def test_objective_linearity_and_consistency_full_size(clouds):
    """constrain = w_dis*CD + w_hd*HD + w_curv*curv and its gradient are linear in the weights."""
    from geoa3_amd import ops
    ori, nrm, adv = clouds
    d_ao, i_ao, d_oa, i_oa = ops.nn1_pair(adv, ori)
    _, knn_ori = ops.knn_planar(ori, ori, K)
    kap = ops.kappa(ori, nrm, knn_ori)
    _, knn_adv = ops.knn_planar(adv, adv, K, knn_ori)
    kw = dict(normal_ori=nrm, kappa_ori=kap, d_ao=d_ao, i_ao=i_ao, d_oa=d_oa, i_oa=i_oa, knn_adv=knn_adv, k=K - 1)
    full = ops.geo_loss_grad(adv, ori, dis_type=1, w_dis=1.0, w_hd=0.1, w_curv=1.0, **kw)
    cd = ops.geo_loss_grad(adv, ori, dis_type=1, w_dis=1.0, **kw)
    hd = ops.geo_loss_grad(adv, ori, dis_type=0, w_hd=1.0, **kw)
    cv = ops.geo_loss_grad(adv, ori, dis_type=0, w_curv=1.0, **kw)
    torch.testing.assert_close(full["constrain"], cd["dis_loss"] + 0.1 * hd["hd_loss"] + cv["curv_loss"],
                               rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(full["grad"], cd["grad"] + 0.1 * hd["grad"] + cv["grad"], rtol=2e-3, atol=1e-6)
    assert torch.equal(cd["dis_loss"], full["dis_loss"]) and torch.equal(hd["hd_loss"], full["hd_loss"])
    torch.testing.assert_close(cd["dis_loss"], d_ao.mean(1) + d_oa.mean(1), rtol=1e-5, atol=1e-8)
    assert torch.equal(hd["hd_loss"], d_ao.max(1)[0])
    # the Hausdorff gradient is supported on ONE point per instance
    assert ((hd["grad"].abs().sum(1) > 0).sum(1) <= 1).all()


def test_pointnet_batch_rows_independent_full_size(clouds):
    from geoa3_amd.data import synthetic_state_dict
    from geoa3_amd.pointnet import PointNet
    ori, _, adv = clouds
    net = PointNet(40)
    net.load_state_dict(synthetic_state_dict(40, seed=0))
    net = net.cuda().eval()
    x = adv.clone().requires_grad_()
    full = net(x)
    w = torch.randn(B, 40, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    (full * w).sum().backward()
    for lo, hi in [(0, 7), (100, 133), (249, 250)]:
        xs = adv[lo:hi].clone().requires_grad_()
        part = net(xs)
        assert torch.equal(part, full[lo:hi])                 # forward rows are bit-identical
        (part * w[lo:hi]).sum().backward()
        assert torch.equal(xs.grad, x.grad[lo:hi])            # the input gradient is deterministic and per-row


def test_pointnet_full_size_is_run_to_run_deterministic(clouds):
    """Three evaluations of the same 250-instance batch: logits and input gradient equal bit for bit (a race in a
    kernel shows here every time; the row-independence test above only sees it when the two launches differ)."""
    from geoa3_amd.data import synthetic_state_dict
    from geoa3_amd.pointnet import PointNet
    _, _, adv = clouds
    net = PointNet(40)
    net.load_state_dict(synthetic_state_dict(40, seed=0))
    net = net.cuda().eval()
    w = torch.randn(B, 40, device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    ref = None
    for _ in range(3):
        x = adv.clone().requires_grad_()
        out = net(x)
        (out * w).sum().backward()
        cur = (out.detach().clone(), x.grad.clone())
        if ref is None:
            ref = cur
        else:
            assert torch.equal(cur[0], ref[0])
            assert torch.equal(cur[1], ref[1])


def test_attack_loop_full_size_is_run_to_run_deterministic(clouds):
    """Five runs of 30 inner iterations of the full objective at B = 250 (deterministic gradient sums, both streams), a
    fresh runner each and nothing between the steps: iterate, Adam state and loss history equal bit for bit.  (This is the
    test that found the cross-XCD visibility bug of NOTEBOOK 5a: one run in 10-25 deviated in one of the last instances.)"""
    import bench
    from geoa3_amd.attack import AttackRunner
    from geoa3_amd.data import synthetic_state_dict
    from geoa3_amd.pointnet import PointNet
    ori, nrm, _ = clouds
    net = PointNet(40)
    net.load_state_dict(synthetic_state_dict(40, seed=0))
    net = net.cuda().eval()
    with torch.no_grad():
        gt = net(ori).argmax(1)
    steps = 30
    init = (torch.randn(B, 3, N, generator=torch.Generator().manual_seed(11)) * 1e-3).cuda()
    res = []
    for _ in range(5):
        cfg = bench.cfg_full_geoa3(steps + 4, N, K - 1)
        r = AttackRunner(net, B, N, cfg, torch.device("cuda"))
        r.setup(ori, nrm, gt, gt)
        r.begin_search_step(init)
        for s in range(steps):
            r.step(s, 0)
        torch.cuda.synchronize()
        res.append({k: r.t[k].clone() for k in ("x", "m", "v", "loss_hist")})
    for other in res[1:]:
        for k in res[0]:
            assert torch.equal(res[0][k], other[k]), k


def test_full_config2_attack_smoke_statistics():
    """2 binary steps x 60 iterations of the real config (B=250): every instance's loss history is finite, the
    success mask matches best_step, and recorded adversarial clouds really fool the network."""
    from argparse import Namespace

    from geoa3_amd.attack import attack
    from geoa3_amd.data import synthetic_clouds, synthetic_state_dict
    from geoa3_amd.pointnet import PointNet
    net = PointNet(40)
    net.load_state_dict(synthetic_state_dict(40, seed=0))
    net = net.cuda().eval()
    ori, nrm = synthetic_clouds(B, N, seed=7)
    with torch.no_grad():
        gt = net(ori.cuda()).argmax(1).cpu()
    cfg = Namespace(attack_label="Untarget", binary_max_steps=2, iter_max_steps=60, lr=0.01, initial_const=10.0,
                    optim="adam", cls_loss_type="CE", confidence=0.0, dis_loss_type="CD", dis_loss_weight=1.0,
                    is_cd_single_side=False, hd_loss_weight=0.1, curv_loss_weight=1.0, curv_loss_knn=16,
                    uniform_loss_weight=0.0, is_use_lr_scheduler=False, cc_linf=0.0, classes=40)
    data = [ori.permute(0, 2, 1).unsqueeze(1).contiguous(), nrm.permute(0, 2, 1).unsqueeze(1).contiguous(),
            gt.view(B, 1)]
    best, target, succ, best_step, all_loss = attack(net, data, cfg, 0, 1, verbose=False)
    loss = np.asarray(all_loss, dtype=np.float32)
    assert loss.shape == (60, B) and np.isfinite(loss).all()
    assert ((np.asarray(best_step) >= 0) == succ).all()
    assert succ.mean() > 0.5
    with torch.no_grad():
        pred = net(best.contiguous()).argmax(1).cpu().numpy()
    # a recorded best_attack was a successful iterate: re-evaluation disagrees with the ground truth
    assert (pred[succ] != gt.numpy()[succ]).all()
    # never-successful instances keep the all-ones placeholder (geoA3_attack.py:226)
    assert (best.cpu().numpy()[~succ] == 1.0).all()
