"""The matrix-core filter of the 1-NN search (csrc/geom_filter.hip) against the all-pairs kernel, bit for bit, and its
time per launch beside the grid walk and the in-kernel sweep.  `python tools/nn1_filter_check.py [--time]` (GPU)."""
import argparse
import sys
import time

import torch

sys.path.insert(0, ".")
from geoa3_amd import ops                                   # noqa: E402
from geoa3_amd.data import synthetic_cad_clouds, synthetic_clouds, CAD_KINDS   # noqa: E402


def dev(x):
    return x.contiguous().cuda()


def cases():
    g = torch.Generator().manual_seed(5)
    for N, scale in ((1024, 0.003), (1024, 0.05), (1024, 0.25), (4096, 0.01), (4096, 0.2), (300, 0.02), (1000, 0.1), (33, 0.1),
                     (2049, 0.05)):
        ori, _ = synthetic_cad_clouds(10, N, seed=N)
        yield "cad N=%d s=%g" % (N, scale), ori + scale * torch.randn(10, 3, N, generator=g), ori
    ori, _ = synthetic_clouds(4, 4096, seed=1)
    yield "ellipsoid 1024 vs 4096", ori[:, :, :1024] + 0.05 * torch.randn(4, 3, 1024, generator=g), ori
    yield "ellipsoid 4096 vs 1024", ori + 0.05 * torch.randn(4, 3, 4096, generator=g), ori[:, :, :1024].contiguous()
    big, _ = synthetic_clouds(2, 6000, seed=3)
    yield "6000 vs 5000 (beyond the grid's 4096)", big + 0.05 * torch.randn(2, 3, 6000, generator=g), big[:, :, :5000].contiguous()
    for nq, m in ((2000, 3000), (1025, 3000), (4000, 4000), (3000, 2000)):   # fewer than 4096: lanes without a 4th query
        yield "ellipsoid %d vs %d" % (nq, m), ori[:1, :, :nq] + 0.05 * torch.randn(1, 3, nq, generator=g), ori[:1, :, :m].contiguous()
    yield "ellipsoid 2500 vs 2049", ori[:, :, :2500] + 0.05 * torch.randn(4, 3, 2500, generator=g), ori[:, :, :2049].contiguous()
    a = ori[:, :, :1024] + 0.03 * torch.randn(4, 3, 1024, generator=g)
    a[:, :, :512] += torch.tensor([3.0, -2.0, 0.5]).view(1, 3, 1)
    a[:, :, 5] = 1e3
    yield "far queries", a, ori[:, :, :1024].contiguous()
    yield "shifted by 100", ori[:, :, :1024] + 100.0 + 0.01 * torch.randn(4, 3, 1024, generator=g), ori[:, :, :1024] + 100.0
    yield "tiny 1e-6", 1e-6 * (ori[:, :, :1024] + 0.01 * torch.randn(4, 3, 1024, generator=g)), 1e-6 * ori[:, :, :1024]
    yield "huge 1e6", 1e6 * (ori[:, :, :1024] + 0.01 * torch.randn(4, 3, 1024, generator=g)), 1e6 * ori[:, :, :1024]
    r = ori[:, :, :1024].clone()
    r[:] = r[:, :, :1]
    a = r.clone()
    a[:, :, ::2] += 0.01
    yield "all points equal", a, r
    r = ori[:, :, :1000].clone()
    r[:, 1:, :] = 0.25
    a = r + 0.02 * torch.randn(4, 3, 1000, generator=g)
    a[:, 2, :] = 0.25
    yield "collinear", a, r
    r = ori[:, :, :1024].clone()
    a = r.clone()                                # identical clouds with duplicates: ties everywhere
    a[:, :, 1] = a[:, :, 0]
    r[:, :, 1] = r[:, :, 0]
    yield "identical + duplicates", a, r
    lat = torch.stack(torch.meshgrid(torch.arange(16.), torch.arange(8.), torch.arange(8.), indexing="ij")).reshape(1, 3, -1) / 16
    yield "lattice (exact ties)", (lat + torch.tensor([1 / 32, 0, 0]).view(1, 3, 1)).repeat(2, 1, 1), lat.repeat(2, 1, 1)
    bad = ori[:, :, :1024].clone()
    bad[0, 0, 3] = float("nan")
    bad[1, 2, 9] = float("inf")
    yield "nan / inf in the searched cloud", ori[:, :, :1024] + 0.01, bad
    yield "nan / inf in the queries", bad, ori[:, :, :1024].contiguous()
    yield "1e30 coordinates", 1e30 * (ori[:, :, :1024] + 0.01), 1e30 * ori[:, :, :1024]


def check(verbose=True):
    """-> number of runs that differ from the all-pairs kernel in any bit (distance or index, either direction)."""
    bad = 0
    for name, a, r in cases():
        A, R = dev(a), dev(r)
        want = ops.nn1_pair(A, R)
        small = max(a.shape[2], r.shape[2]) <= 4096        # (the walk and the in-kernel sweep need a grid: <= 4096 points)
        for label, policy, prior in (("filter", (0.0, 1), None), ("filter+prior", (0.0, 1), "exact"),
                                     ("filter+junk prior", (0.0, 1), "junk"), ("sweep", (0.0, 0), None),
                                     ("walk", (1e9, 0), "exact"), ("shipped", None, "exact"), ("shipped, no prior", None, None)):
            if not small and policy is not None and policy[1] == 0:
                continue
            pr = None
            if prior == "exact":
                pr = (want[1].clone(), want[3].clone())
            elif prior == "junk":
                pr = (torch.full_like(want[1], 10 ** 6), torch.full_like(want[3], -5))
            got = ops.nn1_pair(A, R, method="grid", prior=pr, policy=policy)
            ok = all(torch.equal(w.view(torch.int32), x.view(torch.int32)) for w, x in zip(want, got))
            if not ok:
                bad += 1
                nd = [(w.view(torch.int32) != x.view(torch.int32)).sum().item() for w, x in zip(want, got)]
                print("MISMATCH", name, label, nd)
                for w, x in zip(want, got):
                    ne = (w.view(torch.int32) != x.view(torch.int32)).nonzero()
                    if len(ne):
                        b_, q_ = ne[0].tolist()
                        print("   first difference at", (b_, q_), "all pairs", w[b_, q_].item(), "this policy", x[b_, q_].item())
            if policy is not None and pr is not None and prior == "exact":      # in place: the prior IS the output
                d_ar, i_ar, d_ra, i_ra = want[0].clone(), want[1].clone(), want[2].clone(), want[3].clone()
                from geoa3_amd import _lib
                rc = _lib.load().geoa3_debug_grid_nn1_pair(A.data_ptr(), R.data_ptr(), A.shape[0], A.shape[2], R.shape[2],
                                                           i_ar.data_ptr(), i_ra.data_ptr(), d_ar.data_ptr(), i_ar.data_ptr(),
                                                           d_ra.data_ptr(), i_ra.data_ptr(), 0.0, 1,
                                                           torch.cuda.current_stream().cuda_stream)
                assert rc == 0
                if not (torch.equal(i_ar, want[1]) and torch.equal(i_ra, want[3]) and torch.equal(d_ar, want[0])):
                    bad += 1
                    print("MISMATCH in place", name)
        if verbose:
            print("ok " if not bad else "   ", name)
    if verbose:
        print("mismatching runs:", bad)
    return bad


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def bench(B=250):
    g = torch.Generator().manual_seed(7)
    for N in (1024, 4096):
        for kind in ("ellipsoid", "table", "clusters", "rod", "mixed"):
            if kind == "ellipsoid":
                ori, _ = synthetic_clouds(B, N, seed=2)
            elif kind == "mixed":
                ori, _ = synthetic_cad_clouds(B, N, seed=2)
            else:
                ori, _ = synthetic_cad_clouds(B, N, seed=2, kinds=(kind,))
            for scale in (0.02, 0.2):
                A, R = dev(ori + scale * torch.randn(B, 3, N, generator=g)), dev(ori)
                want = ops.nn1_pair(A, R)
                pr = (want[1], want[3])
                row = []
                for label, policy in (("grid", (1e9, 0)), ("sweep", (0.0, 0)), ("filter", (0.0, 1)), ("r5 policy", (0.12, 0)),
                                      ("f.12", (0.12, 1)), ("f.06", (0.06, 1)), ("f.03", (0.03, 1))):
                    row.append("%s %7.1f" % (label, timeit(lambda: ops.nn1_pair(A, R, method="grid", prior=pr, policy=policy))))
                row.append("filter, no prior %7.1f" % timeit(lambda: ops.nn1_pair(A, R, method="grid", policy=(0.0, 1))))
                junk = (torch.randint(0, N, (B, N), generator=g).int().cuda(), torch.randint(0, N, (B, N), generator=g).int().cuda())
                row.append("filter, random prior %7.1f" % timeit(lambda: ops.nn1_pair(A, R, method="grid", prior=junk, policy=(0.0, 1))))
                print("N=%d %-10s offsets %.2f  us/launch: %s" % (N, kind, scale, "  ".join(row)), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--time", action="store_true")
    ap.add_argument("--no-check", action="store_true")
    ap.add_argument("--dissect", action="store_true", help="filter-all on one ellipsoid batch (variant builds with -DGEOA3_NF_STOP=n)")
    args = ap.parse_args()
    if args.dissect:
        g = torch.Generator().manual_seed(7)
        for N in (1024, 4096):
            ori, _ = synthetic_clouds(250, N, seed=2)
            A, R = dev(ori + 0.2 * torch.randn(250, 3, N, generator=g)), dev(ori)
            want = ops.nn1_pair(A, R)
            pr = (want[1], want[3])
            print("N=%d filter-all %.1f us  (walk %.1f)" % (N, timeit(lambda: ops.nn1_pair(A, R, method="grid", prior=pr, policy=(0.0, 1))),
                                                       timeit(lambda: ops.nn1_pair(A, R, method="grid", prior=pr, policy=(1e9, 0)))))
        sys.exit(0)
    rc = 0 if args.no_check else check()
    if args.time:
        bench()
    sys.exit(1 if rc else 0)
