#!/usr/bin/env python3
"""Benchmark of the GeoA3 inner attack loop on MI355X (BASELINE.json metric: attack-iterations/sec).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE inner iteration of the attack (Attacker/geoA3_attack.py:238-352) applied to the whole batch of
victim instances that lives on a GPU: success check + PointNet forward + CE/CD/HD/curvature objective + input
gradient + Adam step.  Workload = BASELINE.json configs[1]: PointNet, N=1024 points, 250 instances per GPU, full
GeoA3 (CD 1.0 + HD 0.1 + curvature 1.0 with k=16), untargeted CE.  Instances are independent, so N GPUs hold N
independent 250-instance shards (weak scaling, no data-path collective; the global-batch loss divisor and the
one-int last-label broadcast per binary step are what the sharded attack() adds, geoa3_amd/distributed.py).

value = (instances advanced per second over all ranks) / 250 = iterations/sec of a 250-instance batch.
Inputs are synthetic (seeded ellipsoid clouds, calibrated random-init PointNet), resident in HBM before timing.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

INSTANCES = 250
NPOINT = 1024
KNN = 16
CLASSES = 40
PEAK_F32_MFMA = 157.3e12   # MI355X_MICROARCH.md: dense fp32 matrix peak (spec)
PEAK_F16_MFMA = 2.5e15    # dense f16 / bf16 matrix peak (spec, without 2:1 sparsity)
PEAK_HBM = 8.0e12


def cfg_config2(steps):
    """The reference flags of BASELINE.json configs[1] (main_attack.py:317-384 defaults + Untarget)."""
    return argparse.Namespace(attack_label="Untarget", binary_max_steps=1, iter_max_steps=steps, lr=0.01,
                              initial_const=10.0, optim="adam", cls_loss_type="CE", confidence=0.0,
                              dis_loss_type="CD", dis_loss_weight=1.0, is_cd_single_side=False, hd_loss_weight=0.1,
                              curv_loss_weight=1.0, curv_loss_knn=KNN, uniform_loss_weight=0.0,
                              is_use_lr_scheduler=False, cc_linf=0.0, is_pro_grad=False, is_real_offset=False,
                              npoint=NPOINT, classes=CLASSES)


def cpu_baseline(sample_b=8, budget_s=20.0):
    """The CPU port (oracle, reference semantics incl. the b batch-1 success-check forwards and the six dense
    K-NN queries per iteration) timed on a bounded sample of the same workload: one warm-up iteration sizes
    the timed run so that it stays within ~budget_s."""
    import torch
    from oracle import geoa3_oracle as O
    threads = min(os.cpu_count() or 1, 32)   # more intra-op threads than this only adds contention on this path
    torch.set_num_threads(threads)
    sd = O.make_pointnet_state_dict(CLASSES, seed=0)
    net = lambda x: O.pointnet_forward(sd, x)
    ori, nrm = O.make_synthetic_clouds(sample_b, NPOINT, seed=0)
    with torch.no_grad():
        gt = net(ori).argmax(1)
    g = torch.Generator().manual_seed(1)
    init = [torch.randn(sample_b, 3, NPOINT, generator=g) * 1e-3]
    t0 = time.perf_counter()
    O.attack(net, ori, nrm, gt, None, cfg_config2(1), init, faithful_success_check=True)  # warm-up, sizes the run
    warm = time.perf_counter() - t0
    iters = max(1, min(10, int(budget_s / max(warm, 1e-3))))
    t0 = time.perf_counter()
    O.attack(net, ori, nrm, gt, None, cfg_config2(iters), init, faithful_success_check=True)
    dt = time.perf_counter() - t0
    inst_it_per_s = sample_b * iters / dt
    return {"value": inst_it_per_s / INSTANCES, "unit": "attack-iterations/sec (250-instance batch)",
            "cores": threads, "kind": "port",
            "sample": "oracle attack(), %d instances x %d iterations of config 2 (N=1024, CE+CD+HD+curv k=16), "
                      "%.1f s, rate scaled to 250 instances" % (sample_b, iters, dt)}


def pmc_traffic(kernel_substr):
    """HBM bytes per launch of a kernel from the newest committed rocprofv3 PMC summary (profiles/*_pmc.csv:
    separate FETCH_SIZE / WRITE_SIZE passes of this same command; FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md).  None when no summary is present."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(REPO, "profiles", "*_pmc.csv")), key=os.path.getmtime)
    if not files:
        return None, None
    for r in csv.DictReader(open(files[-1])):
        if kernel_substr in r["Kernel"]:
            return (float(r["read_MB_corrected_x2"]) + float(r["write_MB"])) * 1e6, os.path.basename(files[-1])
    return None, None


def main():
    global NPOINT, KNN
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--single-mode", action="store_true",
                    help="skip the second, shorter measurement in the other arithmetic mode of the 1024-wide layers")
    ap.add_argument("--instances", type=int, default=INSTANCES, help="instances per GPU (default 250)")
    ap.add_argument("--npoint", type=int, default=NPOINT, help="points per cloud (configs[4]: 4096)")
    ap.add_argument("--knn", type=int, default=KNN, help="curv_loss_knn (configs[4]: 32)")
    ap.add_argument("--arch", default="PointNet", choices=["PointNet", "PointNetPP"],
                    help="victim (PointNetPP = configs[3]: SSG classifier on the HIP set-abstraction operators)")
    a = ap.parse_args()
    NPOINT, KNN = a.npoint, a.knn

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ndev = max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", (local_rank % ndev) if world > 1 else 0)
    torch.cuda.set_device(dev)
    backend = os.environ.get("GEOA3_BENCH_BACKEND", "nccl")   # "gloo": functional test of the N>1 path on one GPU
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import __graft_entry__
    if not os.path.exists(os.path.join(REPO, "geoa3_amd", "lib", "libgeoa3_hip.so")):
        __graft_entry__.build()
    from geoa3_amd import _lib
    from geoa3_amd.attack import AttackRunner
    from geoa3_amd.pointnet import PointNet
    from geoa3_amd.data import synthetic_clouds, synthetic_state_dict

    B = a.instances
    if a.arch == "PointNet":
        net = PointNet(CLASSES)
        net.load_state_dict(synthetic_state_dict(CLASSES, seed=0, device=dev))
    else:
        from geoa3_amd.pointnet2 import PointNet2ClassificationSSG
        torch.manual_seed(0)
        net = PointNet2ClassificationSSG(use_xyz=True, use_normal=False)
    net = net.to(dev).eval()
    ori, nrm = synthetic_clouds(B, NPOINT, seed=100 + rank)
    ori, nrm = ori.to(dev), nrm.to(dev)
    with torch.no_grad():
        gt = net(ori).argmax(1)
    lib = _lib.load()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(wide_mode, steps, warmup):
        """`warmup` untimed + exactly `steps` timed inner iterations with the 1024-wide layers in `wide_mode`;
        returns (seconds, max over ranks; per-kernel average ms of conv5 / nn1 / knn / T-Net wide)."""
        total = warmup + steps
        cfg = cfg_config2(total + 16)
        if a.arch == "PointNet":
            net.wide_mode = wide_mode
        runner = AttackRunner(net, B, NPOINT, cfg, dev, global_batch=B * world)
        runner.setup(ori, nrm, gt, gt)
        g = torch.Generator(device="cpu").manual_seed(7 + rank)
        init = (torch.randn(B, 3, NPOINT, generator=g) * 1e-3).to(dev)
        runner.begin_search_step(init)
        for s in range(warmup):
            runner.step(s, 0)
        barrier()
        # HIP events around the DOMINANT kernel only inside the timed region (an event pair costs ~6 us of stream
        # time); the other kernels' durations come from a few extra, untimed iterations afterwards
        lib.geoa3_profile_enable(steps)
        lib.geoa3_profile_select(1)
        t0 = time.perf_counter()
        for s in range(warmup, total):
            runner.step(s, 0)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())

        def kernel_ms(tag):
            buf = (C.c_float * steps)()
            n = lib.geoa3_profile_read(tag, buf, steps)
            return (sum(buf[:n]) / n) if n > 0 else None

        conv5_ms = kernel_ms(0)
        extra = min(steps, 10)
        lib.geoa3_profile_select(0xE)
        geo_stream, runner.geo_stream = runner.geo_stream, None   # one stream: durations of the kernels on their own
        for s in range(total, total + extra):
            runner.step(s, 0)
        torch.cuda.synchronize()
        runner.geo_stream = geo_stream
        nn1_ms, knn_ms, tnet_ms = (kernel_ms(t) for t in (1, 2, 3))
        lib.geoa3_profile_select(0xFFFFFFFF)
        lib.geoa3_profile_enable(0)
        return dt, conv5_ms, nn1_ms, knn_ms, tnet_ms, extra

    from geoa3_amd.pointnet import default_wide_mode
    mode = default_wide_mode() if a.arch == "PointNet" else None
    dt, conv5_ms, nn1_ms, knn_ms, tnet_ms, extra = measure(mode, a.steps, a.warmup)
    other = None
    if a.arch == "PointNet" and not a.single_mode:
        # the same loop with the 1024-wide layers in the other arithmetic mode, shorter, reported beside the headline
        omode = "f32" if mode == "f16x2" else "f16x2"
        osteps = max(10, min(a.steps, 40))
        odt, oconv5, _, _, otnet, _ = measure(omode, osteps, min(a.warmup, 5))
        other = (omode, osteps, odt, oconv5, otnet)

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = (B * world * a.steps / dt) / INSTANCES
        conv5_flops = 2.0 * B * NPOINT * 1024 * 384          # algorithmic: 1024 outputs x (3 taps x 128) MACs / point

        def roofline(m, ms):
            """conv5 + max: fp32 MFMA against the fp32 matrix peak; split mode against the dense f16 peak with the
            flops the scheme EXECUTES (3 f16 MFMA products per fp32 product)."""
            if not ms:
                return None
            if m == "f16x2":
                ach = 3.0 * conv5_flops / (ms * 1e-3)
                return {"bound": "mfma", "kernel": "wide_split_kernel<3> (conv5+bn5+relu+max; split-fp16 operands, "
                                                   "3 f16 MFMA products per fp32 product, fp32 accumulate)",
                        "achieved": round(ach / 1e12, 1), "peak": PEAK_F16_MFMA / 1e12, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_F16_MFMA, 4), "avg_launch_ms": round(ms, 4),
                        "algorithmic_flops_per_launch": conv5_flops, "executed_mfma_flops_per_launch": 3.0 * conv5_flops,
                        "fp32_equivalent_TFLOPs": round(conv5_flops / (ms * 1e-3) / 1e12, 1),
                        "note": "MI355X_MICROARCH.md: a tuned 8192^3 bf16 GEMM sustains 1247 TFLOP/s on random data "
                                "(the chip lowers its clock under 16-bit MFMA load)", "traffic": None}
            ach = conv5_flops / (ms * 1e-3)
            return {"bound": "mfma", "kernel": "wide_max2_kernel<3> (conv5+bn5+relu+max, fp32 MFMA)",
                    "achieved": round(ach / 1e12, 2), "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_F32_MFMA, 4), "avg_launch_ms": round(ms, 4),
                    "algorithmic_flops_per_launch": conv5_flops, "traffic": None}

        out = {
            "metric": "attack-iterations/sec (B=250, N=%d)" % NPOINT, "value": round(value, 3),
            "unit": "iterations/s of a 250-instance batch", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "arithmetic": "fp32 MFMA throughout" if mode != "f16x2" else
                          "fp32 values and fp32 accumulation everywhere; the convolutions (1024-wide layers, 64/128-wide "
                          "layers, Gram product) carry each fp32 operand as two fp16 values on the f16 MFMA (3 products "
                          "per fp32 product) -- error against float64 no larger than the fp32 MFMA kernels', "
                          "tools/wide_accuracy.py, DESIGN.md 4a; GEOA3_WIDE_MODE=f32 selects fp32 MFMA (other_wide_mode)",
            "data": "synthetic",
            "config": {"workload": "configs[%d]: PointNet %d-pt, %d instances per GPU, full GeoA3 (CE + CD 1.0 + HD 0.1 + "
                                   "curvature 1.0 k=%d), untargeted" % (1 if NPOINT == 1024 else 4, NPOINT, B, KNN),
                       "instances_per_gpu": B, "npoint": NPOINT, "knn": KNN, "classes": CLASSES,
                       "wide_mode": mode, "parallelism": "instance-sharded x%d" % world},
            "roofline": roofline(mode, conv5_ms),
        }
        if other is not None:
            omode, osteps, odt, oconv5, otnet = other
            out["other_wide_mode"] = {"wide_mode": omode, "value": round((B * world * osteps / odt) / INSTANCES, 3),
                                      "ms_per_step": round(odt / osteps * 1e3, 4), "steps": osteps,
                                      "roofline": roofline(omode, oconv5),
                                      "kernels_ms": {"conv5_wide_max": oconv5, "tnet_wide_max(x2)": otnet}}
        tr, src = pmc_traffic("wide_split_kernel<3" if mode == "f16x2" else "wide_max2_kernel<3")
        if tr is not None and NPOINT == 1024 and B == INSTANCES and out["roofline"]:
            out["roofline"]["traffic"] = tr
            out["roofline"]["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)" % src
        if a.arch != "PointNet":   # configs[3]: the MLPs are MIOpen/hipBLASLt kernels, no single hand-written dominant kernel
            out["config"]["workload"] = out["config"]["workload"].replace("PointNet ", "PointNet++ SSG ").replace(
                "configs[1]", "configs[3]")
            out["roofline"] = None
        if nn1_ms:
            cd_bytes = 40.0 * B * NPOINT
            out["cd_kernel"] = {"kernel": "grid_nn1_kernel (1-NN both directions, uniform-grid search; all-pairs "
                                          "nn1_pair_kernel beyond 4096 points)", "avg_launch_us": round(nn1_ms * 1e3, 2),
                                "hbm_GBps_algorithmic": round(cd_bytes / (nn1_ms * 1e-3) / 1e9, 2),
                                "hbm_frac": round(cd_bytes / (nn1_ms * 1e-3) / PEAK_HBM, 5),
                                "valu_frac": round(8.0 * B * NPOINT * NPOINT / (nn1_ms * 1e-3) / 157.3e12, 4)}
        out["kernels_ms"] = {"conv5_wide_max": conv5_ms, "tnet_wide_max(x2)": tnet_ms, "nn1_pair": nn1_ms, "knn": knn_ms,
                             "note": "conv5: HIP events inside the timed region; the others: %d untimed iterations "
                                     "right after it, on ONE stream (in the timed loop the geometry kernels run on a "
                                     "second stream beside the victim's forward)" % extra}
        if not a.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:  # the baseline never blocks the GPU number
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
