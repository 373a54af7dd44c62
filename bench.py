#!/usr/bin/env python3
"""Benchmark of the GeoA3 inner attack loop on MI355X (BASELINE.json metric: attack-iterations/sec at B=250).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE inner iteration of the attack (Attacker/geoA3_attack.py:238-352) applied to every victim instance
held by a GPU: success check + victim forward + CE/CD/HD/curvature objective + input gradient + Adam step.

Workloads (BASELINE.json `configs`):
  default                         configs[1]: PointNet, 1024 points, 250 instances, full GeoA3 (CD 1.0 + HD 0.1 + curvature
                                  1.0 with k=16), untargeted CE -- the configuration the metric is quoted on
  --gpus N (N > 1)                configs[2]: the SAME 250-instance batch sharded over N GPUs (`--scaling strong`, the
                                  default: rank r holds shard_bounds(250, N)[r] instances, loss divisor 1/250, no
                                  collective in the iteration); `--scaling weak` holds 250 instances on EVERY GPU
  --instances 32                  1-GPU proxy of configs[2]: one rank's shard of the 8-way split (global divisor 250)
  --arch PointNetPP               configs[3]: PointNet++ SSG victim
  --npoint 4096 --knn 32          configs[4]

value = (instances advanced per second over all ranks) / 250 = iterations/sec of the 250-instance batch.
Inputs are synthetic (seeded ellipsoid clouds, calibrated random-init victim), resident in HBM before timing; the timed
window starts from a steady-state iterate (`--presteps` untimed iterations after the initial offsets, default 150: the
data-dependent searches cost more once the offsets span several grid cells than in the first iterations).
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

BATCH = 250                # the batch the metric is quoted on
NPOINT = 1024
KNN = 16
CLASSES = 40
PEAK_F32_MFMA = 157.3e12   # MI355X_MICROARCH.md: dense fp32 matrix peak (spec)
PEAK_F16_MFMA = 2.5e15     # dense f16 / bf16 matrix peak (spec, without 2:1 sparsity)
PEAK_HBM = 8.0e12
PEAK_F32_VALU = 157.3e12

# event-timer tags (include/geoa3_hip_debug.h)
TAG_CONV5, TAG_NN1, TAG_KNN, TAG_TNET, TAG_SA1_BWD, TAG_SA1_FWD, TAG_GEO = 0, 1, 2, 3, 4, 5, 7


def cfg_full_geoa3(steps, npoint=NPOINT, knn=KNN):
    """The reference flags of BASELINE.json configs[1] (main_attack.py:317-384 defaults + Untarget)."""
    return argparse.Namespace(attack_label="Untarget", binary_max_steps=1, iter_max_steps=steps, lr=0.01,
                              initial_const=10.0, optim="adam", cls_loss_type="CE", confidence=0.0,
                              dis_loss_type="CD", dis_loss_weight=1.0, is_cd_single_side=False, hd_loss_weight=0.1,
                              curv_loss_weight=1.0, curv_loss_knn=knn, uniform_loss_weight=0.0,
                              is_use_lr_scheduler=False, cc_linf=0.0, is_pro_grad=False, is_real_offset=False,
                              npoint=npoint, classes=CLASSES)


def cpu_baseline(arch, npoint, knn, budget_s=60.0, threads=0):
    """The CPU port (oracle: reference semantics incl. the b batch-1 success-check forwards and the six dense K-NN
    queries per iteration, Attacker/geoA3_attack.py:238-352) timed on the host cores of this box on a BOUNDED sample of
    the same workload: the thread count is swept once (32 / 64 / 128, a 4-instance iteration each) and the
    best kept; that iteration sizes the sample (b instances x iterations) to ~budget_s of CPU work in THREE
    (T(b, 1), T(b, 1 + iters)) pairs; the figure is the median pair, the spread is reported beside it."""
    import statistics

    import torch
    from oracle import geoa3_oracle as O
    host_cores = os.cpu_count() or 1
    if arch == "PointNet":
        sd = O.make_pointnet_state_dict(CLASSES, seed=0)
        net = lambda x: O.pointnet_forward(sd, x)
    else:
        from oracle import pointnet2_oracle as P2
        sd2 = P2.make_pn2_state_dict(seed=0)
        net = lambda x: P2.pointnet2_ssg_forward(sd2, x)

    def run(b, iters):
        ori, nrm = O.make_synthetic_clouds(b, npoint, seed=0)
        with torch.no_grad():
            gt = net(ori).argmax(1)
        g = torch.Generator().manual_seed(1)
        init = [torch.randn(b, 3, npoint, generator=g) * 1e-3]
        t0 = time.perf_counter()
        O.attack(net, ori, nrm, gt, None, cfg_full_geoa3(iters, npoint, knn), init, faithful_success_check=True)
        return time.perf_counter() - t0

    # torch's intra-op pool stops scaling on this path well below the core count of the GPU box (measured there: the same
    # 4-instance iteration takes 73 s on 256 threads and ~4 s on 32): sweep 32 / 64 / 128 once unless --cpu-threads is given
    cands = [threads] if threads else sorted({min(host_cores, t) for t in (32, 64, 128)})
    sweep = {}
    for t in cands:
        torch.set_num_threads(t)
        run(2, 1)                               # first-touch costs (thread pool, allocator) stay out of the sizing run
        sweep[t] = run(4, 1) / 4.0
    threads = min(sweep, key=sweep.get)
    torch.set_num_threads(threads)
    per_inst_it = sweep[threads]
    # sample: b instances x (1 untimed + `iters` timed) iterations.  The untimed first iteration (step 0: nothing to rank,
    # cold caches at this size) is measured on its own by a run of exactly one iteration and subtracted:
    # timed = T(b, 1 + iters) - T(b, 1), so every timed iteration is a non-step-0 iterate.
    npairs, iters = 3, 2
    b = int(max(4, min(BATCH, budget_s / max(per_inst_it * npairs * (2 + iters), 1e-6))))
    if b == BATCH:
        iters = int(max(2, min(20, budget_s / max(per_inst_it * b * npairs, 1e-6) - 2)))
    pairs = [(run(b, 1), run(b, 1 + iters)) for _ in range(npairs)]
    # a difference far below the proportional share of the long run is noise (thread-pool warm-up, another load on the
    # box), not a fast CPU: such a pair counts with the share
    dts = []
    for t1, tn in pairs:
        share = tn * iters / (1.0 + iters)
        d = tn - t1
        dts.append(d if d >= 0.5 * share else share)
    dt = statistics.median(dts)
    vals = sorted(b * iters / d / BATCH for d in dts)
    return {"value": b * iters / dt / BATCH, "unit": "attack-iterations/sec (250-instance batch)",
            "cores": threads, "host_cores": host_cores, "kind": "port",
            "spread": {"min": vals[0], "max": vals[-1], "pairs": npairs, "rel": (vals[-1] - vals[0]) / max(vals[1], 1e-12)},
            "thread_sweep_s_per_instance_iteration": {str(t): round(v, 3) for t, v in sweep.items()},
            "sample": "oracle attack() (%s victim, N=%d, CE + CD + HD + curvature k=%d, reference success check = b "
                      "batch-1 forwards): b = %d instances x %d iterations (steps 1..%d, after one untimed step-0 iteration "
                      "at this size), median of %d pairs: %.1f s on %d threads (best of %s); instance-iterations/s / 250"
                      % (arch, npoint, knn, b, iters, iters, npairs, dt, threads, sorted(sweep))}


def pmc_traffic(kernel_substr, cfg_tag):
    """HBM bytes per launch of a kernel from the committed rocprofv3 PMC summary of this configuration
    (profiles/round*_<cfg_tag>_pmc.csv, the newest round/version by name: separate FETCH_SIZE / WRITE_SIZE passes of this
    same command, tools/gpu_profiles.sh; FETCH_SIZE doubled per the gfx950 note in MI355X_MICROARCH.md).  None when
    absent."""
    import csv
    import glob
    import re

    def version(f):
        m = re.search(r"round(\d+)_v(\d+)", os.path.basename(f))
        return (int(m.group(1)), int(m.group(2))) if m else (0, 0)

    files = sorted(glob.glob(os.path.join(REPO, "profiles", "round*_%s_pmc.csv" % cfg_tag)), key=version)
    for f in reversed(files):
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Kernel"]:
                return (float(r["read_MB_corrected_x2"]) + float(r["write_MB"])) * 1e6, os.path.basename(f)
    return None, None


def spawn_ranks(n):
    """Run this script as n ranks of one node under torch.distributed.run (rendezvous on 127.0.0.1, a free port) and
    return the launcher's exit code.  The parent stays CPU-only: nothing here imports torch."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--presteps", type=int, default=150,
                    help="untimed iterations before the warm-up so the timed window starts from a steady-state iterate")
    ap.add_argument("--scaling", default="strong", choices=["strong", "weak"],
                    help="N > 1: strong = the 250-instance batch sharded over the GPUs (configs[2]); weak = 250 per GPU")
    ap.add_argument("--global-batch", type=int, default=BATCH, help="instances of the whole job (strong scaling)")
    ap.add_argument("--instances", type=int, default=0,
                    help="instances held by EACH GPU (overrides the split): 32 = one rank's shard of the 8-GPU run")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-proxy-full", action="store_true",
                    help="shard proxy (--instances): skip the full 250-instance comparison run (profiling)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (default: the best of 32 / 64 / 128, swept once; see cpu_baseline)")
    ap.add_argument("--cpu-budget", type=float, default=60.0, help="seconds of CPU work of the baseline sample (three pairs)")
    ap.add_argument("--single-mode", action="store_true",
                    help="skip the second, shorter measurement in the strict fp32-MFMA mode of the convolutions")
    ap.add_argument("--npoint", type=int, default=NPOINT, help="points per cloud (configs[4]: 4096)")
    ap.add_argument("--knn", type=int, default=KNN, help="curv_loss_knn (configs[4]: 32)")
    ap.add_argument("--data", default="ellipsoid", choices=["ellipsoid", "cad"],
                    help="synthetic input: one uniform ellipsoid per instance (the headline), or CAD-like clouds -- boxes, "
                         "tables on thin legs, two clusters of very different density, rods, 5 %% exact duplicates "
                         "(geoa3_amd.data.synthetic_cad_clouds)")
    ap.add_argument("--cad-kinds", default="", help="with --data cad: comma-separated subset of box,table,clusters,rod,ellipsoid2")
    ap.add_argument("--cad-duplicates", type=float, default=0.05, help="with --data cad: fraction of exact duplicate points")
    ap.add_argument("--arch", default="PointNet", choices=["PointNet", "PointNetPP"],
                    help="victim (PointNetPP = configs[3]: SSG classifier)")
    a = ap.parse_args()
    npoint, knn = a.npoint, a.knn

    # a real launcher exports all of RANK, WORLD_SIZE and MASTER_PORT; a stray WORLD_SIZE=1 from a scheduler does not count
    launched = all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_PORT"))
    if a.gpus > 1 and not launched:
        # `python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process group (torch.distributed.run)
        # before this process has imported torch or touched the GPU, and leave with its return code (a process that has
        # initialised the GPU must never exec another program on this pool)
        sys.exit(spawn_ranks(a.gpus))

    import torch
    import torch.distributed as dist
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
    if not launched:
        rank = local_rank = 0
    if a.gpus > 1 and world != a.gpus:
        sys.exit("bench.py: --gpus %d but the launcher started %d ranks" % (a.gpus, world))
    ndev = max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", (local_rank % ndev) if world > 1 else 0)
    torch.cuda.set_device(dev)
    backend = os.environ.get("GEOA3_BENCH_BACKEND", "nccl")   # "gloo": functional test of the N>1 path on one GPU
    # under a launcher (WORLD_SIZE set) the process group is created even for ONE rank: a 1-GPU box then still runs the
    # RCCL initialisation, the barriers and the device-tensor all-reduce of the N > 1 path
    use_dist = launched
    if use_dist:
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
    from geoa3_amd.data import SYNTHETIC_GENERATORS, synthetic_state_dict
    synthetic_clouds = SYNTHETIC_GENERATORS[a.data]
    if a.data == "cad" and (a.cad_kinds or a.cad_duplicates != 0.05):
        import functools
        synthetic_clouds = functools.partial(synthetic_clouds, duplicates=a.cad_duplicates,
                                             **({"kinds": tuple(a.cad_kinds.split(","))} if a.cad_kinds else {}))
    from geoa3_amd.distributed import shard_bounds
    from geoa3_amd.pointnet import PointNet

    # which instances this rank holds, and the divisor of the loss mean (geoA3_attack.py:178 -> 1 / GLOBAL batch)
    if a.instances > 0:            # explicit per-GPU shard (1-GPU proxy of a sharded run, or a custom weak run)
        B, lo = a.instances, 0
        global_batch = max(a.global_batch, B * world) if a.scaling == "strong" else B * world
        mode = "shard-proxy" if world == 1 and B != BATCH else a.scaling
    elif a.scaling == "strong":
        global_batch = a.global_batch
        lo, hi = shard_bounds(global_batch, world)[rank]
        B, mode = hi - lo, "strong"
    else:
        B, lo, global_batch, mode = BATCH, 0, BATCH * world, "weak"
    total_instances = B * world if (a.instances > 0 or mode == "weak") else global_batch

    if a.arch == "PointNet":
        net = PointNet(CLASSES)
        net.load_state_dict(synthetic_state_dict(CLASSES, seed=0, device=dev))
    else:
        from geoa3_amd.pointnet2 import PointNet2ClassificationSSG
        torch.manual_seed(0)
        net = PointNet2ClassificationSSG(use_xyz=True, use_normal=False)
    net = net.to(dev).eval()
    if mode == "strong":           # every rank cuts ITS rows out of the same seeded global batch
        ori, nrm = synthetic_clouds(global_batch, npoint, seed=100)
        ori, nrm = ori[lo:lo + B].contiguous(), nrm[lo:lo + B].contiguous()
    else:
        ori, nrm = synthetic_clouds(B, npoint, seed=100 + rank)
    ori, nrm = ori.to(dev), nrm.to(dev)
    with torch.no_grad():
        gt = net(ori).argmax(1) if B > 0 else torch.zeros(0, dtype=torch.long, device=dev)
    lib = _lib.load()

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    dominant_tag = TAG_CONV5 if a.arch == "PointNet" else TAG_SA1_BWD

    submit = [0.0]

    def measure(wide_mode, steps, warmup, presteps, b=B, pts=None):
        """`presteps` + `warmup` untimed, then exactly `steps` timed inner iterations; -> (seconds, max over ranks;
        per-kernel average ms by event-timer tag)."""
        o, nm, g_ = (ori, nrm, gt) if pts is None else pts
        total = presteps + warmup + steps
        cfg = cfg_full_geoa3(total + 16, npoint, knn)
        if a.arch == "PointNet":
            net.wide_mode = wide_mode
        runner = AttackRunner(net, b, npoint, cfg, dev, global_batch=global_batch if pts is None else BATCH)
        runner.setup(o, nm, g_, g_)
        g = torch.Generator(device="cpu").manual_seed(7 + rank)
        init = (torch.randn(b, 3, npoint, generator=g) * 1e-3).to(dev)
        runner.begin_search_step(init)
        for s in range(presteps + warmup):
            runner.step(s, 0)
        barrier()
        # HIP events around the DOMINANT kernel only inside the timed region (an event pair costs ~6 us of stream
        # time); the other kernels' durations come from a few extra, untimed iterations afterwards
        lib.geoa3_profile_enable(steps)
        lib.geoa3_profile_select(1 << dominant_tag)
        t0 = time.perf_counter()
        for s in range(presteps + warmup, total):
            runner.step(s, 0)
        submit[0] = time.perf_counter() - t0      # host time to ENQUEUE the steps (== dt: the host is the bound)
        barrier()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())

        def kernel_ms(tag):
            buf = (C.c_float * steps)()
            n = lib.geoa3_profile_read(tag, buf, steps)
            return (sum(buf[:n]) / n) if n > 0 else None

        ms = {dominant_tag: kernel_ms(dominant_tag)}
        extra = min(steps, 10)
        lib.geoa3_profile_select(0xFF & ~(1 << dominant_tag))
        geo_stream, runner.geo_stream = runner.geo_stream, None   # one stream: durations of the kernels on their own
        for s in range(total, total + extra):
            runner.step(s, 0)
        torch.cuda.synchronize()
        runner.geo_stream = geo_stream
        for tag in range(8):
            if tag != dominant_tag:
                ms[tag] = kernel_ms(tag)
        lib.geoa3_profile_select(0xFFFFFFFF)
        lib.geoa3_profile_enable(0)
        return dt, ms, extra

    from geoa3_amd.pointnet import default_wide_mode
    wmode = default_wide_mode() if a.arch == "PointNet" else None
    dt, kms, extra = measure(wmode, a.steps, a.warmup, a.presteps)
    host_submit_ms = submit[0] / a.steps * 1e3
    other = None
    if a.arch == "PointNet" and not a.single_mode and mode != "shard-proxy":
        # the same loop with every convolution on the fp32 MFMA (strict fp32 products), shorter, beside the headline
        omode = "f32" if wmode == "f16x2" else "f16x2"
        osteps = max(10, min(a.steps, 40))
        odt, okms, _ = measure(omode, osteps, min(a.warmup, 5), min(a.presteps, 60))
        other = (omode, osteps, odt, okms)
    proxy = None
    if mode == "shard-proxy" and a.arch == "PointNet" and not a.no_proxy_full:
        # the full 250-instance batch on this GPU in the same process: what linear scaling is measured against
        fo, fn = synthetic_clouds(BATCH, npoint, seed=100)
        fo, fn = fo.to(dev), fn.to(dev)
        with torch.no_grad():
            fg = net(fo).argmax(1)
        fsteps = max(10, min(a.steps, 60))
        fdt, _, _ = measure(wmode, fsteps, 5, a.presteps, b=BATCH, pts=(fo, fn, fg))
        proxy = (fsteps, fdt)

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = (total_instances * a.steps / dt) / BATCH
        conv5_flops = 2.0 * B * npoint * 1024 * 384          # algorithmic: 1024 outputs x (3 taps x 128) MACs / point

        def conv5_roofline(m, ms):
            """conv5 + bn5 + relu + max.  `achieved` = ALGORITHMIC flops (2 * B * N * 1024 * 384 per launch)
            / the kernel's measured duration; `peak` = the dense peak of the matrix pipe the kernel runs on.  The
            split mode executes 3 f16 products per fp32 product: `executed_frac` prices those against the same
            peak (how busy the pipe is), `frac` prices the useful work."""
            if not ms:
                return None
            ach = conv5_flops / (ms * 1e-3)
            if m == "f16x2":
                from geoa3_amd.pointnet import wide_shape
                kname = ("wide16_kernel<3> (v_mfma_f32_16x16x32_f16)" if wide_shape("conv5") == 16
                         else "wide_split_kernel<3> (v_mfma_f32_32x32x16_f16)")
                return {"bound": "mfma", "kernel": kname + ": conv5+bn5+relu+max; fp32 operands carried as two fp16 "
                                                           "values on the f16 matrix pipe, fp32 accumulate",
                        "achieved": round(ach / 1e12, 1), "peak": PEAK_F16_MFMA / 1e12, "unit": "TFLOP/s",
                        "frac": round(ach / PEAK_F16_MFMA, 4), "avg_launch_ms": round(ms, 4),
                        "algorithmic_flops_per_launch": conv5_flops,
                        "executed_mfma_flops_per_launch": 3.0 * conv5_flops,
                        "executed_frac": round(3.0 * ach / PEAK_F16_MFMA, 4),
                        "frac_of_fp32_matrix_peak": round(ach / PEAK_F32_MFMA, 3),
                        "note": "MI355X_MICROARCH.md: a tuned 8192^3 bf16 GEMM sustains 1247 TFLOP/s on random data "
                                "(the chip lowers its clock under 16-bit MFMA load), i.e. executed_frac ~0.5 is the "
                                "practical ceiling of this pipe", "traffic": None}
            return {"bound": "mfma", "kernel": "wide_max2_kernel<3> (conv5+bn5+relu+max, fp32 MFMA)",
                    "achieved": round(ach / 1e12, 2), "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
                    "frac": round(ach / PEAK_F32_MFMA, 4), "avg_launch_ms": round(ms, 4),
                    "algorithmic_flops_per_launch": conv5_flops, "traffic": None}

        cfg_idx = 3 if a.arch != "PointNet" else (4 if npoint >= 4096 else (2 if (world > 1 or mode == "shard-proxy") else 1))
        victim = "PointNet" if a.arch == "PointNet" else "PointNet++ SSG"
        out = {
            "metric": "attack-iterations/sec (B=250, N=%d)" % npoint, "value": round(value, 3),
            "unit": "iterations/s of a 250-instance batch", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "host_enqueue_ms_per_step": round(host_submit_ms, 4),
            "higher_is_better": True,
            "ranks_seen": dist.get_world_size() if use_dist else 1,
            "backend": ("rccl (torch.distributed 'nccl')" if backend == "nccl" else backend) if use_dist else None,
            "scaling": "weak" if mode == "weak" else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "arithmetic": "fp32 MFMA throughout" if wmode != "f16x2" else
                          "fp32 values and fp32 accumulation everywhere; the convolutions (1024-wide layers, 64/128-wide "
                          "layers, Gram product) carry each fp32 operand as two fp16 values on the f16 MFMA (3 products "
                          "per fp32 product) -- error against float64 no larger than the fp32 MFMA kernels', "
                          "tools/wide_accuracy.py, DESIGN.md 4a; GEOA3_WIDE_MODE=f32 selects fp32 MFMA (other_wide_mode)",
            "data": "synthetic" if a.data == "ellipsoid" else "synthetic (%s)" % a.data,
            "config": {"workload": "configs[%d]: %s %d-pt, %s, full GeoA3 (CE + CD 1.0 + HD 0.1 + curvature 1.0 k=%d), "
                                   "untargeted" % (cfg_idx, victim, npoint,
                                                   {"strong": "%d instances sharded over %d GPU(s) (%d on rank 0)"
                                                              % (global_batch, world, B),
                                                    "weak": "%d instances on each of %d GPU(s)" % (B, world),
                                                    "shard-proxy": "ONE rank's %d-instance shard of the %d-instance batch "
                                                                   "on 1 GPU (loss divisor 1/%d)" % (B, global_batch,
                                                                                                      global_batch)}[mode],
                                                   knn),
                       "instances_per_gpu": B, "global_batch": global_batch, "npoint": npoint, "knn": knn,
                       "classes": CLASSES, "wide_mode": wmode, "presteps": a.presteps,
                       "parallelism": "instance-sharded x%d" % world},
        }
        if a.arch == "PointNet":
            out["roofline"] = conv5_roofline(wmode, kms.get(TAG_CONV5))
            from geoa3_amd.pointnet import wide_shape
            tr, src = pmc_traffic(("wide16_kernel<3" if wide_shape("conv5") == 16 else "wide_split_kernel<3")
                                  if wmode == "f16x2" else "wide_max2_kernel<3", "c2" if npoint < 4096 else "c5")
            if tr is not None and B == BATCH and npoint in (1024, 4096) and out["roofline"]:
                out["roofline"]["traffic"] = tr
                out["roofline"]["traffic_source"] = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)" % src
        if a.arch == "PointNet" and npoint >= 4096:
            # configs[4]: the self top-(k+1) search was the largest kernel of the iteration until the cell-grid walk took
            # its candidates 64 at a time (now conv5 is, as at 1024 points); it is reported beside the roofline.  Its
            # algorithmic HBM bytes: read B*N*12, write B*N*(k+1)*8 (distances + indices); it is VALU/LDS
            # bound (8*B*N^2 flops all-pairs), so the HBM fraction is small by construction: valu_frac is given beside it
            ms = kms.get(TAG_KNN)
            bytes_ = B * npoint * (12.0 + (knn + 1) * 8.0)
            out["knn_kernel"] = None if not ms else {
                "bound": "hbm", "kernel": "knn_grid_kernel (self top-%d on a 16^3 cell grid, one wavefront per query)" % (knn + 1),
                "achieved": round(bytes_ / (ms * 1e-3) / 1e9, 2), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                "frac": round(bytes_ / (ms * 1e-3) / PEAK_HBM, 5), "avg_launch_ms": round(ms, 4),
                "algorithmic_bytes_per_launch": bytes_,
                "valu_frac_vs_all_pairs_flops": round(8.0 * B * npoint * npoint / (ms * 1e-3) / PEAK_F32_VALU, 4),
                "traffic": None}
            tr, src = pmc_traffic("knn_grid_kernel", "c5")
            if tr is not None and out["knn_kernel"] and B == BATCH:
                out["knn_kernel"]["traffic"], out["knn_kernel"]["traffic_source"] = tr, "profiles/" + src
        if a.arch != "PointNet":
            # configs[3]: level 1's backward (recompute 3->64->64, then the input gradient through 128->64->64->3) is the
            # largest kernel.  Algorithmic flops (input gradient only, no recompute): 2*(128*64 + 64*64 + 64*3) per
            # grouped sample, B*512*64 samples; executed on the f16 pipe with split operands (3 products per product)
            ms = kms.get(TAG_SA1_BWD)
            flops = 2.0 * (128 * 64 + 64 * 64 + 64 * 3) * B * 512 * 64
            out["roofline"] = None if not ms else {
                "bound": "mfma", "kernel": "sa1_bwd_kernel (PointNet++ level 1 input gradient, split-fp16 operands)",
                "achieved": round(flops / (ms * 1e-3) / 1e12, 2), "peak": PEAK_F16_MFMA / 1e12, "unit": "TFLOP/s",
                "frac": round(flops / (ms * 1e-3) / PEAK_F16_MFMA, 4), "avg_launch_ms": round(ms, 4),
                "algorithmic_flops_per_launch": flops, "traffic": None}
            tr, src = pmc_traffic("sa1_bwd_kernel", "c4")
            if tr is not None and out["roofline"] and B == BATCH:
                out["roofline"]["traffic"], out["roofline"]["traffic_source"] = tr, "profiles/" + src
        if other is not None:
            omode, osteps, odt, okms = other
            oroof = conv5_roofline(omode, okms.get(TAG_CONV5))
            otr, osrc = pmc_traffic("wide_max2_kernel<3" if omode == "f32" else "wide16_kernel<3", "c2f32" if omode == "f32" else "c2")
            if oroof and otr is not None and B == BATCH and npoint == NPOINT:
                oroof["traffic"], oroof["traffic_source"] = otr, "profiles/%s (rocprofv3 --pmc passes of this mode)" % osrc
            out["other_wide_mode"] = {"wide_mode": omode, "value": round((total_instances * osteps / odt) / BATCH, 3),
                                      "ms_per_step": round(odt / osteps * 1e3, 4), "steps": osteps,
                                      "roofline": oroof,
                                      "kernels_ms": {"conv5_wide_max": okms.get(TAG_CONV5),
                                                     "tnet_wide_max(x2)": okms.get(TAG_TNET)}}
        if proxy is not None:
            fsteps, fdt = proxy
            full_ms = fdt / fsteps * 1e3
            ranks = float(global_batch) / B
            out["strong_scaling_proxy"] = {
                "full_batch_ms_per_step": round(full_ms, 4), "shard_ms_per_step": round(ms_per_step, 4),
                "ranks_emulated": round(ranks, 2), "linear_shard_ms": round(full_ms / ranks, 4),
                "fraction_of_linear": round(full_ms / ranks / ms_per_step, 4),
                "note": "1-GPU proxy: the %d-instance shard a rank of the %.1f-way split holds vs the whole %d-instance "
                        "batch on the same GPU, same process (no collective runs in the iteration)" % (B, ranks,
                                                                                                        global_batch)}
        nn1_ms = kms.get(TAG_NN1)
        if nn1_ms:
            cd_bytes = 40.0 * B * npoint
            out["cd_kernel"] = {"kernel": "grid_nn1_kernel (1-NN both directions, uniform-grid search; all-pairs "
                                          "nn1_pair_kernel beyond 4096 points)", "avg_launch_us": round(nn1_ms * 1e3, 2),
                                "hbm_GBps_algorithmic": round(cd_bytes / (nn1_ms * 1e-3) / 1e9, 2),
                                "hbm_frac": round(cd_bytes / (nn1_ms * 1e-3) / PEAK_HBM, 5),
                                "valu_frac": round(8.0 * B * npoint * npoint / (nn1_ms * 1e-3) / PEAK_F32_VALU, 4)}
        out["kernels_ms"] = {"conv5_wide_max": kms.get(TAG_CONV5), "tnet_wide_max(x2)": kms.get(TAG_TNET),
                             "nn1_pair": nn1_ms, "knn": kms.get(TAG_KNN), "geo_loss_grad": kms.get(TAG_GEO), "sa1_bwd": kms.get(TAG_SA1_BWD),
                             "sa1_fwd": kms.get(TAG_SA1_FWD),
                             "note": "the roofline kernel: HIP events inside the timed region; the others: %d untimed "
                                     "iterations right after it, on ONE stream (in the timed loop the geometry kernels "
                                     "run on a second stream beside the victim's forward)" % extra}
        if not a.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(a.arch, npoint, knn, budget_s=a.cpu_budget, threads=a.cpu_threads)
            except Exception as e:  # the baseline never blocks the GPU number
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
