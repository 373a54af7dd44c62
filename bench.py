#!/usr/bin/env python3
"""Benchmark of the GeoA3 inner attack loop on MI355X (BASELINE.json metric: attack-iterations/sec at B=250).

    python bench.py --gpus N --steps K --warmup W
    (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is ONE inner iteration of the attack (Attacker/geoA3_attack.py:238-352) applied to every victim instance
held by a GPU: success check + victim forward + CE/CD/HD/curvature objective + input gradient + Adam step.

Workloads (BASELINE.json `configs`):
  default                         configs[1]: PointNet, 1024 points, 250 instances, full GeoA3 (CD 1.0 + HD 0.1 + curvature
                                  1.0 with k=16), untargeted CE -- the configuration the metric is quoted on.  The
                                  headline keys of the JSON line describe THIS workload and nothing else.
                                  With no workload flag the same run then appends `other_configs`: configs[3], configs[4],
                                  the 32-instance shard of configs[2] (with its full-batch comparison) and configs[1] /
                                  configs[4] / configs[3] on CAD-like clouds -- GPU legs only, `--other-steps` timed steps each
                                  (`--no-other-configs` skips them)
  --gpus N (N > 1)                configs[2]: the SAME 250-instance batch sharded over N GPUs (`--scaling strong`, the
                                  default: rank r holds shard_bounds(250, N)[r] instances, loss divisor 1/250, no
                                  collective in the iteration); `--scaling weak` holds 250 instances on EVERY GPU.
                                  The line also carries `multi_gpu`: the OTHER scaling mode measured briefly, per-rank
                                  ms/step, the result gather of the batch over the process group, and an all-gather of
                                  the rank ids
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

# event-timer tags (include/geoa3_hip_debug.h)
TAG_CONV5, TAG_NN1, TAG_KNN, TAG_TNET, TAG_SA1_BWD, TAG_SA1_FWD, TAG_FC, TAG_GEO = 0, 1, 2, 3, 4, 5, 6, 7
TAG_SA2_FWD, TAG_SA2_BWD, TAG_SA2_GRAD = 8, 9, 10
NTAGS = 12
TAG_NAMES = {TAG_CONV5: "conv5_wide_max", TAG_TNET: "tnet_wide_max(x2)", TAG_NN1: "nn1_pair", TAG_KNN: "knn",
             TAG_GEO: "geo_loss_grad", TAG_SA1_BWD: "sa1_bwd", TAG_SA1_FWD: "sa1_fwd", TAG_SA2_FWD: "sa2_fwd",
             TAG_SA2_BWD: "sa2_bwd", TAG_SA2_GRAD: "sa2_bwd_prep"}


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


def committed_pmc_traffic(kernel_substr, cfg_tag):
    """HBM bytes per launch of a kernel from the COMMITTED rocprofv3 PMC summary of this configuration
    (profiles/round*_<cfg_tag>_pmc.csv, the newest round/version by name that exists in the tree: separate FETCH_SIZE /
    WRITE_SIZE passes of the same command, tools/gpu_profiles.sh; FETCH_SIZE doubled per the gfx950 note in
    MI355X_MICROARCH.md).  A constant of the committed profile -- counters cannot be read inside a timed run -- and the
    line says so (`traffic_measured_in_this_run: false`).  (None, None) when absent."""
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


def attach_traffic(roof, kernel_substr, cfg_tag, full_size):
    """roofline.traffic from the committed PMC passes (only for the batch size the passes were taken at)."""
    if not roof:
        return roof
    roof.setdefault("traffic", None)
    if full_size:
        tr, src = committed_pmc_traffic(kernel_substr, cfg_tag)
        if tr is not None:
            roof["traffic"] = tr
            roof["traffic_source"] = "profiles/%s (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command)" % src
            roof["traffic_measured_in_this_run"] = False
    return roof


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


# ------------------------------------------------------------------------------------------------------------------
# rooflines (algorithmic work per launch: SURVEY 8d / DESIGN.md section 4)
# ------------------------------------------------------------------------------------------------------------------
def conv5_roofline(mode, ms, B, npoint):
    """conv5 + bn5 + relu + max.  `achieved` = ALGORITHMIC flops (2 * B * N * 1024 * 384 per launch) / the
    kernel's measured duration; `peak` = the dense peak of the matrix pipe the kernel runs on.  The split mode executes 3
    f16 products per fp32 product: `executed_frac` prices those against the same peak (how busy the pipe is), `frac`
    prices the useful work."""
    if not ms:
        return None
    flops = 2.0 * B * npoint * 1024 * 384          # 1024 outputs x (3 taps x 128) MACs per point
    ach = flops / (ms * 1e-3)
    if mode == "f16x2":
        from geoa3_amd.pointnet import wide_shape
        kname = ("wide16_kernel<3> (v_mfma_f32_16x16x32_f16)" if wide_shape("conv5") == 16
                 else "wide_split_kernel<3> (v_mfma_f32_32x32x16_f16)")
        return {"bound": "mfma", "kernel": kname + ": conv5+bn5+relu+max; fp32 operands carried as two fp16 "
                                                   "values on the f16 matrix pipe, fp32 accumulate",
                "achieved": round(ach / 1e12, 1), "peak": PEAK_F16_MFMA / 1e12, "unit": "TFLOP/s",
                "frac": round(ach / PEAK_F16_MFMA, 4), "avg_launch_ms": round(ms, 4),
                "algorithmic_flops_per_launch": flops,
                "executed_mfma_flops_per_launch": 3.0 * flops,
                "executed_frac": round(3.0 * ach / PEAK_F16_MFMA, 4),
                "frac_of_fp32_matrix_peak": round(ach / PEAK_F32_MFMA, 3),
                "note": "MI355X_MICROARCH.md: a tuned 8192^3 bf16 GEMM sustains 1247 TFLOP/s on random data "
                        "(the chip lowers its clock under 16-bit MFMA load), i.e. executed_frac ~0.5 is the "
                        "practical ceiling of this pipe", "traffic": None}
    return {"bound": "mfma", "kernel": "wide_max2_kernel<3> (conv5+bn5+relu+max, fp32 MFMA)",
            "achieved": round(ach / 1e12, 2), "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
            "frac": round(ach / PEAK_F32_MFMA, 4), "avg_launch_ms": round(ms, 4),
            "algorithmic_flops_per_launch": flops, "traffic": None}


def pn2_rooflines(kms, B):
    """configs[3]: the three large kernels of the PointNet++ iteration, each against the f16 matrix peak with its
    ALGORITHMIC flops (fp32 products; the split operands execute 3 f16 products per product).  M1 = 512 / M2 = 128
    centroids, 64 samples each (PointNetPP_ssg.py:58-76)."""
    spec = {
        TAG_SA1_BWD: ("sa1_bwd_kernel (PointNet++ level 1 input gradient through 128->64->64->3, split-fp16 operands)",
                      2.0 * (128 * 64 + 64 * 64 + 64 * 3) * B * 512 * 64),
        TAG_SA2_FWD: ("sa2_fwd8_kernel (PointNet++ level 2: gather + shift + relu, 128->128, 128->256 + max, split-fp16 "
                      "operands)", 2.0 * (128 * 128 + 128 * 256) * B * 128 * 64),
        TAG_SA2_BWD: ("sa2b_bwd_kernel (PointNet++ level 2 input gradient + the grouping's scatter-add, rows in destination "
                      "order: sparse pooled gradient through W2 rows, W1^T on the matrix core, per-point sums on chip; "
                      "algorithmic flops = the dense form's)", 2.0 * (128 * 128 * 64 + 256 * 128) * B * 128),
    }
    out = {}
    for tag, (name, flops) in spec.items():
        ms = kms.get(tag)
        if ms:
            out[tag] = {"bound": "mfma", "kernel": name, "achieved": round(flops / (ms * 1e-3) / 1e12, 2),
                        "peak": PEAK_F16_MFMA / 1e12, "unit": "TFLOP/s", "frac": round(flops / (ms * 1e-3) / PEAK_F16_MFMA, 4),
                        "avg_launch_ms": round(ms, 4), "algorithmic_flops_per_launch": flops, "traffic": None}
    return out


def knn_kernel_line(ms, B, npoint, knn):
    """configs[4]: the self top-(k+1) search, reported beside the roofline.  Algorithmic HBM bytes: read
    B*N*12, write B*N*(k+1)*8 (distances + indices); an irregular search bound by L2 / LDS latency, so the HBM fraction is
    small by construction."""
    if not ms:
        return None
    bytes_ = B * npoint * (12.0 + (knn + 1) * 8.0)
    return {"bound": "hbm", "kernel": "knn_grid_kernel (self top-%d on a 16^3 cell grid, one wavefront per query)" % (knn + 1),
            "achieved": round(bytes_ / (ms * 1e-3) / 1e9, 2), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
            "frac": round(bytes_ / (ms * 1e-3) / PEAK_HBM, 5), "avg_launch_ms": round(ms, 4),
            "algorithmic_bytes_per_launch": bytes_, "traffic": None}


def cd_kernel_line(ms, B, npoint, vector_only_us=None):
    """The "CD kernel" (BASELINE.json: CD-kernel HBM GB/s): algorithmic bytes 40 * B * N per launch."""
    if not ms:
        return None
    cd_bytes = 40.0 * B * npoint
    filt = npoint <= 1024 or npoint > 4096
    pairs = 2.0 * B * npoint * npoint
    out = {"kernel": ("nn1_filter_kernel (exact 1-NN both directions: approximate distances of ALL pairs on the matrix core, "
                      "exact evaluation of the pairs under the seed's radius; csrc/geom_filter.h)" if filt else
                      "grid_nn1_kernel (exact 1-NN both directions: grid walk, workgroups with crowded boxes through the "
                      "matrix-core filter search of csrc/geom_filter.h)"),
           "avg_launch_us": round(ms * 1e3, 2), "algorithmic_bytes_per_launch": cd_bytes,
           "hbm_GBps_algorithmic": round(cd_bytes / (ms * 1e-3) / 1e9, 2),
           "hbm_frac": round(cd_bytes / (ms * 1e-3) / PEAK_HBM, 5),
           "note": "not HBM-bound by construction (SURVEY 8d: 10 MB per launch for 2*B*N^2 pairs of search)"}
    if vector_only_us is not None:
        out["vector_only_form_us"] = round(vector_only_us, 2)
        out["vector_only_form"] = ("the same search by grid walk + LDS-broadcast sweep on the vector unit alone (north_star's form, "
                                   "geoa3_debug_grid_nn1_pair(..., 0.12, 0)) on the loop's state after the run, alone on the queue; "
                                   "bit-identical results (checked in this run)")
    if filt:   # every pair goes through the filter: its rate against the vector-issue floor
        out["pairs_per_launch"] = pairs
        out["pairs_per_cycle_per_simd"] = round(pairs / (ms * 1e-3) / (1024 * 2.4e9), 2)
        out["issue_floor_pairs_per_cycle_per_simd"] = 19.7   # 1024 pairs per (8 v_min3 + compare + 2) * 4 + 8 MFMA issue cycles
    return out


class Bench:
    """Process-wide state of one bench.py run: the process group, the device, the library, cached victims."""

    def __init__(self, a):
        import torch
        import torch.distributed as dist
        self.a, self.torch, self.dist = a, torch, dist
        # a real launcher exports all of RANK, WORLD_SIZE and MASTER_PORT; a stray WORLD_SIZE=1 from a scheduler does not count
        launched = all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_PORT"))
        self.rank = int(os.environ.get("RANK", "0")) if launched else 0
        local_rank = int(os.environ.get("LOCAL_RANK", "0")) if launched else 0
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if launched else 1
        if a.gpus > 1 and self.world != a.gpus:
            sys.exit("bench.py: --gpus %d but the launcher started %d ranks" % (a.gpus, self.world))
        ndev = max(torch.cuda.device_count(), 1)
        self.dev = torch.device("cuda", (local_rank % ndev) if self.world > 1 else 0)
        torch.cuda.set_device(self.dev)
        self.backend = os.environ.get("GEOA3_BENCH_BACKEND", "nccl")   # "gloo": functional test of the N>1 path on one GPU
        # under a launcher (WORLD_SIZE set) the process group is created even for ONE rank: a 1-GPU box then still runs the
        # RCCL initialisation, the barriers and the device-tensor all-reduce of the N > 1 path
        self.use_dist = launched
        if self.use_dist:
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)
        import __graft_entry__
        if not os.path.exists(os.path.join(REPO, "geoa3_amd", "lib", "libgeoa3_hip.so")):
            __graft_entry__.build()
        from geoa3_amd import _lib
        self.lib = _lib.load()
        self._victims = {}

    def victim(self, arch):
        torch = self.torch
        if arch not in self._victims:
            if arch == "PointNet":
                from geoa3_amd.data import synthetic_state_dict
                from geoa3_amd.pointnet import PointNet
                net = PointNet(CLASSES)
                net.load_state_dict(synthetic_state_dict(CLASSES, seed=0, device=self.dev))
            else:
                from geoa3_amd.pointnet2 import PointNet2ClassificationSSG
                torch.manual_seed(0)
                net = PointNet2ClassificationSSG(use_xyz=True, use_normal=False)
            self._victims[arch] = net.to(self.dev).eval()
        return self._victims[arch]

    def generator(self, data, cad_kinds="", cad_duplicates=0.05):
        from geoa3_amd.data import SYNTHETIC_GENERATORS
        gen = SYNTHETIC_GENERATORS[data]
        if data == "cad" and (cad_kinds or cad_duplicates != 0.05):
            import functools
            gen = functools.partial(gen, duplicates=cad_duplicates, **({"kinds": tuple(cad_kinds.split(","))} if cad_kinds else {}))
        return gen

    def clouds(self, arch, gen, npoint, count, seed, rows=None):
        """-> (ori, nrm, gt) on the device; rows = (lo, hi) cuts a shard out of the seeded batch of `count` clouds."""
        torch = self.torch
        ori, nrm = gen(count, npoint, seed=seed)
        if rows is not None:
            ori, nrm = ori[rows[0]:rows[1]].contiguous(), nrm[rows[0]:rows[1]].contiguous()
        ori, nrm = ori.to(self.dev), nrm.to(self.dev)
        with torch.no_grad():
            gt = (self.victim(arch)(ori).argmax(1) if ori.shape[0] > 0
                  else torch.zeros(0, dtype=torch.long, device=self.dev))
        return ori, nrm, gt

    def barrier(self):
        self.torch.cuda.synchronize()
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, x):
        if not self.use_dist:
            return x
        torch = self.torch
        t = torch.tensor([x], device=self.dev if self.backend == "nccl" else "cpu", dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def measure(self, arch, pts, npoint, knn, global_batch, wide_mode, steps, warmup, presteps, timed_tags):
        """`presteps` + `warmup` untimed, then exactly `steps` timed inner iterations between two barriers
        -> dict(dt = seconds, max over ranks; dt_local; kms = per-kernel average ms by event-timer tag; extra;
        host_enqueue_ms = cost of ENQUEUEING one iteration into an empty queue; host_loop_ms = host time per step inside the
        timed loop, which includes blocking on a full queue)."""
        torch, lib = self.torch, self.lib
        from geoa3_amd.attack import AttackRunner
        ori, nrm, gt = pts
        b = int(ori.shape[0])
        total = presteps + warmup + steps
        n_extra, n_host = min(steps, 10), 10
        cfg = cfg_full_geoa3(total + n_extra + n_host + 4, npoint, knn)
        net = self.victim(arch)
        if arch == "PointNet":
            net.wide_mode = wide_mode
        runner = AttackRunner(net, b, npoint, cfg, self.dev, global_batch=global_batch)
        runner.setup(ori, nrm, gt, gt)
        g = torch.Generator(device="cpu").manual_seed(7 + self.rank)
        init = (torch.randn(b, 3, npoint, generator=g) * 1e-3).to(self.dev)
        runner.begin_search_step(init)
        for s in range(presteps + warmup):
            runner.step(s, 0)
        self.barrier()
        # HIP events around the DOMINANT kernel(s) only inside the timed region (an event pair costs ~6 us of stream
        # time); the other kernels' durations come from a few extra, untimed iterations afterwards
        mask = 0
        for t in timed_tags:
            mask |= 1 << t
        lib.geoa3_profile_enable(steps)
        lib.geoa3_profile_select(mask)
        t0 = time.perf_counter()
        for s in range(presteps + warmup, total):
            runner.step(s, 0)
        host_loop = time.perf_counter() - t0
        self.barrier()
        dt_local = time.perf_counter() - t0
        dt = self.max_over_ranks(dt_local)

        def kernel_ms(tag):
            buf = (C.c_float * steps)()
            n = lib.geoa3_profile_read(tag, buf, steps)
            return (sum(buf[:n]) / n) if n > 0 else None

        kms = {t: kernel_ms(t) for t in timed_tags}
        lib.geoa3_profile_select(((1 << NTAGS) - 1) & ~mask)
        geo_stream, runner.geo_stream = runner.geo_stream, None   # one stream: durations of the kernels on their own
        for s in range(total, total + n_extra):
            runner.step(s, 0)
        torch.cuda.synchronize()
        runner.geo_stream = geo_stream
        for tag in range(NTAGS):
            if tag not in kms:
                kms[tag] = kernel_ms(tag)
        lib.geoa3_profile_select(0xFFFFFFFF)
        lib.geoa3_profile_enable(0)
        # what the host pays to enqueue ONE iteration: the queue is drained before each, the event timers are off
        host = 0.0
        for s in range(total + n_extra, total + n_extra + n_host):
            torch.cuda.synchronize()
            h0 = time.perf_counter()
            runner.step(s, 0)
            host += time.perf_counter() - h0
        torch.cuda.synchronize()
        # the same 1-NN search in north_star's form (grid walk + LDS-broadcast sweep, the vector unit only: round 5's policy)
        # on the loop's state at this point, alone on the queue -- beside the shipped search's `nn1_pair` figure
        vec_us = None
        t = runner.t
        if runner.need_nn and runner.grid_nn1 and runner.ne <= 4096 and runner.n <= 4096:
            both = runner.dis_type == 1 and not getattr(runner.cfg, "is_cd_single_side", False)
            xe = t["x"]
            if xe.shape[2] == runner.ne:
                tmp = [torch.empty_like(t[k]) for k in ("d_ao", "i_ao", "d_oa", "i_oa")]
                ref = [torch.empty_like(t[k]) for k in ("d_ao", "i_ao", "d_oa", "i_oa")]
                stream = torch.cuda.current_stream().cuda_stream
                head = (xe.data_ptr(), runner.ori.data_ptr(), runner.b, runner.ne, runner.n, t["i_ao"].data_ptr(),
                        t["i_oa"].data_ptr() if both else None)

                def outs(bufs):
                    return (bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr() if both else None,
                            bufs[3].data_ptr() if both else None)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                if (lib.geoa3_grid_nn1_pair(*head, *outs(ref), stream) == 0 and
                        lib.geoa3_debug_grid_nn1_pair(*head, *outs(tmp), 0.12, 0, stream) == 0):
                    e0.record()
                    for _ in range(5):
                        lib.geoa3_debug_grid_nn1_pair(*head, *outs(tmp), 0.12, 0, stream)
                    e1.record()
                    torch.cuda.synchronize()
                    vec_us = e0.elapsed_time(e1) / 5 * 1e3
                    if not (torch.equal(tmp[1], ref[1]) and torch.equal(tmp[0].view(torch.int32), ref[0].view(torch.int32))):
                        raise RuntimeError("the vector-only 1-NN search and the shipped one disagree")
        runner.end_search_step()
        del runner
        return {"dt": dt, "dt_local": dt_local, "kms": kms, "extra": n_extra, "host_enqueue_ms": host / n_host * 1e3,
                "host_loop_ms": host_loop / steps * 1e3, "nn1_vector_only_us": vec_us}


def kernels_ms_dict(kms, extra, timed_tags):
    d = {name: kms.get(tag) for tag, name in TAG_NAMES.items() if kms.get(tag) is not None}
    d["note"] = ("%s: HIP events inside the timed region; the others: %d untimed iterations right after it, on ONE stream "
                 "(in the timed loop the geometry kernels run on a second stream beside the victim's forward)"
                 % (", ".join(TAG_NAMES[t] for t in timed_tags), extra))
    return d


def gpu_leg(bench, name, arch, npoint, knn, data, instances, steps, warmup, presteps, cfg_idx, full_compare=False):
    """One entry of `other_configs`: a 1-GPU measurement of another BASELINE.json configuration in this process (same
    barriers, same event timers as the headline; no CPU baseline)."""
    torch = bench.torch
    gen = bench.generator(data)
    pts = bench.clouds(arch, gen, npoint, instances, seed=100)
    from geoa3_amd.pointnet import default_wide_mode
    wmode = default_wide_mode() if arch == "PointNet" else None
    timed = [TAG_CONV5] if arch == "PointNet" else [TAG_SA1_BWD, TAG_SA2_FWD, TAG_SA2_BWD]
    m = bench.measure(arch, pts, npoint, knn, BATCH, wmode, steps, warmup, presteps, timed)
    ms_per_step = m["dt"] / steps * 1e3
    kms = m["kms"]
    victim = "PointNet" if arch == "PointNet" else "PointNet++ SSG"
    what = ("%d instances" % instances if instances == BATCH else
            "ONE rank's %d-instance shard of the %d-instance batch on 1 GPU (loss divisor 1/%d)" % (instances, BATCH, BATCH))
    out = {"name": name,
           "workload": "configs[%d]: %s %d-pt, %s, full GeoA3 (CE + CD 1.0 + HD 0.1 + curvature 1.0 k=%d), untargeted"
                       % (cfg_idx, victim, npoint, what, knn),
           "data": "synthetic" if data == "ellipsoid" else "synthetic (%s)" % data,
           "value": round((instances * steps / m["dt"]) / BATCH, 3), "unit": "iterations/s of a 250-instance batch",
           "ms_per_step": round(ms_per_step, 4), "steps": steps, "warmup": warmup, "presteps": presteps,
           "host_enqueue_ms_per_step": round(m["host_enqueue_ms"], 4)}
    if arch == "PointNet":
        roof = conv5_roofline(wmode, kms.get(TAG_CONV5), instances, npoint)
        out["roofline"] = attach_traffic(roof, "wide16_kernel<3", "c2" if npoint < 4096 else "c5",
                                         instances == BATCH and npoint in (1024, 4096) and data == "ellipsoid")
        if npoint >= 4096:
            out["knn_kernel"] = knn_kernel_line(kms.get(TAG_KNN), instances, npoint, knn)
    else:
        roofs = pn2_rooflines(kms, instances)
        big = max(roofs, key=lambda t: roofs[t]["avg_launch_ms"]) if roofs else None
        out["roofline"] = attach_traffic(roofs.get(big), {TAG_SA1_BWD: "sa1_bwd_kernel", TAG_SA2_FWD: "sa2_fwd8_kernel",
                                                         TAG_SA2_BWD: "sa2b_bwd_kernel"}.get(big, ""), "c4",
                                         instances == BATCH and data == "ellipsoid")
        out["other_large_kernels"] = [roofs[t] for t in roofs if t != big]
    out["cd_kernel"] = cd_kernel_line(kms.get(TAG_NN1), instances, npoint, m.get("nn1_vector_only_us"))
    out["kernels_ms"] = kernels_ms_dict(kms, m["extra"], timed)
    if full_compare:
        # the full 250-instance batch on this GPU in the same process: what linear scaling is measured against
        fpts = bench.clouds(arch, gen, npoint, BATCH, seed=100)
        fm = bench.measure(arch, fpts, npoint, knn, BATCH, wmode, steps, 5, presteps, [TAG_CONV5])
        out["strong_scaling_proxy"] = proxy_dict(fm["dt"] / steps * 1e3, ms_per_step, instances, BATCH)
    torch.cuda.empty_cache()
    return out


def proxy_dict(full_ms, shard_ms, B, global_batch):
    ranks = float(global_batch) / B
    return {"full_batch_ms_per_step": round(full_ms, 4), "shard_ms_per_step": round(shard_ms, 4),
            "ranks_emulated": round(ranks, 2), "linear_shard_ms": round(full_ms / ranks, 4),
            "fraction_of_linear": round(full_ms / ranks / shard_ms, 4),
            "note": "1-GPU proxy: the %d-instance shard a rank of the %.1f-way split holds vs the whole %d-instance "
                    "batch on the same GPU, same process (no collective runs in the iteration)" % (B, ranks, global_batch)}


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
    ap.add_argument("--no-other-configs", action="store_true",
                    help="the default run: only the configs[1] headline, not the `other_configs` legs")
    ap.add_argument("--other-steps", type=int, default=40, help="timed steps of each `other_configs` leg")
    ap.add_argument("--other-presteps", type=int, default=150, help="untimed iterations in front of each `other_configs` leg")
    a = ap.parse_args()
    npoint, knn = a.npoint, a.knn

    launched = all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_PORT"))
    if a.gpus > 1 and not launched:
        # `python bench.py --gpus N` without a launcher: start the N ranks as a CHILD process group (torch.distributed.run)
        # before this process has imported torch or touched the GPU, and leave with its return code (a process that has
        # initialised the GPU must never exec another program on this pool)
        sys.exit(spawn_ranks(a.gpus))

    bench = Bench(a)
    torch, dist = bench.torch, bench.dist
    rank, world, dev, use_dist, backend = bench.rank, bench.world, bench.dev, bench.use_dist, bench.backend
    from geoa3_amd.distributed import shard_bounds
    from geoa3_amd.pointnet import default_wide_mode

    # which instances this rank holds, and the divisor of the loss mean (geoA3_attack.py:178 -> 1 / GLOBAL batch)
    def split(scaling):
        if a.instances > 0:            # explicit per-GPU shard (1-GPU proxy of a sharded run, or a custom weak run)
            B, lo = a.instances, 0
            gb = max(a.global_batch, B * world) if scaling == "strong" else B * world
            mode = "shard-proxy" if world == 1 and B != BATCH else scaling
        elif scaling == "strong":
            gb = a.global_batch
            lo, hi = shard_bounds(gb, world)[rank]
            B, mode = hi - lo, "strong"
        else:
            B, lo, gb, mode = BATCH, 0, BATCH * world, "weak"
        total = B * world if (a.instances > 0 or mode == "weak") else gb
        return B, lo, gb, mode, total

    B, lo, global_batch, mode, total_instances = split(a.scaling)
    gen = bench.generator(a.data, a.cad_kinds, a.cad_duplicates)

    def rank_clouds(B_, lo_, gb_, mode_):
        if mode_ == "strong":          # every rank cuts ITS rows out of the same seeded global batch
            return bench.clouds(a.arch, gen, npoint, gb_, seed=100, rows=(lo_, lo_ + B_))
        return bench.clouds(a.arch, gen, npoint, B_, seed=100 + rank)

    pts = rank_clouds(B, lo, global_batch, mode)
    timed_tags = [TAG_CONV5] if a.arch == "PointNet" else [TAG_SA1_BWD, TAG_SA2_FWD, TAG_SA2_BWD]
    wmode = default_wide_mode() if a.arch == "PointNet" else None
    m = bench.measure(a.arch, pts, npoint, knn, global_batch, wmode, a.steps, a.warmup, a.presteps, timed_tags)
    dt, kms, extra = m["dt"], m["kms"], m["extra"]

    other = None
    if a.arch == "PointNet" and not a.single_mode and mode != "shard-proxy":
        # the same loop with every convolution on the fp32 MFMA (strict fp32 products), shorter, beside the headline
        omode = "f32" if wmode == "f16x2" else "f16x2"
        osteps = max(10, min(a.steps, 40))
        om = bench.measure(a.arch, pts, npoint, knn, global_batch, omode, osteps, min(a.warmup, 5), min(a.presteps, 60),
                           timed_tags)
        other = (omode, osteps, om["dt"], om["kms"])
    proxy = None
    if mode == "shard-proxy" and a.arch == "PointNet" and not a.no_proxy_full:
        fpts = bench.clouds(a.arch, gen, npoint, BATCH, seed=100)
        fsteps = max(10, min(a.steps, 60))
        fm = bench.measure(a.arch, fpts, npoint, knn, BATCH, wmode, fsteps, 5, a.presteps, timed_tags)
        proxy = (fsteps, fm["dt"])
        del fpts

    multi = None
    if use_dist and a.instances == 0:
        multi = multi_gpu_report(bench, a, split, rank_clouds, timed_tags, wmode, m, mode)

    # the default command (no workload flag): the other BASELINE.json configurations under the same clock, GPU legs only
    default_run = (world == 1 and a.arch == "PointNet" and npoint == NPOINT and knn == KNN and a.data == "ellipsoid" and
                   a.instances == 0 and a.global_batch == BATCH and not a.no_other_configs)
    others = None
    if default_run:
        del pts
        torch.cuda.empty_cache()
        st, ws, ps = a.other_steps, min(a.warmup, 10), a.other_presteps
        others = []
        for kw in (dict(name="configs[3] PointNet++ SSG", arch="PointNetPP", npoint=1024, knn=16, data="ellipsoid",
                        instances=BATCH, cfg_idx=3),
                   dict(name="configs[4] N=4096 k=32", arch="PointNet", npoint=4096, knn=32, data="ellipsoid",
                        instances=BATCH, cfg_idx=4),
                   dict(name="configs[2] one rank's 32-instance shard (1-GPU proxy of the 8-GPU run)", arch="PointNet",
                        npoint=1024, knn=16, data="ellipsoid", instances=32, cfg_idx=2, full_compare=True),
                   dict(name="configs[1] on CAD-like clouds", arch="PointNet", npoint=1024, knn=16, data="cad",
                        instances=BATCH, cfg_idx=1),
                   dict(name="configs[4] on CAD-like clouds", arch="PointNet", npoint=4096, knn=32, data="cad",
                        instances=BATCH, cfg_idx=4),
                   dict(name="configs[3] on CAD-like clouds", arch="PointNetPP", npoint=1024, knn=16, data="cad",
                        instances=BATCH, cfg_idx=3)):
            try:
                others.append(gpu_leg(bench, steps=st, warmup=ws, presteps=ps, **kw))
            except Exception as e:   # a failing extra leg never takes the headline down; it is reported as failed
                others.append({"name": kw["name"], "error": repr(e)})

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = (total_instances * a.steps / dt) / BATCH
        cfg_idx = 3 if a.arch != "PointNet" else (4 if npoint >= 4096 else (2 if (world > 1 or mode == "shard-proxy") else 1))
        victim = "PointNet" if a.arch == "PointNet" else "PointNet++ SSG"
        out = {
            "metric": "attack-iterations/sec (B=250, N=%d)" % npoint, "value": round(value, 3),
            "unit": "iterations/s of a 250-instance batch", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "host_enqueue_ms_per_step": round(m["host_enqueue_ms"], 4),
            "host_loop_ms_per_step": round(m["host_loop_ms"], 4),
            "host_note": "host_enqueue = host time to enqueue ONE iteration into a drained queue (10 untimed iterations after "
                         "the timed window); host_loop = host time per step inside the timed loop, which includes blocking "
                         "on a full queue when the GPU is the bound",
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
        full_size = B == BATCH and a.data == "ellipsoid"
        if a.arch == "PointNet":
            from geoa3_amd.pointnet import wide_shape
            kn = (("wide16_kernel<3" if wide_shape("conv5") == 16 else "wide_split_kernel<3") if wmode == "f16x2"
                  else "wide_max2_kernel<3")
            out["roofline"] = attach_traffic(conv5_roofline(wmode, kms.get(TAG_CONV5), B, npoint), kn,
                                             "c2" if npoint < 4096 else "c5", full_size and npoint in (1024, 4096))
            if npoint >= 4096:
                out["knn_kernel"] = attach_traffic(knn_kernel_line(kms.get(TAG_KNN), B, npoint, knn), "knn_grid_kernel", "c5",
                                                   full_size)
        else:
            roofs = pn2_rooflines(kms, B)
            big = max(roofs, key=lambda t: roofs[t]["avg_launch_ms"]) if roofs else None
            out["roofline"] = attach_traffic(roofs.get(big), {TAG_SA1_BWD: "sa1_bwd_kernel", TAG_SA2_FWD: "sa2_fwd8_kernel",
                                                             TAG_SA2_BWD: "sa2b_bwd_kernel"}.get(big, ""), "c4", full_size)
            out["other_large_kernels"] = [roofs[t] for t in roofs if t != big]
        if other is not None:
            omode, osteps, odt, okms = other
            oroof = attach_traffic(conv5_roofline(omode, okms.get(TAG_CONV5), B, npoint),
                                   "wide_max2_kernel<3" if omode == "f32" else "wide16_kernel<3",
                                   "c2f32" if omode == "f32" else "c2", full_size and npoint == NPOINT)
            out["other_wide_mode"] = {"wide_mode": omode, "value": round((total_instances * osteps / odt) / BATCH, 3),
                                      "ms_per_step": round(odt / osteps * 1e3, 4), "steps": osteps,
                                      "roofline": oroof,
                                      "kernels_ms": {"conv5_wide_max": okms.get(TAG_CONV5),
                                                     "tnet_wide_max(x2)": okms.get(TAG_TNET)}}
        if proxy is not None:
            fsteps, fdt = proxy
            out["strong_scaling_proxy"] = proxy_dict(fdt / fsteps * 1e3, ms_per_step, B, global_batch)
        out["cd_kernel"] = cd_kernel_line(kms.get(TAG_NN1), B, npoint, m.get("nn1_vector_only_us"))
        out["kernels_ms"] = kernels_ms_dict(kms, extra, timed_tags)
        if multi is not None:
            out["multi_gpu"] = multi
        if others is not None:
            out["other_configs"] = others
        if not a.no_cpu_baseline and world == 1:
            try:
                out["cpu_baseline"] = cpu_baseline(a.arch, npoint, knn, budget_s=a.cpu_budget, threads=a.cpu_threads)
            except Exception as e:  # the baseline never blocks the GPU number
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


def multi_gpu_report(bench, a, split, rank_clouds, timed_tags, wmode, headline, mode):
    """The extra evidence of an N-rank run (every rank takes part, rank 0 prints): the OTHER scaling mode measured briefly,
    per-rank ms/step of the headline (min / max over ranks), the gather of one batch's results over the process group
    (geoa3_amd.distributed.gather_results_timed: best_attack shards + success + best step + loss history, ~3.5 MB at 250 x
    1024 x 500 iterations) and a device-tensor all-gather of the rank ids (proof that the backend saw N ranks)."""
    torch, dist = bench.torch, bench.dist
    from geoa3_amd.distributed import gather_results_timed, rank_roll_call, shard_bounds
    world, rank, dev = bench.world, bench.rank, bench.dev
    cdev = dev if bench.backend == "nccl" else torch.device("cpu")
    # per-rank step time of the headline window
    t = torch.tensor([headline["dt_local"] / a.steps * 1e3], dtype=torch.float64, device=cdev)
    per_rank = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(per_rank, t)
    per_rank = [float(x.item()) for x in per_rank]
    # the other scaling mode, briefly
    omode = "weak" if mode == "strong" else "strong"
    B2, lo2, gb2, mode2, total2 = split(omode)
    osteps = max(6, min(a.steps, 40))
    pts2 = rank_clouds(B2, lo2, gb2, mode2)
    om = bench.measure(a.arch, pts2, a.npoint, a.knn, gb2, wmode, osteps, min(a.warmup, 5), min(a.presteps, 60), timed_tags)
    del pts2
    # the result gather of ONE batch (configs[2]: the 250-instance batch, 500 iterations of loss history)
    counts = [h - l for l, h in shard_bounds(BATCH, world)]
    bl = counts[rank]
    best = torch.ones(bl, 3, a.npoint, device=dev)
    succ = torch.ones(bl, dtype=torch.uint8, device=dev)
    step = torch.zeros(bl, dtype=torch.int64, device=dev)
    loss = torch.zeros(bl, 500, device=dev)
    gather = gather_results_timed(best, succ, step, loss, counts, repeats=5, sync=torch.cuda.synchronize)
    roll = rank_roll_call(dev if bench.backend == "nccl" else torch.device("cpu"))
    return {"headline_scaling": mode, "ms_per_step_per_rank": {"min": round(min(per_rank), 4), "max": round(max(per_rank), 4),
                                                               "all": [round(x, 4) for x in per_rank]},
            "other_scaling": {"scaling": mode2, "value": round((total2 * osteps / om["dt"]) / BATCH, 3),
                              "ms_per_step": round(om["dt"] / osteps * 1e3, 4), "steps": osteps,
                              "instances_per_gpu": B2, "global_batch": gb2},
            "result_gather": gather, "rank_roll_call": roll}


if __name__ == "__main__":
    main()
