"""Driver-level boundary: ``attack(net, input_data, cfg, i, loader_len, saved_dir)`` with the
reference's signature, return tuple and semantics (Attacker/geoA3_attack.py:182-386), as a device-resident
loop: every inner iteration is seven enqueues into the HIP library and ZERO host synchronisations (the
reference does ~15 000 tiny launches and ~760 ``.item()`` syncs per iteration at b=250).

Reference quirks kept on purpose:
  * the success check of step s ranks the CURRENT iterate with the constrain loss of step s-1
    (1e10 at step 0), geoA3_attack.py:301;
  * the binary search compares the label of the LAST instance at the LAST step against every
    instance's target (geoA3_attack.py:298,375);
  * ``loss = loss_n.mean()`` scales every gradient by 1/b (geoA3_attack.py:178) -- the GLOBAL b when the batch
    is sharded over GPUs, so shards reproduce the single-GPU iterates.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os
from typing import Callable, Optional, Sequence

import numpy as np
import torch

from . import _lib, ops
from ._lib import AttackState, check
from .pointnet import PointNet

Tensor = torch.Tensor


def _cfg(cfg, name, default):
    return getattr(cfg, name, default)


class AttackRunner:
    """Owns the device state of one batch of b attacks in flight and enqueues the loop."""

    def __init__(self, net: PointNet, b: int, n: int, cfg, device, global_batch: Optional[int] = None):
        if _cfg(cfg, "uniform_loss_weight", 0.0) != 0:
            # the reference's uniform_loss (Lib/loss_utils.py:151-189) calls pointnet2_utils without importing it:
            # --uniform_loss_weight != 0 dies with a NameError there (geoA3_attack.py:170)
            raise NotImplementedError("uniform_loss cannot run in the reference either (NameError: pointnet2_utils "
                                      "is never imported in Lib/loss_utils.py); default weight 0")
        if cfg.dis_loss_type == "L2" and cfg.hd_loss_weight != 0:
            raise AssertionError("L2 distance needs hd_loss_weight == 0")  # geoA3_attack.py:140
        if cfg.optim not in ("adam", "sgd"):
            raise AssertionError("Not support such optimizer.")
        self.net, self.cfg, self.b, self.n, self.dev = net, cfg, b, n, device
        self.lib = _lib.load()
        # Native victim: the PointNet whose forward / input-gradient live in the HIP library.  Any other eval-mode
        # nn.Module (PointNet++ SSG on the HIP set-abstraction operators, geoa3_amd/pointnet2.py) is driven through
        # torch autograd: logits = net(x); logits.backward(dlogits) -- still no host synchronisation.
        # ... and the PointNet++ SSG classifier of geoa3_amd/pointnet2.py (eval mode, xyz only) is native too:
        # geoa3_pn2ssg_forward / _backward (csrc/pointnet2_net.hip)
        from .pointnet2 import PointNet2ClassificationSSG
        self.ssg = (isinstance(net, PointNet2ClassificationSSG) and not net.training and not net.use_normal and
                    min(n, int(_cfg(cfg, "npoint", n))) >= PointNet2ClassificationSSG.MIN_POINTS)
        self.native = isinstance(net, PointNet) or self.ssg
        if self.native:
            self.packed = net.packed(device)
            self.classes = self.packed.classes
        else:
            self.packed = None
            self.classes = int(_cfg(cfg, "classes", 40))
            # only d/d input is needed (the reference also forms the unused weight gradients: its parameters keep
            # requires_grad=True): the flags are cleared for the duration of run() and restored after it
            self._grad_flags = None
        self.global_batch = global_batch or b
        self.targeted = cfg.attack_label != "Untarget"
        self.k = int(cfg.curv_loss_knn)
        self.use_curv = cfg.curv_loss_weight != 0
        self.dis_type = {"CD": 1, "L2": 2, "None": 0}[cfg.dis_loss_type]
        self.need_nn = self.dis_type == 1 or cfg.hd_loss_weight != 0 or self.use_curv
        self.iters = int(cfg.iter_max_steps)
        # the objective's gradient summed in a fixed order (geoa3_geo_args.deterministic): iterates are reproducible bit
        # for bit and shard rows equal the full batch's; cfg.deterministic / GEOA3_DETERMINISTIC=0 select the atomics
        self.deterministic = bool(_cfg(cfg, "deterministic", os.environ.get("GEOA3_DETERMINISTIC", "1") != "0"))
        # Dense-cloud path (geoA3_attack.py:283-296): the offset lives on all n points, the objective sees the
        # farthest-point sample of cfg.npoint of them, success is a vote over eval_num resamplings.
        self.npoint = int(_cfg(cfg, "npoint", n))
        # --is_partial_var (geoA3_attack.py:239-262): a [b,3,knn_range] offset on the neighbours of one random clean
        # point, re-drawn (with a fresh optimiser) every 50 steps on top of the iterate reached so far
        self.partial = bool(_cfg(cfg, "is_partial_var", False))
        self.kr = int(_cfg(cfg, "knn_range", 3))
        self.sub = bool(_cfg(cfg, "is_subsample_opt", False)) and n > self.npoint and not self.partial
        self.ne = self.npoint if self.sub else n          # points the objective is evaluated on
        self.eval_num = int(_cfg(cfg, "eval_num", 1))
        if self.sub and not 1 <= self.eval_num <= 64:
            raise ValueError("eval_num must be in 1..64")
        # --is_pre_jitter_input (geoA3_attack.py:312-317): the objective is evaluated at x + tangent-plane noise
        self.jitter = bool(_cfg(cfg, "is_pre_jitter_input", False))
        self.hooks = {}
        # The geometry kernels of an iteration (1-NN, K-NN, losses: VALU / latency bound) do not depend on the victim's
        # forward (MFMA / HBM bound): they are enqueued on a second HIP stream between two events, so the hardware
        # runs them beside it (GEOA3_GEO_STREAM=0: everything on the current stream).  Same kernels, same results.
        # Measured: 2.78 -> 2.72 ms/iteration (N = 4096: 12.7 -> 11.7); joining only after the backward (head without
        # the constrain loss, added afterwards) gained nothing more: the kernels slow each other down.
        self.geo_stream = (torch.cuda.Stream(device=device)
                           if os.environ.get("GEOA3_GEO_STREAM", "1") != "0" and torch.device(device).type == "cuda"
                           else None)
        self.ev_x, self.ev_geo = ((torch.cuda.Event(), torch.cuda.Event()) if self.geo_stream is not None
                                  else (None, None))
        # Small shards (the 32-instance regime of an 8-GPU split) are latency bound: there the geometry chain is as long
        # as the forward, so the head is split (geoa3_attack_head_classify / _finish) and the geometry stream is joined
        # only AFTER the victim's backward.  At 250 instances the kernels are throughput bound and the late join gains
        # nothing (measured), so it is taken for b <= 96 (cfg.late_join overrides).  Same results.
        lj = _cfg(cfg, "late_join", None)           # (tests: cfg.late_join = True / False)
        self.late_join = (b <= 96) if lj is None else bool(lj)
        # the 1-NN tables through the pruned searches (geom_grid.hip / geom_filter.hip; same bits as the all-pairs kernel,
        # which stays the path for clouds of fewer than 32 points or when cfg.brute_force_nn1 is set)
        self.grid_nn1 = min(n, self.ne) >= 32 and not _cfg(cfg, "brute_force_nn1", False)
        # (tools: GEOA3_NN1_POLICY="brute_frac,filter" runs the loop's searches through geoa3_debug_grid_nn1_pair -- same
        # bits, another split between grid walk, in-kernel sweep and filter search: include/geoa3_hip_debug.h)
        pol = os.environ.get("GEOA3_NN1_POLICY")
        self.nn1_policy = (float(pol.split(",")[0]), int(pol.split(",")[1])) if pol else None
        ne = self.ne
        f32 = dict(device=device, dtype=torch.float32)
        i32 = dict(device=device, dtype=torch.int32)
        z = lambda *s: torch.zeros(*s, **f32)
        self.t = t = {}
        for name in ("offset", "m", "v", "x"):
            t[name] = z(b, 3, n)
        for name in ("g_cls", "g_geo"):
            t[name] = z(b, 3, ne)
        if self.sub:
            E = self.eval_num
            t["g_cls_full"], t["g_geo_full"], t["x_cur"] = z(b, 3, n), z(b, 3, n), z(b, 3, ne)
            t["sub_idx"] = torch.zeros(b, ne, **i32)
            t["vote_idx"], t["vote_pts"] = torch.zeros(b * E, ne, **i32), z(b * E, 3, ne)
            t["vote_logits"] = z(b * E, self.classes)
        if self.partial:
            t["part"], t["pm"], t["pv"] = z(b, 3, self.kr), z(b, 3, self.kr), z(b, 3, self.kr)
            t["pidx"], t["periodical"] = torch.zeros(b, self.kr, **i32), z(b, 3, n)
        if self.jitter:
            t["noise"], t["x_eval"] = z(b, 3, ne), z(b, 3, ne)
            if not self.sub:
                t["check_logits"] = z(b, self.classes)
        t["best_attack"] = torch.ones(b, 3, n, **f32)
        for name in ("scale_const", "lower", "upper", "best_loss", "iter_best_loss", "prev_constrain", "cls_loss",
                     "loss_n"):
            t[name] = z(b)
        for name in ("best_step", "best_bs", "iter_best_score", "label"):
            t[name] = torch.zeros(b, **i32)
        t["last_label"] = torch.zeros(1, **i32)
        t["ok"] = torch.zeros(b, **i32)
        t["loss_hist"] = z(self.iters, b)
        t["logits"], t["dlogits"] = z(b, self.classes), z(b, self.classes)
        t["d_ao"], t["d_oa"] = z(b, ne), z(b, n)
        t["i_ao"], t["i_oa"] = torch.zeros(b, ne, **i32), torch.zeros(b, n, **i32)
        if self.sub and _cfg(cfg, "is_pro_grad", False):   # the projections search with all n points
            t["proj_d"], t["proj_i"] = z(b, n), torch.zeros(b, n, **i32)
        else:
            t["proj_d"], t["proj_i"] = t["d_ao"], t["i_ao"]
        self.geo_out = {name: z(b) for name in ("dis_loss", "hd_loss", "curv_loss", "constrain")}
        self.geo_out["grad"] = t["g_geo"]
        # clouds of 1025..4096 points (or k > 32): the records of the fixed-point objective kernel (geoa3_geo_args.scratch)
        self.geo_scratch = ops.geo_scratch(b, ne, device) if ((1024 < ne or self.k > 32) and ne <= 4096 and self.use_curv) else None
        if self.use_curv:
            t["knn"] = [torch.zeros(b, ne, self.k + 1, **i32) for _ in range(2)]
            t["knn_d"] = z(b, ne, self.k + 1)
            self.knn_slab = True
            self.knn_method = int(_cfg(cfg, "knn_method", 0))   # geoa3_knn_self: 0 = by (K, N), 1 = slab, 2 = cell grid
            t["knn_scratch"] = ops.knn_self_scratch(b, ne, device)
        if self.native:
            bw = b * self.eval_num if self.sub else b
            nbytes = (self.lib.geoa3_pn2ssg_workspace_bytes(bw, ne) if self.ssg
                      else self.lib.geoa3_pointnet_workspace_bytes(bw, ne, self.classes))
            self.ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        self.state: Optional[AttackState] = None
        self._ws_fwd_shape = None       # (B, N) of the last victim forward that used self.ws

    # ------------------------------------------------------------------------------------
    def _p(self, x: Optional[Tensor]):
        return None if x is None else x.data_ptr()

    def _freeze(self):
        if not self.native and self._grad_flags is None:
            self._grad_flags = [(prm, prm.requires_grad) for prm in self.net.parameters()]
            for prm, _ in self._grad_flags:
                prm.requires_grad_(False)

    def _unfreeze(self):
        if not self.native and self._grad_flags is not None:
            for prm, flag in self._grad_flags:
                prm.requires_grad_(flag)
            self._grad_flags = None

    def setup(self, pc_ori: Tensor, normal_ori: Tensor, gt: Tensor, target: Tensor):
        cfg, t = self.cfg, self.t
        if self.native:   # re-read the folded weights: a cached runner must see a victim whose weights were reloaded
            self.packed = self.net.packed(self.dev)
            if isinstance(self.net, PointNet):
                from .pointnet import ab_flags
                self.packed.struct.flags = ab_flags()
        self.ori = pc_ori.to(self.dev, torch.float32).contiguous()
        self.nrm = normal_ori.to(self.dev, torch.float32).contiguous()
        self.gt = gt.to(self.dev, torch.int32).contiguous()
        self.target = target.to(self.dev, torch.int32).contiguous()
        t["scale_const"].fill_(float(cfg.initial_const))
        t["lower"].zero_()
        t["upper"].fill_(1e10)
        t["best_loss"].fill_(1e10)
        t["best_attack"].fill_(1.0)
        t["best_step"].fill_(-1)
        t["best_bs"].fill_(-1)
        self.kappa_ori = None
        self.nn1_seeded = False
        if self.use_curv:  # _get_kappa_ori once per batch (geoA3_attack.py:216-217)
            _, knn_ori = ops.knn_planar(self.ori, self.ori, self.k + 1)
            self.kappa_ori = ops.kappa(self.ori, self.nrm, knn_ori)
            self.knn_cur = 0
            self.knn_seeded = not self.sub
            if self.knn_seeded:
                t["knn"][0].copy_(knn_ori)
        cls_type = {"None": 0, "CE": 1, "Margin": 2}[cfg.cls_loss_type]
        self.state = AttackState(
            B=self.b, N=self.n, classes=self.classes, targeted=int(self.targeted), cls_loss_type=cls_type,
            confidence=float(_cfg(cfg, "confidence", 0.0)), inv_global_batch=1.0 / float(self.global_batch),
            gt=self._p(self.gt), target=self._p(self.target), scale_const=self._p(t["scale_const"]),
            lower_bound=self._p(t["lower"]), upper_bound=self._p(t["upper"]), best_loss=self._p(t["best_loss"]),
            best_attack=self._p(t["best_attack"]), best_step=self._p(t["best_step"]), best_bs=self._p(t["best_bs"]),
            iter_best_loss=self._p(t["iter_best_loss"]), iter_best_score=self._p(t["iter_best_score"]),
            prev_constrain=self._p(t["prev_constrain"]), label=self._p(t["label"]), cls_loss=self._p(t["cls_loss"]),
            loss_n=self._p(t["loss_n"]), loss_hist=self._p(t["loss_hist"]), last_label=self._p(t["last_label"]))

    # ------------------------------------------------------------------------------------
    def begin_search_step(self, init_offset: Optional[Tensor]):
        t = self.t
        if self.partial:   # the offset is drawn inside the loop (every 50 steps): start from the clean cloud
            init_offset = torch.zeros(self.b, 3, self.n, device=self.dev)
        init = init_offset.to(self.dev, torch.float32).contiguous()
        self._freeze()   # callers that drive step() themselves (bench.py, tests): undone by end_search_step()
        s = torch.cuda.current_stream().cuda_stream
        check(self.lib.geoa3_attack_begin_search_step(C.byref(self.state), self._p(self.ori), self._p(init),
                                                      self._p(t["offset"]), self._p(t["m"]), self._p(t["v"]),
                                                      self._p(t["x"]), s), "begin_search_step")

    def _native_forward(self, pts: Tensor, out: Tensor, s: int):
        fn = self.lib.geoa3_pn2ssg_forward if self.ssg else self.lib.geoa3_pointnet_forward
        st = self.packed.struct
        shape = (int(pts.shape[0]), int(pts.shape[2]))
        # PointNet: a forward of the same shape as the last one on this workspace finds its arg-max keys zero already
        # (GEOA3_PN_KEYS_CLEAN: no clearing launch); the weight table is shared with the module, so the bit is set for
        # this call only
        clean = not self.ssg and self._ws_fwd_shape == shape
        self._ws_fwd_shape = None
        if clean:
            st.flags |= 4
        try:
            check(fn(C.byref(st), pts.data_ptr(), pts.shape[0], pts.shape[2], out.data_ptr(), self.ws.data_ptr(), s),
                  "victim forward")
        finally:
            if clean:
                st.flags &= ~4
        self._ws_fwd_shape = shape

    def _native_backward(self, pts: Tensor, dlogits: Tensor, dx: Tensor, s: int):
        fn = self.lib.geoa3_pn2ssg_backward if self.ssg else self.lib.geoa3_pointnet_backward
        check(fn(C.byref(self.packed.struct), pts.data_ptr(), dlogits.data_ptr(), pts.shape[0], pts.shape[2],
                 dx.data_ptr(), self.ws.data_ptr(), s), "victim backward")

    def _victim_labels_logits(self, pts: Tensor, out: Tensor):
        """logits of the victim on pts [B',3,m] (no gradient kept) -> out [B',classes]."""
        if self.native:
            self._native_forward(pts, out, torch.cuda.current_stream().cuda_stream)
        else:
            with torch.no_grad():
                out.copy_(self.net(pts))

    def step(self, step: int, search_step: int):
        """One inner iteration (geoA3_attack.py:238-352) for all b instances; only enqueues."""
        g = self.objective(step, search_step)
        if g is not None:
            self.optimiser_step(step, g[0], g[1])

    def objective(self, step: int, search_step: int):
        """Success check + `_forward_step` + backward of one iteration (geoA3_attack.py:238-326) on the current
        iterate t["x"]: fills logits / cls_loss / loss_n / the distance losses and returns the two gradient parts
        (g_cls [b,3,n] already scaled by 1/b, g_geo [b,3,n] = d constrain / dx un-scaled; either may be None), or None
        when the step is complete (--is_partial_var keeps its own optimiser)."""
        from . import utility as U
        cfg, t, lib = self.cfg, self.t, self.lib
        s = torch.cuda.current_stream().cuda_stream
        st = C.byref(self.state)
        x = t["x"]                      # the full iterate pc_ori + offset, [b,3,n]
        if self.partial and step % 50 == 0:   # geoA3_attack.py:240-262
            hook = self.hooks.get("partial_points")
            p0 = int(hook(search_step, step)) if hook else int(np.random.randint(self.n))
            _, nbr = ops.knn_planar(self.ori[:, :, p0:p0 + 1].contiguous(), self.ori, self.kr + 1)
            t["pidx"].copy_(nbr[:, 0, 1:])
            hook = self.hooks.get("partial_inits")
            t["part"].copy_(hook(search_step, step) if hook else torch.randn(self.b, 3, self.kr, device=self.dev) * 1e-3)
            t["pm"].zero_()
            t["pv"].zero_()
            t["periodical"].copy_(x)    # the iterate of the last forward (its optimiser step is dropped, as :259-262)
            check(lib.geoa3_attack_partial_step(st, None, None, t["pidx"].data_ptr(), self.kr,
                                                t["periodical"].data_ptr(), t["part"].data_ptr(), None, None,
                                                x.data_ptr(), -1, 0.0, 1.0, 0.0, 0, s), "attack_partial_step")
            self.part_t = 0
        xe, ne = x, self.ne             # what the objective is evaluated on, [b,3,ne]
        vote_logits = None
        if self.sub:
            # geoA3_attack.py:283-284: the objective's sample; :289-295: eval_num resamplings per instance for the
            # success vote.  The reference draws the start indices with torch.randint on its device.
            hook = self.hooks.get("sub_starts")
            start = hook(search_step, step) if hook else torch.randint(self.n, (self.b,), device=self.dev)
            start = start.to(self.dev, torch.int32).contiguous()
            check(lib.geoa3_fps_sample(x.data_ptr(), self.b, self.n, ne, start.data_ptr(), t["sub_idx"].data_ptr(),
                                       t["x_cur"].data_ptr(), s), "fps_sample")
            xe = t["x_cur"]
            E = self.eval_num
            hook = self.hooks.get("vote_starts")
            vstart = hook(search_step, step) if hook else torch.randint(self.n, (self.b, E), device=self.dev)
            vstart = vstart.to(self.dev, torch.int32).reshape(-1).contiguous()
            xrep = x if E == 1 else x.unsqueeze(1).expand(self.b, E, 3, self.n).reshape(self.b * E, 3, self.n)
            check(lib.geoa3_fps_sample(xrep.data_ptr(), self.b * E, self.n, ne, vstart.data_ptr(),
                                       t["vote_idx"].data_ptr(), t["vote_pts"].data_ptr(), s), "fps_sample")
            self._victim_labels_logits(t["vote_pts"], t["vote_logits"])
            vote_logits = t["vote_logits"]
        if self.jitter:
            if not self.sub:   # the success check sees the iterate WITHOUT the jitter (geoA3_attack.py:297 vs :317)
                self._victim_labels_logits(x, t["check_logits"])
                vote_logits = t["check_logits"]
            if step % int(cfg.calculate_project_jitter_noise_iter) == 0:
                hook = self.hooks.get("jitter_noise")
                if hook:
                    t["noise"].copy_(hook(search_step, step, xe))
                else:
                    aux = self.hooks["jitter_aux"](search_step, step) if "jitter_aux" in self.hooks else None
                    t["noise"].copy_(U.estimate_perpendicular(xe, int(cfg.jitter_k), float(cfg.jitter_sigma),
                                                              float(cfg.jitter_clip), aux=aux))
            torch.add(xe, t["noise"], out=t["x_eval"])
            xe = t["x_eval"]
        logits_ag = x_leaf = None
        main = torch.cuda.current_stream()
        sg, geo_ctx = s, None
        if self.geo_stream is not None:     # xe is final on the main stream: the geometry may start
            self.ev_x.record(main)
            self.geo_stream.wait_event(self.ev_x)
            sg = self.geo_stream.cuda_stream
        if self.native:
            self._native_forward(xe, t["logits"], s)
        else:
            x_leaf = xe.detach().clone().requires_grad_()
            with torch.enable_grad():
                logits_ag = self.net(x_leaf)
            if logits_ag.shape != t["logits"].shape:
                raise _lib.Geoa3Error("the victim returned logits of shape %s, expected %s (set cfg.classes)"
                                      % (tuple(logits_ag.shape), tuple(t["logits"].shape)))
            t["logits"].copy_(logits_ag.detach())
        with (torch.cuda.stream(self.geo_stream) if self.geo_stream is not None else contextlib.nullcontext()):
            constrain = None
            if self.need_nn:
                both = self.dis_type == 1 and not cfg.is_cd_single_side
                if self.grid_nn1:   # seeded with the tables of the previous iteration (in place)
                    pa = t["i_ao"].data_ptr() if self.nn1_seeded else None
                    pr = t["i_oa"].data_ptr() if self.nn1_seeded and both else None
                    out = (t["d_ao"].data_ptr(), t["i_ao"].data_ptr(), t["d_oa"].data_ptr() if both else None,
                           t["i_oa"].data_ptr() if both else None)
                    if self.nn1_policy is None:
                        check(lib.geoa3_grid_nn1_pair(xe.data_ptr(), self.ori.data_ptr(), self.b, ne, self.n, pa, pr, *out,
                                                      sg), "grid_nn1_pair")
                    else:
                        check(lib.geoa3_debug_grid_nn1_pair(xe.data_ptr(), self.ori.data_ptr(), self.b, ne, self.n, pa, pr,
                                                            *out, *self.nn1_policy, sg), "grid_nn1_pair")
                    self.nn1_seeded = not self.sub
                else:
                    check(lib.geoa3_nn1_pair(xe.data_ptr(), self.ori.data_ptr(), self.b, ne, self.n,
                                             t["d_ao"].data_ptr(), t["i_ao"].data_ptr(),
                                             t["d_oa"].data_ptr() if both else None,
                                             t["i_oa"].data_ptr() if both else None, sg), "nn1_pair")
            knn_adv = None
            if self.use_curv:
                prior, out = t["knn"][self.knn_cur], t["knn"][1 - self.knn_cur]
                check(lib.geoa3_knn_self(xe.data_ptr(), self.b, ne, self.k + 1,
                                         prior.data_ptr() if self.knn_seeded else None, t["knn_d"].data_ptr(),
                                         out.data_ptr(), t["knn_scratch"].data_ptr() if self.knn_slab else None,
                                         self.knn_method, sg),
                      "knn_self")
                self.knn_seeded = True
                self.knn_cur = 1 - self.knn_cur
                knn_adv = out
            if self.dis_type != 0 or cfg.hd_loss_weight != 0 or self.use_curv:
                ops.geo_loss_grad(xe, self.ori, normal_ori=self.nrm if self.use_curv else None,
                                  kappa_ori=self.kappa_ori, d_ao=t["d_ao"] if self.need_nn else None,
                                  i_ao=t["i_ao"] if self.need_nn else None,
                                  d_oa=t["d_oa"] if self.dis_type == 1 and not cfg.is_cd_single_side else None,
                                  i_oa=t["i_oa"] if self.dis_type == 1 and not cfg.is_cd_single_side else None,
                                  knn_adv=knn_adv, k=self.k if self.use_curv else 0, dis_type=self.dis_type,
                                  single_side=bool(cfg.is_cd_single_side), w_dis=float(cfg.dis_loss_weight),
                                  w_hd=float(cfg.hd_loss_weight), w_curv=float(cfg.curv_loss_weight), out=self.geo_out,
                                  deterministic=self.deterministic, scratch=self.geo_scratch)
                constrain = self.geo_out["constrain"]
        late = self.late_join and self.geo_stream is not None
        if self.geo_stream is not None:     # join: the bookkeeping needs the constrain loss, the update the gradient
            self.ev_geo.record(self.geo_stream)
            if not late:
                main.wait_event(self.ev_geo)
        if late:   # the classification half of the head now, the victim's backward beside the geometry kernels
            check(lib.geoa3_attack_head_classify(st, t["logits"].data_ptr(), self._p(vote_logits),
                                                 self.eval_num if self.sub else 1, t["dlogits"].data_ptr(),
                                                 t["ok"].data_ptr(), s), "attack_head_classify")
        else:
            check(lib.geoa3_attack_head_vote(st, t["logits"].data_ptr(), self._p(vote_logits),
                                             self.eval_num if self.sub else 1,
                                             self._p(constrain), x.data_ptr(), step, search_step,
                                             t["dlogits"].data_ptr(), s), "attack_head")
        g_cls = None
        if cfg.cls_loss_type != "None":
            if self.native:
                self._native_backward(xe, t["dlogits"], t["g_cls"], s)
            else:
                logits_ag.backward(t["dlogits"])
                t["g_cls"].copy_(x_leaf.grad)
            g_cls = t["g_cls"]
        if late:
            main.wait_event(self.ev_geo)
            check(lib.geoa3_attack_head_finish(st, t["ok"].data_ptr(), self._p(constrain), x.data_ptr(), step,
                                               search_step, s), "attack_head_finish")
        g_geo = t["g_geo"] if constrain is not None else None
        if self.sub:   # torch.gather's backward (Lib/utility.py:185): scatter the sample's gradient to the full cloud
            for src, dst in ((g_cls, "g_cls_full"), (g_geo, "g_geo_full")):
                if src is not None:
                    check(lib.geoa3_pn2_gather_points_grad(src.data_ptr(), t["sub_idx"].data_ptr(), self.b, 3, self.n,
                                                           ne, t[dst].data_ptr(), s), "gather_points_grad")
            g_cls = t["g_cls_full"] if g_cls is not None else None
            g_geo = t["g_geo_full"] if g_geo is not None else None
        if self.partial:
            self.part_t += 1
            if (step + 1) % 50 != 0:    # the step before a re-draw is never used (periodical_pc = input_all, :260)
                lr = cfg.lr * (0.9990 ** (self.part_t - 1) if _cfg(cfg, "is_use_lr_scheduler", False) else 1.0)
                if cfg.optim == "adam":
                    args = (0, lr / (1.0 - 0.9 ** self.part_t), math.sqrt(1.0 - 0.999 ** self.part_t), 0.0, 0)
                else:
                    args = (1, lr, 1.0, 0.9, int(self.part_t == 1))
                check(lib.geoa3_attack_partial_step(st, self._p(g_cls), self._p(g_geo), t["pidx"].data_ptr(), self.kr,
                                                    t["periodical"].data_ptr(), t["part"].data_ptr(),
                                                    t["pm"].data_ptr(), t["pv"].data_ptr(), x.data_ptr(), *args, s),
                      "attack_partial_step")
            return None  # the projections / lp_clip act on a padded copy in the reference: no effect (:341-352)
        return g_cls, g_geo

    def optimiser_step(self, step: int, g_cls: Optional[Tensor], g_geo: Optional[Tensor]):
        """Adam / SGD on the offset with g = g_cls + (scale_const / b) g_geo, then the flag-gated projections
        (geoA3_attack.py:326-352)."""
        cfg, t, lib = self.cfg, self.t, self.lib
        s = torch.cuda.current_stream().cuda_stream
        st = C.byref(self.state)
        x = t["x"]
        pro_grad = bool(_cfg(cfg, "is_pro_grad", False))
        # optimiser scalars in double, as torch.optim.Adam forms them
        lr = cfg.lr * (0.9990 ** step if _cfg(cfg, "is_use_lr_scheduler", False) else 1.0)
        if cfg.optim == "adam":
            tt = step + 1
            step_size, sqrt_bc2, optim = lr / (1.0 - 0.9 ** tt), math.sqrt(1.0 - 0.999 ** tt), 0
        else:
            step_size, sqrt_bc2, optim = lr, 1.0, 1
        check(lib.geoa3_attack_update(st, self._p(g_cls), self._p(g_geo),
                                      self.ori.data_ptr(), t["offset"].data_ptr(), t["m"].data_ptr(),
                                      t["v"].data_ptr(), x.data_ptr(), optim, step_size, sqrt_bc2,
                                      0.0 if pro_grad else float(_cfg(cfg, "cc_linf", 0.0)), s), "attack_update")
        if pro_grad:   # geoA3_attack.py:341-352: (real offset,) projection onto the normal, then lp_clip
            pd, pi = t["proj_d"], t["proj_i"]
            if _cfg(cfg, "is_real_offset", False):
                check(lib.geoa3_nn1_pair(x.data_ptr(), self.ori.data_ptr(), self.b, self.n, self.n,
                                         pd.data_ptr(), pi.data_ptr(), None, None, s), "nn1_pair")
                check(lib.geoa3_attack_project(0, self.ori.data_ptr(), None, pi.data_ptr(),
                                               t["offset"].data_ptr(), x.data_ptr(), self.b, self.n, 0.0, s),
                      "attack_project")
            check(lib.geoa3_nn1_pair(t["offset"].data_ptr(), self.ori.data_ptr(), self.b, self.n, self.n,
                                     pd.data_ptr(), pi.data_ptr(), None, None, s), "nn1_pair")
            check(lib.geoa3_attack_project(1, self.ori.data_ptr(), self.nrm.data_ptr(), pi.data_ptr(),
                                           t["offset"].data_ptr(), x.data_ptr(), self.b, self.n,
                                           float(_cfg(cfg, "cc_linf", 0.0)), s), "attack_project")

    def end_search_step(self, sync_last_label: Optional[Callable[[Tensor], None]] = None):
        if sync_last_label is not None:
            sync_last_label(self.t["last_label"])
        s = torch.cuda.current_stream().cuda_stream
        check(self.lib.geoa3_attack_binary_update(C.byref(self.state), s), "binary_update")
        self._unfreeze()

    def info_line(self, search_step, step, i, loader_len) -> str:
        """The reference's progress line (geoA3_attack.py:129,138,154,163,365); the ONLY host sync in the loop."""
        cfg, t = self.cfg, self.t
        vals = torch.stack([t["loss_n"].mean() * (self.b / float(self.global_batch)), t["cls_loss"].mean(),
                            self.geo_out["dis_loss"].mean(), self.geo_out["hd_loss"].mean(),
                            self.geo_out["curv_loss"].mean()]).tolist()
        info = "[{5}/{6}][{0}/{1}][{2}/{3}] \t loss: {4:6.4f}\t".format(
            search_step + 1, cfg.binary_max_steps, step + 1, cfg.iter_max_steps, vals[0], i, loader_len)
        info += "cls_loss: {0:6.4f}\t".format(vals[1])
        if cfg.dis_loss_type == "CD":
            info += "cd_loss: {0:6.4f}\t".format(vals[2])
        elif cfg.dis_loss_type == "L2":
            info += "l2_loss: {0:6.4f}\t".format(vals[2])
        if cfg.hd_loss_weight != 0:
            info += "hd_loss : {0:6.4f}\t".format(vals[3])
        if cfg.curv_loss_weight != 0:
            info += "curv_loss : {0:6.4f}\t".format(vals[4])
        return info

    def run(self, init_offsets: Optional[Sequence[Tensor]] = None, i: int = 0, loader_len: int = 1,
            verbose: bool = False, sync_last_label: Optional[Callable[[Tensor], None]] = None,
            on_step: Optional[Callable[[int, int], None]] = None, hooks: Optional[dict] = None):
        """hooks (parity runs): the reference's random draws as inputs -- sub_starts(search_step, step) -> [b],
        vote_starts(search_step, step) -> [b, eval_num] (torch.randint of farthest_points_sample),
        jitter_aux(search_step, step) -> (aux1, aux2) [b,ne] or jitter_noise(search_step, step, x) -> [b,3,ne];
        partial_points(search_step, step) -> int (np.random.randint) and partial_inits(search_step, step) ->
        [b,3,knn_range] (nn.init.normal_) for --is_partial_var."""
        cfg = self.cfg
        self.hooks = dict(hooks or {})
        self._freeze()
        try:
            self._run(init_offsets, i, loader_len, verbose, sync_last_label, on_step)
        finally:
            self._unfreeze()

    def _run(self, init_offsets, i, loader_len, verbose, sync_last_label, on_step):
        cfg = self.cfg
        for search_step in range(int(cfg.binary_max_steps)):
            if self.partial:
                init = None
            elif init_offsets is not None:
                init = init_offsets[search_step]
            else:  # nn.init.normal_(offset, mean=0, std=1e-3), geoA3_attack.py:264-266
                init = torch.randn(self.b, 3, self.n, device=self.dev) * 1e-3
            self.begin_search_step(init)
            for step in range(self.iters):
                self.step(step, search_step)
                if on_step is not None:
                    on_step(search_step, step)
                if verbose and (step % 50 == 0 or step == self.iters - 1):
                    print(self.info_line(search_step, step, i, loader_len))
            self.end_search_step(sync_last_label)

    def results(self):
        """-> the reference 5-tuple (geoA3_attack.py:386)."""
        t = self.t
        success = (t["best_loss"].cpu().numpy() < 1e10)
        best_step = t["best_step"].cpu().tolist()
        all_loss = t["loss_hist"].cpu().tolist()
        # a fresh tensor per call, as the reference allocates one (geoA3_attack.py:226): a cached runner refills its
        # own buffer at the next setup()
        return t["best_attack"].clone(), self.target.long(), success, best_step, all_loss


# cfg fields that shape an AttackRunner's buffers / kernel choices at construction
RUNNER_CFG_FIELDS = ("attack_label", "iter_max_steps", "curv_loss_knn", "curv_loss_weight", "dis_loss_type",
                     "hd_loss_weight", "cls_loss_type", "optim", "uniform_loss_weight", "npoint", "is_partial_var",
                     "knn_range", "is_subsample_opt", "eval_num", "is_pre_jitter_input", "is_pro_grad",
                     "brute_force_nn1", "classes", "deterministic", "late_join", "knn_method")


def unpack_input(input_data, targeted: bool):
    """The DataLoader batch of main_attack.py -> ([b,3,n] pc, [b,3,n] normal, gt [b], target [b])
    (geoA3_attack.py:196-214)."""
    pc, normal, gt_labels = input_data[0], input_data[1], input_data[2]
    if pc.size(3) == 3:
        pc = pc.permute(0, 1, 3, 2)
    if normal.size(3) == 3:
        normal = normal.permute(0, 1, 3, 2)
    bs, l, _, n = pc.size()
    b = bs * l
    pc_ori = pc.reshape(b, 3, n)
    normal_ori = normal.reshape(b, 3, n)
    gt = gt_labels.reshape(-1)
    target = input_data[3].reshape(-1) if targeted else gt
    return pc_ori, normal_ori, gt, target


def attack(net, input_data, cfg, i, loader_len, saved_dir=None, *, init_offsets=None, verbose=True,
           global_batch=None, sync_last_label=None, runner_cache: Optional[dict] = None,
           hooks: Optional[dict] = None):
    """Drop-in for geoA3_attack.attack (same positional arguments, same 5-tuple).  Keyword-only extras:
    init_offsets (list of [b,3,n] step-0 offsets, one per binary step, for reproducible parity runs),
    global_batch / sync_last_label (set by geoa3_amd.distributed when the batch is sharded)."""
    targeted = cfg.attack_label != "Untarget"
    pc_ori, normal_ori, gt, target = unpack_input(input_data, targeted)
    b, _, n = pc_ori.shape
    device = next(net.parameters()).device
    if device.type != "cuda":
        raise _lib.Geoa3Error("attack() needs the victim network on the GPU")
    # everything AttackRunner.__init__ derives from cfg is part of the key: a changed flag builds a new runner
    key = (b, n, id(net), global_batch) + tuple(
        (name, repr(getattr(cfg, name, None))) for name in RUNNER_CFG_FIELDS)
    runner = runner_cache.get(key) if runner_cache is not None else None
    if runner is None:
        runner = AttackRunner(net, b, n, cfg, device, global_batch)
        if runner_cache is not None:
            runner_cache[key] = runner
    runner.cfg = cfg
    runner.setup(pc_ori, normal_ori, gt, target)
    runner.run(init_offsets, i, loader_len, verbose, sync_last_label, hooks=hooks)
    return runner.results()


def attack_sharded(net, input_data, cfg, i, loader_len, saved_dir=None, *, init_offsets=None, verbose=False,
                   group=None):
    """attack() with the batch sharded by instance over the ranks of an initialised process group (one process
    per GPU, RCCL).  Every rank passes the SAME full batch and receives the full 5-tuple; a rank computes only
    its own block of instances (geoa3_amd.distributed)."""
    import torch.distributed as dist
    from .distributed import sharded_attack
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return attack(net, input_data, cfg, i, loader_len, saved_dir, init_offsets=init_offsets, verbose=verbose)
    targeted = cfg.attack_label != "Untarget"
    pc_ori, normal_ori, gt, target = unpack_input(input_data, targeted)
    b, _, n = pc_ori.shape
    device = next(net.parameters()).device
    if init_offsets is None:   # one shared draw so every rank sees the same global initial offsets
        g = torch.Generator(device="cpu").manual_seed(int(getattr(cfg, "id", 0)) * 1000003 + int(i))
        init_offsets = [torch.randn(b, 3, n, generator=g) * 1e-3 for _ in range(int(cfg.binary_max_steps))]

    def run_shard(pc, normal, g_, t_, inits, global_batch, sync):
        bl = pc.shape[0]
        if bl == 0:   # more ranks than instances: still take part in the per-binary-step broadcasts
            dummy = torch.zeros(1, dtype=torch.int32, device=device)
            for _ in range(int(cfg.binary_max_steps)):
                sync(dummy)
            return (torch.zeros(0, 3, n, device=device), t_.to(device).long(), np.zeros(0, bool), [],
                    [[] for _ in range(int(cfg.iter_max_steps))])
        runner = AttackRunner(net, bl, n, cfg, device, global_batch)
        runner.setup(pc, normal, g_, t_)
        runner.run([o.to(device) for o in inits], i, loader_len, verbose and dist.get_rank(group) == 0, sync)
        return runner.results()

    return sharded_attack(run_shard, pc_ori, normal_ori, gt, target, init_offsets, group)
