"""Instance sharding of one attack batch over the GPUs of a node: one process per GPU,
`torch.distributed` (backend "nccl" = RCCL over xGMI on the GPU box; "gloo" in the CPU tests).

Every quantity of the loop is per instance, so there is NO collective in the per-iteration data path.
Two couplings of the reference are kept exact:
  1. `loss = loss_n.mean()` (Attacker/geoA3_attack.py:178): each shard divides by the GLOBAL batch size;
  2. the binary-search `output_label` quirk (geoA3_attack.py:298,375): the rank that owns the global last
     instance broadcasts ONE int32 per binary step.
Results are all-gathered once per batch (best_attack shards, success, best step, loss history).

The per-shard computation is injected (`run_shard`) so that this host logic is testable on CPU with gloo.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(b: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous blocks of ceil/floor(b/world) instances; the remainder goes to the first ranks."""
    base, rem = divmod(b, world)
    out, lo = [], 0
    for r in range(world):
        hi = lo + base + (1 if r < rem else 0)
        out.append((lo, hi))
        lo = hi
    return out


def owner_of_last_instance(b: int, world: int) -> int:
    bounds = shard_bounds(b, world)
    return max(r for r, (lo, hi) in enumerate(bounds) if hi > lo and hi == b)


def make_last_label_sync(b_global: int, group=None) -> Callable[[torch.Tensor], None]:
    """-> sync(t): t is the local [1] int32 `last_label`; after the call every rank holds the label of the
    GLOBAL last instance (a 4-byte broadcast, once per binary step)."""
    world = dist.get_world_size(group)
    src = owner_of_last_instance(b_global, world)

    def sync(t: torch.Tensor) -> None:
        if t.is_cuda and _host_staged(group):
            h = t.cpu()
            dist.broadcast(h, src=src, group=group)
            t.copy_(h)
        else:
            dist.broadcast(t, src=src, group=group)

    return sync


def _host_staged(group=None) -> bool:
    """gloo carries device tensors only for broadcast / all_reduce: under it (functional runs of the N > 1 path on one
    GPU, CPU tests) the few KB-MB of per-batch results are staged through the host; RCCL moves device buffers."""
    return dist.get_backend(group) == "gloo"


def _gather_rows(x: torch.Tensor, counts: Sequence[int], group=None) -> torch.Tensor:
    """all_gather of row blocks with different row counts (pads to the largest block)."""
    world = len(counts)
    mx = max(counts)
    dev = x.device
    if x.is_cuda and _host_staged(group):
        x = x.cpu()
    pad = torch.zeros((mx,) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    pad[: x.shape[0]] = x
    bufs = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(bufs, pad, group=group)
    return torch.cat([bufs[r][: counts[r]] for r in range(world)], dim=0).to(dev)


def sharded_attack(run_shard: Callable, pc: torch.Tensor, normal: torch.Tensor, gt: torch.Tensor,
                   target: torch.Tensor, init_offsets: Sequence[torch.Tensor], group=None):
    """Run one batch of b attacks sharded over the process group.

    run_shard(pc, normal, gt, target, init_offsets, global_batch, sync_last_label)
        -> (best_attack [bl,3,n], target [bl], success bool[bl], best_step list[bl], all_loss [iters][bl])
    is the single-device attack (geoa3_amd.attack.attack on a GPU; the oracle in the CPU tests).
    Returns the reference 5-tuple for the WHOLE batch on every rank."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    b = pc.shape[0]
    bounds = shard_bounds(b, world)
    lo, hi = bounds[rank]
    counts = [h - l for l, h in bounds]
    sync = make_last_label_sync(b, group)
    sl = slice(lo, hi)
    best, tgt, succ, step, loss = run_shard(pc[sl], normal[sl], gt[sl], target[sl],
                                            [o[sl] for o in init_offsets], b, sync)
    dev = best.device
    best_all = _gather_rows(best.contiguous(), counts, group)
    tgt_all = _gather_rows(tgt.contiguous(), counts, group)
    succ_all = _gather_rows(torch.as_tensor(np.asarray(succ), device=dev).to(torch.uint8), counts, group)
    step_all = _gather_rows(torch.as_tensor(step, device=dev, dtype=torch.int64), counts, group)
    loss_t = torch.as_tensor(np.asarray(loss, dtype=np.float32), device=dev).t().contiguous()   # [bl, iters]
    loss_all = _gather_rows(loss_t, counts, group).t()
    return (best_all, tgt_all, succ_all.cpu().numpy().astype(bool), step_all.cpu().tolist(), loss_all.cpu().tolist())


def gather_results_timed(best: torch.Tensor, succ: torch.Tensor, step: torch.Tensor, loss: torch.Tensor,
                         counts: Sequence[int], repeats: int = 5, sync: Callable[[], None] = None, group=None) -> dict:
    """The per-batch result gather of `sharded_attack` on tensors of the caller's shapes (best_attack [bl,3,n], success
    [bl] uint8, best step [bl] int64, loss history [bl,iters]), timed: one untimed pass, then `repeats` passes between two
    barriers.  -> {"ms": average per gather (max over ranks), "bytes": payload of the whole batch, "ranks": world}.
    bench.py prints it in the rank-0 line of an N-rank run (the one collective of the path besides the 4-byte label
    broadcast per binary step)."""
    import time
    world = dist.get_world_size(group)
    sync = sync or (lambda: None)

    def once():
        return [_gather_rows(x.contiguous(), counts, group) for x in (best, succ, step, loss)]

    outs = once()
    sync()
    dist.barrier(group)
    t0 = time.perf_counter()
    for _ in range(repeats):
        once()
    sync()
    dist.barrier(group)
    dt = (time.perf_counter() - t0) / repeats
    t = torch.tensor([dt], dtype=torch.float64, device=best.device if not _host_staged(group) else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    nbytes = sum(int(o.numel()) * o.element_size() for o in outs)
    return {"ms": float(t.item()) * 1e3, "bytes": nbytes, "ranks": world, "rows": [int(o.shape[0]) for o in outs][0],
            "GBps": nbytes / max(float(t.item()), 1e-12) / 1e9}


def rank_roll_call(device, group=None) -> dict:
    """All-gather of one int32 per rank holding the rank id (a DEVICE tensor under RCCL): what came back is the proof
    that the backend connected `world` distinct ranks.  -> {"world", "ranks": [...], "ok", "backend"}."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    mine = torch.tensor([rank], dtype=torch.int32, device=device)
    got = [torch.full_like(mine, -1) for _ in range(world)]
    dist.all_gather(got, mine, group=group)
    ranks = [int(g.item()) for g in got]
    return {"world": world, "ranks": ranks, "ok": ranks == list(range(world)), "backend": dist.get_backend(group),
            "device": str(mine.device)}
