"""CPU, world_size 2, gloo: the sharding host logic (global loss divisor, last-label broadcast, ragged gather)
reproduces the un-sharded run.  The per-shard computation is the ORACLE here (the HIP path needs a GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from geoa3_amd.distributed import (gather_results_timed, owner_of_last_instance, rank_roll_call, shard_bounds,
                                   sharded_attack)
from oracle import geoa3_oracle as O


def test_shard_bounds():
    assert shard_bounds(250, 8) == [(0, 32), (32, 64), (64, 95), (95, 126), (126, 157), (157, 188), (188, 219),
                                    (219, 250)]
    assert shard_bounds(5, 2) == [(0, 3), (3, 5)]
    assert shard_bounds(1, 2) == [(0, 1), (1, 1)] and owner_of_last_instance(1, 2) == 0
    assert owner_of_last_instance(250, 8) == 7


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup_case():
    cfg = O.AttackCfg(curv_loss_knn=4, binary_max_steps=3, iter_max_steps=5, lr=0.002, initial_const=2000.0)
    sd = O.make_pointnet_state_dict(40, seed=0)
    ori, nrm = O.make_synthetic_clouds(5, 64, seed=35)
    g = torch.Generator().manual_seed(5)
    inits = [torch.randn(5, 3, 64, generator=g) * 1e-3 for _ in range(3)]
    return cfg, sd, ori, nrm, inits


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg, sd, ori, nrm, inits = _setup_case()
    net = lambda x: O.pointnet_forward(sd, x)
    with torch.no_grad():
        gt = net(ori).argmax(1)

    def run_shard(pc, normal, g, t, init, global_batch, sync):
        calls = {"s": 0}

        def hook(local_label):
            tt = torch.tensor([local_label], dtype=torch.int32)
            sync(tt)
            return int(tt.item())

        if pc.shape[0] == 0:   # an empty shard still takes part in the broadcasts
            for _ in range(cfg.binary_max_steps):
                hook(-1)
            z = torch.zeros(0, 3, ori.shape[2])
            return z, g, np.zeros(0, bool), [], [[] for _ in range(cfg.iter_max_steps)]
        return O.attack(net, pc, normal, g, None, cfg, init, loss_divisor=global_batch, last_label_hook=hook)

    out = sharded_attack(run_shard, ori, nrm, gt, gt, inits)
    # the evidence bench.py prints in an N-rank run: the timed result gather of a batch and the roll call of the rank ids
    counts = [h - l for l, h in shard_bounds(250, world)]
    bl = counts[rank]
    gather = gather_results_timed(torch.full((bl, 3, 64), float(rank)), torch.ones(bl, dtype=torch.uint8),
                                  torch.zeros(bl, dtype=torch.int64), torch.zeros(bl, 50), counts, repeats=2)
    roll = rank_roll_call(torch.device("cpu"))
    if rank == 0:
        q.put((out[0].numpy(), out[1].numpy(), out[2], out[3], np.asarray(out[4], dtype=np.float32), gather, roll))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_rank_sharded_attack_equals_full_batch():
    cfg, sd, ori, nrm, inits = _setup_case()
    net = lambda x: O.pointnet_forward(sd, x)
    with torch.no_grad():
        gt = net(ori).argmax(1)
    full = O.attack(net, ori, nrm, gt, None, cfg, inits)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    best, tgt, succ, step, loss, gather, roll = q.get(timeout=500)
    assert roll == {"world": 2, "ranks": [0, 1], "ok": True, "backend": "gloo", "device": "cpu"}
    assert gather["ranks"] == 2 and gather["rows"] == 250 and gather["ms"] > 0
    assert gather["bytes"] == 250 * (3 * 64 * 4 + 1 + 8 + 50 * 4)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_allclose(best, full[0].numpy(), rtol=0, atol=1e-5)
    assert (tgt == full[1].numpy()).all()
    assert (succ == full[2]).all() and list(step) == list(full[3])
    np.testing.assert_allclose(loss, np.asarray(full[4], dtype=np.float32), rtol=1e-4, atol=1e-5)  # CPU GEMMs depend on batch size
