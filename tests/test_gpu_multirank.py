"""GPU, world_size 2 on ONE device (gloo carries the collectives): the N > 1 path end to end through the HIP library --
bench.py's strong-scaling mode (configs[2]: the 250-instance batch split by shard_bounds) and attack_sharded() against
the un-sharded attack() on the same inputs (shard rows bit-identical, one int32 broadcast per binary step,
one all-gather of the results)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _env():
    env = dict(os.environ)
    env.update(GEOA3_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
               PYTHONPATH=REPO + os.pathsep + env.get("PYTHONPATH", ""))
    return env


@pytest.mark.timeout(900)
def test_bench_two_ranks_strong_scaling_on_one_gpu():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps",
           "6", "--warmup", "2", "--presteps", "4", "--single-mode", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=_env(), cwd=REPO, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]       # rank 0 prints ONE line
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["steps"] == 6
    assert out["config"]["global_batch"] == 250 and out["config"]["instances_per_gpu"] == 125
    assert "configs[2]" in out["config"]["workload"]
    # value = iterations/s of the WHOLE 250-instance batch = steps / seconds in strong mode
    assert abs(out["value"] - 1e3 / out["ms_per_step"]) < 1e-2 * out["value"]
    assert out["roofline"]["frac"] > 0 and out["roofline"]["executed_frac"] == pytest.approx(3 * out["roofline"]["frac"], rel=1e-2)
    # the N-rank evidence: both scaling modes, per-rank step times, the timed result gather, the roll call of rank ids
    mg = out["multi_gpu"]
    assert mg["headline_scaling"] == "strong" and mg["other_scaling"]["scaling"] == "weak"
    assert mg["other_scaling"]["instances_per_gpu"] == 250 and mg["other_scaling"]["global_batch"] == 500
    assert mg["other_scaling"]["value"] > 0 and mg["other_scaling"]["ms_per_step"] > out["ms_per_step"]
    pr = mg["ms_per_step_per_rank"]
    assert len(pr["all"]) == 2 and 0 < pr["min"] <= pr["max"] <= out["ms_per_step"] * 1.001
    assert mg["rank_roll_call"]["ranks"] == [0, 1] and mg["rank_roll_call"]["ok"]
    assert mg["result_gather"]["rows"] == 250 and mg["result_gather"]["ranks"] == 2 and mg["result_gather"]["ms"] > 0
    assert mg["result_gather"]["bytes"] == 250 * (3 * 1024 * 4 + 1 + 8 + 500 * 4)
    assert "other_configs" not in out              # only the plain 1-GPU default command appends them


@pytest.mark.timeout(900)
def test_bench_spawns_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (the way the driver calls `--gpus 1`): the parent must start the two
    ranks itself (a child torch.distributed.run, before anything touches the GPU) and rank 0's line must say so."""
    env = _env()
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--presteps",
           "2", "--single-mode", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=REPO, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["backend"] == "gloo"
    assert out["config"]["instances_per_gpu"] == 125 and out["scaling"] == "strong"


@pytest.mark.timeout(900)
def test_bench_one_rank_over_rccl():
    """The nccl (= RCCL) path on the one GPU a test box has: under a launcher bench.py creates the process group for a
    single rank too -- RCCL initialisation with device_id, the barriers and the device-tensor all-reduce of the timing run
    for real (several ranks on one device are refused by RCCL, so gloo carries the 2-rank tests above)."""
    env = _env()
    env.pop("GEOA3_BENCH_BACKEND")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps",
           "4", "--warmup", "1", "--presteps", "2", "--single-mode", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=REPO, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith('{"metric"')][0])
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1 and out["backend"].startswith("rccl")
    assert out["config"]["instances_per_gpu"] == 250


_RCCL_WORKER = r"""
import os, sys, pickle
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r})
from oracle import geoa3_oracle as O
from geoa3_amd.attack import AttackRunner, attack, unpack_input
from geoa3_amd.distributed import sharded_attack
from geoa3_amd.pointnet import PointNet
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", device_id=dev)
net = PointNet(40); net.load_state_dict(O.make_pointnet_state_dict(40, seed=0)); net = net.cuda().eval()
cfg = O.AttackCfg(curv_loss_knn=8, binary_max_steps=2, iter_max_steps=5, lr=0.005, initial_const=500.0)
ori, nrm = O.make_synthetic_clouds(5, 128, seed=43)
with torch.no_grad():
    gt = net(ori.cuda()).argmax(1).cpu()
g = torch.Generator().manual_seed(6)
inits = [torch.randn(5, 3, 128, generator=g) * 1e-3 for _ in range(2)]
data = [ori.permute(0, 2, 1).unsqueeze(1).contiguous(), nrm.permute(0, 2, 1).unsqueeze(1).contiguous(), gt.view(5, 1)]
pc, nm, g_, t_ = unpack_input(data, False)

def run_shard(pc, normal, gt_, tgt_, inits_, global_batch, sync):
    r = AttackRunner(net, pc.shape[0], pc.shape[2], cfg, dev, global_batch)
    r.setup(pc, normal, gt_, tgt_)
    r.run([o.to(dev) for o in inits_], sync_last_label=sync)
    return r.results()

out = sharded_attack(run_shard, pc, nm, g_, t_, inits)      # broadcast + all_gather of DEVICE tensors through RCCL
full = attack(net, data, cfg, 0, 1, init_offsets=[i.cuda() for i in inits], verbose=False)
res = dict(backend=dist.get_backend(), bit_equal=bool(torch.equal(out[0].cpu(), full[0].cpu())),
           succ=bool((np.asarray(out[2]) == np.asarray(full[2])).all()), steps=list(out[3]) == list(full[3]),
           loss=float(np.abs(np.asarray(out[4], dtype=np.float64) - np.asarray(full[4], dtype=np.float64)).max()))
pickle.dump(res, open({out!r}, "wb"))
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.timeout(900)
def test_sharded_attack_collectives_over_rccl_one_rank(tmp_path):
    """geoa3_amd.distributed on the nccl backend (one rank): the last-label broadcast and the ragged all-gathers move
    device tensors through RCCL and return the single-process result bit for bit."""
    script, outp = tmp_path / "worker.py", tmp_path / "res.pkl"
    script.write_text(_RCCL_WORKER.format(repo=REPO, out=str(outp)))
    env = _env()
    env.pop("GEOA3_BENCH_BACKEND")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=REPO, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    import pickle
    res = pickle.load(open(outp, "rb"))
    assert res["backend"] == "nccl" and res["bit_equal"] and res["succ"] and res["steps"] and res["loss"] == 0.0, res


_WORKER = r"""
import os, sys, pickle
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, {repo!r})
from oracle import geoa3_oracle as O
from geoa3_amd.attack import attack, attack_sharded
from geoa3_amd.pointnet import PointNet
dist.init_process_group("gloo")
rank = dist.get_rank()
torch.cuda.set_device(0)
net = PointNet(40); net.load_state_dict(O.make_pointnet_state_dict(40, seed=0)); net = net.cuda().eval()
cfg = O.AttackCfg(curv_loss_knn=8, binary_max_steps=3, iter_max_steps=6, lr=0.005, initial_const=500.0)
ori, nrm = O.make_synthetic_clouds(7, 128, seed=41)
with torch.no_grad():
    gt = net(ori.cuda()).argmax(1).cpu()
g = torch.Generator().manual_seed(6)
inits = [torch.randn(7, 3, 128, generator=g) * 1e-3 for _ in range(3)]
data = [ori.permute(0, 2, 1).unsqueeze(1).contiguous(), nrm.permute(0, 2, 1).unsqueeze(1).contiguous(), gt.view(7, 1)]
out = attack_sharded(net, data, cfg, 0, 1, init_offsets=inits)
if rank == 0:
    full = attack(net, data, cfg, 0, 1, init_offsets=[i.cuda() for i in inits], verbose=False)
    res = dict(bit_equal=bool(torch.equal(out[0].cpu(), full[0].cpu())), succ=(np.asarray(out[2]) == np.asarray(full[2])).all(),
               steps=list(out[3]) == list(full[3]), tgt=bool(torch.equal(out[1].cpu(), full[1].cpu())),
               loss=float(np.abs(np.asarray(out[4], dtype=np.float64) - np.asarray(full[4], dtype=np.float64)).max()),
               nsucc=int(np.asarray(full[2]).sum()), shape=tuple(out[0].shape))
    pickle.dump(res, open({out!r}, "wb"))
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.timeout(900)
def test_attack_sharded_two_ranks_equals_full_batch(tmp_path):
    """7 instances over 2 ranks (4 + 3: a ragged split), 3 binary steps: the gathered 5-tuple equals the single-process
    attack() bit for bit (global loss divisor, last-label broadcast, ragged all-gather)."""
    script, outp = tmp_path / "worker.py", tmp_path / "res.pkl"
    script.write_text(_WORKER.format(repo=REPO, out=str(outp)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), str(script)]
    r = subprocess.run(cmd, capture_output=True, text=True, env=_env(), cwd=REPO, timeout=800)
    assert r.returncode == 0, r.stderr[-3000:]
    import pickle
    res = pickle.load(open(outp, "rb"))
    assert res["shape"] == (7, 3, 128)
    assert res["bit_equal"] and res["succ"] and res["steps"] and res["tgt"], res
    assert res["loss"] == 0.0, res
