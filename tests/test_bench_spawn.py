"""`python bench.py --gpus N` without a launcher starts its N ranks itself (a child torch.distributed.run) BEFORE anything
imports torch or touches the GPU, and leaves with the child's return code.  CPU only: the launcher call is intercepted."""
import importlib
import os
import subprocess
import sys

import pytest

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_gpus_n_without_world_size_spawns_a_child_launcher(monkeypatch):
    sys.path.insert(0, REPO)
    bench = importlib.import_module("bench")
    calls = []

    def fake_call(cmd, env=None):
        calls.append((cmd, env))
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    torch_loaded_before = "torch" in sys.modules
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                       # the launcher's return code is the parent's
    (cmd, env), = calls
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    i = cmd.index(os.path.join(REPO, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    if not torch_loaded_before:                    # the parent stays CPU-only and torch-free up to the spawn
        assert "torch" not in sys.modules


def test_under_a_launcher_no_second_spawn(monkeypatch):
    """With WORLD_SIZE set (the driver's torchrun, or our own child) main() must go on to the measurement, not spawn."""
    sys.path.insert(0, REPO)
    bench = importlib.import_module("bench")
    monkeypatch.setattr(bench, "spawn_ranks", lambda n: (_ for _ in ()).throw(AssertionError("spawned twice")))
    for k, v in (("WORLD_SIZE", "2"), ("RANK", "0"), ("LOCAL_RANK", "0"), ("MASTER_PORT", "29999")):
        monkeypatch.setenv(k, v)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    import torch
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: (_ for _ in ()).throw(RuntimeError("reached the measurement")))
    with pytest.raises(RuntimeError, match="reached the measurement"):
        bench.main()


def test_stray_world_size_is_not_a_launcher(monkeypatch):
    """WORLD_SIZE=1 exported by a scheduler without RANK / MASTER_PORT: plain `python bench.py` stays single-process
    (no env:// rendezvous), and `--gpus N` still spawns its own ranks."""
    sys.path.insert(0, REPO)
    bench = importlib.import_module("bench")
    for k in ("RANK", "LOCAL_RANK", "MASTER_PORT"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("WORLD_SIZE", "1")
    spawned = []
    monkeypatch.setattr(bench, "spawn_ranks", lambda n: spawned.append(n) or 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    with pytest.raises(SystemExit):
        bench.main()
    assert spawned == [2]
    import torch
    import torch.distributed as dist
    monkeypatch.setattr(dist, "init_process_group", lambda *a, **k: (_ for _ in ()).throw(AssertionError("rendezvous")))
    monkeypatch.setattr(torch.cuda, "set_device", lambda d: (_ for _ in ()).throw(RuntimeError("reached the measurement")))
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    with pytest.raises(RuntimeError, match="reached the measurement"):
        bench.main()
