"""bench.py's launcher logic without a GPU: `--gpus N` is the world size — under a launcher
whose WORLD_SIZE differs the run is refused before anything is imported; without a launcher and N > 1 the command
line for N ranks is built (checked here by intercepting the subprocess call)."""
import os
import subprocess
import sys
import types

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world_size_mismatch_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=env, capture_output=True,
                         text=True, timeout=120)
    assert out.returncode == 2
    assert "--gpus 8 but WORLD_SIZE=1" in out.stderr
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]   # no line with another n_gpus


def test_gpus_n_builds_an_n_rank_launch(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 0
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "7"])
    with pytest.raises(SystemExit) as done:
        bench.launch_ranks(types.SimpleNamespace(gpus=4))
    assert done.value.code == 0
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-4:] == ["--gpus", "4", "--steps", "7"] and cmd[-5].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_single_gpu_needs_no_launcher(monkeypatch):
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench.launch_ranks(types.SimpleNamespace(gpus=1)) is None
    monkeypatch.setenv("WORLD_SIZE", "2")
    assert bench.launch_ranks(types.SimpleNamespace(gpus=2)) is None
