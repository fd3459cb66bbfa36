"""bench.py's launcher logic (no GPU): `--gpus N` without a launcher starts N ranks through torch.distributed.run as a child
process; a WORLD_SIZE that disagrees with --gpus is an error, also when it is 1."""
import os
import subprocess
import sys

import pytest

from conftest import REPO


def test_world_size_mismatch_is_an_error_even_for_one_rank():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "--gpus 2 but WORLD_SIZE=1" in (p.stderr + p.stdout)


def test_self_launch_command(monkeypatch):
    sys.path.insert(0, REPO)
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7
    monkeypatch.setattr(subprocess, "call", fake_call)
    with pytest.raises(SystemExit) as e:
        bench.self_launch(4, ["--gpus", "4", "--steps", "3"])
    assert e.value.code == 7                                           # the parent exits with the child's code
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-5:] == [os.path.join(REPO, "bench.py"), "--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_self_launches_when_no_launcher(monkeypatch):
    sys.path.insert(0, REPO)
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "1"])
    called = {}

    def fake_launch(n, argv):
        called["n"], called["argv"] = n, argv
        raise SystemExit(0)
    monkeypatch.setattr(bench, "self_launch", fake_launch)
    with pytest.raises(SystemExit):
        bench.main()
    assert called == {"n": 2, "argv": ["--gpus", "2", "--steps", "1"]}
