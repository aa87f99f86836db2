"""bench.py's own launcher, the part that needs no GPU: `--gpus N` without WORLD_SIZE must refuse loudly on a node with fewer
than N GPUs (here: none) instead of quietly running one rank, and `--gpus` must default to "whatever the launcher started"."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env():
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "FHIP_BENCH_SHARE_GPU"):
        env.pop(k, None)
    return env


def test_gpus_n_without_enough_gpus_is_refused():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], capture_output=True, text=True,
                         timeout=300, env=_env(), cwd=ROOT)
    assert out.returncode != 0
    assert "--gpus 8" in out.stderr and "GPU(s)" in out.stderr and "{" not in out.stdout


def test_gpus_flag_and_world_size_must_agree():
    env = dict(_env(), WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True, text=True,
                         timeout=300, env=env, cwd=ROOT)
    assert out.returncode != 0 and "must agree" in out.stderr


def test_launch_command_is_the_contract_command(monkeypatch, capsys):
    """launch_ranks builds exactly the driver contract's command line (torch.distributed.run, one node, N ranks, 127.0.0.1)."""
    sys.path.insert(0, ROOT)
    import bench
    seen = {}

    class FakeProc:
        def __init__(self, cmd, env=None, **kw):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = iter(["[Gloo] Rank 0 is connected\n", '{"n_gpus": 2}\n'])

        def wait(self):
            return 0

    import subprocess as sp
    monkeypatch.setattr(sp, "Popen", FakeProc)
    monkeypatch.setenv("FHIP_BENCH_SHARE_GPU", "1")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3"])
    assert bench.launch_ranks(2) == 0
    out = capsys.readouterr()
    assert out.out == '{"n_gpus": 2}\n' and "[Gloo]" in out.err  # only the JSON line reaches stdout
    c = seen["cmd"]
    assert c[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=2" in c
    assert c[c.index("--master-addr") + 1] == "127.0.0.1" and c[-4:] == ["--gpus", "2", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
