"""SURVEY.md 8(e) / VERDICT r03 #7: `bench.py --gpus 2` end to end as the driver launches it (one process per rank under
torch.distributed.run), rehearsed on the ONE GPU of the test box: FHIP_BENCH_SHARE_GPU=1 puts both ranks on cuda:0 and the collectives on gloo,
so the N > 1 code path -- model broadcast from rank 0, barriers, max-over-ranks timing, config 5's global batch of 512 split over the ranks,
the shard check -- runs for real.  Its throughput numbers mean nothing and are not asserted."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu(cuda):
    from feathercnn_amd import model_zoo
    env = dict(os.environ, FHIP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-steady", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]  # rank 0 prints ONE JSON line, rank 1 none
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak" and r["value"] > 0
    assert r["config"]["global_batch"] == 64 and r["config"]["per_gpu_batch"] == 32            # VGG-16, 32 per GPU, weak
    g512 = r["nets"]["resnet50_global512"]
    assert g512["global_batch"] == 512 and g512["scaling"] == "strong" and g512["per_gpu_batch"] == 256
    assert r["nets"]["resnet50"]["scaling"] == "weak" and r["nets"]["resnet50"]["global_batch"] == 128
    # the one collective of the path: the .bin of the headline net, broadcast once from rank 0
    assert r["weight_broadcast"]["bytes"] == len(model_zoo.MODELS["vgg16"]()[1])
    # rank 1's results on its shard equal rank 0's on the same images (src/layers/conv_layer.h:107: images are independent)
    sc = r["shard_check"]
    assert sc["ok"] and sc["global_batch"] == 5 and sc["shares"] == [3, 2] and sc["max_norm_err"] <= 1e-5, sc
    assert "cpu_baseline" not in r and r["roofline"]["frac"] > 0


@pytest.mark.gpu
def test_bench_gpus_flag_starts_the_ranks_itself(cuda):
    """plain `python bench.py --gpus 2` -- no torch.distributed.run in front -- must start two ranks by itself (launch_ranks) and
    print one JSON line with n_gpus == 2; the same one-GPU rehearsal switch.  Also: the compact per-net summary sits inside `config` and
    `roofline` (the objects a consumer that trims unknown top-level keys still keeps), and config 5's expectation is written into the line."""
    from feathercnn_amd import model_zoo
    env = dict(os.environ, FHIP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-steady", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]  # the launcher's stdout is rank 0's JSON line and nothing else
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["value"] > 0
    assert r["shard_check"]["ok"] and r["weight_broadcast"]["bytes"] == len(model_zoo.MODELS["vgg16"]()[1])
    other = r["config"]["other_nets"]
    assert other["resnet50_g512"]["img_s"] > 0 and other["resnet50_b64"]["img_s"] > 0 and other["resnet50_g512_one_gpu"]["img_s"] > 0
    exp = r["nets"]["resnet50_global512"]["expected_from_1gpu"]
    assert exp["measured_efficiency_at_this_n"] > 0 and exp["predicted_efficiency_at_8_gpus"] > 0
    assert r["roofline"]["also"]["vgg16_b32"][0]["kernel"] == "tile_gemm"


@pytest.mark.gpu
def test_bench_gpus_flag_must_agree_with_the_launcher(cuda):
    """--gpus 4 under a launcher that started 2 ranks: refused before anything is measured (the flag can never disagree with the run)."""
    env = dict(os.environ, FHIP_BENCH_SHARE_GPU="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True, text=True, timeout=300,
                         env=env, cwd=ROOT)
    assert out.returncode != 0 and "must agree" in out.stderr
