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


def _run(cmd, env, tmp_path):
    detail = str(tmp_path / "detail.json")
    out = subprocess.run(cmd + ["--detail-out", detail], capture_output=True, text=True, timeout=1500, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    return out, detail


def _line(out, only_line=False):
    lines = [ln for ln in out.stdout.splitlines() if (ln.strip() if only_line else ln.startswith("{"))]
    assert len(lines) == 1 and lines[0].startswith("{"), out.stdout[-2000:]  # rank 0 prints ONE JSON line, the other ranks none
    assert len(lines[0].encode()) <= 8192, len(lines[0])  # the driver must be able to parse it
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_share_one_gpu(cuda, tmp_path):
    from feathercnn_amd import model_zoo
    env = dict(os.environ, FHIP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-steady", "--no-cpu-baseline"]
    out, detail = _run(cmd, env, tmp_path)
    r = _line(out)
    d = json.load(open(detail))
    assert r["n_gpus"] == 2 and r["steps"] == 2 and r["warmup"] == 1 and r["scaling"] == "weak" and r["value"] > 0
    assert r["config"]["global_batch"] == 64 and r["config"]["per_gpu_batch"] == 32            # VGG-16, 32 per GPU, weak
    g512 = d["nets"]["resnet50_global512"]
    assert g512["global_batch"] == 512 and g512["scaling"] == "strong" and g512["per_gpu_batch"] == 256
    assert d["nets"]["resnet50"]["scaling"] == "weak" and d["nets"]["resnet50"]["global_batch"] == 128
    # the one collective of the path: the .bin of the headline net, broadcast once from rank 0
    assert r["weight_broadcast"]["bytes"] == len(model_zoo.MODELS["vgg16"]()[1])
    # rank 1's results on its shard equal rank 0's on the same images (src/layers/conv_layer.h:107: images are independent)
    sc = r["shard_check"]
    assert sc["ok"] and sc["global_batch"] == 5 and sc["shares"] == [3, 2] and sc["max_norm_err"] <= 1e-5, sc
    assert "cpu_baseline" not in r and r["roofline"]["frac"] > 0 and len(r["roofline"]["frac_passes"]) >= 5


@pytest.mark.gpu
def test_bench_gpus_flag_starts_the_ranks_itself(cuda, tmp_path):
    """plain `python bench.py --gpus 2` -- no torch.distributed.run in front -- must start two ranks by itself (launch_ranks) and
    print one JSON line with n_gpus == 2; the same one-GPU rehearsal switch.  The compact per-net summary sits inside `config` and `roofline`,
    and both scaling efficiencies of config 5 are in the line with their definitions."""
    from feathercnn_amd import model_zoo
    env = dict(os.environ, FHIP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-steady", "--no-cpu-baseline"]
    out, detail = _run(cmd, env, tmp_path)
    r = _line(out, only_line=True)  # the launcher's stdout is rank 0's JSON line and nothing else
    assert r["n_gpus"] == 2 and r["scaling"] == "weak" and r["value"] > 0
    assert r["shard_check"]["ok"] and r["weight_broadcast"]["bytes"] == len(model_zoo.MODELS["vgg16"]()[1])
    other = r["config"]["other_nets"]
    assert other["resnet50_g512"]["img_s"] > 0 and other["resnet50_b64"]["img_s"] > 0 and other["resnet50_g512_one_gpu"]["img_s"] > 0
    assert other["resnet50_b64_one_gpu"]["img_s"] > 0 and other["vgg16_b32_one_gpu"]["img_s"] > 0
    sc = other["resnet50_scaling"]
    assert sc["strong_eff"] > 0 and sc["weak_eff"] > 0 and sc["predicted_strong_eff_at_8_gpus"] > 0 and sc["strong_def"] and sc["weak_def"]
    assert r["roofline"]["also"]["vgg16_b32"][0]["kernel"] == "tile_gemm"
    assert json.load(open(detail))["nets"]["resnet50_global512"]["expected_from_1gpu"]["n_gpus"] == 2


@pytest.mark.gpu
def test_bench_eight_ranks_rehearsal_on_one_gpu(cuda, tmp_path):
    """the REAL 8-rank shape before an 8-GPU node sees it -- launcher, port choice, eight per-rank arenas, config 5 split 64 per
    rank, the 17-image ragged shard check, the line size -- rehearsed on one GPU (every rank on cuda:0, collectives over gloo)."""
    env = dict(os.environ, FHIP_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-steady", "--no-cpu-baseline"]
    out, detail = _run(cmd, env, tmp_path)
    r = _line(out, only_line=True)
    assert r["n_gpus"] == 8 and r["scaling"] == "weak" and r["value"] > 0 and r["config"]["global_batch"] == 256
    sc = r["shard_check"]
    assert sc["ok"] and sc["global_batch"] == 17 and sum(sc["shares"]) == 17 and len(sc["shares"]) == 8 and sc["max_norm_err"] <= 1e-5, sc
    other = r["config"]["other_nets"]
    assert other["resnet50_g512"]["per_gpu_batch"] == 64 and other["resnet50_g512"]["img_s"] > 0   # config 5 as the 8-GPU node will run it
    assert other["resnet50_b64"]["per_gpu_batch"] == 64 and other["resnet50_g512_one_gpu"]["per_gpu_batch"] == 512
    s8 = other["resnet50_scaling"]
    assert s8["n_gpus"] == 8 and s8["strong_eff"] > 0 and s8["weak_eff"] > 0
    d = json.load(open(detail))
    assert d["nets"]["resnet50_global512"]["global_batch"] == 512 and d["nets"]["resnet50"]["global_batch"] == 512


@pytest.mark.gpu
def test_bench_gpus_flag_must_agree_with_the_launcher(cuda):
    """--gpus 4 under a launcher that started 2 ranks: refused before anything is measured (the flag can never disagree with the run)."""
    env = dict(os.environ, FHIP_BENCH_SHARE_GPU="1", WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"], capture_output=True, text=True, timeout=300,
                         env=env, cwd=ROOT)
    assert out.returncode != 0 and "must agree" in out.stderr


@pytest.mark.gpu
def test_bench_single_rank_line_is_compact_and_complete(cuda, tmp_path):
    """The N = 1 form of the driver's command (shortened: headline net only, no CPU leg): ONE stdout line of at most 8 KiB that json.loads, with the contract keys,
    `roofline` carrying the spread of its 7 attribution passes, and everything long in the detail side file the line names."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "FHIP_BENCH_SHARE_GPU"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--headline-only", "--no-steady", "--no-cpu-baseline"]
    out, detail = _run(cmd, env, tmp_path)
    r = _line(out, only_line=True)
    assert r["n_gpus"] == 1 and r["steps"] == 3 and r["warmup"] == 1 and r["value"] > 0 and r["dtype"] == "f32" and r["vs_baseline"] is None
    assert r["config"]["net"] == "vgg16" and r["config"]["per_gpu_batch"] == 32 and r["detail"] == detail
    ro = r["roofline"]
    assert ro["bound"] == "mfma" and len(ro["frac_passes"]) == 7 and ro["frac_min"] <= ro["frac"] <= ro["frac_max"] and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 2e-3
    d = json.load(open(detail))
    assert len(d["tables"]["vgg16"]) >= 15 and d["rooflines"]["vgg16"][0]["note"]
