"""The ONE stdout line of bench.py: round 5's line was 20 KB and the driver could not parse it.  benchkit/report.py now
builds a compact line (contract keys + config + roofline + cpu_baseline + a few short N > 1 objects) capped at 8 KiB, and a detail side file
with everything else.  Here: a FULL synthetic result -- every key the N = 1 and the N = 8 paths produce, with round-5-sized strings and
tables -- goes through compose + fit_line; size, json.loads and the keys a consumer needs are asserted.  No GPU, no torch."""
import json

import pytest

from benchkit import report


def _roof(kernel, bound, frac, passes=False):
    r = {"kernel": kernel, "bound": bound, "achieved": 110.99 if bound == "mfma" else 4614.5, "peak": 157.3 if bound == "mfma" else 8000.0,
         "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": frac, "traffic": 439355835, "work_per_step": 223774507008.0, "ms_per_step": 2.0162,
         "note": "algorithmic FLOPs " + "x" * 400, "sustained_peak_measured": 134.4, "shader_mhz_under_mfma_load": 2180, "frac_of_sustained": 0.8258,
         "traffic_unit": "HBM bytes per launch (launch-weighted mean over the row's kernels)",
         "traffic_source": "profiles/r06_vgg16/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; 2*FETCH + WRITE)" + "y" * 100,
         "traffic_head": "766e5de23b33", "traffic_profiled_at": "766e5de23b33", "traffic_fingerprint": "3016be4bb61aa56e"}
    if passes:
        r.update({"frac_passes": [0.7012, 0.7056, 0.7061, 0.6989, 0.7049, 0.7055, 0.7058], "frac_survey_8d_formula": 0.7056, "frac_survey_8d_note": "z" * 300})
    return r


def _net(name, batch, world=1, scaling="weak", detail=True, global_batch=None):
    e = {"net": name, "images_per_s": 9921.83 * world, "ms_per_step": 3.2252, "steps": 100, "warmup": 5, "per_gpu_batch": batch,
         "global_batch": global_batch or batch * world, "scaling": scaling, "sub_batches": 2 if name == "mobilenet_v1" else 1,
         "steady_state": {"steps": 200, "images_per_s": 9936.98 * world, "ms_per_step": 3.22}}
    if detail:
        e.update({"workload": f"{name} whole net (40 layers in the model file, 20 after fusion level 3), batch {batch} per GPU, 224x224x3, fp32, synthetic ncnn .param/.bin" + "w" * 80,
                  "conv_tflops_direct": 98.7, "stage_ms_per_step": {"wino_gemm": 2.01, "wino_input": 0.22, "wino_chain": 0.87, "igemm": 0.1},
                  "layer_type_ms_per_step": {"Convolution/WINOGRADF63": 3.0, "InnerProduct": 0.2, "Pooling": 0.05},
                  "device_memory": {"arena": 1 << 30, "weights": 553 << 20},
                  "rooflines": [_roof("Winograd tile GEMM: wino_gemm_glds_kernel (C >= 128, K > 64) / gemm_mfma_kernel<WinoGemmPolicy>", "mfma", 0.7056, True),
                                dict(_roof("1x1 implicit GEMM: gemm_mfma_kernel<ConvGemmPolicy<1|2|5>> (+ split-K reduce) / stream_gemm_kernel", "mfma", 0.5345),
                                     frac_of_tighter_bound=0.5521, hbm_bound_layers={"layers": 4, "frac_hbm": 0.52}, layers=36),
                                _roof("depthwise: depthwise3x3_flat_kernel (7 / 14 / 28-pixel planes) / depthwise3x3_band_kernel", "hbm", 0.6443),
                                _roof("fused depthwise 3x3 + 1x1: gemm_mfma_kernel<ConvGemmPolicy<3|4>> / dwpw_band_kernel", "hbm", 0.3824),
                                _roof("fused depthwise 3x3 + 1x1 (the same launches, matrix side)", "mfma", 0.3276),
                                _roof("wino_input_from_first_staged_kernel (first layer computed inside the input transform)", "hbm", 0.4434),
                                _roof("wino_chain_kernel (output transform [+ max pooling] + next layer's input transform)", "hbm", 0.5768)],
                  "table": [{"layer": f"conv{i}", "type": "Convolution", "algo": "WINOGRADF63", "ms": 0.2, "C": 64, "K": 64, "H": 224, "winograd": "F(6x6,3x3)" * 5}
                            for i in range(60)]})
    return e


def _cpu():
    sweep = [{"procs": p, "images_per_s": 4.9 * p ** 0.7, "wall_images_per_s": 4.0, "mean_forward_s": 0.2, "best_forward_s": 0.19, "timed_forwards_per_worker": 3, "wall_s": 2.0}
             for p in (1, 8, 16, 32, 64, 128, 256)]
    return {"value": 48.04, "unit": "images/s", "cores": 16, "kind": "reference", "sample": "vgg16 whole net through the reference feather::Net " + "s" * 900,
            "sample_short": "vgg16 whole net, reference feather::Net, 1 image x 3 timed forwards per single-thread process, best of P=[1, 8, 16, 32, 64, 128], 49s",
            "sweep": sweep, "single_core_images_per_s": 4.95, "cpu_model": "AMD EPYC 9575F 64-Core Processor", "host_cores": 256, "nproc_images_per_s": 20.8,
            "nproc_point": sweep[-1], "why_best_is_not_nproc": "n" * 500}


ARGS = {"mode": "net", "steps": 20, "warmup": 5, "no_graph": False, "reference_selection": False, "no_overlap": False, "fusion": 3}
TREE = {"git_head": "766e5de23b33", "git_dirty": False, "source_fingerprint": "3016be4bb61aa56e", "from": "build stamp"}
CALIB = {"tflops": 134.4, "shader_mhz": 2180, "note": "c" * 300}


def _check_line(text, world):
    assert len(text.encode()) <= report.LINE_LIMIT == 8192
    assert "\n" not in text
    r = json.loads(text)
    for k in report.CONTRACT_KEYS:
        assert k in r, k
    assert r["n_gpus"] == world and r["higher_is_better"] is True and r["vs_baseline"] is None and r["dtype"] == "f32"
    assert r["config"]["workload"] and "model" not in r["config"]
    ro = r["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in ro, k
    assert ro["bound"] == "mfma" and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-3

    def strings(o):
        if isinstance(o, dict):
            for v in o.values():
                yield from strings(v)
        elif isinstance(o, list):
            for v in o:
                yield from strings(v)
        elif isinstance(o, str):
            yield o
    assert max(len(s) for s in strings(r)) <= report.STR_LIMIT
    return r


def test_line_n1_full_result_fits_and_parses():
    head = _net("vgg16", 32)
    extras = {"resnet50": _net("resnet50", 64), "mobilenet_v1": _net("mobilenet_v1", 256),
              "resnet50_global512": _net("resnet50", 512, scaling="strong", detail=False)}
    line, detail = report.compose(head, extras, args=ARGS, world=1, cpu=_cpu(), tree=TREE, calibration=CALIB, detail_path="bench_detail.json", traffic_note="t" * 500)
    text = report.fit_line(line)
    r = _check_line(text, 1)
    assert "shed" not in r  # the full N = 1 line fits without dropping anything
    assert r["roofline"]["frac_min"] == 0.6989 and r["roofline"]["frac_max"] == 0.7061 and len(r["roofline"]["frac_passes"]) == 7
    assert r["roofline"]["also"]["resnet50_b64"][0]["kernel"] == "tile_gemm"
    cb = r["cpu_baseline"]
    assert cb["value"] == 48.04 and cb["cores"] == 16 and cb["kind"] == "reference" and cb["sample"] and cb["single_core_images_per_s"] == 4.95 and cb["nproc_images_per_s"] == 20.8
    assert "sweep" not in cb and "why_best_is_not_nproc" not in cb
    other = r["config"]["other_nets"]
    assert other["resnet50_b64"]["img_s"] > 0 and other["resnet50_b64"]["gemm1x1_frac"] == 0.5345 and other["mobilenet_v1_b256"]["dwpw_hbm_frac"] == 0.3824
    assert other["resnet50_g512"]["per_gpu_batch"] == 512
    sc = other["resnet50_scaling"]
    assert sc["predicted_strong_eff_at_8_gpus"] == 1.0 and "strong_def" in sc and "weak_def" in sc
    assert r["detail"] == "bench_detail.json"
    # everything long lives in the detail object
    assert len(detail["tables"]["vgg16"]) == 60 and "sweep" in detail["cpu_baseline"] and len(detail["rooflines"]["resnet50"]) == 7
    json.dumps(detail)


def test_line_n8_every_key_of_the_multi_rank_path_fits_and_parses():
    world = 8
    head = dict(_net("vgg16", 32, world), weight_broadcast={"ms": 61.2, "bytes": 553432168, "collective": "1 flat RCCL broadcast of the .bin from rank 0"})
    extras = {"resnet50_global512": dict(_net("resnet50", 64, world, scaling="strong", global_batch=512), weight_broadcast={"ms": 9.0, "bytes": 102 << 20}),
              "resnet50": _net("resnet50", 64, world, detail=False),
              "resnet50_global512_one_gpu": _net("resnet50", 512, 1, scaling="strong", detail=False),
              "resnet50_one_gpu": _net("resnet50", 64, 1, detail=False), "vgg16_one_gpu": _net("vgg16", 32, 1, detail=False)}
    shard = {"net": "vgg16", "global_batch": 17, "shares": [3, 2, 2, 2, 2, 2, 2, 2], "max_norm_err": 0.0, "ok": True, "what": "q" * 200}
    aff = {"numa_node": 0, "cpus": list(range(64))}
    line, detail = report.compose(head, extras, args=ARGS, world=world, shard_ok=shard, affinity=aff, tree=TREE, calibration=CALIB, detail_path="bench_detail.json")
    r = _check_line(report.fit_line(line), world)
    assert "cpu_baseline" not in r and r["shard_check"]["ok"] is True and r["weight_broadcast"]["bytes"] == 553432168
    other = r["config"]["other_nets"]
    assert other["resnet50_g512"]["per_gpu_batch"] == 64 and other["resnet50_g512_one_gpu"]["img_s"] > 0 and other["vgg16_b32_one_gpu"]["img_s"] > 0
    sc = other["resnet50_scaling"]
    assert sc["n_gpus"] == 8 and sc["strong_eff"] == pytest.approx(1.0) and sc["weak_eff"] == pytest.approx(1.0) and sc["strong_def"] and sc["weak_def"]
    assert detail["config"]["rank0_cpu_affinity"] == aff


def test_a_failed_solo_point_is_reported_not_fatal():
    head = _net("vgg16", 32, 2)
    extras = {"resnet50_global512": _net("resnet50", 256, 2, scaling="strong", global_batch=512), "resnet50": _net("resnet50", 64, 2, detail=False),
              "resnet50_global512_one_gpu": {"error": "RuntimeError('HIP out of memory')"}}
    line, _ = report.compose(head, extras, args=ARGS, world=2, shard_ok={"ok": True, "max_norm_err": 0.0}, tree=TREE)
    r = _check_line(report.fit_line(line), 2)
    assert "HIP out of memory" in r["config"]["other_nets"]["resnet50_global512_one_gpu"]["error"] and "resnet50_scaling" not in r["config"]["other_nets"]


def test_fit_line_sheds_optional_keys_instead_of_overflowing():
    head = _net("vgg16", 32)
    extras = {f"net{i}": _net("resnet50", 64) for i in range(40)}  # far more than any real run carries
    for i, e in enumerate(extras.values()):
        e["per_gpu_batch"] = 64 + i  # distinct tags
    line, _ = report.compose(head, extras, args=ARGS, world=1, cpu=_cpu(), tree=TREE, calibration=CALIB, detail_path="d.json")
    assert len(json.dumps(line)) > 3 * report.LINE_LIMIT
    r = _check_line(report.fit_line(line), 1)
    assert "roofline.also" in r["shed"] and r["value"] == head["images_per_s"] and r["cpu_baseline"]["value"] == 48.04
    assert all("img_s" in v for k, v in r["config"]["other_nets"].items() if k != "resnet50_scaling")


def test_convstack_line():
    head = {"net": "vgg16", "images_per_s": 12000.0, "ms_per_step": 2.6, "per_gpu_batch": 32, "global_batch": 32, "scaling": "weak", "workload": "vgg16 conv layers (13)",
            "launch": "hipGraph replay per step", "rooflines": [_roof("Winograd tile GEMM", "mfma", 0.7056)], "roofline": _roof("Winograd tile GEMM", "mfma", 0.7056), "table": []}
    line, _ = report.compose(head, {}, args=dict(ARGS, mode="convstack"), world=1, cpu={"value": None, "error": "boom"}, tree=TREE)
    r = _check_line(report.fit_line(line), 1)
    assert "conv stack" in r["metric"] and r["cpu_baseline"] == {"value": None, "error": "boom"}


def test_only_the_cpu_baseline_leg_touches_the_oracle():
    """bench.py may use oracle/ in its cpu_baseline leg only: of benchkit's modules exactly cpu.py imports it, bench.py itself does not."""
    import os
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pat = re.compile(r"^\s*(from|import)\s+oracle\b", re.M)
    hits = [f for f in sorted(os.listdir(os.path.join(root, "benchkit"))) if f.endswith(".py") and pat.search(open(os.path.join(root, "benchkit", f)).read())]
    assert hits == ["cpu.py"]
    assert not pat.search(open(os.path.join(root, "bench.py")).read())
