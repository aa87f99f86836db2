"""What rank 0 prints.  Pure Python over plain dicts: no torch, no GPU, so tests/test_bench_report_cpu.py can build a full synthetic result
for N = 1 and N = 8 and check the line the driver has to parse.

The stdout line carries the contract keys + `config` (incl. the compact `other_nets`) + `roofline` (with `also`, `traffic` and the spread
over the eager attribution passes) + `cpu_baseline` (value, cores, the one-core and the P = nproc points) and a few short N > 1 objects --
and NOTHING else.  Everything long (`rooflines`, `nets`, per-layer tables, notes, the CPU sweep) goes to the detail side file whose path the
line names.  Round 5's line grew to 20 KB and the driver could not parse it: the line is now capped at LINE_LIMIT bytes by
construction, every string at STR_LIMIT characters, and `fit_line` sheds optional keys instead of ever printing more.
"""
from __future__ import annotations

import json

LINE_LIMIT = 8192
STR_LIMIT = 120

CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                 "data", "config")

SHORT = {"Winograd tile GEMM": "tile_gemm", "1x1 implicit GEMM": "gemm1x1", "depthwise:": "dw", "fused depthwise 3x3 + 1x1:": "dwpw_hbm",
         "fused depthwise 3x3 + 1x1 (": "dwpw_mfma", "wino_input_": "wino_input", "wino_chain_kernel": "wino_chain"}


def clip(s, n=STR_LIMIT):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def key_of(row):
    return next((v for k, v in SHORT.items() if row["kernel"].startswith(k)), row["kernel"][:24])


def median(v):
    v = sorted(v)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def net_tag(name, e):
    if e.get("scaling") == "strong":
        return f"{name.split('_global')[0]}_g{e['global_batch']}" + ("_one_gpu" if name.endswith("_one_gpu") else "")
    return f"{name.split('_one_gpu')[0]}_b{e['per_gpu_batch']}" + ("_one_gpu" if name.endswith("_one_gpu") else "")


def compact_roofline(r):
    """The headline roofline object of the line: SURVEY.md 8(d)'s keys + the spread of `frac` over the attribution passes."""
    out = {"kernel": clip(r["kernel"]), "bound": r["bound"], "achieved": r["achieved"], "peak": r["peak"], "unit": r["unit"], "frac": r["frac"],
           "traffic": r.get("traffic"), "work_per_step": r.get("work_per_step"), "ms_per_step": r.get("ms_per_step")}
    p = r.get("frac_passes")
    if p:
        # `frac` is already the median pass (benchkit/attribution.py); min / max / the passes themselves say how much a box moves it
        out.update({"frac_min": min(p), "frac_max": max(p), "frac_passes": p, "frac_is": f"median of {len(p)} eager attribution passes"})
    for k_in, k_out in (("frac_of_sustained", "frac_of_sustained"), ("sustained_peak_measured", "sustained_peak_measured"),
                        ("shader_mhz_under_mfma_load", "shader_mhz"), ("frac_survey_8d_formula", "frac_8d_formula"),
                        ("traffic_source", "traffic_source"), ("traffic_stale", "traffic_stale"), ("traffic_profiled_at", "traffic_profiled_at")):
        if r.get(k_in) is not None:
            out[k_out] = clip(r[k_in]) if isinstance(r[k_in], str) else r[k_in]
    if out.get("traffic") and r.get("launches_per_step"):
        out["launches_per_step"] = r["launches_per_step"]
    return out


def compact_cpu(c):
    if not c:
        return None
    if c.get("value") is None:
        return {"value": None, "error": clip(c.get("error", ""))}
    out = {"value": c["value"], "unit": c.get("unit", "images/s"), "cores": c["cores"], "kind": c["kind"], "sample": clip(c.get("sample_short") or c.get("sample", ""))}
    for k in ("single_core_images_per_s", "nproc_images_per_s", "host_cores"):
        if c.get(k) is not None:
            out[k] = c[k]
    if c.get("cpu_model"):
        out["cpu_model"] = clip(c["cpu_model"], 60)
    return out


def scaling_summary(nets_out, world):
    """configs[4] (ResNet-50, 512 images in total over the GPUs) and its weak twin (64 per GPU), both efficiencies with their definitions.
    At N = 1 only the prediction exists: at 8 GPUs every GPU runs 64 images per step and the data path has no collective, so the expected
    per-GPU rate is the one-GPU rate at batch 64 and the expected strong efficiency is img/s(b64) / img/s(b512) on one GPU."""
    g512_1 = nets_out.get("resnet50_global512_one_gpu") if world > 1 else nets_out.get("resnet50_global512")
    b64_1 = nets_out.get("resnet50_one_gpu") if world > 1 else nets_out.get("resnet50")
    if not (g512_1 and b64_1) or b64_1.get("per_gpu_batch") != 64:
        return None
    out = {"strong_def": "img_s(N GPUs, 512 images in total) / (N x img_s(1 GPU, 512 images))",
           "weak_def": "img_s(N GPUs, 64 images per GPU) / (N x img_s(1 GPU, 64 images))",
           "one_gpu_g512_img_s": g512_1["images_per_s"], "one_gpu_b64_img_s": b64_1["images_per_s"],
           "predicted_strong_eff_at_8_gpus": round(b64_1["images_per_s"] / g512_1["images_per_s"], 4),
           "b64_img_s_needed_for_0.9": round(0.9 * g512_1["images_per_s"], 1)}
    if world > 1:
        g, w = nets_out.get("resnet50_global512"), nets_out.get("resnet50")
        if g:
            out["strong_eff"] = round(g["images_per_s"] / (world * g512_1["images_per_s"]), 4)
            out["strong_img_s"] = g["images_per_s"]
        if w:
            out["weak_eff"] = round(w["images_per_s"] / (world * b64_1["images_per_s"]), 4)
            out["weak_img_s"] = w["images_per_s"]
        out["n_gpus"] = world
    return out


def compose(head, extras, *, args, world, cpu=None, shard_ok=None, affinity=None, tree=None, calibration=None, detail_path=None,
            traffic_note=""):
    """-> (line, detail): the dict printed as the ONE stdout line (pass it through fit_line) and the dict written to the side file.
    head / extras[name] are benchkit.timing.measure_net results (rank 0's, with detail where it was taken)."""
    head = dict(head)
    head_net = head["net"]
    convstack = args["mode"] == "convstack"
    line = {"metric": "images/sec fp32 forward @224x224" + (" (conv stack)" if convstack else ""),
            "value": head["images_per_s"], "unit": "images/s", "n_gpus": world, "steps": args["steps"], "warmup": args["warmup"],
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": head["scaling"], "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": clip(head.get("workload", "")), "net": head_net, "mode": args["mode"], "per_gpu_batch": head["per_gpu_batch"],
                       "global_batch": head["global_batch"], "parallelism": f"batch-shard x{world}",
                       "launch": "eager launches" if args.get("no_graph") else "hipGraph replay per step",
                       "conv_routing": "reference SelectAlgo rule" if args.get("reference_selection") else "fhip_conv_select_algo_tuned"}}
    detail = {k: line[k] for k in CONTRACT_KEYS if k != "config"}
    detail["config"] = dict(line["config"], workload=head.get("workload", ""),
                            streams="one" if (args.get("no_overlap") or convstack) else "main + one side stream for arena-free branch convolutions")
    if affinity is not None:
        detail["config"]["rank0_cpu_affinity"] = affinity
    table = head.pop("table", [])
    for k in ("stage_ms_per_step", "layer_type_ms_per_step", "conv_tflops_direct", "conv_gflops_per_s_direct", "device_memory", "weight_broadcast", "launch"):
        if k in head:
            detail[k] = head[k]
    roofs = head.get("rooflines", [])
    dom = None  # the dominant kernel of the headline net: tile GEMM (MFMA) for VGG / ResNet, depthwise (HBM) for MobileNet
    for r in roofs:
        if head_net == "mobilenet_v1" and r["kernel"].startswith("depthwise"):
            dom = r
    dom = dom or (roofs[0] if roofs else head.get("roofline"))
    detail["roofline"] = dom
    detail["rooflines"] = {head_net: roofs}
    detail["traffic_note"] = traffic_note
    if calibration:
        detail["mfma_calibration"] = calibration
    keep = ("images_per_s", "ms_per_step", "per_gpu_batch", "global_batch", "scaling", "sub_batches", "steps", "warmup", "steady_state", "error")
    nets_out = {head_net: {k: head[k] for k in keep if k in head}}
    tables = {head_net: table}
    for name, e in extras.items():
        nets_out[name] = {k: e[k] for k in keep if k in e}
        for k in ("workload", "stage_ms_per_step", "layer_type_ms_per_step", "conv_tflops_direct", "device_memory"):
            if k in e:
                nets_out[name][k] = e[k]
        if "rooflines" in e:
            detail["rooflines"][name] = e["rooflines"]
        tables[name] = e.get("table", [])
    detail["nets"] = nets_out
    detail["tables"] = tables
    # ---- the compact per-net summary: inside `config` (img/s + the fraction of each hot kernel) and `roofline.also` (achieved / unit / ms)
    other, also = {}, {}
    for name, e in nets_out.items():
        if "images_per_s" not in e:
            other[name] = {"error": clip(e.get("error", "not measured"))}
            continue
        tag = net_tag(name, e)
        row = {"img_s": e["images_per_s"], "ms_per_step": e["ms_per_step"], "per_gpu_batch": e["per_gpu_batch"]}
        if e.get("steady_state"):
            row["img_s_steady"] = e["steady_state"]["images_per_s"]
        for r in detail["rooflines"].get(name, []):
            k = key_of(r)
            row[k + "_frac"] = r["frac"]
            if k == "gemm1x1" and r.get("frac_of_tighter_bound") is not None:
                row["gemm1x1_frac_of_tighter_bound"] = r["frac_of_tighter_bound"]
            if k == "tile_gemm" and "frac_survey_8d_formula" in r:
                row["tile_gemm_frac_8d_formula"] = r["frac_survey_8d_formula"]
            also.setdefault(tag, []).append({"kernel": k, "bound": r["bound"], "frac": r["frac"], "achieved": r["achieved"], "unit": r["unit"],
                                             "ms_per_step": r["ms_per_step"]})
        if name != head_net:
            other[tag] = row
        else:
            line["config"]["headline_net"] = dict(row, tag=tag)
    sc = scaling_summary(nets_out, world)
    if sc:
        other["resnet50_scaling"] = sc
        detail["nets"].setdefault("resnet50_global512", {})["expected_from_1gpu"] = sc
    line["config"]["other_nets"] = other
    detail["config"]["other_nets"] = other
    if dom:
        line["roofline"] = dict(compact_roofline(dom), also=also)
    if cpu is not None:
        line["cpu_baseline"] = compact_cpu(cpu)
        detail["cpu_baseline"] = cpu
    if shard_ok is not None:
        detail["shard_check"] = shard_ok
        line["shard_check"] = {k: (clip(v) if isinstance(v, str) else v) for k, v in shard_ok.items() if k in ("ok", "max_norm_err", "global_batch", "shares", "error")}
    if "weight_broadcast" in head:
        line["weight_broadcast"] = {k: head["weight_broadcast"][k] for k in ("ms", "bytes") if k in head["weight_broadcast"]}
    if calibration and calibration.get("tflops"):
        line["mfma_calibration"] = {"tflops": calibration["tflops"], "shader_mhz": calibration.get("shader_mhz")}
    if tree:
        line["tree"] = tree
        detail["tree"] = tree
    if detail_path:
        line["detail"] = detail_path
    return line, detail


# optional parts of the line, in the order they are shed when it would not fit (the contract keys, roofline's and cpu_baseline's own
# SURVEY.md 8(d) keys and config.other_nets' img/s are never shed)
_SHED = (("roofline", "also"), ("mfma_calibration",), ("tree",), ("roofline", "frac_passes"), ("config", "headline_net"), ("weight_broadcast",),
         ("roofline", "traffic_source"), ("roofline", "traffic_stale"), ("cpu_baseline", "sample"), ("cpu_baseline", "cpu_model"),
         ("config", "conv_routing"), ("config", "launch"), ("shard_check", "shares"))


def fit_line(line, limit=LINE_LIMIT):
    """json text of `line`, at most `limit` bytes: optional keys are dropped (in _SHED order, then other_nets rows are reduced to img/s) until it
    fits; `shed` in the line lists what went.  Raises only if the contract keys alone do not fit -- which cannot happen."""
    line = json.loads(json.dumps(line))  # deep copy; also proves it serialises
    text = json.dumps(line, separators=(",", ":"))
    shed = []
    for path in _SHED:
        if len(text.encode()) <= limit:
            break
        d = line
        for k in path[:-1]:
            d = d.get(k) if isinstance(d, dict) else None
        if isinstance(d, dict) and path[-1] in d:
            del d[path[-1]]
            shed.append(".".join(path))
            line["shed"] = shed
            text = json.dumps(line, separators=(",", ":"))
    if len(text.encode()) > limit:
        other = line.get("config", {}).get("other_nets", {})
        for tag, row in other.items():
            other[tag] = {k: v for k, v in row.items() if k in ("img_s", "ms_per_step", "per_gpu_batch", "strong_eff", "weak_eff", "predicted_strong_eff_at_8_gpus", "error")}
        shed.append("config.other_nets.*: img/s only")
        line["shed"] = shed
        text = json.dumps(line, separators=(",", ":"))
    if len(text.encode()) > limit:
        raise AssertionError(f"bench line is {len(text.encode())} bytes after shedding every optional key (limit {limit})")
    assert json.loads(text)["value"] == line["value"]
    return text
