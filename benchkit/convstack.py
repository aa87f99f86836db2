"""--mode convstack: the convolution layers of a net alone, each through ConvBooster::Forward (the hot path in isolation)."""
from __future__ import annotations

import sys
import time

from . import PEAK_HBM_GBS, PEAK_MFMA_F32_TFLOPS
from .roofs import roofline_hbm, roofline_mfma
from .timing import per_gpu_batch


def setup_convstack(a, env):
    """Conv-stack mode (the hot path in isolation).  -> (step, finalize, batch)"""
    import torch

    from feathercnn_amd import ConvLayer, booster, nets
    from feathercnn_amd import WINOGRADF63, DEPTHWISE, IM2COL, ALGO_NAMES
    from feathercnn_amd.shard import broadcast_weights
    dev, rank, world = env["dev"], env["rank"], env["world"]
    batch = per_gpu_batch(a.net, a, env, a.global_batch, a.batch)
    layers = nets.NETS[a.net]()

    # ---- weights: generated on rank 0, broadcast once over RCCL (the only collective of this path) -------------------
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)  # ranks start from DIFFERENT weights: only the broadcast makes them agree
    raw = []
    for layer in layers:
        name, c, k, h, ks, s, p, g = layer
        prm = nets.layer_param(layer, batch)
        cpg = c // g
        w = (torch.rand((prm.output_channels, cpg, ks, ks), device=dev, generator=gen) * 2 - 1) / (cpg * ks * ks) ** 0.5
        b = (torch.rand((prm.output_channels,), device=dev, generator=gen) * 2 - 1) * 0.1
        raw.append((prm, w, b))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    bcast_bytes = broadcast_weights([t for _, w, b in raw for t in (w, b)], src=0)  # ONE flat RCCL broadcast
    torch.cuda.synchronize()
    t_bcast = time.perf_counter() - t0
    built = []
    max_scratch, max_out = 0, 0
    for layer, (prm, w, b) in zip(layers, raw):
        name, c, k, h, ks, s, p, g = layer
        lyr = ConvLayer(prm, w, b, tuned=not a.reference_selection)
        x = torch.rand((batch, c, h, h), device=dev, generator=gen) * 2 - 1
        built.append((layer, prm, lyr, x))
        max_scratch = max(max_scratch, lyr.buffer_bytes)
        max_out = max(max_out, batch * prm.output_channels * prm.output_h * prm.output_w)
    scratch = torch.empty(max(max_scratch // 4, 1), dtype=torch.float32, device=dev)  # one shared arena (mempool.cpp:88-92)
    out = torch.empty(max_out, dtype=torch.float32, device=dev)

    def eager_step():
        for _, prm, lyr, x in built:
            lyr.booster.Forward(prm, out, x, lyr.packed, scratch, lyr.bias)

    # One step = one hipGraph replay: Forward never allocates and has no host-side state, so it is capturable as is.
    step, graph_used = eager_step, False
    if not a.no_graph:
        try:
            eager_step()
            torch.cuda.synchronize()
            cg = torch.cuda.CUDAGraph()
            with torch.cuda.graph(cg):
                eager_step()
            step, graph_used = cg.replay, True
        except Exception as e:  # capture is an optimisation, never a requirement
            print(f"bench: hipGraph capture unavailable ({e!r}); timing eager launches", file=sys.stderr)
            step, graph_used = eager_step, False

    def finalize(ms_per_step):
        res = {"metric": "images/sec fp32 forward (conv stack) @224x224", "launch": "hipGraph replay per step" if graph_used else "eager launches",
               "workload": f"{a.net} conv layers ({len(layers)}), batch {batch} per GPU, 224x224x3, bias+ReLU fused, fp32"}
        if rank != 0:
            return res
        table = []
        reps = max(3, min(a.steps, 10))
        booster.stage_timing(True)
        flops_direct_total, gemm_flops, gemm_ms = 0.0, 0.0, 0.0
        dw_bytes, dw_ms = 0.0, 0.0
        stage_tot = {}
        for layer, prm, lyr, x in built:
            booster.stage_timing_collect()
            for _ in range(reps):
                lyr.booster.Forward(prm, out, x, lyr.packed, scratch, lyr.bias)
            st = booster.stage_timing_collect()
            per = {k: v[0] / reps for k, v in st.items() if v[1]}
            for k, v in per.items():
                stage_tot[k] = stage_tot.get(k, 0.0) + v
            algo = lyr.booster.algo
            fl = prm.GetFLOPS() * batch
            flops_direct_total += fl
            row = {"layer": layer[0], "algo": ALGO_NAMES[algo], "C": prm.input_channels, "K": prm.output_channels,
                   "H": prm.input_h, "k": prm.kernel_h, "s": prm.stride_h, "ms": round(sum(per.values()), 4),
                   "direct_gflops_per_s": round(fl / max(sum(per.values()), 1e-9) / 1e6, 1), "stages_ms": {k: round(v, 4) for k, v in per.items()}}
            if algo == WINOGRADF63:
                pl = booster.winograd_plan(prm)
                gf = 2.0 * 64 * prm.output_channels * prm.input_channels * pl.tiles_per_image * batch
                gemm_flops += gf
                gemm_ms += per.get("wino_gemm", 0.0)
                row["tile_gemm_tflops"] = round(gf / max(per.get("wino_gemm", 1e-9), 1e-9) / 1e9, 2)
                row["tile_gemm_mfma_frac"] = round(row["tile_gemm_tflops"] / PEAK_MFMA_F32_TFLOPS, 4)
                hbm_in = 4.0 * (prm.input_channels * prm.input_h * prm.input_w + 64 * prm.input_channels * pl.tiles_per_image) * batch
                hbm_out = 4.0 * (64 * prm.output_channels * pl.tiles_per_image + prm.output_channels * prm.output_h * prm.output_w) * batch
                row["input_xform_gbs"] = round(hbm_in / max(per.get("wino_input", 1e-9), 1e-9) / 1e6, 1)
                row["output_xform_gbs"] = round(hbm_out / max(per.get("wino_output", 1e-9), 1e-9) / 1e6, 1)
            elif algo == DEPTHWISE:
                by = 4.0 * (prm.input_channels * prm.input_h * prm.input_w + prm.output_channels * prm.output_h * prm.output_w) * batch \
                    + 4.0 * 10 * prm.input_channels
                dw_bytes += by
                dw_ms += per.get("depthwise", 0.0)
                row["hbm_gbs"] = round(by / max(per.get("depthwise", 1e-9), 1e-9) / 1e6, 1)
                row["hbm_frac"] = round(row["hbm_gbs"] / PEAK_HBM_GBS, 4)
            elif algo == IM2COL:
                row["igemm_tflops"] = round(fl / max(per.get("igemm", 1e-9), 1e-9) / 1e9, 2)
                row["igemm_mfma_frac"] = round(row["igemm_tflops"] / PEAK_MFMA_F32_TFLOPS, 4)
            table.append(row)
        booster.stage_timing(False)
        res["stage_ms_per_step"] = {k: round(v, 4) for k, v in stage_tot.items()}
        roofs = []
        if gemm_flops and gemm_ms:
            roofs.append(roofline_mfma("Winograd tile GEMM", gemm_flops, gemm_ms, "2*64*K*C*T*N over the Winograd layers / their tile-GEMM event durations"))
        if dw_bytes and dw_ms:
            roofs.append(roofline_hbm("depthwise: depthwise3x3_flat_kernel (7 / 14 / 28-pixel planes) / depthwise3x3_band_kernel (112 / 56 pixels, stride 1) / depthwise3x3_direct_kernel", dw_bytes, dw_ms, "4*(C*Hin*Win + C*Ho*Wo)*N + 40*C over the depthwise layers / their event durations"))
        res["rooflines"] = roofs
        res["roofline"] = (roofs[1] if a.net == "mobilenet_v1" and len(roofs) > 1 else roofs[0]) if roofs else None
        res["conv_gflops_per_s_direct"] = round(flops_direct_total * world / (ms_per_step * 1e6), 1)
        res["conv_direct_frac_of_mfma_peak"] = round(flops_direct_total / (ms_per_step * 1e6) / 1e3 / PEAK_MFMA_F32_TFLOPS, 4)
        res["table"] = table
        if world > 1:
            res["weight_broadcast"] = {"ms": round(t_bcast * 1e3, 3), "bytes": bcast_bytes, "collective": "1 flat RCCL broadcast from rank 0"}
        return res

    return step, finalize, batch
