"""Per-kernel attribution of one step of a net: HIP events on the launch stream around every kernel and every layer."""
from __future__ import annotations

from . import PEAK_HBM_GBS, PEAK_MFMA_F32_TFLOPS
from .roofs import roofline_hbm, roofline_mfma


def _median(v):
    v = sorted(v)
    n = len(v)
    return v[n // 2] if n % 2 else 0.5 * (v[n // 2 - 1] + v[n // 2])


def attribute(net, reps):
    """After the timed region: eager forwards with HIP events on the launch stream around every kernel (stage timers of the C-ABI)
    and around every layer, joined with each convolution's geometry as it runs -> per-kernel algorithmic work / measured time.
    (The timed steps replay a hipGraph, which cannot carry per-kernel events.)"""
    from feathercnn_amd import ALGO_NAMES, DEPTHWISE, IM2COL, WINOGRADF63, booster
    algo_id = {v: k for k, v in ALGO_NAMES.items()}
    net.set_graph(False)
    booster.stage_timing(True)
    booster.stage_timing_collect()
    passes = []  # one {stage: ms} per eager pass: the spread of the figure across passes is reported next to its median
    for _ in range(reps):
        net.Forward()
        st = booster.stage_timing_collect()
        passes.append({k: v[0] for k, v in st.items() if v[1]})
    booster.stage_timing(False)
    stage = {k: _median([q[k] for q in passes if k in q]) for k in passes[0]}
    per_layer = None
    for _ in range(reps):
        timed = net.forward_timed()
        ms = [t[3] for t in timed]
        per_layer = ms if per_layer is None else [x + y for x, y in zip(per_layer, ms)]
    per_layer = [x / reps for x in per_layer]
    info = net.layers()
    convs = net.conv_params()
    fused_pw = net.fused_pointwise()
    siblings = net.siblings()      # 1: this 1x1 layer's launch also computes the next layer, 2: that next layer (launches nothing)
    residuals = net.residuals()    # 1: an Eltwise SUM operand (output-sized) is read and added in this layer's GEMM epilogue
    chains = net.chains(raw=True)  # 2 = the pair "first layer computed inside the next layer's input transform"
    chain_bytes = first_bytes = 0.0
    fz_flops = fz_bytes = fz_ms = 0.0
    by_type, table = {}, []
    gemm_flops = gemm_flops_64 = k2_bytes = dw_bytes = dw_ms = pw_flops = pw_ms = pw_bound_ms = direct = 0.0
    pw_hbm_bytes = pw_hbm_ms = 0.0
    pw_rows, pw_hbm_rows = [], []
    for i, ((typ, nm, algo), ms) in enumerate(zip(info, per_layer)):
        key = typ + ("/" + algo if algo else "")
        by_type[key] = by_type.get(key, 0.0) + ms
        row = {"layer": nm, "type": typ, "algo": algo, "ms": round(ms, 4)}
        if i in convs:
            p, n = convs[i]
            fl = 2.0 * p.output_channels * (p.input_channels // max(p.group, 1)) * p.output_h * p.output_w * p.kernel_h * p.kernel_w * n
            direct += fl
            row.update({"C": p.input_channels, "K": p.output_channels, "H": p.input_h, "k": p.kernel_h, "s": p.stride_h, "batch": n,
                        "direct_tflops": round(fl / max(ms, 1e-9) / 1e9, 2)})
            a_id = algo_id.get(algo)
            if chains.get(i, (0, 0))[1] == 2:
                row["computed_inside_next_input_transform"] = True  # launches nothing (fhip_winograd_f63_input_from_first)
                first_bytes += 4.0 * p.input_channels * p.input_h * p.input_w * n  # the image, read by the consumer's input transform
            if i in fused_pw and not fused_pw[i][1]:
                # absorbed pair that runs its two kernels one after the other at this shape: one layer time for both, priced by neither roofline
                q = fused_pw[i][0]
                direct += 2.0 * q.output_channels * q.input_channels * q.output_h * q.output_w * n
                row["sequential_pair_K"] = q.output_channels
                dw_bytes += 4.0 * (p.input_channels * p.input_h * p.input_w + p.output_channels * p.output_h * p.output_w) * n + 40.0 * p.input_channels
            elif i in fused_pw:
                # a 3x3 depthwise layer and the 1x1 convolution behind it running as ONE kernel (fhip_conv_forward_dw_pw): the pair's
                # compulsory bytes are its input and its output, its matrix work the pointwise GEMM
                q = fused_pw[i][0]
                pfl = 2.0 * q.output_channels * q.input_channels * q.output_h * q.output_w * n
                direct += pfl
                fz_flops += pfl
                fz_bytes += 4.0 * (p.input_channels * p.input_h * p.input_w + q.output_channels * q.output_h * q.output_w) * n
                fz_ms += ms
                row.update({"fused_pointwise_K": q.output_channels, "pair_gbs": round(4.0 * (p.input_channels * p.input_h * p.input_w + q.output_channels *
                            q.output_h * q.output_w) * n / max(ms, 1e-9) / 1e6, 1), "pair_mfma_frac": round(pfl / max(ms, 1e-9) / 1e9 / PEAK_MFMA_F32_TFLOPS, 4)})
            elif a_id == WINOGRADF63:
                # tiles and frequency points as the library runs the layer (fhip_winograd_f63_plan): 64 points on 6 x 6-output tiles, or -- planes
                # of 7 / 8 output pixels per side, round 4 -- 36 points on 4 x 4-output tiles; the work counted is the work executed
                import ctypes
                from feathercnn_amd import _lib
                pl_ = _lib.fhip_winograd_plan()
                if _lib.load_library().fhip_winograd_f63_plan(ctypes.byref(p), n, ctypes.byref(pl_)) != 0:
                    raise SystemExit("bench: fhip_winograd_f63_plan failed on a layer the net runs as Winograd")
                tiles, nxi = pl_.tiles_per_image, pl_.frequency_points
                gemm_flops += 2.0 * nxi * p.output_channels * p.input_channels * tiles * n
                # SURVEY.md 8(d)'s literal formula (64 points on ceil(Ho/6) * ceil(Wo/6) tiles) next to the executed work
                gemm_flops_64 += 2.0 * 64 * p.output_channels * p.input_channels * (-(-p.output_h // 6)) * (-(-p.output_w // 6)) * n
                row["winograd"] = f"F({pl_.tile_outputs}x{pl_.tile_outputs},3x3), {nxi} frequency points, {tiles} tiles per image"
                v_in, v_out = chains.get(i, (0, 0))
                if not v_in:
                    k2_bytes += 4.0 * (p.input_channels * p.input_h * p.input_w + nxi * p.input_channels * tiles) * n
                elif v_in == 2:
                    first_bytes += 4.0 * nxi * p.input_channels * tiles * n  # V written by the fused first-layer + input transform
                else:
                    chain_bytes += 4.0 * nxi * p.input_channels * tiles * n   # V' written by the chained transform of the layer before
                if v_out:
                    chain_bytes += 4.0 * nxi * p.output_channels * tiles * n  # M read by this layer's chained transform
                    row["chained_to_next"] = True
            elif a_id == DEPTHWISE:
                dw_bytes += 4.0 * (p.input_channels * p.input_h * p.input_w + p.output_channels * p.output_h * p.output_w) * n + 40.0 * p.input_channels
                dw_ms += ms
            elif a_id == IM2COL and p.kernel_h == 1 and p.kernel_w == 1:
                pw_flops += fl
                pw_ms += ms
                by = 4.0 * ((p.input_channels + p.output_channels) * p.output_h * p.output_w * n + p.input_channels * p.output_channels)
                if residuals.get(i) == 1:
                    # the fused residual operand: one more output-sized tensor this launch reads (fhip_conv_forward_residual)
                    by += 4.0 * p.output_channels * p.output_h * p.output_w * n
                    row["fused_residual"] = True
                if siblings.get(i) == 2:
                    # computed by the launch of the layer before (fhip_conv_forward_siblings): its work joins that row, it has no time of its own
                    row["computed_with_previous_layer"] = True
                    prev = table[-1]
                    prev["sibling_K"] = p.output_channels
                    fl += prev.pop("_fl")
                    by += prev.pop("_by") - 4.0 * p.input_channels * p.output_h * p.output_w * n  # the shared input is read once
                    ms, row_ = prev["ms"], prev
                    pw_rows.pop()
                    pw_bound_ms -= prev.pop("_bound_ms")
                    if prev.pop("_was_hbm"):
                        pw_hbm_rows.pop()
                        pw_hbm_bytes -= prev["_by0"]
                        pw_hbm_ms -= prev["_ms0"]
                    prev.pop("_by0", None)
                    prev.pop("_ms0", None)
                else:
                    row_ = row
                row_["mfma_frac"] = round(fl / max(ms, 1e-9) / 1e9 / PEAK_MFMA_F32_TFLOPS, 4)
                pw_rows.append(row_["mfma_frac"])
                # the layer's own lower bound: its matrix work at the MFMA peak or its compulsory bytes (the pixels the stride keeps, the
                # output, the weights and -- round 5 -- the fused residual operand) at the HBM peak, whichever is longer
                t_mfma, t_hbm = fl / (PEAK_MFMA_F32_TFLOPS * 1e9), by / (PEAK_HBM_GBS * 1e6)
                pw_bound_ms += max(t_mfma, t_hbm)
                row_["bound"] = "hbm" if t_hbm > t_mfma else "mfma"
                row_["bound_frac"] = round(max(t_mfma, t_hbm) / max(ms, 1e-9), 4)
                row_["hbm_bytes"] = by
                row_["frac_hbm"] = round(by / max(ms, 1e-9) / 1e6 / PEAK_HBM_GBS, 4)
                if t_hbm > t_mfma:
                    pw_hbm_rows.append(row_["frac_hbm"])
                    pw_hbm_bytes += by
                    pw_hbm_ms += ms
                if siblings.get(i) == 1:
                    row["_fl"], row["_by"], row["_bound_ms"], row["_was_hbm"], row["_by0"], row["_ms0"] = fl, by, max(t_mfma, t_hbm), t_hbm > t_mfma, by, ms
        table.append(row)
    roofs = []
    if gemm_flops and stage.get("wino_gemm"):
        roofs.append(dict(roofline_mfma("Winograd tile GEMM: wino_gemm_glds_kernel (C >= 128, K > 64) / gemm_mfma_kernel<WinoGemmPolicy>", gemm_flops,
                                   stage["wino_gemm"], "algorithmic FLOPs 2*xi*K*C*T*N (xi = 64 frequency points, T = ceil(Ho/6)*ceil(Wo/6) tiles; 7- and 8-pixel planes: xi = 36, T = ceil(Ho/4)*ceil(Wo/4)) summed over the Winograd layers of a step / "
                                   "sum of their tile-GEMM HIP-event durations on the launch stream"),
                          frac_passes=[round(gemm_flops / q["wino_gemm"] / 1e9 / PEAK_MFMA_F32_TFLOPS, 4) for q in passes if q.get("wino_gemm")],
                          frac_survey_8d_formula=round(gemm_flops_64 / stage["wino_gemm"] / 1e9 / PEAK_MFMA_F32_TFLOPS, 4),
                          frac_survey_8d_note="the same durations against SURVEY.md 8(d)'s literal 2*64*K*C*ceil(Ho/6)*ceil(Wo/6)*N (layers that run "
                          "F(4x4,3x3) execute 36 points on more tiles; `frac` counts the work executed)"))
    if pw_flops and pw_ms:
        r = roofline_mfma("1x1 implicit GEMM: gemm_mfma_kernel<ConvGemmPolicy<1|2|5>> (+ split-K reduce) / stream_gemm_kernel (C >= 256, 128 <= K <= 512)", pw_flops, pw_ms,
                          "ConvParam::GetFLOPS 2*K*C*Ho*Wo*N summed over the 1x1 convolution layers of a step / sum of their per-layer "
                          "HIP-event durations (bias, ReLU, folded BatchNorm and fused residual included)")
        r["layers"] = len(pw_rows)
        r["layer_frac_min"] = min(pw_rows)
        r["layer_frac_mean"] = round(sum(pw_rows) / len(pw_rows), 4)
        # sum of the layers' own lower bounds (max of MFMA time at 157.3 TF and HBM time at 8 TB/s, per layer) / measured time: what the
        # MFMA fraction alone understates for the layers that are bandwidth-bound at this batch (ResNet-50's 64 -> 256 @56x56)
        r["frac_of_tighter_bound"] = round(pw_bound_ms / pw_ms, 4)
        if pw_hbm_rows:
            # the layers whose compulsory bytes (fused residual operand included) take longer at 8 TB/s than their matrix work at 157.3 TF
            r["hbm_bound_layers"] = {"layers": len(pw_hbm_rows), "ms_per_step": round(pw_hbm_ms, 4), "bytes_per_step": pw_hbm_bytes,
                                     "frac_hbm": round(pw_hbm_bytes / pw_hbm_ms / 1e6 / PEAK_HBM_GBS, 4), "layer_frac_hbm_min": min(pw_hbm_rows)}
            r["layer_frac_min_is"] = "mfma fraction of the slowest layer; hbm_bound_layers.layer_frac_hbm_min is the HBM fraction of the slowest HBM-bound layer"
        roofs.append(r)
    if dw_bytes and stage.get("depthwise"):
        roofs.append(roofline_hbm("depthwise: depthwise3x3_flat_kernel (7 / 14 / 28-pixel planes) / depthwise3x3_band_kernel (112 / 56 pixels, stride 1) / depthwise3x3_direct_kernel", dw_bytes, stage["depthwise"], "compulsory bytes 4*(C*Hin*Win + C*Ho*Wo)*N + 40*C "
                                  "summed over the depthwise launches of a step (the layers fused into their 1x1 convolution have none) / sum of "
                                  "their HIP-event durations on the launch stream"))
    if fz_ms:
        roofs.append(roofline_hbm("fused depthwise 3x3 + 1x1: gemm_mfma_kernel<ConvGemmPolicy<3|4>> / dwpw_band_kernel (32-channel pair on 112-pixel rows)", fz_bytes, fz_ms, "input of the depthwise + output "
                                  "of the pointwise layer (the depthwise output never exists) summed over the fused pairs / their HIP-event durations"))
        roofs.append(roofline_mfma("fused depthwise 3x3 + 1x1 (the same launches, matrix side)", fz_flops, fz_ms, "2*K*C*Ho*Wo*N of the pointwise "
                                   "halves / the same durations"))
    if (k2_bytes or first_bytes) and stage.get("wino_input"):
        roofs.append(roofline_hbm("wino_input_from_first_staged_kernel (first layer computed inside the input transform: vector-ALU bound, "
                                  "1296 FMAs per 64 V values)" if first_bytes else
                                  "wino_input_staged_kernel (planes staged through LDS) / wino_input_transform_kernel / wino43_input_transform_kernel",
                                  k2_bytes + first_bytes, stage["wino_input"],
                                  "4*(C*H*W + 64*C*T)*N summed over the Winograd layers that run an input transform (for the fused first layer: the "
                                  "image + the consumer's V) / sum of the input-transform HIP-event durations"))
    if chain_bytes and stage.get("wino_chain"):
        roofs.append(roofline_hbm("wino_chain_kernel (output transform [+ max pooling] + next layer's input transform)", chain_bytes,
                                  stage["wino_chain"], "4*64*(K*T + C'*T')*N -- M read, next layer's V written; the activation between the two layers "
                                  "never exists -- summed over the chained layer boundaries / sum of their HIP-event durations"))
    return {"stage_ms_per_step": {k: round(v, 4) for k, v in stage.items()},
            "layer_type_ms_per_step": {k: round(v, 4) for k, v in sorted(by_type.items(), key=lambda kv: -kv[1])},
            "rooflines": roofs, "conv_direct_flops_per_step": direct, "table": table}
