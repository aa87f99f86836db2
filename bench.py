#!/usr/bin/env python
"""bench.py -- images/s of the fp32 forward pass on MI355X, per the driver contract.

A "step" is one forward pass of the benchmark network over one synthetic batch already resident in HBM.  Default
workload = BASELINE.json configs[1]: VGG-16, batch 32 per GPU, fp32, 224x224.  Two modes:

  --mode net (default)  the WHOLE network through the feather::Net runtime (include/feather_hip/feather_net.h): every
                        convolution through the ConvBooster hot path plus the layers between them (pooling, FC,
                        softmax, BN/Scale, eltwise ...), chained, from a synthetic ncnn .param/.bin model.
  --mode convstack      only the convolution layers, each through ConvBooster::Forward with bias + ReLU fused and its
                        input re-drawn (not chained) -- the hot path in isolation, SURVEY.md 8(d).

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)
  python bench.py --net resnet50|mobilenet_v1|vgg16|squeezenet_v1.1 [--batch B] [--mode convstack] [--fusion 0|1|2]

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline     : the dominant kernel (tile GEMM on fp32 MFMA for VGG/ResNet, depthwise on HBM for MobileNet), achieved =
                 algorithmic FLOPs (bytes) of all its launches in a step / their HIP-event durations, measured live;
  cpu_baseline : the REAL reference (oracle/_ref, FeatherCNN's AVX2 path compiled from /root/reference) timed on this
                 host's cores on a bounded sample (1 image through the same conv stack per process).
Multi-GPU: the batch dimension is sharded (fixed per-GPU batch => weak scaling); raw weights are generated on rank 0
and broadcast once over RCCL/xGMI, every rank runs its own Init; there is no steady-state collective.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_F32_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CU x 4 SIMD x 64 FLOP/clk x 2.4 GHz
PEAK_HBM_GBS = 8000.0         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec

DEFAULT_BATCH = {"vgg16": 32, "resnet50": 64, "mobilenet_v1": 256, "squeezenet_v1.1": 64}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--net", default="vgg16", choices=list(DEFAULT_BATCH))
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the BASELINE.json config)")
    ap.add_argument("--global-batch", type=int, default=0, help="fix the TOTAL batch instead (strong scaling, SURVEY.md 8d config 5: "
                    "--net resnet50 --global-batch 512 --gpus 8); rank r takes shard_range(global, r, N) images")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of one hipGraph replay per step")
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes for the CPU baseline (default: all cores, max 64)")
    ap.add_argument("--layers-out", default="", help="write the per-layer table (JSON) here")
    ap.add_argument("--mode", default="net", choices=["net", "convstack"])
    ap.add_argument("--no-overlap", action="store_true", help="net mode: keep every layer on one stream (no branch concurrency)")
    ap.add_argument("--reference-selection", action="store_true",
                    help="route convolutions with the reference's SelectAlgo rule instead of the MI355X cost model (fhip_conv_select_algo_tuned)")
    ap.add_argument("--fusion", type=int, default=2, help="net mode: 0 none, 1 the reference's TryFuse patterns, 2 also fold BN/Scale into conv weights")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------------------------
def cpu_worker(args):
    """One single-threaded process of the CPU baseline: the reference ConvBooster over the net's conv stack, 1 image."""
    net, core, reps = args
    try:
        os.sched_setaffinity(0, {core})
    except Exception:
        pass
    os.environ["OMP_NUM_THREADS"] = "1"
    import oracle
    from feathercnn_amd import nets
    from oracle import conv_geom, synth
    lib = oracle.ref() if oracle.have_ref() else None
    total = 0.0
    for layer in nets.NETS[net]():
        _, c, k, h, ks, s, p, g = layer
        geom = conv_geom(c, k, h, ks, s, p, group=g, bias=1, act=1)
        x, w, b = synth(geom, 1)
        if lib is not None:
            best, mean = lib.time_forward(geom, x[0], w, b, warmup=1, reps=reps)
            total += mean
        else:
            t0 = time.perf_counter()
            oracle.port().forward(geom, x, w, b)
            total += time.perf_counter() - t0
    return total  # seconds per image (conv stack only)


def cpu_baseline(net, procs):
    import multiprocessing as mp

    import oracle
    ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(ncpu))
    procs = procs or min(ncpu, 64)
    kind = "reference" if oracle.have_ref() else "port"
    reps = 2 if kind == "reference" else 1
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with ctx.Pool(1) as pool:  # single core first: the uncontended per-core number
        single = pool.map(cpu_worker, [(net, cores[0], reps)])[0]
    with ctx.Pool(procs) as pool:
        per = pool.map(cpu_worker, [(net, cores[i % len(cores)], reps) for i in range(procs)])
    wall = time.perf_counter() - t0
    value = sum(1.0 / t for t in per)
    model = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    model = line.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"value": round(value, 3), "unit": "images/s", "cores": procs, "kind": kind,
            "sample": f"{net} conv stack, 1 image per process, {procs} independent single-thread processes pinned to distinct "
                      f"cores (reference AVX Winograd is single-thread only), warmup 1 + {reps} timed reps per layer, "
                      f"{wall:.1f}s wall",
            "single_core_images_per_s": round(1.0 / single, 3), "cpu_model": model, "host_cores": ncpu}


def pmc_traffic(net, bound):
    """roofline.traffic: HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of this same
    command (profiles/<round>_<net>/traffic.json, written by tools/profile.sh + tools/summarize_prof.py: separate --pmc
    passes for FETCH_SIZE and WRITE_SIZE, read side doubled per MI355X_MICROARCH.md's gfx950 note).  null if no profile."""
    tag = {"vgg16": "vgg16", "resnet50": "resnet50", "mobilenet_v1": "mobilenet"}.get(net, net)
    best = None
    pdir = os.path.join(ROOT, "profiles")
    for d in sorted(os.listdir(pdir)) if os.path.isdir(pdir) else []:
        f = os.path.join(pdir, d, "traffic.json")
        if d.endswith("_" + tag) and os.path.exists(f):
            best = f  # the latest round wins
    if not best:
        return {"traffic": None}
    want = "WinoGemmPolicy" if bound == "mfma" else "depthwise"
    tot, n = 0.0, 0
    for k, v in json.load(open(best)).items():
        if want in k:
            tot += v["hbm_bytes_per_launch"] * v["launches_profiled"]
            n += v["launches_profiled"]
    if not n:
        return {"traffic": None}
    return {"traffic": round(tot / n), "traffic_unit": "bytes per launch (avg over the kernel's launches)",
            "traffic_source": os.path.relpath(best, ROOT)}


# ----------------------------------------------------------------------------------------------------------------------
def net_cpu_worker(args):
    """One single-threaded process of the whole-net CPU baseline: the reference feather::Net on one image."""
    param_path, bin_path, core, reps, in_name, out_name, shape = args
    try:
        os.sched_setaffinity(0, {core})
    except Exception:
        pass
    os.environ["OMP_NUM_THREADS"] = "1"
    import numpy as np
    from oracle import netcheck
    x = np.random.default_rng(core).uniform(-1, 1, (1,) + tuple(shape)).astype(np.float32)
    if netcheck.have_ref_net():
        ref = netcheck.RefNet(None, None, param_path, bin_path)
        ref.run(in_name, x, out_name)           # Reshape + Init + first Forward (untimed)
        t = ref.time_forward(reps)              # 1 warm-up + best of `reps`
        ref.close()
        return t
    port = netcheck.PortNet(open(param_path, "rb").read(), open(bin_path, "rb").read())
    t0 = time.perf_counter()
    port.run(in_name, x, out_name)
    return time.perf_counter() - t0


def net_cpu_baseline(net_name, model, procs):
    """The REAL reference runtime (feather::Net, AVX2) on this host: P independent single-thread processes, 1 image each."""
    import multiprocessing as mp
    import tempfile

    from oracle import netcheck
    p, b, i, o = model
    cores = sorted(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else list(range(os.cpu_count() or 1))
    ncpu = len(cores)
    avail = 0
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable"):
                avail = int(line.split()[1]) * 1024
    except OSError:
        pass
    per_proc = 6 * len(b) + (1 << 30)  # raw blobs + packed copies + transient Mat + python
    procs = procs or max(1, min(ncpu, 64, int(avail * 0.5 // per_proc) if avail else 8))
    kind = "reference" if netcheck.have_ref_net() else "port"
    reps = 2
    ctx = mp.get_context("spawn")
    t0 = time.perf_counter()
    with tempfile.TemporaryDirectory() as d:
        pp, bp = os.path.join(d, "m.param"), os.path.join(d, "m.bin")
        open(pp, "wb").write(p)
        open(bp, "wb").write(b)
        job = lambda c: (pp, bp, c, reps, i, o, (3, 224, 224))  # noqa: E731
        with ctx.Pool(1) as pool:
            single = pool.map(net_cpu_worker, [job(cores[0])])[0]
        with ctx.Pool(procs) as pool:
            per = pool.map(net_cpu_worker, [job(cores[k % ncpu]) for k in range(procs)])
    wall = time.perf_counter() - t0
    model_name = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model_name = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"value": round(sum(1.0 / t for t in per), 3), "unit": "images/s", "cores": procs, "kind": kind,
            "sample": f"{net_name} whole net through the reference feather::Net (N = 1, no fusion: the reference never runs its "
                      f"fusion pass), 1 image per process, {procs} independent single-thread processes pinned to distinct cores "
                      f"(reference AVX Winograd is single-thread only), 1 warm-up + best of {reps} forwards each, {wall:.1f}s wall",
            "single_core_images_per_s": round(1.0 / single, 3), "cpu_model": model_name, "host_cores": ncpu}


# ----------------------------------------------------------------------------------------------------------------------
def winograd_and_depthwise_work(net_name, batch, tuned=False):
    """Algorithmic work of the roofline kernels per step, from the conv shape list:
    tile-GEMM FLOPs 2*64*K*C*ceil(Ho/6)*ceil(Wo/6)*N over the layers SelectAlgo routes to WINOGRADF63, depthwise bytes
    4*(C*Hin*Win + C*Ho*Wo)*N + 40*C over the DEPTHWISE layers, direct-conv FLOPs (ConvParam::GetFLOPS) over all."""
    from feathercnn_amd import ConvBooster, booster, nets, WINOGRADF63, DEPTHWISE
    gemm_flops = dw_bytes = direct = 0.0
    n_conv = 0
    for layer in nets.NETS[net_name]():
        prm = nets.layer_param(layer, batch)
        cb = ConvBooster()
        cb.SelectAlgo(prm, tuned)
        direct += prm.GetFLOPS() * batch
        n_conv += 1
        if cb.algo == WINOGRADF63:
            pl = booster.winograd_plan(prm)
            gemm_flops += 2.0 * 64 * prm.output_channels * prm.input_channels * pl.tiles_per_image * batch
        elif cb.algo == DEPTHWISE:
            dw_bytes += 4.0 * (prm.input_channels * prm.input_h * prm.input_w + prm.output_channels * prm.output_h * prm.output_w) * batch \
                + 40.0 * prm.input_channels
    return gemm_flops, dw_bytes, direct, n_conv


def make_roofline(net_name, gemm_flops, gemm_ms, dw_bytes, dw_ms):
    roofline = None
    if net_name == "mobilenet_v1" and dw_ms > 0:
        ach = dw_bytes / dw_ms / 1e6
        roofline = {"kernel": "depthwise3x3_direct_kernel", "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS,
                    "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None,
                    "note": "compulsory bytes 4*(C*Hin*Win + C*Ho*Wo)*N + 40*C summed over the 13 depthwise launches of a step / "
                            "sum of their HIP-event durations on the launch stream, taken in an eager pass of the same step right after the "
                            "timed region"}
    elif gemm_ms > 0:
        ach = gemm_flops / gemm_ms / 1e9
        roofline = {"kernel": "Winograd tile GEMM: wino_gemm_glds_kernel (C >= 128, K > 64) / gemm_mfma_kernel<WinoGemmPolicy>", "bound": "mfma", "achieved": round(ach, 2),
                    "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_F32_TFLOPS, 4), "traffic": None,
                    "note": "algorithmic FLOPs 2*64*K*C*ceil(Ho/6)*ceil(Wo/6)*N summed over the Winograd layers of a step / "
                            "sum of their tile-GEMM HIP-event durations on the launch stream, taken in an eager pass of the same step right "
                            "after the timed region (the timed steps replay a hipGraph, which cannot carry per-kernel events)"}
    if roofline is not None:
        roofline.update(pmc_traffic(net_name, roofline["bound"]))
    return roofline


# ----------------------------------------------------------------------------------------------------------------------
def per_gpu_batch(a, env):
    """Weak scaling: the configured per-GPU batch.  Strong scaling (--global-batch): this rank's shard of the total."""
    if a.global_batch:
        from feathercnn_amd.shard import shard_range
        lo, hi = shard_range(a.global_batch, env["rank"], env["world"])
        if hi - lo < 1:
            raise SystemExit("bench: --global-batch smaller than the number of GPUs")
        return hi - lo
    return a.batch or DEFAULT_BATCH[a.net]


def setup_net(a, env):
    """Whole-net mode.  -> (step, finalize)"""
    import numpy as np
    import torch

    from feathercnn_amd import booster, model_zoo
    from feathercnn_amd.net import Net
    dev, rank, world = env["dev"], env["rank"], env["world"]
    batch = per_gpu_batch(a, env)
    build = model_zoo.MODELS[a.net]
    # ---- model: generated on rank 0, the .bin broadcast once over RCCL (the only collective of this path) ------------
    from feathercnn_amd.shard import broadcast_model
    model, t_bcast, bcast_bytes = broadcast_model(build, dev, src=0)
    p, b, in_name, out_name = model
    net = Net(fusion=a.fusion, graph=not a.no_graph, tuned=not a.reference_selection, concurrency=not a.no_overlap)
    net.LoadParam(p)
    net.LoadWeights(b)
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321 + rank)
    x = torch.rand((batch, 3, 224, 224), device=dev, generator=gen) * 2 - 1
    net.FeedInput(in_name, x)
    net.Forward()  # Reshape + Init (weight upload and transforms) + first forward; graph capture happens here
    torch.cuda.synchronize()
    prob = net.Extract(out_name)
    if not np.isfinite(prob).all() or abs(float(prob[0].sum()) - 1.0) > 1e-3:
        raise SystemExit("bench: the net's output is not a probability vector")

    def finalize(ms_per_step):
        res = {"metric": "images/sec fp32 forward @224x224", "launch": "hipGraph replay per step" if not a.no_graph else "eager launches"}
        gemm_flops, dw_bytes, direct, n_conv = winograd_and_depthwise_work(a.net, batch, not a.reference_selection)
        layers = net.layers()
        res["workload"] = (f"{a.net} whole net ({len(netcheck_layers(p))} layers in the model file, {len(layers)} after fusion level "
                           f"{a.fusion}, {n_conv} convolutions), batch {batch} per GPU, 224x224x3, fp32, synthetic ncnn .param/.bin")
        if rank != 0:
            return res
        reps = max(3, min(a.steps, 10))
        net.set_graph(False)
        booster.stage_timing(True)
        booster.stage_timing_collect()
        for _ in range(reps):
            net.Forward()
        st = booster.stage_timing_collect()
        booster.stage_timing(False)
        stage = {k: v[0] / reps for k, v in st.items() if v[1]}
        timed = net.forward_timed()
        by_type = {}
        for typ, nm, algo, ms in timed:
            key = typ + ("/" + algo if algo else "")
            by_type[key] = by_type.get(key, 0.0) + ms
        res["stage_ms_per_step"] = {k: round(v, 4) for k, v in stage.items()}
        res["layer_type_ms_per_step"] = {k: round(v, 4) for k, v in sorted(by_type.items(), key=lambda kv: -kv[1])}
        res["roofline"] = make_roofline(a.net, gemm_flops, stage.get("wino_gemm", 0.0), dw_bytes, stage.get("depthwise", 0.0))
        res["conv_gflops_per_s_direct"] = round(direct * world / (ms_per_step * 1e6), 1)
        res["device_memory"] = net.memory()
        res["table"] = [{"layer": nm, "type": typ, "algo": algo, "ms": round(ms, 4)} for typ, nm, algo, ms in timed]
        if world > 1:
            res["weight_broadcast"] = {"ms": round(t_bcast * 1e3, 3), "bytes": bcast_bytes, "collective": "1 flat RCCL broadcast of the .bin from rank 0"}
        res["cpu_baseline_fn"] = lambda: net_cpu_baseline(a.net, model, a.cpu_procs)
        return res

    return net.Forward, finalize, batch


def netcheck_layers(param_text):
    return [ln for ln in param_text.decode().splitlines()[2:] if ln.strip()]


def setup_convstack(a, env):
    """Conv-stack mode (the hot path in isolation).  -> (step, finalize, batch)"""
    import torch

    from feathercnn_amd import ConvLayer, booster, nets
    from feathercnn_amd import WINOGRADF63, DEPTHWISE, IM2COL, ALGO_NAMES
    from feathercnn_amd.shard import broadcast_weights
    dev, rank, world = env["dev"], env["rank"], env["world"]
    batch = per_gpu_batch(a, env)
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
        res["roofline"] = make_roofline(a.net, gemm_flops, gemm_ms, dw_bytes, dw_ms)
        res["conv_gflops_per_s_direct"] = round(flops_direct_total * world / (ms_per_step * 1e6), 1)
        res["conv_direct_frac_of_mfma_peak"] = round(flops_direct_total / (ms_per_step * 1e6) / 1e3 / PEAK_MFMA_F32_TFLOPS, 4)
        res["table"] = table
        if world > 1:
            res["weight_broadcast"] = {"ms": round(t_bcast * 1e3, 3), "bytes": bcast_bytes, "collective": "1 flat RCCL broadcast from rank 0"}
        res["cpu_baseline_fn"] = lambda: cpu_baseline(a.net, a.cpu_procs)
        return res

    return step, finalize, batch


# ----------------------------------------------------------------------------------------------------------------------
def main():
    a = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # FHIP_BENCH_SHARE_GPU=1 is a rehearsal switch for boxes with ONE GPU: every rank uses cuda:0 and the collectives go over
    # gloo, so the N > 1 code path (model broadcast, barriers, max-over-ranks timing) can be exercised end to end.  Its numbers
    # mean nothing; the driver's multi-GPU runs never set it.
    share = os.environ.get("FHIP_BENCH_SHARE_GPU") == "1"
    if share:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n_gpus = world
    env = {"dev": dev, "rank": rank, "world": world}
    step, finalize, batch = (setup_net if a.mode == "net" else setup_convstack)(a, env)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / a.steps * 1e3
    total_images = a.global_batch if a.global_batch else n_gpus * batch
    value = total_images * a.steps / dt

    # ---- per-stage / per-layer HIP-event timing (separate pass, after the timed region) -------------------------------
    extra = finalize(ms_per_step)
    if world > 1:
        dist.barrier()

    if rank == 0:
        table = extra.pop("table", [])
        cpu_fn = extra.pop("cpu_baseline_fn", None)
        res = {
            "metric": extra.pop("metric"), "value": round(value, 2), "unit": "images/s",
            "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "strong" if a.global_batch else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": extra.pop("workload"), "net": a.net, "mode": a.mode, "per_gpu_batch": batch, "global_batch": total_images,
                       "parallelism": f"batch-shard x{n_gpus}", "launch": extra.pop("launch"),
                       "conv_routing": "reference SelectAlgo rule" if a.reference_selection else "fhip_conv_select_algo_tuned (Winograd also on 4..8-pixel 3x3 layers)",
                       "streams": "one" if (a.no_overlap or a.mode != "net") else "main + one side stream for arena-free branch convolutions"},
        }
        roofline = extra.pop("roofline", None)
        res.update(extra)
        res["roofline"] = roofline
        if not a.no_cpu_baseline and n_gpus == 1 and cpu_fn is not None:
            try:
                res["cpu_baseline"] = cpu_fn()
            except Exception as e:  # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"value": None, "error": repr(e)}
        if a.layers_out:
            os.makedirs(os.path.dirname(os.path.abspath(a.layers_out)), exist_ok=True)
            with open(a.layers_out, "w") as f:
                json.dump({"net": a.net, "mode": a.mode, "batch": batch, "layers": table}, f, indent=1)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
