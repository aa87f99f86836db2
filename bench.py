#!/usr/bin/env python
"""bench.py -- images/s of the conv hot path on MI355X, per the driver contract.

A "step" is one pass of the conv stack of the benchmark network over one synthetic batch: every convolution layer
of the net runs once through ConvBooster::Forward (bias + ReLU fused), inputs already resident in HBM, layer inputs
re-drawn per layer (not chained: the layers between convs are out of this tier's scope).  Default
workload = BASELINE.json configs[1]: VGG-16, batch 32 per GPU, fp32, 224x224.

  python bench.py --gpus N --steps K --warmup W            (N > 1: launched by torch.distributed.run, one rank per GPU)
  python bench.py --net resnet50|mobilenet_v1|vgg16 [--batch B]

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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of one hipGraph replay per step")
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes for the CPU baseline (default: all cores, max 64)")
    ap.add_argument("--layers-out", default="", help="write the per-layer table (JSON) here")
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
def main():
    a = parse()
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    n_gpus = world

    from feathercnn_amd import ConvLayer, booster, nets
    from feathercnn_amd import WINOGRADF63, DEPTHWISE, IM2COL, ALGO_NAMES

    batch = a.batch or DEFAULT_BATCH[a.net]
    layers = nets.NETS[a.net]()

    # ---- weights: generated on rank 0, broadcast once over RCCL (the only collective of this path) -------------------
    from feathercnn_amd.shard import broadcast_weights
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
        lyr = ConvLayer(prm, w, b)
        x = torch.rand((batch, c, h, h), device=dev, generator=gen) * 2 - 1
        built.append((layer, prm, lyr, x))
        max_scratch = max(max_scratch, lyr.buffer_bytes)
        max_out = max(max_out, batch * prm.output_channels * prm.output_h * prm.output_w)
    scratch = torch.empty(max(max_scratch // 4, 1), dtype=torch.float32, device=dev)  # one shared arena (mempool.cpp:88-92)
    out = torch.empty(max_out, dtype=torch.float32, device=dev)

    def eager_step():
        for _, prm, lyr, x in built:
            lyr.booster.Forward(prm, out, x, lyr.packed, scratch, lyr.bias)

    # One step = one hipGraph replay: the layer launches of a step are captured once (HIP graphs instead of a tracing
    # compiler); Forward never allocates and has no host-side state, so it is capturable as is.  --no-graph times the
    # eager launch loop instead.
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
    value = n_gpus * batch * a.steps / dt

    # ---- per-stage / per-layer HIP-event timing (separate pass, after the timed region) -------------------------------
    roofline = None
    table = []
    if rank == 0:
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
        if a.net == "mobilenet_v1" and dw_ms > 0:
            ach = dw_bytes / dw_ms / 1e6
            roofline = {"kernel": "depthwise3x3_lds_kernel", "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS,
                        "unit": "GB/s", "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None,
                        "note": "compulsory bytes 4*(C*Hin*Win + C*Ho*Wo)*N + 40*C summed over the 13 depthwise launches of a step"}
        elif gemm_ms > 0:
            ach = gemm_flops / gemm_ms / 1e9
            roofline = {"kernel": "gemm_mfma_kernel<WinoGemmPolicy> (Winograd tile GEMM)", "bound": "mfma", "achieved": round(ach, 2),
                        "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_MFMA_F32_TFLOPS, 4), "traffic": None,
                        "note": "algorithmic FLOPs 2*64*K*C*ceil(Ho/6)*ceil(Wo/6)*N summed over the Winograd layers of a step / "
                                "sum of their tile-GEMM HIP-event durations"}
        stage_ms = {k: round(v, 4) for k, v in stage_tot.items()}
        if roofline is not None:
            roofline.update(pmc_traffic(a.net, roofline["bound"]))
    if world > 1:
        dist.barrier()

    if rank == 0:
        res = {
            "metric": "images/sec fp32 forward (conv stack) @224x224", "value": round(value, 2), "unit": "images/s",
            "n_gpus": n_gpus, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{a.net} conv layers ({len(layers)}), batch {batch} per GPU, 224x224x3, bias+ReLU fused, fp32",
                       "net": a.net, "per_gpu_batch": batch, "global_batch": batch * n_gpus, "parallelism": f"batch-shard x{n_gpus}",
                       "launch": "hipGraph replay per step" if graph_used else "eager launches"},
            "conv_gflops_per_s_direct": round(flops_direct_total * n_gpus / (ms_per_step * 1e6), 1) if rank == 0 else None,
            "conv_direct_frac_of_mfma_peak": round(flops_direct_total / (ms_per_step * 1e6) / 1e3 / PEAK_MFMA_F32_TFLOPS, 4),
            "stage_ms_per_step": stage_ms,
            "roofline": roofline,
        }
        if world > 1:
            res["weight_broadcast"] = {"ms": round(t_bcast * 1e3, 3), "bytes": bcast_bytes, "collective": "1 flat RCCL broadcast from rank 0"}
        if not a.no_cpu_baseline and n_gpus == 1:
            try:
                res["cpu_baseline"] = cpu_baseline(a.net, a.cpu_procs)
            except Exception as e:  # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"value": None, "error": repr(e)}
        if a.layers_out:
            os.makedirs(os.path.dirname(os.path.abspath(a.layers_out)), exist_ok=True)
            with open(a.layers_out, "w") as f:
                json.dump({"net": a.net, "batch": batch, "layers": table}, f, indent=1)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
