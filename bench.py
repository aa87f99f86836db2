#!/usr/bin/env python
"""bench.py -- images/s of the fp32 forward pass on MI355X, per the driver contract.

A "step" is one forward pass of a benchmark network over one synthetic batch already resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
      The headline `value` is BASELINE.json configs[1]: VGG-16, 32 images per GPU, fp32, 224x224, the whole network through the
      feather::Net runtime (every convolution through the ConvBooster hot path plus the layers between them), timed exactly as the
      contract says: W untimed steps, then K steps between barrier + synchronize, max over ranks; weak scaling at N > 1.
      The SAME process then times the other nets BASELINE.json's metric names, with the same procedure, and reports them under
      "nets" / "rooflines":  N = 1: ResNet-50 b64 (configs[2]), MobileNet-V1 b256 (configs[3]) and ResNet-50 with 512 images (the
      one-GPU point of configs[4]);  N > 1: configs[4] itself -- ResNet-50, 512 images in total sharded over the ranks (strong
      scaling) -- and its weak point (64 per GPU).
  python bench.py --net resnet50|mobilenet_v1|vgg16|squeezenet_v1.1 [--batch B | --global-batch G] [--fusion 0|1|2|3]
      Only that net (its images/s becomes `value`).
  python bench.py --mode convstack [--net ...]
      Only the convolution layers, each through ConvBooster::Forward with bias + ReLU fused and its input re-drawn (not chained).

Rank 0 prints ONE JSON line.  Besides the contract keys it carries
  roofline     : the dominant kernel of the headline net (Winograd tile GEMM on fp32 MFMA for VGG-16);
  rooflines    : per net, every hot kernel priced against its roofline -- tile GEMM and 1x1 implicit GEMM (MFMA; the latter with
                 ConvParam::GetFLOPS), depthwise and the Winograd input transform (HBM).  achieved = algorithmic FLOPs (bytes) of the
                 kernel's launches in a step / their HIP-event durations on the launch stream, taken in eager passes of the same
                 step right after the timed region (a replayed hipGraph cannot carry per-kernel events);  traffic = null: PMC
                 counters cannot be read in-process, the rocprofv3 passes of this command are committed under profiles/;
  cpu_baseline : the REAL reference runtime (oracle/_ref: FeatherCNN's feather::Net + AVX2 booster compiled from /root/reference)
                 on this host, one single-thread process per core (its AVX Winograd is single-thread only), one image each.
Multi-GPU: the batch dimension is sharded; the model is generated on rank 0 and its .bin broadcast once over RCCL/xGMI, every
rank runs its own Init; there is no steady-state collective.  `python bench.py --gpus N` without a launcher starts its N ranks itself
(launch_ranks: re-execution under torch.distributed.run, rank 0's JSON line is the only stdout); under a launcher --gpus must equal
WORLD_SIZE.  A node with fewer than N GPUs is refused (FHIP_BENCH_SHARE_GPU=1: one-GPU rehearsal over gloo).
Where to look first: `config.other_nets` (ResNet-50 b64, MobileNet-V1 b256, ResNet-50 with 512 images: images/s + the fraction of each hot
kernel) and `roofline.also` repeat, inside the two objects every consumer keeps, what `nets` / `rooflines` hold in full;
`nets.resnet50_global512.expected_from_1gpu` states config 5's expected strong-scaling efficiency from the one-GPU figures.
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
# sub-batch replicas of the net per GPU (fhip_net_set_sub_batches), measured with tools/dual_stream_bench.py: MobileNet-V1 b256 gains 8 %
# with two (its HBM-bound depthwise kernels run under the other share's MFMA-bound 1x1 kernels), VGG-16 and ResNet-50 gain nothing
SUB_BATCHES = {"mobilenet_v1": 2}
STEADY_STEPS = 200  # length of the cross-check region timed after the contract's K steps ("steady_state" in the JSON line)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (default: WORLD_SIZE when launched by torch.distributed.run, "
                    "else 1).  Started WITHOUT a launcher and N > 1, bench.py starts the N ranks itself (launch_ranks)")
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--net", default=None, choices=list(DEFAULT_BATCH), help="time ONLY this net (default: VGG-16 as the headline value, then ResNet-50 and MobileNet-V1 in the same process)")
    ap.add_argument("--headline-only", action="store_true", help="skip the additional nets of the default run")
    ap.add_argument("--batch", type=int, default=0, help="per-GPU batch (default: the BASELINE.json config)")
    ap.add_argument("--global-batch", type=int, default=0, help="fix the TOTAL batch instead (strong scaling, SURVEY.md 8d config 5: "
                    "--net resnet50 --global-batch 512 --gpus 8); rank r takes shard_range(global, r, N) images")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-steady", action="store_true", help="skip the 200-step steady-state cross-check region (profiling runs: every extra step is traced)")
    ap.add_argument("--no-graph", action="store_true", help="time eager launches instead of one hipGraph replay per step")
    ap.add_argument("--cpu-procs", type=int, default=0, help="processes for the CPU baseline (default: one per host core, bounded by free memory)")
    ap.add_argument("--layers-out", default="", help="write the per-layer table (JSON) here")
    ap.add_argument("--manifest-out", default="", help="write what ran (tree head, source fingerprint, per-net batch / replicas, fusion level) here: "
                    "tools/profile.sh hands it to tools/summarize_prof.py, which stamps traffic.json with it")
    ap.add_argument("--mode", default="net", choices=["net", "convstack"])
    ap.add_argument("--no-overlap", action="store_true", help="net mode: keep every layer on one stream (no branch concurrency)")
    ap.add_argument("--reference-selection", action="store_true",
                    help="route convolutions with the reference's SelectAlgo rule instead of the MI355X cost model (fhip_conv_select_algo_tuned)")
    ap.add_argument("--sub-batches", type=int, default=0, help="replicas of the net on streams of their own, each taking a share of the batch "
                    "(fhip_net_set_sub_batches); 0 = per-net default: 2 for mobilenet_v1 (HBM-bound depthwise under MFMA-bound 1x1 layers), else 1")
    ap.add_argument("--fusion", type=int, default=3, help="net mode: 0 none, 1 the reference's TryFuse patterns, 2 + BN/Scale folded into conv weights, "
                    "conv+pool, conv+add, depthwise+pointwise, 3 + chained Winograd layers (feather_net.h, fhip_net_set_fusion)")
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
    procs = procs or ncpu
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


# ----------------------------------------------------------------------------------------------------------------------
def net_cpu_baseline(net_name, model, procs, budget=30.0):
    """The REAL reference runtime (feather::Net, AVX2) on this host's cores, SURVEY.md 8(d): the model is loaded once in a helper
    process (oracle/cpu_bench.py: no torch, no HIP), which fork()s P single-thread workers pinned to distinct cores -- the weights are
    shared copy-on-write -- for P in {1, 8, 16, 32, 64, 128, host cores}; every worker does 1 warm-up + 3 timed forwards of one image.
    Reported: the best aggregate of the sweep with its P, the whole sweep, the one-core figure.  Bounded to ~`budget` seconds."""
    import subprocess
    import tempfile

    from oracle import netcheck
    p, b, i, o = model
    if not netcheck.have_ref_net():  # no compiled reference here: the restatement, one image, one core
        import numpy as np
        port = netcheck.PortNet(p, b)
        x = np.random.default_rng(7).uniform(-1, 1, (1, 3, 224, 224)).astype(np.float32)
        t0 = time.perf_counter()
        port.run(i, x, o)
        dt = time.perf_counter() - t0
        return {"value": round(1.0 / dt, 3), "unit": "images/s", "cores": 1, "kind": "port",
                "sample": f"{net_name} whole net through the numpy/C restatement, 1 image, 1 process, {dt:.1f}s"}
    with tempfile.TemporaryDirectory() as d:
        pp, bp = os.path.join(d, "m.param"), os.path.join(d, "m.bin")
        open(pp, "wb").write(p)
        open(bp, "wb").write(b)
        cmd = [sys.executable, "-m", "oracle.cpu_bench", "--param", pp, "--bin", bp, "--input", i, "--output", o, "--budget", str(budget)]
        if procs:
            cmd += ["--procs", f"1,{procs}"]
        env = dict(os.environ, OMP_NUM_THREADS="1")
        t0 = time.perf_counter()
        out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=budget * 4 + 120)
        wall = time.perf_counter() - t0
    if out.returncode != 0:
        raise RuntimeError("cpu baseline helper failed: " + out.stderr[-400:])
    r = json.loads(out.stdout.strip().splitlines()[-1])
    best = r["best"]
    one = next((s_ for s_ in r["sweep"] if s_["procs"] == 1), None)
    return {"value": best["images_per_s"], "unit": "images/s", "cores": best["procs"], "kind": "reference",
            "sample": f"{net_name} whole net through the reference feather::Net (N = 1, no fusion: the reference never runs its fusion pass), "
                      f"1 image per process; model loaded once, then P fork()ed single-thread workers pinned to distinct cores (weights shared "
                      f"copy-on-write; the reference's AVX Winograd is single-thread only); {r['warmup']} warm-up + {r['reps']} timed forwards "
                      f"per worker, aggregate = sum of 1 / mean forward time; best of the sweep P = {[s_['procs'] for s_ in r['sweep']]}"
                      + (f" (P = {r['skipped']} not run in the sweep: time cap {r['budget_s']:.0f}s or aggregate already under half of the best)" if r["skipped"] else "")
                      + (f"; P = nproc = {r['nproc_point']['procs']} run once outside the sweep (nproc_point: 1 warm-up + 1 timed forward per worker)"
                         if (r.get("nproc_point") or {}).get("outside_sweep") else "")
                      + f"; {r['sweep_s']:.1f}s sweep + {r['load_s']:.1f}s load, {wall:.1f}s wall",
            "sweep": r["sweep"], "single_core_images_per_s": one["images_per_s"] if one else None, "cpu_model": r["cpu_model"],
            "host_cores": r["host_cores"],
            # SURVEY.md 8(d) names P = nproc; that point is kept here next to the best of the sweep
            "nproc_images_per_s": (r.get("nproc_point") or {}).get("images_per_s"), "nproc_point": r.get("nproc_point"),
            "why_best_is_not_nproc": "every process streams the whole model (VGG-16: 550 MB of weights, re-read per image at N = 1) and its own "
                                     "Winograd scratch through a memory system shared by all cores of the two sockets: the aggregate peaks where "
                                     "that saturates (the sweep shows where) and falls beyond it; hardware threads past the physical cores add nothing"}


# ----------------------------------------------------------------------------------------------------------------------
_SUSTAINED = {}


def sustained_mfma():
    """fhip_calibrate_mfma_f32, once per process: what a kernel made of nothing but fp32 MFMAs reaches on THIS device (the chip clocks
    to its power budget under full-chip matrix load), and the shader clock it ran at."""
    if not _SUSTAINED:
        from feathercnn_amd import booster
        try:
            tf, mhz = booster.calibrate_mfma_f32()
            _SUSTAINED.update({"tflops": round(tf, 1), "shader_mhz": round(mhz)})
        except Exception as e:  # a measurement aid must never take the benchmark down
            _SUSTAINED.update({"tflops": None, "error": repr(e)})
    return _SUSTAINED


def roofline_mfma(kernel, flops, ms, note):
    ach = flops / ms / 1e9
    r = {"kernel": kernel, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s",
         "frac": round(ach / PEAK_MFMA_F32_TFLOPS, 4), "traffic": None, "work_per_step": flops, "ms_per_step": round(ms, 4), "note": note}
    sus = sustained_mfma()
    if sus.get("tflops"):
        # next to the nominal peak (2.4 GHz): the measured ceiling of a pure-MFMA kernel on this device in this process
        r["sustained_peak_measured"] = sus["tflops"]
        r["shader_mhz_under_mfma_load"] = sus["shader_mhz"]
        r["frac_of_sustained"] = round(ach / sus["tflops"], 4)
    return r


def roofline_hbm(kernel, nbytes, ms, note):
    ach = nbytes / ms / 1e6
    return {"kernel": kernel, "bound": "hbm", "achieved": round(ach, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(ach / PEAK_HBM_GBS, 4), "traffic": None, "work_per_step": nbytes, "ms_per_step": round(ms, 4), "note": note}


# kernels behind each roofline row, as rocprofv3 names them (profiles/<round>_<net>/traffic.json keys)
TRAFFIC_KERNELS = {"Winograd tile GEMM": ("wino_gemm_glds", "WinoGemmPolicy"), "1x1 implicit GEMM": ("ConvGemmPolicy<1", "ConvGemmPolicy<2", "ConvGemmPolicy<5", "stream_gemm_kernel"),
                   "depthwise": ("depthwise3x3_",), "fused depthwise 3x3 + 1x1": ("ConvGemmPolicy<3", "ConvGemmPolicy<4", "dwpw_band_kernel"),
                   "wino_input_": ("wino_input_staged_kernel", "wino_input_transform_kernel", "wino43_input_transform_kernel", "wino_input_from_first"), "wino_chain_kernel": ("wino_chain_kernel",)}


def _round_of(path):
    """profiles/r12_vgg16/traffic.json -> 12 (numeric, so r10 sorts after r9)."""
    import re
    m = re.match(r"r(\d+)_", os.path.basename(os.path.dirname(path)))
    return int(m.group(1)) if m else -1


def attach_traffic(net_name, roofs, batch=None, sub_batches=1, fusion=None):
    """roofline.traffic: HBM bytes per launch of the row's kernels from the rocprofv3 PMC passes of THIS command (2 * FETCH_SIZE + WRITE_SIZE,
    separate --pmc passes, the gfx950 correction of MI355X_MICROARCH.md; tools/profile.sh + tools/summarize_prof.py).  PMC counters cannot be
    read inside the benchmark process, so the figure comes from the digest committed under profiles/ (newest round that has one for this net)
    -- and ONLY when that digest describes the tree that is running: its `_meta.source_fingerprint` (sha256 over the kernel and runtime
    sources, feathercnn_amd/provenance.py) must equal the live one and its profiled batch / fusion level the measured ones.  Otherwise
    traffic stays null and `traffic_stale` says why.  `traffic_head` = git commit of the running tree (the digest is valid for it because
    the fingerprints are equal), `traffic_profiled_at` = the commit the profile was taken on.  Launch-weighted mean over the kernels of
    the row; `achieved` and `frac` stay live measurements."""
    import glob
    from feathercnn_amd import provenance
    # newest round first (numerically); within a round the single-stream profile (what the per-kernel attribution runs) before the replica one
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_{net_name}", "traffic.json")) +
                   glob.glob(os.path.join(ROOT, "profiles", f"r*_{net_name}_single_stream", "traffic.json")),
                   key=lambda q: (_round_of(q), q.endswith("_single_stream/traffic.json")))
    if not cands:
        return
    path = cands[-1]
    try:
        dig = json.load(open(path))
    except (OSError, ValueError):
        return
    meta = dig.get("_meta") or {}
    here = provenance.tree_head()
    stale = None
    if not meta.get("source_fingerprint"):
        stale = "the digest carries no source fingerprint (profiled before round 4)"
    elif meta["source_fingerprint"] != here["source_fingerprint"]:
        stale = f"profiled on sources {meta['source_fingerprint']} (commit {meta.get('git_head')}), running {here['source_fingerprint']}"
    else:
        prof = (meta.get("nets") or {}).get(net_name) or {}
        if batch is not None and prof.get("per_gpu_batch") not in (None, batch):
            stale = f"profiled at batch {prof.get('per_gpu_batch')}, measured at {batch}"
        elif fusion is not None and meta.get("fusion") not in (None, fusion):
            stale = f"profiled at fusion level {meta.get('fusion')}, measured at {fusion}"
    src = os.path.relpath(path, ROOT)
    for r in roofs:
        pats = next((v for k, v in TRAFFIC_KERNELS.items() if r["kernel"].startswith(k)), None)
        if not pats:
            continue
        if stale:
            r["traffic"] = None
            r["traffic_stale"] = f"{src}: {stale}"
            continue
        rows = [v for k, v in dig.items() if k != "_meta" and any(q in k for q in pats)]
        n = sum(v["launches_profiled"] for v in rows)
        if n:
            r["traffic"] = round(sum(v["hbm_bytes_per_launch"] * v["launches_profiled"] for v in rows) / n)
            r["traffic_unit"] = "HBM bytes per launch (launch-weighted mean over the row's kernels)"
            r["traffic_source"] = src + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command; 2*FETCH + WRITE)"
            r["traffic_head"] = here["git_head"]
            r["traffic_profiled_at"] = meta.get("git_head")
            r["traffic_fingerprint"] = here["source_fingerprint"]


TRAFFIC_NOTE = ("traffic (HBM bytes per launch from rocprofv3 PMC passes) cannot be collected inside this process: it is read from the digest of "
                "the same command committed under profiles/ (traffic_source) and attached only when the digest's source fingerprint "
                "(feathercnn_amd/provenance.py: sha256 over the kernel + runtime sources) equals the running tree's and the profiled batch / "
                "fusion level are the measured ones -- otherwise traffic is null and traffic_stale says why; traffic_head = commit of the running "
                "tree, traffic_profiled_at = commit the counters were collected on")


def attribute(net, reps):
    """After the timed region: eager forwards with HIP events on the launch stream around every kernel (stage timers of the C-ABI)
    and around every layer, joined with each convolution's geometry as it runs -> per-kernel algorithmic work / measured time.
    (The timed steps replay a hipGraph, which cannot carry per-kernel events.)"""
    from feathercnn_amd import ALGO_NAMES, DEPTHWISE, IM2COL, WINOGRADF63, booster
    algo_id = {v: k for k, v in ALGO_NAMES.items()}
    net.set_graph(False)
    booster.stage_timing(True)
    booster.stage_timing_collect()
    for _ in range(reps):
        net.Forward()
    st = booster.stage_timing_collect()
    booster.stage_timing(False)
    stage = {k: v[0] / reps for k, v in st.items() if v[1]}
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


# ----------------------------------------------------------------------------------------------------------------------
def per_gpu_batch(net_name, a, env, global_batch=0, batch=0):
    """Weak scaling: the configured per-GPU batch.  Strong scaling (global batch): this rank's shard of the total."""
    if global_batch:
        from feathercnn_amd.shard import shard_range
        lo, hi = shard_range(global_batch, env["rank"], env["world"])
        if hi - lo < 1:
            raise SystemExit("bench: the global batch is smaller than the number of GPUs")
        return hi - lo
    return batch or DEFAULT_BATCH[net_name]


def timed_region(step, steps, warmup, env):
    """The contract's timing: W untimed steps, then exactly K steps bracketed by barrier + synchronize, max over ranks."""
    import torch
    import torch.distributed as dist
    world, dev = env["world"], env["dev"]

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def measure_net(net_name, a, env, steps, warmup, global_batch=0, batch=0, detail=True, steady=0, model=None):
    """One benchmark network through the feather::Net runtime: build (rank 0) + one RCCL broadcast of the .bin, timed region,
    per-kernel attribution.  -> result dict (rank 0 carries the detail).  `model` given: no broadcast (the caller already holds the
    model -- the one-rank reference point that rank 0 times by itself inside an N > 1 run, env["world"] == 1 there)."""
    import numpy as np
    import torch

    from feathercnn_amd import model_zoo
    from feathercnn_amd.net import Net
    from feathercnn_amd.shard import broadcast_model
    dev, rank, world = env["dev"], env["rank"], env["world"]
    nb = per_gpu_batch(net_name, a, env, global_batch, batch)
    if model is None:
        model, t_bcast, bcast_bytes = broadcast_model(model_zoo.MODELS[net_name], dev, src=0)
    else:
        t_bcast, bcast_bytes = 0.0, 0
    p, b, in_name, out_name = model
    replicas = a.sub_batches if a.sub_batches > 0 else SUB_BATCHES.get(net_name, 1)
    replicas = max(1, min(replicas, nb))

    def make_net(r):
        n_ = Net(fusion=a.fusion, graph=not a.no_graph, tuned=not a.reference_selection, concurrency=not a.no_overlap, sub_batches=r)
        n_.LoadParam(p)
        n_.LoadWeights(b)
        return n_

    net = make_net(replicas)
    gen = torch.Generator(device=dev)
    gen.manual_seed(4321 + rank)
    x = torch.rand((nb, 3, 224, 224), device=dev, generator=gen) * 2 - 1
    net.FeedInput(in_name, x)
    net.Forward()  # Reshape + Init (weight upload and transforms) + first forward; graph capture happens here
    torch.cuda.synchronize()
    prob = net.Extract(out_name)
    if not np.isfinite(prob).all() or abs(float(prob[0].sum()) - 1.0) > 1e-3:
        raise SystemExit(f"bench: {net_name}: the net's output is not a probability vector")
    sustained_mfma()  # the device's pure-MFMA ceiling (reported next to every MFMA roofline) is measured BEFORE the timed region, once per process
    dt = timed_region(net.Forward, steps, warmup, env)
    total_images = global_batch if global_batch else world * nb
    res = {"net": net_name, "images_per_s": round(total_images * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps,
           "warmup": warmup, "per_gpu_batch": nb, "global_batch": total_images, "scaling": "strong" if global_batch else "weak",
           "sub_batches": replicas}
    if steady:
        # cross-check of the contract's K-step figure: the same bracketed region again, long enough (>= 0.5 s of GPU time) that clock
        # ramp and launch jitter average out.  `value` stays the K-step number.
        n2 = max(steady, steps)
        dt2 = timed_region(net.Forward, n2, 0, env)
        res["steady_state"] = {"steps": n2, "images_per_s": round(total_images * n2 / dt2, 2), "ms_per_step": round(dt2 / n2 * 1e3, 4)}
    if rank == 0 and detail:
        if replicas > 1:
            # kernels of concurrent replicas share the chip, so their individual durations are not a roofline measurement: the
            # per-kernel attribution runs the same batch through a single-stream net (same kernels, same shapes but the batch)
            net.close()
            net = make_net(1)
            net.FeedInput(in_name, x)
            net.Forward()
            torch.cuda.synchronize()
        att = attribute(net, max(3, min(steps, 5)))
        attach_traffic(net_name, att["rooflines"], batch=nb, sub_batches=replicas, fusion=a.fusion)
        n_model_layers = len(netcheck_layers(p))
        res["workload"] = (f"{net_name} whole net ({n_model_layers} layers in the model file, {len(net.layers())} after fusion level {a.fusion}), "
                           f"batch {nb} per GPU" + (f" as {replicas} concurrent sub-batch replicas of the net (fhip_net_set_sub_batches)" if replicas > 1 else "")
                           + ", 224x224x3, fp32, synthetic ncnn .param/.bin")
        res["conv_tflops_direct"] = round(att.pop("conv_direct_flops_per_step") * world / (dt / steps) / 1e12, 2)
        res.update(att)
        res["device_memory"] = net.memory()
        if world > 1:
            res["weight_broadcast"] = {"ms": round(t_bcast * 1e3, 3), "bytes": bcast_bytes, "collective": "1 flat RCCL broadcast of the .bin from rank 0"}
    net.close()
    del net, x
    torch.cuda.empty_cache()
    return res, model


def shard_check(net_name, model, a, env):
    """N > 1 only: the property the batch shard rests on (the reference runs one image at a time, src/layers/conv_layer.h:107, so images are
    independent).  Every rank draws the SAME seeded global batch (2 * world + 1 images: ragged shares), runs its shard_range of it through
    its own net (weights from the broadcast), the shards are gathered on rank 0 and compared with rank 0's run of the whole batch.
    -> {"global_batch", "max_norm_err", "ok"} on rank 0 (not timed, not part of `value`)."""
    import numpy as np
    import torch
    import torch.distributed as dist

    from feathercnn_amd.net import Net
    from feathercnn_amd.shard import shard_range
    dev, rank, world = env["dev"], env["rank"], env["world"]
    p, b, in_name, out_name = model
    G = 2 * world + 1
    x_all = np.random.default_rng(97).uniform(-1, 1, (G, 3, 224, 224)).astype(np.float32)

    def run(x):
        n_ = Net(fusion=a.fusion, graph=False, tuned=not a.reference_selection, concurrency=not a.no_overlap)
        n_.LoadParam(p)
        n_.LoadWeights(b)
        n_.FeedInput(in_name, torch.from_numpy(x).to(dev))
        n_.Forward()
        y = np.array(n_.Extract(out_name), dtype=np.float32).reshape(x.shape[0], -1)
        n_.close()
        return y
    lo, hi = shard_range(G, rank, world)
    cdev = dev if dist.get_backend() == "nccl" else torch.device("cpu")  # gloo (the one-GPU rehearsal) gathers host tensors
    mine, err_local = None, None
    try:
        mine = run(x_all[lo:hi])
    except Exception as e:  # caught HERE, so that every rank still reaches the collectives below in the same order
        err_local = repr(e)
    # agree on success BEFORE any data collective: a rank that failed must not leave the others waiting in all_gather
    flag = torch.tensor([0 if mine is None else mine.shape[1]], dtype=torch.int64, device=cdev)
    lo_flag = flag.clone()
    dist.all_reduce(lo_flag, op=dist.ReduceOp.MIN)
    if int(lo_flag.item()) == 0:
        return {"net": net_name, "ok": None, "error": err_local or "another rank failed its shard"} if rank == 0 else None
    width = mine.shape[1]
    pad = torch.zeros((G // world + 1, width), dtype=torch.float32, device=cdev)
    pad[:hi - lo] = torch.from_numpy(mine).to(cdev)
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    if rank != 0:
        return None
    whole = run(x_all)
    got = np.concatenate([parts[r][:shard_range(G, r, world)[1] - shard_range(G, r, world)[0]].cpu().numpy() for r in range(world)])
    err = float(np.abs(got - whole).max() / max(float(np.abs(whole).max()), 1e-30))
    return {"net": net_name, "global_batch": G, "shares": [shard_range(G, r, world)[1] - shard_range(G, r, world)[0] for r in range(world)],
            "max_norm_err": err, "ok": bool(err <= 1e-5),
            "what": "every rank's shard of one seeded batch, gathered, vs rank 0's run of the whole batch (same broadcast weights)"}


def netcheck_layers(param_text):
    return [ln for ln in param_text.decode().splitlines()[2:] if ln.strip()]


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


# ----------------------------------------------------------------------------------------------------------------------
def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N ranks here -- this process becomes
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>`,
    one rank per GPU over RCCL, exactly the command the driver contract names -- and hand its exit status on.  Rank 0's JSON line goes to
    this process's stdout unchanged.  Fails loudly when the node has fewer than N GPUs (FHIP_BENCH_SHARE_GPU=1: the one-GPU rehearsal,
    every rank on cuda:0 over gloo; its numbers mean nothing)."""
    import socket
    import subprocess

    import torch
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if os.environ.get("FHIP_BENCH_SHARE_GPU") != "1" and have < n:
        raise SystemExit(f"bench.py --gpus {n}: this node shows {have} GPU(s); one rank per GPU is the only mode that measures anything "
                         "(FHIP_BENCH_SHARE_GPU=1 rehearses the N-rank path on one GPU)")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print(f"bench: --gpus {n} without a launcher: starting {n} ranks ({' '.join(cmd[1:9])} ...)", file=sys.stderr, flush=True)
    # rank 0's JSON line is the ONLY thing this process prints on stdout; whatever else the ranks or their libraries write there (gloo's
    # connection notes in the one-GPU rehearsal, for one) goes to stderr
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    for line in proc.stdout:
        out = sys.stdout if line.startswith("{") else sys.stderr
        out.write(line)
        out.flush()
    return proc.wait()


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and (a.gpus or 1) > 1:
        raise SystemExit(launch_ranks(a.gpus))
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus is not None and a.gpus != world:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s) (WORLD_SIZE): the flag and the run must agree")
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
    env = {"dev": dev, "rank": rank, "world": world}
    affinity = None
    if world > 1 and not share:
        # one process per GPU on a two-socket host: this rank's host thread stays on the cores of its GPU's NUMA node
        from feathercnn_amd.shard import pin_rank_to_gpu_numa_node
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        try:
            ids = []
            for i in range(min(local_world, torch.cuda.device_count())):
                pr = torch.cuda.get_device_properties(i)
                ids.append(f"{getattr(pr, 'pci_domain_id', 0):04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0")
        except Exception:
            ids = []
        affinity = pin_rank_to_gpu_numa_node(local, local_world, ids)
    explicit = a.net is not None
    head_net = a.net or "vgg16"

    if a.mode == "convstack":
        a.net = head_net
        step, finalize, batch = setup_convstack(a, env)
        dt = timed_region(step, a.steps, a.warmup, env)
        head = finalize(dt / a.steps * 1e3)
        total = a.global_batch if a.global_batch else world * batch
        head.update({"net": head_net, "images_per_s": round(total * a.steps / dt, 2), "ms_per_step": round(dt / a.steps * 1e3, 4),
                     "per_gpu_batch": batch, "global_batch": total, "scaling": "strong" if a.global_batch else "weak"})
        model, extras = None, {}
    else:
        # ---- headline: BASELINE.json configs[1] (VGG-16, 32 images per GPU) unless --net says otherwise; weak scaling at N > 1
        head, model = measure_net(head_net, a, env, a.steps, a.warmup, a.global_batch, a.batch, steady=0 if a.no_steady else STEADY_STEPS)
        extras = {}
        if not explicit and not a.headline_only:
            # ---- the other nets BASELINE.json's metric names, same process, same timing procedure
            if world == 1:
                for name in ("resnet50", "mobilenet_v1"):  # not the contract's headline: never fewer than 100 timed steps
                    extras[name], _ = measure_net(name, a, env, max(a.steps, 100), a.warmup, steady=0 if a.no_steady else STEADY_STEPS)
                # the one-GPU point of configs[4]'s strong-scaling curve (ResNet-50, 512 images in total)
                extras["resnet50_global512"], _ = measure_net("resnet50", a, env, max(a.steps // 5, 5), max(a.warmup // 2, 1), global_batch=512,
                                                              detail=False)
            else:
                # configs[4]: ResNet-50, 512 images in total sharded over the ranks (strong), plus its weak point (64 per GPU)
                extras["resnet50_global512"], r50_model = measure_net("resnet50", a, env, a.steps, a.warmup, global_batch=512)
                extras["resnet50"], _ = measure_net("resnet50", a, env, a.steps, a.warmup, detail=False)
                # the ONE-GPU point of the same strong-scaling curve, in the same run: rank 0 alone runs all 512 images (no collective inside:
                # its env says world = 1 and it already holds the model); the other ranks wait at the barrier below
                if rank == 0:
                    extras["resnet50_global512_one_gpu"], _ = measure_net("resnet50", a, dict(env, world=1), max(a.steps // 5, 5), max(a.warmup // 2, 1),
                                                                          global_batch=512, detail=False, model=r50_model)
                dist.barrier()
    shard_ok = None
    if world > 1:
        if a.mode == "net":
            # errors of a rank's own run are caught inside and agreed on with an all_reduce before the gather (no mismatched collectives)
            shard_ok = shard_check(head_net, model, a, env)
        dist.barrier()

    if rank == 0:
        res = {
            "metric": "images/sec fp32 forward @224x224" + (" (conv stack)" if a.mode == "convstack" else ""),
            "value": head["images_per_s"], "unit": "images/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": head["scaling"], "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": head.get("workload", ""), "net": head_net, "mode": a.mode, "per_gpu_batch": head["per_gpu_batch"],
                       "global_batch": head["global_batch"], "parallelism": f"batch-shard x{world}",
                       "launch": "eager launches" if a.no_graph else "hipGraph replay per step",
                       "conv_routing": "reference SelectAlgo rule" if a.reference_selection else "fhip_conv_select_algo_tuned (Winograd also on 4..8-pixel 3x3 layers)",
                       "streams": "one" if (a.no_overlap or a.mode != "net") else "main + one side stream for arena-free branch convolutions"},
        }
        if affinity is not None:
            res["config"]["rank0_cpu_affinity"] = affinity
        if shard_ok is not None:
            res["shard_check"] = shard_ok
        table = head.pop("table", [])
        cpu_fn = head.pop("cpu_baseline_fn", None)
        for k in ("stage_ms_per_step", "layer_type_ms_per_step", "conv_tflops_direct", "conv_gflops_per_s_direct", "device_memory", "weight_broadcast"):
            if k in head:
                res[k] = head[k]
        roofs = head.get("rooflines", [])
        # the dominant kernel of the headline net: tile GEMM (MFMA) for VGG / ResNet, depthwise (HBM) for MobileNet
        dom = None
        for r in roofs:
            if head_net == "mobilenet_v1" and r["kernel"].startswith("depthwise"):
                dom = r
        res["roofline"] = dom or (roofs[0] if roofs else head.get("roofline"))
        res["rooflines"] = {head_net: roofs}
        res["traffic_note"] = TRAFFIC_NOTE
        res["mfma_calibration"] = dict(sustained_mfma(), note="fhip_calibrate_mfma_f32: a kernel of nothing but v_mfma_f32_32x32x2_f32 chains at 3 "
                                       "waves per SIMD on every CU; its TFLOP/s and the shader clock it ran at (nominal peak 157.3 TFLOP/s assumes 2.4 GHz)")
        nets_out = {head_net: {k: head[k] for k in ("images_per_s", "ms_per_step", "per_gpu_batch", "global_batch", "scaling", "sub_batches", "steady_state") if k in head}}
        tables = {head_net: table}
        for name, e in extras.items():
            nets_out[name] = {k: e[k] for k in ("images_per_s", "ms_per_step", "per_gpu_batch", "global_batch", "scaling", "sub_batches", "steps", "warmup", "steady_state") if k in e}
            for k in ("workload", "stage_ms_per_step", "layer_type_ms_per_step", "conv_tflops_direct", "device_memory"):
                if k in e:
                    nets_out[name][k] = e[k]
            if "rooflines" in e:
                res["rooflines"][name] = e["rooflines"]
            tables[name] = e.get("table", [])
        res["nets"] = nets_out
        # ---- compact per-net summary inside the two objects every consumer of this line keeps (`config`, `roofline`): the other metric nets
        # ---- of BASELINE.json next to the headline, with the fraction of each one's hot kernels (full detail stays in nets / rooflines)
        short = {"Winograd tile GEMM": "tile_gemm", "1x1 implicit GEMM": "gemm1x1", "depthwise:": "dw", "fused depthwise 3x3 + 1x1:": "dwpw_hbm",
                 "fused depthwise 3x3 + 1x1 (": "dwpw_mfma", "wino_input_": "wino_input", "wino_chain_kernel": "wino_chain"}

        def key_of(r):
            return next((v for k, v in short.items() if r["kernel"].startswith(k)), r["kernel"][:24])

        other, also = {}, {}
        for name, e in nets_out.items():
            tag = f"{name}_b{e['per_gpu_batch']}" if e.get("scaling") != "strong" else f"{name.split('_global')[0]}_g{e['global_batch']}" + ("_one_gpu" if name.endswith("_one_gpu") else "")
            row = {"img_s": e["images_per_s"], "ms_per_step": e["ms_per_step"]}
            if e.get("steady_state"):
                row["img_s_steady"] = e["steady_state"]["images_per_s"]
            for r in res["rooflines"].get(name, []):
                k = key_of(r)
                row[k + "_frac"] = r["frac"]
                if k == "gemm1x1":
                    row["gemm1x1_frac_of_tighter_bound"] = r.get("frac_of_tighter_bound")
                    if r.get("hbm_bound_layers"):
                        row["gemm1x1_hbm_bound_layers_frac_hbm"] = r["hbm_bound_layers"]["frac_hbm"]
                if k == "tile_gemm" and "frac_survey_8d_formula" in r:
                    row["tile_gemm_frac_8d_formula"] = r["frac_survey_8d_formula"]
                also.setdefault(tag, []).append({"kernel": k, "bound": r["bound"], "frac": r["frac"], "achieved": r["achieved"], "unit": r["unit"],
                                                 "ms_per_step": r["ms_per_step"]})
            if name != head_net or e.get("scaling") == "strong":
                other[tag] = row
            else:
                res["config"]["headline_net"] = dict(row, tag=tag)
        # configs[4] (ResNet-50, 512 images in total over 8 GPUs, strong scaling): the number the curve will be judged against, written down
        # BEFORE the 8-GPU node exists.  Definition: efficiency(n) = img/s(n GPUs, global batch 512) / (n x img/s(1 GPU, global batch 512)).
        # At n = 8 every GPU runs 64 images per step, so -- the data path has no collective -- the expected per-GPU rate is the 1-GPU rate
        # at batch 64, and the expected efficiency is img/s(b64) / img/s(b512) on one GPU.
        g512 = nets_out.get("resnet50_global512_one_gpu") if world > 1 else nets_out.get("resnet50_global512")
        b64 = nets_out.get("resnet50")
        if g512 and b64 and b64.get("per_gpu_batch") == 64:
            exp = {"definition": "strong-scaling efficiency at n GPUs = img/s(n GPUs, global batch 512) / (n x img/s(1 GPU, global batch 512))",
                   "one_gpu_global512_img_s": g512["images_per_s"], "one_gpu_b64_img_s": b64["images_per_s"] / (world if b64.get("scaling") == "weak" else 1),
                   "predicted_img_s_at_8_gpus": round(8 * b64["images_per_s"] / (world if b64.get("scaling") == "weak" else 1), 1),
                   "predicted_efficiency_at_8_gpus": round(b64["images_per_s"] / (world if b64.get("scaling") == "weak" else 1) / g512["images_per_s"], 4),
                   "per_gpu_b64_img_s_needed_for_0.9": round(0.9 * g512["images_per_s"], 1),
                   "why_below_1": "a GPU at 64 images per step runs shorter launches (block turnover, tails) than at 512: the loss is on-chip, not in a collective"}
            if world > 1 and "resnet50_global512" in nets_out:
                exp["measured_img_s_at_this_n"] = nets_out["resnet50_global512"]["images_per_s"]
                exp["measured_efficiency_at_this_n"] = round(nets_out["resnet50_global512"]["images_per_s"] / (world * g512["images_per_s"]), 4)
            nets_out["resnet50_global512"]["expected_from_1gpu"] = exp
            other.setdefault("resnet50_g512", {})["expected_from_1gpu"] = {k: exp[k] for k in exp if k not in ("definition", "why_below_1")}
        res["config"]["other_nets"] = other
        if res.get("roofline"):
            res["roofline"] = dict(res["roofline"], also=also)
        if not a.no_cpu_baseline and world == 1:
            try:
                if a.mode == "convstack":
                    res["cpu_baseline"] = cpu_baseline(head_net, a.cpu_procs)
                else:
                    res["cpu_baseline"] = net_cpu_baseline(head_net, model, a.cpu_procs)
            except Exception as e:  # the baseline must never take the GPU number down with it
                res["cpu_baseline"] = {"value": None, "error": repr(e)}
        from feathercnn_amd import provenance
        res["tree"] = provenance.tree_head()
        if a.manifest_out:
            os.makedirs(os.path.dirname(os.path.abspath(a.manifest_out)), exist_ok=True)
            with open(a.manifest_out, "w") as f:
                json.dump(dict(res["tree"], fusion=a.fusion, argv=sys.argv[1:],
                               nets={k: {q: v[q] for q in ("per_gpu_batch", "sub_batches", "global_batch") if q in v} for k, v in nets_out.items()}), f, indent=1)
        if a.layers_out:
            os.makedirs(os.path.dirname(os.path.abspath(a.layers_out)), exist_ok=True)
            with open(a.layers_out, "w") as f:
                json.dump({"mode": a.mode, "tables": tables}, f, indent=1)
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
