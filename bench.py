#!/usr/bin/env python
"""bench.py -- images/s of the fp32 forward pass on MI355X, per the driver contract.

A "step" is one forward pass of a benchmark network over one synthetic batch already resident in HBM.

  python bench.py [--gpus N] [--steps K] [--warmup W]
      The headline `value` is BASELINE.json configs[1]: VGG-16, 32 images per GPU, fp32, 224x224, the whole network through the
      feather::Net runtime (every convolution through the ConvBooster hot path plus the layers between them), timed exactly as the
      contract says: W untimed steps, then K steps between barrier + synchronize, max over ranks; weak scaling at N > 1.
      The SAME process then times the other nets BASELINE.json's metric names with the same procedure:  N = 1: ResNet-50 b64
      (configs[2]), MobileNet-V1 b256 (configs[3]) and ResNet-50 with 512 images (the one-GPU point of configs[4]);  N > 1: configs[4]
      itself -- ResNet-50, 512 images in total sharded over the ranks (strong scaling) -- its weak point (64 per GPU), and the one-GPU
      points of both curves (rank 0 alone), so that weak AND strong efficiency come out of ONE run.
  python bench.py --net resnet50|mobilenet_v1|vgg16|squeezenet_v1.1 [--batch B | --global-batch G] [--fusion 0|1|2|3]
      Only that net (its images/s becomes `value`).
  python bench.py --mode convstack [--net ...]
      Only the convolution layers, each through ConvBooster::Forward with bias + ReLU fused and its input re-drawn (not chained).

Rank 0 prints ONE JSON line of at most 8 KiB (benchkit/report.py enforces it): the contract keys, `config` (with `other_nets`: images/s and
the roofline fraction of each hot kernel of the other metric nets, and `resnet50_scaling`: weak and strong efficiency with their
definitions), `roofline` (the dominant kernel of the headline net -- the Winograd tile GEMM on fp32 MFMA for VGG-16 -- with `frac` = the
median of >= 5 eager attribution passes next to frac_min / frac_max / frac_passes, `traffic` from the committed PMC digest, `also` = the
other hot kernels) and `cpu_baseline` (the REAL reference runtime, oracle/_ref, on this host's cores).  Everything long -- every roofline
row with its notes, per-layer tables, the CPU sweep -- goes to the side file named by `detail` (--detail-out, default bench_detail.json).

Multi-GPU: the batch dimension is sharded; the model is generated on rank 0 and its .bin broadcast once over RCCL/xGMI, every rank runs its
own Init; there is no steady-state collective.  `python bench.py --gpus N` without a launcher starts its N ranks itself; under a launcher
--gpus must equal WORLD_SIZE.  A node with fewer than N GPUs is refused (FHIP_BENCH_SHARE_GPU=1: one-GPU rehearsal over gloo).

The work is in benchkit/ (timing, attribution, roofs, cpu, convstack, launcher, report); this file parses the flags and runs the sequence.
"""
from __future__ import annotations

import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from benchkit import DEFAULT_BATCH, PEAK_HBM_GBS, PEAK_MFMA_F32_TFLOPS, STEADY_STEPS, SUB_BATCHES  # noqa: E402,F401  (re-exported for tools/)
from benchkit import report  # noqa: E402
from benchkit.launcher import launch_ranks as _launch_ranks  # noqa: E402

ATTRIBUTION_PASSES = 7  # eager passes behind every roofline row; roofline.frac = their median (>= 5)


def launch_ranks(n):
    return _launch_ranks(n, os.path.abspath(__file__))


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
    ap.add_argument("--detail-out", default="", help="side file for everything that is not in the stdout line (default: bench_detail.json in the "
                    "working directory; the line names it under `detail`)")
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


def solo(fn, extras, name):
    """A measurement rank 0 takes ALONE inside an N > 1 run (the other ranks wait at the next barrier): an exception here must not keep rank 0
    from that barrier, so it is recorded in the line instead."""
    try:
        extras[name], _ = fn()
    except Exception as e:  # noqa: BLE001
        extras[name] = {"error": repr(e)[:200]}


def main():
    a = parse()
    if "WORLD_SIZE" not in os.environ and (a.gpus or 1) > 1:
        raise SystemExit(launch_ranks(a.gpus))
    import torch
    import torch.distributed as dist

    from benchkit import timing
    from benchkit.roofs import TRAFFIC_NOTE, sustained_mfma

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
    steady = 0 if a.no_steady else STEADY_STEPS
    quick = (max(a.steps // 5, 5), max(a.warmup // 2, 1))  # steps / warm-up of the 512-image one-GPU points (30 ms per step)

    if a.mode == "convstack":
        from benchkit.convstack import setup_convstack
        a.net = head_net
        step, finalize, batch = setup_convstack(a, env)
        dt = timing.timed_region(step, a.steps, a.warmup, env)
        head = finalize(dt / a.steps * 1e3)
        total = a.global_batch if a.global_batch else world * batch
        head.update({"net": head_net, "images_per_s": round(total * a.steps / dt, 2), "ms_per_step": round(dt / a.steps * 1e3, 4),
                     "per_gpu_batch": batch, "global_batch": total, "scaling": "strong" if a.global_batch else "weak"})
        model, extras = None, {}
    else:
        # ---- headline: BASELINE.json configs[1] (VGG-16, 32 images per GPU) unless --net says otherwise; weak scaling at N > 1
        head, model = timing.measure_net(head_net, a, env, a.steps, a.warmup, a.global_batch, a.batch, steady=steady, passes=ATTRIBUTION_PASSES)
        extras = {}
        if not explicit and not a.headline_only:
            # ---- the other nets BASELINE.json's metric names, same process, same timing procedure
            if world == 1:
                for name in ("resnet50", "mobilenet_v1"):  # not the contract's headline: never fewer than 100 timed steps
                    extras[name], _ = timing.measure_net(name, a, env, max(a.steps, 100), a.warmup, steady=steady, passes=ATTRIBUTION_PASSES)
                # the one-GPU point of configs[4]'s strong-scaling curve (ResNet-50, 512 images in total)
                extras["resnet50_global512"], _ = timing.measure_net("resnet50", a, env, *quick, global_batch=512, detail=False)
            else:
                # configs[4]: ResNet-50, 512 images in total sharded over the ranks (strong), plus its weak point (64 per GPU)
                extras["resnet50_global512"], r50_model = timing.measure_net("resnet50", a, env, a.steps, a.warmup, global_batch=512, passes=ATTRIBUTION_PASSES)
                extras["resnet50"], _ = timing.measure_net("resnet50", a, env, a.steps, a.warmup, detail=False)
                # the ONE-GPU points of both curves, in the same run: rank 0 alone (its env says world = 1 and it already holds the model, so
                # there is no collective inside); the other ranks wait at the barrier below
                if rank == 0:
                    one = dict(env, world=1)
                    solo(lambda: timing.measure_net("resnet50", a, one, *quick, global_batch=512, detail=False, model=r50_model), extras, "resnet50_global512_one_gpu")
                    solo(lambda: timing.measure_net("resnet50", a, one, a.steps, a.warmup, detail=False, model=r50_model), extras, "resnet50_one_gpu")
                    solo(lambda: timing.measure_net(head_net, a, one, a.steps, a.warmup, detail=False, model=model), extras, head_net + "_one_gpu")
                dist.barrier()
    shard_ok = None
    if world > 1:
        if a.mode == "net":
            # errors of a rank's own run are caught inside and agreed on with an all_reduce before the gather (no mismatched collectives)
            shard_ok = timing.shard_check(head_net, model, a, env)
        dist.barrier()

    if rank == 0:
        cpu = None
        if not a.no_cpu_baseline and world == 1:
            from benchkit import cpu as cpu_leg
            try:
                cpu = cpu_leg.cpu_baseline(head_net, a.cpu_procs) if a.mode == "convstack" else cpu_leg.net_cpu_baseline(head_net, model, a.cpu_procs)
            except Exception as e:  # the baseline must never take the GPU number down with it
                cpu = {"value": None, "error": repr(e)}
        from feathercnn_amd import provenance
        tree = provenance.tree_head()
        args = {"mode": a.mode, "steps": a.steps, "warmup": a.warmup, "no_graph": a.no_graph, "reference_selection": a.reference_selection,
                "no_overlap": a.no_overlap, "fusion": a.fusion}
        calib = dict(sustained_mfma(), note="fhip_calibrate_mfma_f32: a kernel of nothing but v_mfma_f32_32x32x2_f32 chains at 3 waves per SIMD on every "
                     "CU; its TFLOP/s and the shader clock it ran at (nominal peak 157.3 TFLOP/s assumes 2.4 GHz)")
        detail_path = a.detail_out or "bench_detail.json"
        line, detail = report.compose(head, extras, args=args, world=world, cpu=cpu, shard_ok=shard_ok, affinity=affinity, tree=tree,
                                      calibration=calib, detail_path=detail_path, traffic_note=TRAFFIC_NOTE)
        text = report.fit_line(line)  # <= 8 KiB, or optional keys are shed; never more
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                json.dump(detail, f, indent=1)
        except OSError as e:  # a read-only tree must not cost the line
            print(f"bench: could not write {detail_path}: {e!r}", file=sys.stderr)
        if a.manifest_out:
            os.makedirs(os.path.dirname(os.path.abspath(a.manifest_out)), exist_ok=True)
            with open(a.manifest_out, "w") as f:
                json.dump(dict(tree, fusion=a.fusion, argv=sys.argv[1:],
                               nets={k: {q: v[q] for q in ("per_gpu_batch", "sub_batches", "global_batch") if q in v} for k, v in detail["nets"].items()}), f, indent=1)
        if a.layers_out:
            os.makedirs(os.path.dirname(os.path.abspath(a.layers_out)), exist_ok=True)
            with open(a.layers_out, "w") as f:
                json.dump({"mode": a.mode, "tables": detail["tables"]}, f, indent=1)
        print(text, flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
