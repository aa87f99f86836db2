"""The contract's timed region, one benchmark network through the feather::Net runtime, and the N > 1 shard check."""
from __future__ import annotations

import time

from . import DEFAULT_BATCH, SUB_BATCHES
from .attribution import attribute
from .roofs import attach_traffic, sustained_mfma


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


def measure_net(net_name, a, env, steps, warmup, global_batch=0, batch=0, detail=True, steady=0, model=None, passes=5):
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
        att = attribute(net, max(5, passes))
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
