"""Batch-dimension sharding of the conv hot path across the GPUs of one node.

Inference over a batch has no cross-image dependency (the reference is literally per image,
reference src/layers/conv_layer.h:107), so the data path needs NO collective: rank r of W computes images
[r*N/W, (r+1)*N/W).  The only exchange is the one-time broadcast of the raw weights from rank 0
(RCCL over xGMI on the GPU box, gloo in the CPU tests); every rank then runs its own ConvBooster::Init.
"""
from __future__ import annotations


def shard_range(global_batch: int, rank: int, world: int):
    """[lo, hi) of the images rank `rank` owns; sizes differ by at most one and cover the batch exactly."""
    if not (0 <= rank < world) or global_batch < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(global_batch, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def broadcast_weights(tensors, src: int = 0):
    """Broadcast every tensor of `tensors` (in place) from rank `src`.  One flat message per dtype/device group keeps
    the number of collectives (and xGMI ring latencies) small: a net's conv weights go out as a single buffer."""
    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return 0
    tensors = [t for t in tensors if t is not None]
    if not tensors:
        return 0
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.broadcast(flat, src)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view_as(t))
        off += n
    return flat.numel() * flat.element_size()


def broadcast_model(build, device=None, src: int = 0):
    """Whole-net flavour of the same exchange: rank `src` builds the synthetic model (``build()`` ->
    (param, bin, input, output), feathercnn_amd/model_zoo.py), every other rank builds only the .param and the .bin's size
    (``build(dry=True)``) and receives the .bin as ONE flat byte broadcast.  -> (model, seconds, bytes broadcast)."""
    import time

    import torch
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size() == 1:
        return build(), 0.0, 0
    rank = dist.get_rank()
    if rank == src:
        model = build()
        blob = torch.frombuffer(bytearray(model[1]), dtype=torch.uint8)
    else:
        param, nbytes, i, o = build(dry=True)
        blob = torch.empty(nbytes, dtype=torch.uint8)
    if device is not None:
        blob = blob.to(device)
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    dist.broadcast(blob, src)
    if device is not None:
        torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if rank != src:
        model = (param, blob.cpu().numpy().tobytes(), i, o)
    return model, dt, blob.numel()
