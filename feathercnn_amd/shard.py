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


# ---- host side of one-process-per-GPU: keep every rank's host thread (kernel launches, graph replays, input staging) on the cores of
# ---- the NUMA node its GPU hangs off, and away from the other ranks' cores (an 8-GPU MI355X node is a 2-socket host)
def parse_cpulist(text: str):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11] (the kernel's cpulist format)."""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.extend(range(int(lo), int(hi or lo) + 1))
    return cpus


def core_of(cpu: int):
    """Physical-core key of a logical CPU: the lowest thread sibling (topology/thread_siblings_list), or the CPU itself when the
    platform does not say (then every logical CPU counts as a core of its own)."""
    try:
        return min(parse_cpulist(open(f"/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list").read()))
    except (OSError, ValueError):
        return cpu


def rank_cpus(node_cpus, ranks_on_node: int, index_on_node: int, allowed=None, core_key=core_of):
    """The share of a NUMA node's cores that the `index_on_node`-th of `ranks_on_node` ranks gets, split by PHYSICAL core: Linux usually
    numbers SMT siblings as a second range ('0-31,128-159'), so a split of the sorted cpulist would hand one rank the physical cores and
    another their hyperthread siblings -- the same cores.  A rank gets whole cores (every allowed sibling of each); restricted to `allowed`
    (the CPUs the process may run on) and never empty while `allowed` leaves the node any CPU."""
    cpus = sorted(c for c in node_cpus if allowed is None or c in allowed)
    if not cpus or ranks_on_node < 1 or not (0 <= index_on_node < ranks_on_node):
        return []
    cores = {}
    for c in cpus:
        cores.setdefault(core_key(c), []).append(c)
    keys = sorted(cores)
    per = max(1, len(keys) // ranks_on_node)
    lo = min(index_on_node * per, len(keys) - per)
    return sorted(c for k in keys[lo:lo + per] for c in cores[k])


def format_cpulist(cpus):
    """[0, 1, 2, 3, 8, 10, 11] -> '0-3,8,10-11' (inverse of parse_cpulist)."""
    out, cpus = [], sorted(cpus)
    i = 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def gpu_numa_node(pci_bus_id: str):
    """NUMA node of a PCI device ('0000:75:00.0'), or -1 when the platform does not say."""
    try:
        return int(open(f"/sys/bus/pci/devices/{pci_bus_id.lower()}/numa_node").read().strip())
    except (OSError, ValueError):
        return -1


def pin_rank_to_gpu_numa_node(local_rank: int, local_world: int, pci_bus_ids):
    """os.sched_setaffinity for this rank: the cores of its GPU's NUMA node, split between the local ranks whose GPUs share that node.
    `pci_bus_ids[i]` = PCI address of local rank i's device.  Falls back to an even split of the allowed cores when the platform
    reports no NUMA nodes.  -> what was done (goes into bench.py's JSON line); never raises."""
    import os
    try:
        allowed = set(os.sched_getaffinity(0))
        nodes = [gpu_numa_node(b) for b in pci_bus_ids]
        mine = nodes[local_rank] if local_rank < len(nodes) else -1
        if mine >= 0:
            node_cpus = parse_cpulist(open(f"/sys/devices/system/node/node{mine}/cpulist").read())
            peers = [r for r in range(local_world) if r < len(nodes) and nodes[r] == mine]
            cpus = rank_cpus(node_cpus, len(peers), peers.index(local_rank), allowed)
            how = f"NUMA node {mine} of GPU {pci_bus_ids[local_rank]}, share {peers.index(local_rank) + 1}/{len(peers)}"
        else:
            cpus = rank_cpus(sorted(allowed), local_world, local_rank, allowed)
            how = f"no NUMA information: share {local_rank + 1}/{local_world} of the allowed cores"
        if cpus:
            os.sched_setaffinity(0, cpus)
        return {"pinned": bool(cpus), "cpus": format_cpulist(cpus), "count": len(cpus), "how": how}
    except Exception as e:  # affinity is an optimisation
        return {"pinned": False, "error": repr(e)}
