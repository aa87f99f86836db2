"""The N > 1 path on CPU: two gloo ranks shard a batch and receive rank 0's weights through the same helper bench.py uses on
RCCL (feathercnn_amd/shard.py).  The data path itself has no collective."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from feathercnn_amd.shard import broadcast_model, broadcast_weights, shard_range


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 64, 512, 513):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)  # every rank starts from DIFFERENT weights
        weights = [torch.rand(8, 4, 3, 3, generator=g), torch.rand(8, generator=g), None, torch.rand(16, 8, 1, 1, generator=g)]
        nbytes = broadcast_weights(weights, src=0)
        lo, hi = shard_range(10, rank, world)
        batch = torch.arange(10 * 3, dtype=torch.float32).reshape(10, 3)[lo:hi]
        digest = float(sum(t.double().sum() for t in weights if t is not None))
        # a data-path-free reduction only to CHECK the shards (the product path has no collective)
        s = torch.tensor([batch.sum().item()], dtype=torch.float64)
        dist.all_reduce(s)
        # whole-net flavour: only rank 0 draws the weights; the others get the .bin as one flat broadcast and their
        # host-side readers must accept it
        import hashlib

        from feathercnn_amd import model_zoo
        from feathercnn_amd.net import Net
        (param, blob, _, _), _, sent = broadcast_model(model_zoo.tiny_allsorts)
        net = Net()
        net.LoadParam(param)
        net.LoadWeights(blob)
        q.put((rank, nbytes, digest, lo, hi, float(s.item()), sent, hashlib.sha256(param + blob).hexdigest(), len(net.layers())))
    finally:
        dist.destroy_process_group()


def test_two_ranks_gloo_broadcast_and_shard():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, nb0, d0, lo0, hi0, s0, *m0), (r1, nb1, d1, lo1, hi1, s1, *m1) = res
    from feathercnn_amd import model_zoo
    assert m0 == m1 and m0[0] == len(model_zoo.tiny_allsorts()[1]) and m0[2] == 28, "rank 1 did not receive rank 0's model"
    assert nb0 == nb1 == (8 * 4 * 9 + 8 + 16 * 8) * 4
    assert d0 == d1, "rank 1 did not receive rank 0's weights"
    g = torch.Generator().manual_seed(100)
    want = float(torch.rand(8, 4, 3, 3, generator=g).double().sum() + torch.rand(8, generator=g).double().sum()
                 + torch.rand(16, 8, 1, 1, generator=g).double().sum())
    assert abs(d0 - want) < 1e-9
    assert (lo0, hi0, lo1, hi1) == (0, 5, 5, 10)
    assert s0 == s1 == float(np.arange(30).sum())


def test_rank_affinity_shares_numa_cores_between_local_ranks():
    """bench.py pins every rank's host thread to the cores of its GPU's NUMA node (8 processes on a two-socket host): the split is
    contiguous, disjoint between the ranks of a node, inside the allowed set, and never empty."""
    from feathercnn_amd.shard import parse_cpulist, pin_rank_to_gpu_numa_node, rank_cpus
    assert parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11] and parse_cpulist("") == []
    from feathercnn_amd.shard import format_cpulist
    node0 = parse_cpulist("0-63,128-191")
    flat = lambda c: c                       # every logical CPU a core of its own
    smt = lambda c: c % 128                  # the usual Linux numbering: CPU c and c + 128 are the two threads of core c
    shares = [rank_cpus(node0, 4, k, core_key=flat) for k in range(4)]
    assert all(len(s) == 32 for s in shares) and len(set().union(*map(set, shares))) == 128
    # SMT siblings stay with their core: no rank gets only the hyperthreads of a peer's cores
    shares = [rank_cpus(node0, 4, k, core_key=smt) for k in range(4)]
    assert all(len(s) == 32 for s in shares) and len(set().union(*map(set, shares))) == 128
    assert all({c % 128 for c in s if c < 128} == {c % 128 for c in s if c >= 128} for s in shares)
    assert shares[1] == list(range(16, 32)) + list(range(144, 160)) and format_cpulist(shares[1]) == "16-31,144-159"
    assert format_cpulist([0, 1, 2, 3, 8, 10, 11]) == "0-3,8,10-11" and format_cpulist([]) == ""
    assert rank_cpus(node0, 4, 1, allowed=set(range(8)), core_key=flat) == [2, 3]          # a restricted cpuset is respected
    assert rank_cpus([5], 3, 2) == [5] and rank_cpus([], 2, 0) == [] and rank_cpus(node0, 2, 2) == []
    import os
    before = os.sched_getaffinity(0)
    try:
        r = pin_rank_to_gpu_numa_node(1, 2, ["ffff:ff:1f.0", "ffff:ff:1e.0"])   # no such devices: falls back to an even split
        assert r["pinned"] and r["count"] >= 1 and set(os.sched_getaffinity(0)) <= set(before)
    finally:
        os.sched_setaffinity(0, before)
