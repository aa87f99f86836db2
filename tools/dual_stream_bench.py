"""One Net at batch B against R replica Nets at batch B/R, each on its own stream (hipGraph replay), launched back to back:
do concurrent half-batch forwards fill the tails of the small kernels?   python tools/dual_stream_bench.py resnet50 64"""
import sys
import time

import numpy as np
import torch

from feathercnn_amd import model_zoo
from feathercnn_amd.net import Net

name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
p, b, i, o = model_zoo.MODELS[name]()
rng = np.random.default_rng(0)


def build(n):
    net = Net(fusion=3, graph=True, tuned=True, concurrency=True)
    net.LoadParam(p)
    net.LoadWeights(b)
    net.FeedInput(i, torch.from_numpy(rng.uniform(-1, 1, (n, 3, 224, 224)).astype(np.float32)).cuda())
    for _ in range(3):
        net.Forward()
    torch.cuda.synchronize()
    return net


def measure(nets, steps=50):
    for _ in range(5):
        for n in nets:
            n.Forward()
    for n in nets:
        pass
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        for n in nets:
            n.Forward()
    for n in nets:
        pass
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


for r in (1, 2, 4, 1, 2):
    nets = [build(batch // r) for _ in range(r)]
    t = measure(nets)
    print(f"{name} b{batch}: {r} replica(s) x b{batch // r}: {t * 1e3:.3f} ms per {batch} images = {batch / t:.0f} img/s")
    for n in nets:
        n.close()
