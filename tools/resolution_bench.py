#!/usr/bin/env python
"""tools/resolution_bench.py -- the three benchmark nets at input sizes around 224 (bench.py's configuration: fusion 3, MI355X routing, graph):
images/s and the rate per pixel relative to 224 x 224.  Shows what the plane-size-specialised kernels' fall-backs cost."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feathercnn_amd import model_zoo  # noqa: E402
from feathercnn_amd.net import Net  # noqa: E402

BATCH = {"vgg16": 32, "resnet50": 64, "mobilenet_v1": 256}
SUB = {"mobilenet_v1": 2}


def rate(name, size, steps=20):
    p, b, i, o = model_zoo.MODELS[name](size=size)
    batch = BATCH[name]
    x = torch.rand((batch, 3, size, size), device="cuda") * 2 - 1
    net = Net(fusion=3, graph=True, tuned=True, concurrency=True, sub_batches=SUB.get(name, 1))
    net.LoadParam(p)
    net.LoadWeights(b)
    net.FeedInput(i, x)
    for _ in range(4):
        net.Forward()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        net.Forward()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    net.close()
    return batch * steps / dt


def main():
    out = {}
    for name in ("vgg16", "resnet50", "mobilenet_v1"):
        base = None
        for size in (224, 192, 256, 160, 288):
            if name == "vgg16" and size == 288:
                continue
            r = rate(name, size)
            base = base or r
            out[f"{name}@{size}"] = {"img_s": round(r, 1), "per_pixel_vs_224": round(r * size * size / (base * 224 * 224), 3)}
            print(f"{name:13s} b{BATCH[name]:<4d} {size:4d} px  {r:10.1f} img/s   {out[f'{name}@{size}']['per_pixel_vs_224']:.3f} of the 224-pixel rate per pixel", flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
