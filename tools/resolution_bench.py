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


def rate(name, size, steps=20, batch=None, layers=False):
    p, b, i, o = model_zoo.MODELS[name](size=size)
    batch = batch or BATCH[name]
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
    by = None
    if layers:  # eager per-layer HIP-event times, summed by layer type / route (each carries ~5 us of event overhead: compare like with like)
        if SUB.get(name, 1) > 1:  # kernels of concurrent replicas share the chip: the per-layer pass runs the batch through a single-stream net
            net.close()
            net = Net(fusion=3, graph=False, tuned=True, concurrency=True, sub_batches=1)
            net.LoadParam(p)
            net.LoadWeights(b)
            net.FeedInput(i, x)
            net.Forward()
        net.set_graph(False)
        by = {}
        for _ in range(3):
            for (typ, nm, algo), t in zip(net.layers(), net.forward_timed()):
                by[typ + ("/" + algo if algo else "")] = by.get(typ + ("/" + algo if algo else ""), 0.0) + t[3] / 3
    net.close()
    return (batch * steps / dt, by) if layers else batch * steps / dt


def iso_pixel():
    """The same nets with the batch scaled so that every size processes the pixels of the 224-px configuration: what is left of the
    per-pixel difference is the plane-size specialisation (kernel fall-backs), not the shorter launches of a smaller input."""
    out = {}
    for name in ("vgg16", "resnet50", "mobilenet_v1"):
        base = None
        for size in (224, 160, 192, 256, 288):
            if name == "vgg16" and size == 288:
                continue
            batch = max(1, round(BATCH[name] * 224 * 224 / (size * size)))
            r, by = rate(name, size, batch=batch, layers=True)
            px = r * size * size
            base = base or (px, by)
            out[f"{name}@{size}"] = {"batch": batch, "img_s": round(r, 1), "per_pixel_vs_224": round(px / base[0], 3)}
            worst = sorted(((by[k] / max(base[1].get(k, 0.0), 1e-9), k, by[k]) for k in by if base[1].get(k, 0.0) > 0.02), reverse=True)[:4]
            print(f"{name:13s} b{batch:<4d} {size:4d} px  {r:10.1f} img/s   {px / base[0]:.3f} of the 224-pixel rate per pixel at equal pixels per step;  "
                  "slowest layer classes vs 224 (ms ratio at equal pixels): " + ", ".join(f"{k} x{q:.2f} ({ms:.3f} ms)" for q, k, ms in worst), flush=True)
    print(json.dumps(out))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "iso":
        return iso_pixel()
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
