"""Whole-net timing on the GPU: images/s and a per-layer-type breakdown (HIP events around every layer).
usage: python tools/net_bench.py [net[:batch[:fusion]] ...]    e.g.  vgg16:32 resnet50:64:2 mobilenet_v1:256:2"""
import os
import sys
import time
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from feathercnn_amd import model_zoo  # noqa: E402
from feathercnn_amd.net import Net  # noqa: E402


def run(name, batch, fusion, steps=10):
    t0 = time.time()
    p, b, i, o = model_zoo.MODELS[name]()
    net = Net(fusion=fusion, graph=False, tuned=True)
    net.LoadParam(p)
    net.LoadWeights(b)
    del b
    x = torch.rand((batch, 3, 224, 224), device="cuda") * 2 - 1
    net.FeedInput(i, x)
    net.Forward()
    torch.cuda.synchronize()
    setup = time.time() - t0
    for _ in range(2):
        net.Forward()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(steps):
        net.Forward()
    torch.cuda.synchronize()
    eager = (time.time() - t) / steps
    timed = net.forward_timed()
    timed = net.forward_timed()
    by = defaultdict(float)
    for typ, nm, algo, ms in timed:
        by[typ + ("/" + algo if algo else "")] += ms
    tot = sum(by.values())
    print(f"== {name} b{batch} fusion={fusion}: eager {eager * 1e3:.3f} ms/step = {batch / eager:.0f} img/s   (setup {setup:.1f}s, "
          f"{len(timed)} layers, mem {net.memory()})")
    for k, v in sorted(by.items(), key=lambda kv: -kv[1]):
        print(f"   {k:28s} {v:8.3f} ms  {100 * v / tot:5.1f}%")
    worst = sorted(timed, key=lambda r: -r[3])[:8]
    print("   slowest:", ", ".join(f"{nm}({typ[:4]}) {ms:.3f}" for typ, nm, algo, ms in worst))
    net.close()
    net = Net(fusion=fusion, graph=True, tuned=True)
    net.LoadParam(p)
    net.LoadWeights(model_zoo.MODELS[name]()[1])
    net.FeedInput(i, x)
    for _ in range(3):
        net.Forward()
    torch.cuda.synchronize()
    t = time.time()
    for _ in range(steps):
        net.Forward()
    torch.cuda.synchronize()
    g = (time.time() - t) / steps
    print(f"   hipGraph replay: {g * 1e3:.3f} ms/step = {batch / g:.0f} img/s")
    net.close()


if __name__ == "__main__":
    specs = sys.argv[1:] or ["vgg16:32", "resnet50:64:2", "mobilenet_v1:256:2"]
    for s in specs:
        parts = s.split(":")
        run(parts[0], int(parts[1]) if len(parts) > 1 else 32, int(parts[2]) if len(parts) > 2 else 1)
