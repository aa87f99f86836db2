#!/usr/bin/env python
"""tools/res_layers.py <net> <size> [batch] -- per-layer eager HIP-event times of one net at one input size (bench.py's configuration), to locate the
layers whose plane-size-specialised kernels fall back at that size."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feathercnn_amd import model_zoo  # noqa: E402
from feathercnn_amd.net import Net  # noqa: E402

name, size = sys.argv[1], int(sys.argv[2])
batch = int(sys.argv[3]) if len(sys.argv) > 3 else {"vgg16": 32, "resnet50": 64, "mobilenet_v1": 256}[name]
p, b, i, o = model_zoo.MODELS[name](size=size)
net = Net(fusion=3, graph=False, tuned=True, concurrency=True, sub_batches=1)
net.LoadParam(p)
net.LoadWeights(b)
net.FeedInput(i, torch.rand((batch, 3, size, size), device="cuda") * 2 - 1)
for _ in range(3):
    net.Forward()
torch.cuda.synchronize()
acc = None
for _ in range(5):
    t = [q[3] for q in net.forward_timed()]
    acc = t if acc is None else [a + c for a, c in zip(acc, t)]
convs = net.conv_params()
print(f"# {name} {size}px b{batch}")
for k, ((typ, nm, algo), ms) in enumerate(zip(net.layers(), acc)):
    g = ""
    if k in convs:
        q, n = convs[k]
        g = f"C={q.input_channels} K={q.output_channels} H={q.input_h} k={q.kernel_h} s={q.stride_h} g={q.group}"
    print(f"{nm[:26]:26s} {typ[:12]:12s} {str(algo or ''):10s} {ms / 5 * 1e3:8.1f} us  {g}")
print(f"total {sum(acc) / 5:.3f} ms")
