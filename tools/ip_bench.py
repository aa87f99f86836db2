"""tools/ip_bench.py -- InnerProduct shapes of VGG-16 (fc6 / fc7 / fc8) through fhip_conv_forward at batch 32: microseconds per call (HIP events, 50 calls).
Run under `rocprofv3 --kernel-trace --stats` to split a call into its kernels."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from feathercnn_amd import ConvLayer, ConvParam

dev = torch.device("cuda:0")
for name, c, k in (("fc6", 25088, 4096), ("fc7", 4096, 4096), ("fc8", 4096, 1000)):
    for batch in (32, 33):  # 33: the LDS-tiled kernel (the route stops at 32)
        p = ConvParam.make(c, k, 1, 1, 1, 0, bias=True, act=1, batch=batch)
        w = torch.rand((k, c, 1, 1), device=dev) - 0.5
        b = torch.rand((k,), device=dev)
        layer = ConvLayer(p, w, b)
        x = torch.rand((batch, c, 1, 1), device=dev)
        for _ in range(3):
            layer.Forward(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        layer.booster  # noqa
        from feathercnn_amd.booster import _stream  # the library launches on torch's current stream
        e0.record()
        for _ in range(50):
            layer.Forward(x)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 50 * 1e3
        print(f"{name} batch {batch}: {us:8.1f} us per call, weights {4.0 * c * k / us / 1e6:7.0f} GB/s")
