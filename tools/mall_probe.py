"""tools/mall_probe.py -- does VGG-16's conv1_2 (V and M of 757 MB each at batch 32) run faster in sub-batches whose V + M fit the 256 MB
memory-side cache?  conv1_1 (inside the input transform) -> conv1_2 -> pool -> conv2_1 through fhip_conv_forward_chained, batch 32 at once
against 2 x 16, 4 x 8 and 8 x 4 (the same memory reused by every sub-batch: torch's caching allocator hands the scratch back).
HIP events around the whole sequence, median of REPS."""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from feathercnn_amd import ConvLayer, ConvParam  # noqa: E402
from feathercnn_amd.booster import forward_chained  # noqa: E402

WINO = 4
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)


def layer(ic, oc, h):
    w = torch.from_numpy((rng.standard_normal((oc, ic, 3, 3)) / np.sqrt(9 * ic)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.uniform(-0.1, 0.1, oc).astype(np.float32)).to(dev)
    prm = ConvParam(output_channels=oc, input_channels=ic, input_h=h, input_w=h, kernel_h=3, kernel_w=3, stride_h=1, stride_w=1, pad_left=1,
                    pad_right=1, pad_top=1, pad_bottom=1, group=1, bias_term=True, activation=1)
    return ConvLayer(prm, w, b, algo=WINO), prm, w, b


def run(x, sub, stacks):
    outs = []
    for i in range(0, x.shape[0], sub):
        layers, first = stacks
        outs.append(forward_chained(layers, x[i:i + sub], pools=[True, False], first=first))
    return outs


def main():
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 15
    N = 32
    x = torch.from_numpy(rng.uniform(-1, 1, (N, 3, 224, 224)).astype(np.float32)).to(dev)
    _, p11, w11, b11 = layer(3, 64, 224)
    l12 = layer(64, 64, 224)[0]
    l21 = layer(64, 128, 112)[0]
    stacks = ([l12, l21], (p11, w11, b11))
    ref = torch.cat(run(x, N, stacks))
    for sub in (32, 16, 8, 4):
        got = torch.cat(run(x, sub, stacks))
        assert torch.equal(got, ref), sub
        # replayed as a graph: no host time between the launches
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run(x, sub, stacks)
            with torch.cuda.graph(g, stream=side):
                keep = run(x, sub, stacks)
        torch.cuda.current_stream().wait_stream(side)
        g.replay()
        torch.cuda.synchronize()
        assert torch.equal(torch.cat(keep), ref), ("graph", sub)
        times = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            g.replay()
            e1.record()
            torch.cuda.synchronize()
            times.append(e0.elapsed_time(e1))
        times.sort()
        print(f"sub-batch {sub:2d} x {N // sub}: median {times[len(times) // 2] * 1000:8.1f} us  min {times[0] * 1000:8.1f} us")


main()
