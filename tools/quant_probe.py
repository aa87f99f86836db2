"""Does the grid of a 1x1 GEMM pay for its last partial round?  256 -> 1024 channels, batch 64, image sizes chosen so that the
number of 128x64 tiles walks across multiples of 768 (3 resident blocks on each of 256 CUs)."""
import numpy as np
import torch

from feathercnn_amd import ConvLayer, ConvParam

dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
C, K, batch = 256, 1024, 64
w = torch.from_numpy((rng.standard_normal((K, C, 1, 1)) / 16).astype(np.float32)).to(dev)
b = torch.from_numpy(rng.uniform(-0.1, 0.1, K).astype(np.float32)).to(dev)
for h, wd in [(12, 12), (12, 14), (12, 15), (12, 16), (14, 14), (10, 20), (13, 16), (12, 18), (15, 16), (16, 18), (18, 18)]:
    prm = ConvParam(output_channels=K, input_channels=C, input_h=h, input_w=wd, kernel_h=1, kernel_w=1, stride_h=1, stride_w=1, pad_left=0,
                    pad_right=0, pad_top=0, pad_bottom=0, group=1, bias_term=True, activation=1, batch=batch)
    l = ConvLayer(prm, w, b, tuned=True)
    x = torch.from_numpy(rng.uniform(-1, 1, (batch, C, h, wd)).astype(np.float32)).to(dev)
    out = torch.empty(l.out_shape(), dtype=torch.float32, device=dev)
    scratch = torch.empty(max(l.buffer_bytes // 4, 1), dtype=torch.float32, device=dev)
    for _ in range(5):
        l.Forward(x, out, scratch)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 50
    for _ in range(reps):
        l.Forward(x, out, scratch)
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / reps
    n = batch * h * wd
    tiles = (K // 128) * ((n + 63) // 64)
    fl = 2.0 * K * C * n
    print(f"{h:2d}x{wd:2d}: {tiles:5d} tiles = {tiles / 768:5.2f} rounds  {t * 1e3:7.1f} us  {fl / t / 1e9:6.1f} TF  {t * 1e3 / tiles * 768:6.1f} us per round-equivalent")
