"""Per-stage times (K2 input transform / K3 tile GEMM / K4 output transform) of single Winograd layers.
Usage: python tools/wino_stage_bench.py [r50|vgg]"""
import sys

import numpy as np
import torch

from feathercnn_amd import ConvLayer, ConvParam
from feathercnn_amd.booster import WINOGRADF63, stage_timing, stage_timing_collect

which = sys.argv[1] if len(sys.argv) > 1 else "r50"
SHAPES = {"r50": [(64, 64, 56, 64), (128, 128, 28, 64), (256, 256, 14, 64), (512, 512, 7, 64)],
          "mall": [(64, 64, 224, 32), (64, 64, 224, 16), (64, 64, 224, 8), (64, 64, 224, 4), (64, 128, 112, 32), (64, 128, 112, 8), (128, 128, 112, 32), (128, 128, 112, 8)],
          "vgg": [(64, 64, 224, 32), (64, 128, 112, 32), (128, 128, 112, 32), (128, 256, 56, 32), (256, 256, 56, 32), (256, 512, 28, 32),
                  (512, 512, 28, 32), (512, 512, 14, 32)]}[which]
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
for ic, oc, h, batch in SHAPES:
    w = torch.from_numpy((rng.standard_normal((oc, ic, 3, 3)) / np.sqrt(9 * ic)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.uniform(-0.1, 0.1, oc).astype(np.float32)).to(dev)
    prm = ConvParam(output_channels=oc, input_channels=ic, input_h=h, input_w=h, kernel_h=3, kernel_w=3, stride_h=1, stride_w=1, pad_left=1,
                    pad_right=1, pad_top=1, pad_bottom=1, group=1, bias_term=True, activation=1, batch=batch)
    l = ConvLayer(prm, w, b, algo=WINOGRADF63)
    x = torch.from_numpy(rng.uniform(-1, 1, (batch, ic, h, h)).astype(np.float32)).to(dev)
    out = torch.empty(l.out_shape(), dtype=torch.float32, device=dev)
    scratch = torch.empty(l.buffer_bytes // 4, dtype=torch.float32, device=dev)
    for _ in range(3):
        l.Forward(x, out, scratch)
    torch.cuda.synchronize()
    stage_timing(True)
    stage_timing_collect()
    reps = 20
    for _ in range(reps):
        l.Forward(x, out, scratch)
    torch.cuda.synchronize()
    st = stage_timing_collect()
    stage_timing(False)
    T = ((h + 5) // 6) ** 2
    k2b = 4.0 * (ic * h * h + 64 * ic * T) * batch
    k4b = 4.0 * (64 * oc * T + oc * h * h) * batch
    fl = 2.0 * 64 * oc * ic * T * batch
    k2, k3, k4 = (st[k][0] / reps for k in ("wino_input", "wino_gemm", "wino_output"))
    print(f"C{ic:4d} K{oc:4d} {h:3d}px b{batch}: K2 {k2*1e3:6.1f} us {k2b/k2/1e9:5.2f} TB/s | K3 {k3*1e3:6.1f} us {fl/k3/1e9:6.1f} TF | "
          f"K4 {k4*1e3:6.1f} us {k4b/k4/1e9:5.2f} TB/s")
