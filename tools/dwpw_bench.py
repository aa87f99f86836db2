#!/usr/bin/env python
"""tools/dwpw_bench.py -- MobileNet-V1's three fused depthwise + pointwise pairs at batch 256: fhip_conv_forward_dw_pw against the depthwise
kernel followed by the 1x1 GEMM (same library), HIP events over `reps` launches, interleaved rounds.  With FHIP_LIB_VARIANTS="a b ..." the
script is run once per library build by tools/dwpw_ab.sh (ablation builds of conv_gemm_policy.h)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feathercnn_amd import DEPTHWISE, IM2COL, ConvLayer, ConvParam, _lib  # noqa: E402

PAIRS = [("conv2 dw112s1+pw32-64", 32, 64, 112, 1), ("conv3 dw112s2+pw64-128", 64, 128, 112, 2), ("conv4 dw56s1+pw128-128", 128, 128, 56, 1), ("conv5 dw56s2+pw128-256", 128, 256, 56, 2)]


def main():
    batch = int(os.environ.get("BATCH", "256"))
    reps = int(os.environ.get("REPS", "20"))
    dev = torch.device("cuda:0")
    lib = _lib.load_library()
    rng = np.random.default_rng(1)
    for name, c, k, h, s in PAIRS:
        t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
        pd = ConvParam(output_channels=c, input_channels=c, input_h=h, input_w=h, kernel_h=3, kernel_w=3, stride_h=s, stride_w=s, pad_left=1, pad_right=1,
                       pad_top=1, pad_bottom=1, group=c, bias_term=True, activation=1, batch=batch)
        pd.AssignOutputDim()
        pp = ConvParam(output_channels=k, input_channels=c, input_h=pd.output_h, input_w=pd.output_w, kernel_h=1, kernel_w=1, stride_h=1, stride_w=1, group=1,
                       bias_term=True, activation=1, batch=batch)
        ld = ConvLayer(pd, t((rng.uniform(-1, 1, (c, 1, 3, 3)) / 3).astype(np.float32)), t(rng.uniform(-.2, .2, c).astype(np.float32)), algo=DEPTHWISE)
        lp = ConvLayer(pp, t((rng.uniform(-1, 1, (k, c, 1, 1)) / np.sqrt(c)).astype(np.float32)), t(rng.uniform(-.1, .1, k).astype(np.float32)), algo=IM2COL)
        x = torch.rand((batch, c, h, h), device=dev) * 2 - 1
        mid = torch.empty((batch, c, pd.output_h, pd.output_w), device=dev)
        out = torch.empty((batch, k, pd.output_h, pd.output_w), device=dev)
        cd, cp = pd._c(), pp._c()

        def fused():
            rc = lib.fhip_conv_forward_dw_pw(ctypes.byref(cd), ctypes.byref(cp), batch, out.data_ptr(), x.data_ptr(), ld.packed.data_ptr(), ld.bias.data_ptr(),
                                             lp.packed.data_ptr(), lp.bias.data_ptr(), None)
            if rc != 0:
                two()  # this build does not fuse the pair

        def two():
            ld.Forward(x, out=mid)
            lp.Forward(mid, out=out)

        def dw():
            ld.Forward(x, out=mid)

        def pw():
            lp.Forward(mid, out=out)

        res = {}
        for rnd in range(3):
            for nm, f in (("fused", fused), ("dw+pw", two), ("dw", dw), ("pw", pw)):
                for _ in range(3):
                    f()
                torch.cuda.synchronize()
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                for _ in range(reps):
                    f()
                b.record()
                torch.cuda.synchronize()
                res.setdefault(nm, []).append(a.elapsed_time(b) / reps * 1e3)
        by = 4.0 * batch * (c * h * h + k * pd.output_h * pd.output_w)
        fl = 2.0 * k * c * pd.output_h * pd.output_w * batch
        med = {n: sorted(v)[1] for n, v in res.items()}
        print(f"{name:26s} fused {med['fused']:7.1f} us ({by / med['fused'] / 1e6:5.2f} TB/s, {fl / med['fused'] / 1e6:5.1f} TF)   dw+pw {med['dw+pw']:7.1f}   dw {med['dw']:6.1f}  pw {med['pw']:6.1f}", flush=True)


if __name__ == "__main__":
    main()
