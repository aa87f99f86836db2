#!/usr/bin/env python
"""tools/band_timeline.py -- phase timeline of the band-staged depthwise + pointwise experiment (library built with -DFHIP_EXPERIMENT_DWPW_BAND
-DFHIP_BAND_TIMELINE): block 3's first wave of each role stamps s_memtime (100 MHz) at the end of every phase of its first 64 chunks."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from feathercnn_amd import DEPTHWISE, IM2COL, ConvLayer, ConvParam, _lib  # noqa: E402

c, k, h, s, batch = [int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (32, 64, 112, 1, 256))]
dev = torch.device("cuda:0")
lib = _lib.load_library()
rng = np.random.default_rng(1)
t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
pd = ConvParam(output_channels=c, input_channels=c, input_h=h, input_w=h, kernel_h=3, kernel_w=3, stride_h=s, stride_w=s, pad_left=1, pad_right=1, pad_top=1,
               pad_bottom=1, group=c, bias_term=True, activation=1, batch=batch)
pd.AssignOutputDim()
pp = ConvParam(output_channels=k, input_channels=c, input_h=pd.output_h, input_w=pd.output_w, kernel_h=1, kernel_w=1, stride_h=1, stride_w=1, group=1, bias_term=True,
               activation=1, batch=batch)
ld = ConvLayer(pd, t((rng.uniform(-1, 1, (c, 1, 3, 3)) / 3).astype(np.float32)), t(rng.uniform(-.2, .2, c).astype(np.float32)), algo=DEPTHWISE)
lp = ConvLayer(pp, t((rng.uniform(-1, 1, (k, c, 1, 1)) / np.sqrt(c)).astype(np.float32)), t(rng.uniform(-.1, .1, k).astype(np.float32)), algo=IM2COL)
x = torch.rand((batch, c, h, h), device=dev) * 2 - 1
out = torch.empty((batch, k, pd.output_h, pd.output_w), device=dev)
cd, cp = pd._c(), pp._c()
for _ in range(3):
    assert lib.fhip_conv_forward_dw_pw(ctypes.byref(cd), ctypes.byref(cp), batch, out.data_ptr(), x.data_ptr(), ld.packed.data_ptr(), ld.bias.data_ptr(),
                                       lp.packed.data_ptr(), lp.bias.data_ptr(), None) == 0
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * (3 * 64 * 4))()
assert ctypes.CDLL(_lib.lib_path()).fhip_debug_band_timeline(buf) == 0
tl = np.array(buf, dtype=np.int64).reshape(3, 64, 4)
t0 = tl[tl > 0].min()
us = lambda v: (v - t0) * 0.01  # noqa: E731
print("chunk | consumer A: mfma-done barrier-passed [stores-done] | consumer B: same | producer: dw-done stash-done fetch-issued barrier-passed   (us since first stamp)")
for j in range(24):
    a, b, p = tl[0, j], tl[1, j], tl[2, j]
    f = lambda v: f"{us(v):7.2f}" if v > 0 else "      -"  # noqa: E731
    print(f"{j:3d}   | {f(a[0])} {f(a[1])} {f(a[2])} | {f(b[0])} {f(b[1])} {f(b[2])} | {f(p[0])} {f(p[1])} {f(p[2])} {f(p[3])}")
