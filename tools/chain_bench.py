"""VGG-16's 3x3 layers (conv1_2 .. conv5_3, batch 32) as separate Winograd layers vs one chained run (fhip_conv_forward_chained).
Usage: python tools/chain_bench.py [batch]"""
import sys

import numpy as np
import torch

from feathercnn_amd import ConvLayer, ConvParam
from feathercnn_amd.booster import WINOGRADF63, forward_chained, stage_timing, stage_timing_collect

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
SPEC = [(64, 64, 224, True), (64, 128, 112, False), (128, 128, 112, True), (128, 256, 56, False), (256, 256, 56, False), (256, 256, 56, True),
        (256, 512, 28, False), (512, 512, 28, False), (512, 512, 28, True), (512, 512, 14, False), (512, 512, 14, False), (512, 512, 14, True)]
rng = np.random.default_rng(0)
layers, pools = [], []
for ic, oc, h, pool in SPEC:
    w = torch.from_numpy((rng.standard_normal((oc, ic, 3, 3)) / np.sqrt(9 * ic)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.uniform(-0.1, 0.1, oc).astype(np.float32)).to(dev)
    prm = ConvParam(output_channels=oc, input_channels=ic, input_h=h, input_w=h, kernel_h=3, kernel_w=3, stride_h=1, stride_w=1, pad_left=1,
                    pad_right=1, pad_top=1, pad_bottom=1, group=1, bias_term=True, activation=1, batch=batch)
    layers.append(ConvLayer(prm, w, b, algo=WINOGRADF63))
    pools.append(pool)
x = torch.from_numpy(rng.uniform(-1, 1, (batch, 64, 224, 224)).astype(np.float32)).to(dev)
scratch = torch.empty(max(l.buffer_bytes for l in layers) // 4, dtype=torch.float32, device=dev)
lib = __import__("feathercnn_amd._lib", fromlist=["x"]).load_library()
import ctypes

from feathercnn_amd.booster import _ptr, _stream


def separate():
    t = x
    for l, pool in zip(layers, pools):
        p = l.param
        oh, ow = (p.output_h // 2, p.output_w // 2) if pool else (p.output_h, p.output_w)
        out = torch.empty((batch, p.output_channels, oh, ow), dtype=torch.float32, device=dev)
        c = p._c()
        f = lib.fhip_conv_forward_maxpool2 if pool else lib.fhip_conv_forward
        assert f(ctypes.byref(c), WINOGRADF63, batch, _ptr(out), _ptr(t), _ptr(l.packed), _ptr(scratch), _ptr(l.bias), _stream()) == 0
        t = out
    return t


def chained_from_second():
    return forward_chained(layers, x, pools)


def timeit(f, reps=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


a, b = separate(), chained_from_second()
torch.cuda.synchronize()
print("equal:", torch.equal(a, b), "max diff", (a - b).abs().max().item())
for name, f in (("separate", separate), ("chained", chained_from_second), ("separate", separate), ("chained", chained_from_second)):
    print(f"{name:10s} {timeit(f):.3f} ms")
stage_timing(True)
for name, f in (("separate", separate), ("chained", chained_from_second)):
    stage_timing_collect()
    for _ in range(5):
        f()
    torch.cuda.synchronize()
    st = stage_timing_collect()
    print(name, {k: (round(v[0] / 5, 3), v[1] // 5) for k, v in st.items() if v[1]})
stage_timing(False)
# the chained transform alone, link by link
from feathercnn_amd.booster import winograd_plan
for i in range(len(layers) - 1):
    a_, b_ = layers[i].param, layers[i + 1].param
    pa, pb = winograd_plan(a_), winograd_plan(b_)
    m = torch.randn(pa.m_bytes // 4, device=dev)
    v = torch.empty(pb.v_bytes // 4, device=dev)
    ca, cb = a_._c(), b_._c()
    f = lambda: lib.fhip_winograd_f63_output_to_next_input(ctypes.byref(ca), ctypes.byref(cb), batch, _ptr(v), _ptr(m), _ptr(layers[i].bias), int(pools[i]), _stream())
    t = timeit(f, 10)
    mb = (64 * a_.output_channels * pa.columns_padded + 64 * b_.input_channels * pb.columns_padded) * 4 / 1e6
    print(f"link {i}: K {a_.output_channels} {a_.output_h}px pool {int(pools[i])}: {t*1e3:.1f} us  {mb:.0f} MB  {mb/t/1e3:.2f} TB/s")
