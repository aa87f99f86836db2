"""tools/first_bench.py -- VGG-16 b32 conv1_1 + conv1_2's input transform, separate (conv_smallc_kernel, wino_input_transform_kernel) against
the fused kernel (wino_first.h), interleaved on one box: microseconds per launch (HIP events, median of `reps`)."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from feathercnn_amd import ConvLayer, ConvParam, _lib  # noqa: E402
from feathercnn_amd.booster import _ptr, _stream, winograd_plan  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(0)

    def prm(ic, oc):
        return ConvParam(output_channels=oc, input_channels=ic, input_h=224, input_w=224, kernel_h=3, kernel_w=3, stride_h=1, stride_w=1, pad_left=1,
                         pad_right=1, pad_top=1, pad_bottom=1, group=1, bias_term=True, activation=1, batch=batch)
    w0 = torch.from_numpy((rng.standard_normal((64, 3, 3, 3)) / 5).astype(np.float32)).to(dev)
    b0 = torch.from_numpy(rng.uniform(-0.1, 0.1, 64).astype(np.float32)).to(dev)
    w1 = torch.from_numpy((rng.standard_normal((64, 64, 3, 3)) / 24).astype(np.float32)).to(dev)
    p0 = prm(3, 64)
    first = ConvLayer(p0, w0, b0)
    nxt = ConvLayer(prm(64, 64), w1, b0, algo=4)
    first.param.batch = batch
    first.buffer_bytes, _ = first.booster.GetBufferSize(first.param)
    x = torch.from_numpy(rng.uniform(-1, 1, (batch, 3, 224, 224)).astype(np.float32)).to(dev)
    mid = first.Forward(x)
    pl = winograd_plan(nxt.param)
    v = torch.empty(pl.v_bytes // 4, dtype=torch.float32, device=dev)
    lib = _lib.load_library()
    cn, cf = nxt.param._c(), p0._c()

    def conv1():
        first.Forward(x, out=mid)

    def k2():
        assert lib.fhip_winograd_f63_input_transform(ctypes.byref(cn), batch, _ptr(v), _ptr(mid), _stream()) == 0

    def fused():
        assert lib.fhip_winograd_f63_input_from_first(ctypes.byref(cf), ctypes.byref(cn), batch, _ptr(v), _ptr(x), _ptr(w0), _ptr(b0), _stream()) == 0

    def timed(f):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        f()
        b.record()
        b.synchronize()
        return a.elapsed_time(b) * 1e3
    m = torch.empty(pl.m_bytes // 4, dtype=torch.float32, device=dev)

    def gemm():
        assert lib.fhip_winograd_f63_tile_gemm(ctypes.byref(cn), batch, _ptr(m), _ptr(nxt.packed), _ptr(v), _stream()) == 0
    res = {"conv1_1": [], "input transform": [], "fused": [], "gemm after input transform": [], "gemm after fused": [], "gemm after gemm": []}
    for f in (conv1, k2, fused):
        f()
    for _ in range(reps):
        res["conv1_1"].append(timed(conv1))
        res["input transform"].append(timed(k2))
        res["fused"].append(timed(fused))
        k2()
        res["gemm after input transform"].append(timed(gemm))
        fused()
        res["gemm after fused"].append(timed(gemm))
        res["gemm after gemm"].append(timed(gemm))
    med = {k: float(np.median(t)) for k, t in res.items()}
    print(f"batch {batch}: conv1_1 {med['conv1_1']:.1f} us + input transform {med['input transform']:.1f} us = "
          f"{med['conv1_1'] + med['input transform']:.1f} us;  fused {med['fused']:.1f} us  (min {min(res['fused']):.1f})")
    print("tile GEMM of conv1_2 right after the input transform %.1f us, after the fused kernel %.1f us, after itself %.1f us" %
          (med["gemm after input transform"], med["gemm after fused"], med["gemm after gemm"]))


if __name__ == "__main__":
    main()
