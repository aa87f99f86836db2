#!/usr/bin/env python
"""tools/overlap_probe.py -- VERDICT r05 next #3 lever (c): what could ANY scheme that lets layer i + 1's first blocks start under layer i's tail buy?
Upper bound, measured without building one: ResNet-50 b64's consecutive 1x1 layers (expand + residual, then the next block's reduce) run
  seq : A then B on ONE stream (what the net does; B depends on A)
  par : A on one stream, B on another, nothing between them (no dependency at all: B's blocks fill A's tail and A's next launch fills B's)
over `reps` launches each.  par is the best any dependency-aware overlap could reach for this pair -- it waits for nothing -- so seq - par bounds the
gain per seam from above.  Also: A and B each alone.  All through the product's C-ABI (ConvLayer), each layer with a scratch buffer of its own."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes  # noqa: E402

from feathercnn_amd import IM2COL, ConvLayer, ConvParam, _lib  # noqa: E402

PAIRS = [("res2 64->256+res | 256->64 @56", 64, 256, 56), ("res3 128->512+res | 512->128 @28", 128, 512, 28),
         ("res4 256->1024+res | 1024->256 @14", 256, 1024, 14), ("res5 512->2048+res | 2048->512 @7", 512, 2048, 7)]


def layer(c, k, h, batch, dev, rng):
    p = ConvParam(output_channels=k, input_channels=c, input_h=h, input_w=h, kernel_h=1, kernel_w=1, stride_h=1, stride_w=1, group=1, bias_term=True,
                  activation=1, batch=batch)
    w = torch.from_numpy((rng.uniform(-1, 1, (k, c, 1, 1)) / np.sqrt(c)).astype(np.float32)).to(dev)
    b = torch.from_numpy(rng.uniform(-.1, .1, k).astype(np.float32)).to(dev)
    return ConvLayer(p, w, b, algo=IM2COL)


def main():
    batch, reps = int(os.environ.get("BATCH", "64")), int(os.environ.get("REPS", "50"))
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    for name, c, k, h in PAIRS:
        la, lb = layer(c, k, h, batch, dev, rng), layer(k, c, h, batch, dev, rng)
        x = torch.rand((batch, c, h, h), device=dev) * 2 - 1
        res = torch.rand((batch, k, h, h), device=dev) * 2 - 1
        mid = torch.empty((batch, k, h, h), device=dev)
        mid2 = torch.rand((batch, k, h, h), device=dev)  # par: B reads an independent tensor of the same shape
        out = torch.empty((batch, c, h, h), device=dev)
        has_res = True
        lib = _lib.load_library()
        sa = torch.empty(max(la.buffer_bytes // 4, 64), dtype=torch.float32, device=dev)
        sb = torch.empty(max(lb.buffer_bytes // 4, 64), dtype=torch.float32, device=dev)
        ca = la.param._c()

        def A():
            rc = lib.fhip_conv_forward_residual(ctypes.byref(ca), IM2COL, batch, ctypes.c_void_p(mid.data_ptr()), ctypes.c_void_p(x.data_ptr()),
                                                ctypes.c_void_p(la.packed.data_ptr()), ctypes.c_void_p(sa.data_ptr()), ctypes.c_void_p(la.bias.data_ptr()),
                                                ctypes.c_void_p(res.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc

        def B(src=mid):
            lb.Forward(src, out=out, scratch=sb)

        def timed(fn):
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / reps * 1e3

        def seq():
            for _ in range(reps):
                A()
                B()

        def only(f):
            def run():
                for _ in range(reps):
                    f()
            return run

        def par():
            cur = torch.cuda.current_stream()
            s1.wait_stream(cur)
            s2.wait_stream(cur)
            for _ in range(reps):
                with torch.cuda.stream(s1):
                    A()
                with torch.cuda.stream(s2):
                    B(mid2)
            cur.wait_stream(s1)
            cur.wait_stream(s2)

        for f in (seq, par):
            f()
        r = {}
        for rnd in range(3):
            for nm, f in (("seq", seq), ("par", par), ("A", only(A)), ("B", only(B))):
                r.setdefault(nm, []).append(timed(f))
        m = {n: sorted(v)[1] for n, v in r.items()}
        print(f"{name:36s} A {m['A']:6.1f} + B {m['B']:6.1f} = {m['A'] + m['B']:6.1f};  seq {m['seq']:6.1f} us   par {m['par']:6.1f} us   "
              f"bound on the gain per seam pair: {m['seq'] - m['par']:5.1f} us ({(m['seq'] - m['par']) / m['seq'] * 100:4.1f} %)" + ("" if has_res else "  [no residual entry point]"), flush=True)


if __name__ == "__main__":
    main()
