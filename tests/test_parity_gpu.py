"""Parity of the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star): normalised max error max|y_gpu - y_ref| / max|y_ref| <= 1e-4
per layer output, in fp32.  The checker is the REAL reference (oracle/_ref, compiled from /root/reference) when its
.so travelled with the snapshot, else the plain-C restatement; both sides are also held against an fp64 direct conv.
"""
import numpy as np
import pytest

import oracle
from oracle import conv_geom, nerr, synth

pytestmark = pytest.mark.gpu

TOL = 1e-4


def run_gpu(g, x, w, b, dev, algo=None):
    import torch
    from feathercnn_amd import ConvLayer, ConvParam
    p = ConvParam(output_channels=g.oc, input_channels=g.ic, input_h=g.ih, input_w=g.iw, kernel_h=g.kh, kernel_w=g.kw,
                  stride_h=g.sh, stride_w=g.sw, pad_left=g.pl, pad_right=g.pr, pad_top=g.pt, pad_bottom=g.pb, group=g.group,
                  bias_term=bool(g.bias), activation=g.act, batch=x.shape[0])
    xt = torch.from_numpy(x).to(dev)
    wt = torch.from_numpy(w).to(dev)
    bt = torch.from_numpy(b).to(dev) if g.bias else None
    layer = ConvLayer(p, wt, bt, algo=algo)
    y = layer.Forward(xt)
    torch.cuda.synchronize()
    return y.cpu().numpy(), layer.booster.algo


def check(g, batch, dev, checker, port, algo=None, seed=1234):
    x, w, b = synth(g, batch, seed)
    y, used = run_gpu(g, x, w, b, dev, algo)
    ref = checker.forward(g, x, w, b, algo=-1 if algo is None else algo)
    f64 = port.direct_f64(g, x, w, b) if algo != oracle.NAIVE else None
    assert y.shape == ref.shape
    assert np.isfinite(y).all()
    e_ref = nerr(y, ref)
    assert e_ref <= TOL, f"{g} algo={used}: normalised error vs oracle {e_ref:.3e}"
    if f64 is not None:
        e64 = nerr(y, f64)
        assert e64 <= TOL, f"{g} algo={used}: normalised error vs fp64 direct conv {e64:.3e}"
    return used


WINO = [
    (conv_geom(16, 16, 9, 3, 1, 1), 2),            # smallest legal Winograd input (h > 8), edge tiles only
    (conv_geom(8, 8, 31, 3, 1, 1, w=17), 3),       # ragged, non-square
    (conv_geom(4, 4, 10, 3, 1, 1), 1),             # C = K = 4: all padding in the GEMM tiles
    (conv_geom(64, 64, 56, 3, 1, 1), 2),           # ResNet 3x3 @56
    (conv_geom(16, 64, 55, 3, 1, 1), 2),           # SqueezeNet fire expand3x3
    (conv_geom(48, 192, 13, 3, 1, 1), 5),          # K = 192 (1.5 row tiles), C = 48 (3 k-tiles)
    (conv_geom(64, 128, 28, 3, 1, 1), 3),
    (conv_geom(128, 256, 14, 3, 1, 1), 4),
    (conv_geom(512, 512, 14, 3, 1, 1), 2),         # VGG conv5
    (conv_geom(64, 64, 224, 3, 1, 1), 1),          # VGG conv1_2 (1444 tiles)
    (conv_geom(20, 36, 12, 3, 1, 0), 2),           # no padding, C % 16 != 0
    (conv_geom(12, 8, 20, 3, 1, 2), 2),            # pad 2 (wider than the kernel needs)
]


@pytest.mark.parametrize("g,batch", WINO, ids=lambda v: str(v) if isinstance(v, int) else f"{v.ic}x{v.oc}@{v.ih}x{v.iw}p{v.pl}")
def test_winograd_f63(g, batch, cuda, checker, port):
    assert check(g, batch, cuda, checker, port) == oracle.WINOGRADF63


IM2COL = [
    (conv_geom(256, 64, 56, 1, 1, 0), 2),          # ResNet 1x1 reduce (MODE 2: float4 column loads)
    (conv_geom(64, 256, 56, 1, 1, 0), 2),
    (conv_geom(256, 512, 56, 1, 2, 0), 2),         # strided 1x1 projection (MODE 1)
    (conv_geom(512, 2048, 7, 1, 1, 0), 3),         # HW = 49, not a multiple of 4 (MODE 1)
    (conv_geom(512, 1000, 13, 1, 1, 0), 2),        # SqueezeNet conv10, K = 1000
    (conv_geom(3, 64, 224, 7, 2, 3), 2),           # ResNet conv1
    (conv_geom(3, 64, 224, 3, 2, 0), 1),           # SqueezeNet conv1
    (conv_geom(3, 32, 224, 3, 2, 1), 2),           # MobileNet conv1
    (conv_geom(3, 64, 224, 3, 1, 1), 1),           # VGG conv1_1 (C % 4 != 0 -> IM2COL)
    (conv_geom(512, 512, 7, 3, 1, 1), 2),          # H <= 8 -> IM2COL
    (conv_geom(6, 10, 12, 3, 1, 1), 3),
    (conv_geom(5, 7, 9, 5, 2, 2, w=14), 2),        # 5x5 stride 2 ragged
    (conv_geom(1, 1, 5, 1, 1, 0, group=1), 1),     # degenerate: group == input_channels == 1 -> reference says DEPTHWISE
]


@pytest.mark.parametrize("g,batch", IM2COL, ids=lambda v: str(v) if isinstance(v, int) else f"{v.ic}x{v.oc}@{v.ih}k{v.kh}s{v.sh}")
def test_im2col_route(g, batch, cuda, checker, port):
    used = check(g, batch, cuda, checker, port)
    assert used == checker.select_algo(g)


DW = [
    (conv_geom(32, 32, 112, 3, 1, 1, group=32), 2),
    (conv_geom(64, 64, 112, 3, 2, 1, group=64), 2),
    (conv_geom(128, 128, 56, 3, 1, 1, group=128), 3),
    (conv_geom(256, 256, 28, 3, 2, 1, group=256), 3),
    (conv_geom(512, 512, 14, 3, 1, 1, group=512), 2),
    (conv_geom(512, 512, 14, 3, 2, 1, group=512), 2),   # OW = 7 -> generic path
    (conv_geom(1024, 1024, 7, 3, 1, 1, group=1024), 2),
    (conv_geom(8, 8, 16, 5, 1, 2, group=8), 2),         # 5x5
    (conv_geom(8, 8, 16, 3, 1, 0, group=8), 2),         # no padding
    (conv_geom(16, 16, 7, 7, 1, 0, group=16), 3),       # global kernel (avx/depthwise.cpp:30-54)
    (conv_geom(24, 24, 20, 3, 1, 1, group=24, w=36), 5),  # chunk tail: planes % planes_per_chunk != 0
]


@pytest.mark.parametrize("g,batch", DW, ids=lambda v: str(v) if isinstance(v, int) else f"dw{v.ic}@{v.ih}k{v.kh}s{v.sh}p{v.pl}")
def test_depthwise(g, batch, cuda, checker, port):
    assert check(g, batch, cuda, checker, port) == oracle.DEPTHWISE


@pytest.mark.parametrize("bias,act", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("kind", ["wino", "im2col", "dw"])
def test_epilogue_variants(kind, bias, act, cuda, checker, port):
    g = {"wino": conv_geom(32, 32, 20, 3, 1, 1, bias=bias, act=act),
         "im2col": conv_geom(32, 48, 20, 1, 1, 0, bias=bias, act=act),
         "dw": conv_geom(32, 32, 20, 3, 1, 1, group=32, bias=bias, act=act)}[kind]
    check(g, 2, cuda, checker, port)


def test_force_select_cross_check(cuda, checker, port):
    """ForceSelectAlgo (booster.h:162): the same ConvParam through NAIVE / IM2COL / WINOGRADF63 must agree --
    how 'booster facilitates unit testing' (booster.h:15-16).  NAIVE ignores activation (avx/booster.cpp:41-61)."""
    g = conv_geom(16, 32, 18, 3, 1, 1)
    x, w, b = synth(g, 2)
    outs = {a: run_gpu(g, x, w, b, cuda, a)[0] for a in (oracle.NAIVE, oracle.IM2COL, oracle.WINOGRADF63)}
    assert nerr(outs[oracle.IM2COL], outs[oracle.WINOGRADF63]) <= TOL
    assert (outs[oracle.NAIVE] < 0).any(), "NAIVE must not apply ReLU"
    assert nerr(np.maximum(outs[oracle.NAIVE], 0), outs[oracle.IM2COL]) <= TOL
    check(g, 2, cuda, checker, port, algo=oracle.NAIVE)


def test_unsupported(cuda):
    from feathercnn_amd import ConvBooster, ConvParam, SGECONV, WINOGRADF23, WINOGRADF63FUSED
    p = ConvParam.make(8, 8, 16, 3, 1, 1, group=2)
    b = ConvBooster()
    assert b.SelectAlgo(p) == -1  # partial group, avx/booster.cpp:304-308
    for a in (SGECONV, WINOGRADF23, WINOGRADF63FUSED):
        assert b.ForceSelectAlgo(a) == -1
