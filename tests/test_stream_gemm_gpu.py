"""The register-streamed 1x1 GEMM route of the implicit GEMM (stream_gemm.h): deep-reduction, narrow-output 1x1 / stride-1 layers.
Against the checker (real reference when built) through the ordinary C-ABI forward, on shapes that exercise pixel tiles straddling
images, a ragged last tile, both request-ring depths, missing bias / ReLU, and the geometry rules that keep a layer OFF the route."""
import ctypes

import numpy as np
import pytest

from oracle import conv_geom, nerr, synth

pytestmark = pytest.mark.gpu
TOL = 1e-4
IM2COL = 1


def _param(g, batch, bias=True, relu=True):
    from feathercnn_amd import ConvParam
    return ConvParam(output_channels=g.oc, input_channels=g.ic, input_h=g.ih, input_w=g.iw, kernel_h=g.kh, kernel_w=g.kw, stride_h=g.sh, stride_w=g.sw,
                     pad_left=g.pl, pad_right=g.pr, pad_top=g.pt, pad_bottom=g.pb, group=g.group, bias_term=bias, activation=1 if relu else 0, batch=batch)


def _streams(prm, batch):
    from feathercnn_amd import _lib
    prm.AssignOutputDim()
    c = prm._c()
    return bool(_lib.load_library().fhip_conv_streams_1x1(ctypes.byref(c), IM2COL, batch))


# name, C, K, H, W, batch, bias, relu
ON_ROUTE = [
    ("r50_res4x_2a_b64", 1024, 256, 14, 14, 64, True, True),      # 98 pixel tiles, every other one straddles two images
    ("r50_res3x_2a_b8", 512, 128, 28, 28, 8, True, True),
    ("ragged_last_tile", 512, 512, 14, 14, 21, True, True),       # 4116 pixels = 32 tiles + 20 pixels
    ("ring_depth_8", 272, 160, 10, 10, 41, True, True),            # C % 32 != 0 -> the 8-deep ring; K = 5 m-groups -> a wave sits out
    ("no_bias", 256, 128, 16, 16, 16, False, True),
    ("no_relu", 256, 192, 16, 18, 15, True, False),
    ("mb_512_512", 512, 512, 14, 14, 32, True, True),
    # Ho*Wo % 4 != 0: the last pixel group of every image is loaded shifted back and stores only its new pixels
    ("deep_7x7_b256", 1024, 512, 7, 7, 256, True, True),           # 49 = 12 groups + 1 pixel
    ("ragged_54", 256, 128, 6, 9, 80, True, True),                # 54 = 13 groups + 2 pixels
    ("ragged_27_no_bias", 512, 256, 3, 9, 160, False, False),      # 27 = 6 groups + 3 pixels
    ("ragged_ring_8", 272, 160, 5, 5, 170, True, True),            # 25 = 6 groups + 1 pixel, 8-deep ring
]


@pytest.mark.parametrize("case", ON_ROUTE, ids=[c[0] for c in ON_ROUTE])
def test_streamed_route_matches_reference(cuda, checker, case):
    import torch

    from feathercnn_amd import ConvLayer
    name, C, K, H, W, batch, bias, relu = case
    g = conv_geom(C, K, H, 1, 1, 0, bias=int(bias), act=int(relu), w=W)
    x, w, b = synth(g, batch, seed=11)
    prm = _param(g, batch, bias, relu)
    assert _streams(prm, batch), name
    layer = ConvLayer(prm, torch.from_numpy(w).to(cuda), torch.from_numpy(b).to(cuda) if bias else None, algo=IM2COL)
    assert layer.buffer_bytes == 0  # the streamed kernel never splits the reduction
    y = layer.Forward(torch.from_numpy(x).to(cuda))
    torch.cuda.synchronize()
    ref = checker.forward(g, x, w, b if bias else None, algo=IM2COL)
    assert nerr(y.cpu().numpy(), ref) <= TOL, name
    # deterministic: the same launch twice is bit-identical
    y2 = layer.Forward(torch.from_numpy(x).to(cuda))
    torch.cuda.synchronize()
    assert torch.equal(y, y2)


OFF_ROUTE = [
    ("shallow_c", 128, 128, 28, 28, 64),
    ("wide_k", 256, 1024, 14, 14, 64),
    ("narrow_k", 256, 64, 56, 56, 64),
    ("tiny_plane", 512, 256, 1, 3, 2000),   # fewer than 4 pixels per image
    ("few_pixels", 512, 256, 14, 14, 8),
    ("k_not_32", 512, 144, 14, 14, 64),
]


@pytest.mark.parametrize("case", OFF_ROUTE, ids=[c[0] for c in OFF_ROUTE])
def test_other_layers_stay_on_the_tiled_kernel(cuda, case):
    name, C, K, H, W, batch = case
    g = conv_geom(C, K, H, 1, 1, 0)
    assert not _streams(_param(g, batch), batch), name


def test_strided_and_padded_1x1_stay_off(cuda):
    assert not _streams(_param(conv_geom(512, 256, 28, 1, 2, 0), 64), 64)
    assert not _streams(_param(conv_geom(512, 256, 14, 1, 1, 1), 64), 64)
    assert not _streams(_param(conv_geom(512, 256, 14, 3, 1, 1), 64), 64)
