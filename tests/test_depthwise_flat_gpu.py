"""The flat small-plane depthwise kernel (depthwise3x3_flat_kernel: 3x3, pad 1, stride 1 / 2 on 14 x 14 and 28 x 28 planes -- MobileNet-V1's
conv6 ... conv13) against the CPU oracle through the C-ABI: plane counts that are not a multiple of the chunk (4 / 5 / 15 / 20 planes),
fewer planes than one chunk, every epilogue variant, and the neighbouring geometries that must NOT take the route (7 x 7, non-square,
other pads) -- same tolerance as every parity test (normalised max error <= 1e-4; measured ~1e-7)."""
import numpy as np
import pytest

import oracle
from oracle import Geom, conv_geom, nerr, synth

from test_parity_gpu import check, run_gpu

pytestmark = pytest.mark.gpu

FLAT = [
    (conv_geom(512, 512, 14, 3, 1, 1, group=512), 3),    # MobileNet conv8..12 shape; 1536 planes = 102.4 chunks of 15
    (conv_geom(512, 512, 14, 3, 2, 1, group=512), 3),    # conv13: 76.8 chunks of 20
    (conv_geom(256, 256, 28, 3, 1, 1, group=256), 2),    # conv6
    (conv_geom(256, 256, 28, 3, 2, 1, group=256), 3),    # conv7: 153.6 chunks of 5
    (conv_geom(7, 7, 14, 3, 1, 1, group=7), 1),          # fewer planes than one chunk
    (conv_geom(7, 7, 14, 3, 2, 1, group=7), 1),
    (conv_geom(3, 3, 28, 3, 1, 1, group=3), 1),
    (conv_geom(3, 3, 28, 3, 2, 1, group=3), 1),
    (conv_geom(37, 37, 14, 3, 1, 1, group=37), 5),       # 185 planes: odd channel count, chunk tail of 5
    (conv_geom(37, 37, 14, 3, 2, 1, group=37), 5),
    (conv_geom(13, 13, 28, 3, 1, 1, group=13), 7),       # 91 planes: tail of 3 (stride 1, chunks of 4)
    (conv_geom(13, 13, 28, 3, 2, 1, group=13), 7),       # tail of 1 (chunks of 5)
    (conv_geom(1, 1, 14, 3, 1, 1, group=1), 1),          # one plane (the reference routes group == C == 1 to DEPTHWISE)
    # 7 x 7 stride 1 (row-per-lane path of the flat kernel, chunks of 36 planes)
    (conv_geom(1024, 1024, 7, 3, 1, 1, group=1024), 2),  # MobileNet conv14 shape
    (conv_geom(37, 37, 7, 3, 1, 1, group=37), 5),        # 185 planes: 5 chunks + 5 planes, odd channel count
    (conv_geom(1, 1, 7, 3, 1, 1, group=1), 1),           # one plane: 49 floats = 12 float4 + 1
    (conv_geom(3, 3, 7, 3, 1, 1, group=3), 1),           # 147 floats: ragged tail of the float4 copies
    # band kernel: 112- and 56-pixel planes, stride 1 (16- / 28-row bands)
    (conv_geom(32, 32, 112, 3, 1, 1, group=32), 2),      # MobileNet conv2_dw shape
    (conv_geom(5, 5, 112, 3, 1, 1, group=5), 3),
    (conv_geom(128, 128, 56, 3, 1, 1, group=128), 2),    # conv4_dw shape
    (conv_geom(9, 9, 56, 3, 1, 1, group=9), 1),
]


@pytest.mark.parametrize("g,batch", FLAT, ids=lambda v: str(v) if isinstance(v, int) else f"dw{v.ic}@{v.ih}s{v.sh}")
def test_flat_route_matches_oracle(g, batch, cuda, checker, port):
    assert check(g, batch, cuda, checker, port, seed=31) == oracle.DEPTHWISE


@pytest.mark.parametrize("bias,act", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("h,s", [(14, 1), (14, 2), (28, 1), (28, 2), (7, 1), (56, 1), (112, 1)])
def test_flat_route_epilogues(h, s, bias, act, cuda, checker, port):
    check(conv_geom(24, 24, h, 3, s, 1, group=24, bias=bias, act=act), 3, cuda, checker, port, seed=5)


NEIGHBOURS = [
    conv_geom(24, 24, 7, 3, 2, 1, group=24),                    # 7 x 7 stride 2
    Geom(24, 24, 56, 56, 3, 3, 1, 1, 1, 0, 1, 0, 24, 1, 1),     # 56 x 56, pad 1 0 1 0
    Geom(8, 8, 112, 56, 3, 3, 1, 1, 1, 1, 1, 1, 8, 1, 1),       # 112 x 56, not square
    Geom(24, 24, 14, 28, 3, 3, 1, 1, 1, 1, 1, 1, 24, 1, 1),     # 14 x 28, not square
    Geom(24, 24, 14, 14, 3, 3, 1, 1, 1, 0, 1, 0, 24, 1, 1),     # pad_right = pad_bottom = 0
    Geom(24, 24, 28, 28, 3, 3, 2, 2, 1, 0, 1, 0, 24, 1, 1),     # stride 2, pad 1 0 1 0 (TensorFlow-style SAME)
    conv_geom(24, 24, 14, 3, 1, 0, group=24),                   # no padding
    conv_geom(24, 24, 15, 3, 1, 1, group=24),                   # 15 x 15
]


@pytest.mark.parametrize("g", NEIGHBOURS, ids=lambda g: f"dw{g.ih}x{g.iw}s{g.sh}p{g.pl}{g.pr}{g.pt}{g.pb}")
def test_geometries_next_to_the_flat_route(g, cuda, checker, port):
    check(g, 3, cuda, checker, port, seed=9)


def test_flat_route_is_deterministic_and_batch_independent(cuda, port):
    """Image n of a batch equals the same image run alone (chunks cut across images, never across results)."""
    g = conv_geom(40, 40, 14, 3, 1, 1, group=40)
    x, w, b = synth(g, 6, seed=77)
    y, _ = run_gpu(g, x, w, b, cuda)
    y2, _ = run_gpu(g, x, w, b, cuda)
    assert np.array_equal(y, y2)
    for n in (0, 3, 5):
        yn, _ = run_gpu(g, x[n:n + 1], w, b, cuda)
        assert np.array_equal(yn[0], y[n])
    assert nerr(y, port.direct_f64(g, x, w, b)) <= 1e-6
