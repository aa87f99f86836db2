"""ConvGemmPolicy<5> (round 4): 1x1 / stride-1 / unpadded convolutions on planes whose size is not a multiple of 4 pixels (ResNet-50's 7 x 7
stage) run on pixel SLOTS -- ceil(Ho*Wo / 4) groups of 4 per image, the last group shifted back inside the image -- with 16-byte unaligned
loads / stores instead of MODE 1's scalar ones.  Checked against the oracle through the C-ABI: every remainder class of Ho*Wo mod 4, the
64-row and the 128-row tile, split-K (its partial sums live in slot space), batches that leave the last column tile ragged, the fused
residual (bit for bit against conv-then-add), and neighbours that must keep their old route (fewer than 4 pixels, narrow grids)."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from oracle import conv_geom, nerr, synth

pytestmark = pytest.mark.gpu

CASES = [
    (64, 96, 7, 7, 9),      # 49 = 4 * 12 + 1: the ResNet-50 plane, 128-row tile, ragged last column tile
    (48, 40, 7, 7, 5),      # 64-row tile (K <= 64)
    (32, 160, 5, 5, 6),     # 25 = 4 * 6 + 1
    (32, 128, 3, 5, 11),    # 15 = 4 * 3 + 3
    (16, 130, 3, 6, 7),     # 18 = 4 * 4 + 2; K not a multiple of the row tile
    (24, 72, 9, 9, 3),      # 81
    (2048, 64, 7, 7, 8),    # deep reduction on a short grid: split-K, partial sums in slot space
    (512, 256, 7, 7, 2),    # split-K on the 128-row tile
    (20, 36, 1, 3, 40),     # 3 pixels per image: stays on the scalar route (a group of 4 would leave the image)
    (16, 32, 5, 1, 2),      # 10 columns in all: the narrow 64 x 32 tile keeps MODE 1
]


def _layer(cuda, c, k, h, w, batch, act=1):
    from feathercnn_amd import ConvLayer, ConvParam
    from feathercnn_amd.booster import IM2COL
    g = conv_geom(c, k, h, 1, 1, 0, act=act, w=w)
    x, wt, b = synth(g, batch, seed=c + k)
    p = ConvParam(output_channels=k, input_channels=c, input_h=h, input_w=w, kernel_h=1, kernel_w=1, stride_h=1, stride_w=1, pad_left=0, pad_right=0,
                  pad_top=0, pad_bottom=0, group=1, bias_term=True, activation=act, batch=batch)
    return g, x, wt, b, p, ConvLayer(p, torch.from_numpy(wt).to(cuda), torch.from_numpy(b).to(cuda), algo=IM2COL)


@pytest.mark.parametrize("c,k,h,w,batch", CASES)
def test_ragged_planes_match_the_oracle(cuda, c, k, h, w, batch):
    g, x, wt, b, p, layer = _layer(cuda, c, k, h, w, batch)
    want = oracle.best().forward(g, x, wt, b)
    got = layer.Forward(torch.from_numpy(x).to(cuda)).cpu().numpy()
    assert got.shape == want.shape and nerr(got, want) <= 1e-5, (c, k, h, w, batch)
    # garbage in the scratch arena (split-K partial sums are written before they are read) and determinism
    scratch = torch.full((max(layer.buffer_bytes // 4, 1),), float("nan"), device=cuda)
    again = layer.Forward(torch.from_numpy(x).to(cuda), scratch=scratch).cpu().numpy()
    assert np.array_equal(got, again)


@pytest.mark.parametrize("c,k,h,w,batch", [(64, 256, 7, 7, 6), (40, 48, 5, 5, 9), (2048, 128, 7, 7, 4)])
def test_ragged_planes_with_fused_residual(cuda, c, k, h, w, batch):
    from feathercnn_amd import _lib
    from feathercnn_amd.booster import IM2COL
    lib = _lib.load_library()
    g, x, wt, b, p0, plain = _layer(cuda, c, k, h, w, batch, act=0)
    xt = torch.from_numpy(x).to(cuda)
    y = plain.Forward(xt)
    res = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, tuple(y.shape)).astype(np.float32)).to(cuda)
    for act in (0, 1):
        _, _, _, _, p, layer = _layer(cuda, c, k, h, w, batch, act=act)
        cp = p._c()
        out = torch.full_like(y, float("nan"))
        scratch = torch.empty(max(layer.buffer_bytes // 4, 1), device=cuda)
        assert lib.fhip_conv_forward_residual(ctypes.byref(cp), IM2COL, batch, out.data_ptr(), xt.data_ptr(), layer.packed.data_ptr(), scratch.data_ptr(),
                                              layer.bias.data_ptr(), res.data_ptr(), None) == 0
        torch.cuda.synchronize()
        want = y + res
        assert torch.equal(out, torch.clamp_min(want, 0) if act else want), (c, k, h, w, batch, act)
