"""A net's first convolution computed inside the input transform of the Winograd layer behind it (wino_first.h,
fhip_winograd_f63_input_from_first): the pair must give what the two layers give one after the other -- to rounding (the first
layer's 27-term sums are added in another order than the implicit GEMM's), and the reference's result within the parity bar."""
import numpy as np
import pytest

from oracle import conv_geom, nerr

pytestmark = pytest.mark.gpu

WINO = 4  # FHIP_WINOGRADF63


def _first(cuda, ic, oc, h, w, bias=True, relu=True, seed=0):
    import torch

    from feathercnn_amd import ConvLayer, ConvParam
    rng = np.random.default_rng(seed)
    wt = (rng.standard_normal((oc, ic, 3, 3)) / np.sqrt(9 * ic)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, oc).astype(np.float32) if bias else None
    prm = ConvParam(output_channels=oc, input_channels=ic, input_h=h, input_w=w, kernel_h=3, kernel_w=3, stride_h=1, stride_w=1, pad_left=1,
                    pad_right=1, pad_top=1, pad_bottom=1, group=1, bias_term=bias, activation=1 if relu else 0)
    wd = torch.from_numpy(wt).to(cuda)
    bd = None if b is None else torch.from_numpy(b).to(cuda)
    return ConvLayer(prm, wd, bd), prm, wd, bd, wt, b


def _wino(cuda, ic, oc, h, w, seed=1):
    import torch

    from feathercnn_amd import ConvLayer, ConvParam
    rng = np.random.default_rng(seed)
    wt = (rng.standard_normal((oc, ic, 3, 3)) / np.sqrt(9 * ic)).astype(np.float32)
    b = rng.uniform(-0.5, 0.5, oc).astype(np.float32)
    prm = ConvParam(output_channels=oc, input_channels=ic, input_h=h, input_w=w, kernel_h=3, kernel_w=3, stride_h=1, stride_w=1, pad_left=1,
                    pad_right=1, pad_top=1, pad_bottom=1, group=1, bias_term=True, activation=1)
    return ConvLayer(prm, torch.from_numpy(wt).to(cuda), torch.from_numpy(b).to(cuda), algo=WINO), wt, b


# (name, batch, image channels, H, W, first layer's output channels, bias, relu, pool behind the Winograd layer)
CASES = [
    ("vgg_conv1_small_batch", 2, 3, 224, 224, 64, True, True, True),
    ("rgb_odd_height", 3, 3, 37, 50, 20, True, True, False),
    ("two_channels", 2, 2, 30, 28, 8, True, True, False),
    ("two_channels_no_bias", 2, 2, 12, 12, 5, False, True, False),
    ("four_channels_linear", 1, 4, 18, 26, 33, True, False, False),  # no ReLU: negative values survive, the padding must still be zero
    ("one_tile", 5, 3, 6, 6, 4, True, True, False),
    ("tiny", 2, 3, 3, 4, 3, True, True, False),
    ("wide_rows_direct_form", 1, 4, 14, 1200, 6, True, True, False),  # 4 x 22 x 1210 floats do not fit the LDS: the L1 form
    ("small_plane_direct_form", 2, 3, 40, 14, 18, True, True, False),  # 21 tiles per image: lanes run across images (direct form)
    ("two_blocks_per_image", 3, 3, 64, 64, 24, True, True, False),  # 121 tiles: blocks of 63 + 58 tiles (shared columns), 2 channel groups
    ("row_end_inside_image", 2, 3, 48, 48, 20, True, True, False),  # W = 6 TX: the row-end tile's column 6 is image column 47 -- staged form WITHOUT shared columns
    ("unshared_w42", 2, 3, 40, 42, 9, True, True, False),  # W = 6 TX again (TX = 7, 49 tiles in one block)
    ("shared_63_tiles_exact", 1, 3, 54, 50, 5, True, False, False),  # TX = 9, TY = 9: 81 tiles = 63 + 18; tile 63 starts a tile row (the helper copies tile 62)
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_first_layer_inside_the_input_transform(cuda, checker, case):
    import torch

    from feathercnn_amd.booster import can_fuse_first_winograd, forward_chained
    name, batch, ic, h, w, oc, bias, relu, pool = case
    first, fprm, fw, fb, w1, b1 = _first(cuda, ic, oc, h, w, bias=bias, relu=relu, seed=11)
    nxt, w2, b2 = _wino(cuda, oc, 12, h, w, seed=12)
    assert can_fuse_first_winograd(fprm, nxt, batch)
    x = np.random.default_rng(3).uniform(-1, 1, (batch, ic, h, w)).astype(np.float32)
    xd = torch.from_numpy(x).to(cuda)
    # the two layers one after the other
    first.param.batch = batch
    first.buffer_bytes, _ = first.booster.GetBufferSize(first.param)
    mid = first.Forward(xd)
    want = forward_chained([nxt], mid, [pool])
    got = forward_chained([nxt], xd, [pool], first=(fprm, fw, fb))
    torch.cuda.synchronize()
    assert got.shape == want.shape
    assert nerr(got.cpu().numpy(), want.cpu().numpy()) <= 2e-5, name  # conv1's rounding differences, amplified by the Winograd transforms
    # and the reference
    ref_mid = checker.forward(conv_geom(ic, oc, h, 3, 1, 1, bias=1 if bias else 0, act=1 if relu else 0, w=w), x, w1, b1)
    ref = checker.forward(conv_geom(oc, 12, h, 3, 1, 1, w=w), ref_mid, w2, b2)
    if pool:
        ref = ref.reshape(batch, 12, h // 2, 2, w // 2, 2).max(axis=(3, 5))
    assert nerr(got.cpu().numpy(), ref) <= 1e-4, name


def test_stage_level_v_matches_the_input_transform(cuda):
    """V written by the fused kernel against fhip_winograd_f63_input_transform of the first layer's output, columns < P."""
    import ctypes

    import torch

    from feathercnn_amd import _lib
    from feathercnn_amd.booster import _ptr, _stream, winograd_plan
    batch, ic, h, w, oc = 3, 3, 28, 30, 16
    first, fprm, fw, fb, _, _ = _first(cuda, ic, oc, h, w, seed=21)
    nxt, _, _ = _wino(cuda, oc, 8, h, w, seed=22)
    xd = torch.from_numpy(np.random.default_rng(4).uniform(-1, 1, (batch, ic, h, w)).astype(np.float32)).to(cuda)
    first.param.batch = batch
    first.buffer_bytes, _ = first.booster.GetBufferSize(first.param)
    mid = first.Forward(xd)
    nxt.param.batch = fprm.batch = batch
    pl = winograd_plan(nxt.param)
    lib = _lib.load_library()
    v_ref = torch.zeros(pl.v_bytes // 4, dtype=torch.float32, device=cuda)
    v = torch.zeros_like(v_ref)
    cn, cf = nxt.param._c(), fprm._c()
    assert lib.fhip_winograd_f63_input_transform(ctypes.byref(cn), batch, _ptr(v_ref), _ptr(mid), _stream()) == 0
    assert lib.fhip_winograd_f63_input_from_first(ctypes.byref(cf), ctypes.byref(cn), batch, _ptr(v), _ptr(xd), _ptr(fw), _ptr(fb), _stream()) == 0
    torch.cuda.synchronize()
    from feathercnn_amd.booster import winograd_rows
    a = winograd_rows(v, pl, oc)[:, :, :pl.columns].cpu().numpy()
    b = winograd_rows(v_ref, pl, oc)[:, :, :pl.columns].cpu().numpy()
    assert nerr(a, b) <= 2e-6


def test_refusals(cuda):
    from feathercnn_amd import ConvParam
    from feathercnn_amd.booster import can_fuse_first_winograd

    def prm(ic, h, w, k=3, stride=1, pad=1, oc=8):
        q = ConvParam(output_channels=oc, input_channels=ic, input_h=h, input_w=w, kernel_h=k, kernel_w=k, stride_h=stride, stride_w=stride,
                      pad_left=pad, pad_right=pad, pad_top=pad, pad_bottom=pad, group=1, bias_term=True, activation=1)
        q.AssignOutputDim()
        return q
    nxt, _, _ = _wino(cuda, 8, 8, 28, 28)
    assert can_fuse_first_winograd(prm(3, 28, 28), nxt, 4)
    assert not can_fuse_first_winograd(prm(5, 28, 28), nxt, 4)          # more than 4 image channels
    one, _, _ = _wino(cuda, 1, 8, 28, 28)
    assert not can_fuse_first_winograd(prm(1, 28, 28, oc=1), one, 4)    # 1 channel, group 1: a depthwise layer to ConvParam
    assert not can_fuse_first_winograd(prm(3, 28, 28, oc=9), nxt, 4)    # channel counts do not meet
    assert not can_fuse_first_winograd(prm(3, 28, 28, pad=0), nxt, 4)   # changes the size
    assert not can_fuse_first_winograd(prm(3, 56, 56, stride=2), nxt, 4)
    assert not can_fuse_first_winograd(prm(3, 28, 28, k=5, pad=2), nxt, 4)
    odd, _, _ = _wino(cuda, 8, 8, 28, 27)
    assert not can_fuse_first_winograd(prm(3, 28, 27), odd, 4)          # odd width: the float2 image reads straddle the edge
    big, _, _ = _wino(cuda, 8, 8, 224, 224)
    assert can_fuse_first_winograd(prm(3, 224, 224), big, 256)
    assert not can_fuse_first_winograd(prm(4, 224, 224), big, 1400)     # image tensor of 1 GiB or more
