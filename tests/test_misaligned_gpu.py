"""The alignment contract of include/feather_hip/feather_hip.h ("tensor pointers need 4-byte alignment, nothing more"): every route of the
hot path with its input AND output tensors placed 4 bytes past a 16-byte boundary must give exactly the values of the aligned run (and so the
checker's).  Routes: flat / band / direct / chunk depthwise, the fused depthwise + pointwise GEMM and its band-staged form, 1x1 implicit GEMM
(aligned-plane and ragged-plane modes, with residual), the streamed 1x1 kernel, the 7x7 and 3x3 first-layer kernel, Winograd F(6,3) and F(4,3)."""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _off(t, k=1):
    """A copy of tensor t whose storage starts k floats past an aligned allocation."""
    import torch
    big = torch.empty(t.numel() + 8, dtype=t.dtype, device=t.device)
    v = big[k:k + t.numel()].view(t.shape)
    v.copy_(t)
    assert v.data_ptr() % 16 == 4 * k
    return v


CASES = [  # name, C, K, H, kernel, stride, pad, group, batch
    ("dw flat 14", 24, 24, 14, 3, 1, 1, 24, 3), ("dw flat 20 s2", 16, 16, 20, 3, 2, 1, 16, 3), ("dw band 56", 8, 8, 56, 3, 1, 1, 8, 2),
    ("dw direct 30", 8, 8, 30, 3, 1, 1, 8, 2), ("dw chunk 7 s2", 12, 12, 7, 3, 2, 1, 12, 5),
    ("1x1 aligned planes", 64, 128, 28, 1, 1, 0, 1, 2), ("1x1 ragged planes", 64, 128, 7, 1, 1, 0, 1, 4), ("1x1 stride 2", 64, 128, 28, 1, 2, 0, 1, 2),
    ("1x1 streamed", 256, 128, 14, 1, 1, 0, 1, 8), ("first 7x7 s2", 3, 64, 64, 7, 2, 3, 1, 2), ("first 3x3", 3, 32, 32, 3, 1, 1, 1, 2),
    ("winograd f63", 32, 64, 28, 3, 1, 1, 1, 2), ("winograd f63 small K", 16, 32, 20, 3, 1, 1, 1, 2), ("winograd f43", 32, 128, 7, 3, 1, 1, 1, 4),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_routes_at_4_byte_alignment(cuda, case):
    import torch

    import oracle
    from feathercnn_amd import ConvLayer, ConvParam
    from oracle import conv_geom, nerr, synth
    name, c, k, h, ks, s, p, g, batch = case
    geom = conv_geom(c, k, h, ks, s, p, group=g)
    x, w, b = synth(geom, batch, seed=len(name))
    dev = torch.device("cuda:0")
    prm = ConvParam(output_channels=k, input_channels=c, input_h=h, input_w=h, kernel_h=ks, kernel_w=ks, stride_h=s, stride_w=s, pad_left=p, pad_right=p,
                    pad_top=p, pad_bottom=p, group=g, bias_term=True, activation=1, batch=batch)
    layer = ConvLayer(prm, torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev), tuned=True)
    xa = torch.from_numpy(x).to(dev)
    ya = layer.Forward(xa)
    layer.bias = _off(layer.bias)
    yo = _off(torch.zeros_like(ya))
    layer.Forward(_off(xa), out=yo)
    torch.cuda.synchronize()
    assert torch.equal(ya, yo), f"{name}: the 4-byte-aligned run differs from the aligned one"
    assert nerr(ya.cpu().numpy(), oracle.best().forward(geom, x, w, b)) <= 1e-4


def test_fused_pairs_and_residual_at_4_byte_alignment(cuda):
    import torch

    from feathercnn_amd import DEPTHWISE, IM2COL, ConvLayer, ConvParam, _lib
    dev = torch.device("cuda:0")
    lib = _lib.load_library()
    rng = np.random.default_rng(5)
    t = lambda a: torch.from_numpy(a.astype(np.float32)).to(dev)  # noqa: E731
    # depthwise + pointwise pairs: the GEMM-fused form (64 -> 128 at 56 px, stride 1 and 2) and the band-staged form (32 -> 64 on 112-pixel rows)
    for c, k, h, s, batch in ((64, 128, 56, 1, 2), (64, 128, 56, 2, 2), (32, 64, 112, 1, 2)):
        pd = ConvParam(output_channels=c, input_channels=c, input_h=h, input_w=h, kernel_h=3, kernel_w=3, stride_h=s, stride_w=s, pad_left=1, pad_right=1,
                       pad_top=1, pad_bottom=1, group=c, bias_term=True, activation=1, batch=batch)
        pd.AssignOutputDim()
        pp = ConvParam(output_channels=k, input_channels=c, input_h=pd.output_h, input_w=pd.output_w, kernel_h=1, kernel_w=1, stride_h=1, stride_w=1, group=1,
                       bias_term=True, activation=1, batch=batch)
        ld = ConvLayer(pd, t(rng.uniform(-1, 1, (c, 1, 3, 3)) / 3), t(rng.uniform(-.2, .2, c)), algo=DEPTHWISE)
        lp = ConvLayer(pp, t(rng.uniform(-1, 1, (k, c, 1, 1)) / np.sqrt(c)), t(rng.uniform(-.1, .1, k)), algo=IM2COL)
        cd, cp = pd._c(), pp._c()
        if not lib.fhip_conv_can_fuse_dw_pw(ctypes.byref(cd), ctypes.byref(cp), batch):
            continue
        x = torch.rand((batch, c, h, h), device=dev) * 2 - 1
        outs = []
        for shift in (False, True):
            xi = _off(x) if shift else x
            o = torch.zeros((batch, k, pd.output_h, pd.output_w), device=dev)
            o = _off(o) if shift else o
            rc = lib.fhip_conv_forward_dw_pw(ctypes.byref(cd), ctypes.byref(cp), batch, ctypes.c_void_p(o.data_ptr()), ctypes.c_void_p(xi.data_ptr()),
                                             ctypes.c_void_p(ld.packed.data_ptr()), ctypes.c_void_p(ld.bias.data_ptr()), ctypes.c_void_p(lp.packed.data_ptr()),
                                             ctypes.c_void_p(lp.bias.data_ptr()), None)
            assert rc == 0
            outs.append(o)
        torch.cuda.synchronize()
        assert torch.equal(outs[0], outs[1]), (c, k, h, s)
    # 1x1 convolution with the fused residual operand
    c, k, h, batch = 64, 256, 28, 2
    p = ConvParam(output_channels=k, input_channels=c, input_h=h, input_w=h, kernel_h=1, kernel_w=1, stride_h=1, stride_w=1, group=1, bias_term=True, activation=1,
                  batch=batch)
    lyr = ConvLayer(p, t(rng.uniform(-1, 1, (k, c, 1, 1)) / 8), t(rng.uniform(-.1, .1, k)), algo=IM2COL)
    x, res = torch.rand((batch, c, h, h), device=dev), torch.rand((batch, k, h, h), device=dev)
    cpar = p._c()
    scratch = torch.empty(max(lyr.buffer_bytes // 4, 64), device=dev)
    outs = []
    for shift in (False, True):
        xi, ri = (_off(x), _off(res)) if shift else (x, res)
        o = torch.zeros((batch, k, h, h), device=dev)
        o = _off(o) if shift else o
        assert lib.fhip_conv_forward_residual(ctypes.byref(cpar), IM2COL, batch, ctypes.c_void_p(o.data_ptr()), ctypes.c_void_p(xi.data_ptr()),
                                              ctypes.c_void_p(lyr.packed.data_ptr()), ctypes.c_void_p(scratch.data_ptr()), ctypes.c_void_p(lyr.bias.data_ptr()),
                                              ctypes.c_void_p(ri.data_ptr()), None) == 0
        outs.append(o)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
