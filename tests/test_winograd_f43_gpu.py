"""F(4x4,3x3) on planes of 7 or 8 output pixels per side (round 4): the WINOGRADF63 route of this library runs such layers -- ResNet-50's res5
3x3 convolutions under the tuned routing -- with 2 x 2 tiles of 4 x 4 outputs and 36 frequency points instead of 2 x 2 tiles of 6 x 6 and 64
(fhip_winograd_plan.frequency_points / tile_outputs).  Checked through the C-ABI against the oracle (the reference computes these layers with
IM2COL, avx/booster.cpp:289), against this library's own IM2COL route, and stage by stage against a numpy restatement of Lavin's matrices."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from oracle import conv_geom, nerr, synth

pytestmark = pytest.mark.gpu

_BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], np.float64)
_G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]], np.float64)
_AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], np.float64)

# C, K, H, W, pad, batch
CASES = [(16, 24, 7, 7, 1, 5), (512, 512, 7, 7, 1, 4), (64, 64, 8, 8, 1, 40), (20, 36, 7, 8, 1, 3), (8, 12, 9, 9, 0, 2), (32, 160, 8, 7, 1, 70),
         (16, 32, 7, 7, 1, 300),   # 1200 columns, <= 64 output channels: V / M in column blocks of 1024 with 36 frequency points
         (12, 20, 10, 9, 0, 3)]    # 8 x 7 outputs from an unpadded 10 x 9 image


def _param(c, k, h, w, pad, batch, act=1, bias=True):
    from feathercnn_amd import ConvParam
    return ConvParam(output_channels=k, input_channels=c, input_h=h, input_w=w, kernel_h=3, kernel_w=3, stride_h=1, stride_w=1, pad_left=pad, pad_right=pad,
                     pad_top=pad, pad_bottom=pad, group=1, bias_term=bias, activation=act, batch=batch)


@pytest.mark.parametrize("c,k,h,w,pad,batch", CASES)
def test_f43_planes_match_the_oracle_and_the_im2col_route(cuda, c, k, h, w, pad, batch):
    from feathercnn_amd import ConvLayer, booster
    from feathercnn_amd.booster import IM2COL, WINOGRADF63
    g = conv_geom(c, k, h, 3, 1, pad, w=w)
    x, wt, b = synth(g, batch, seed=c * 7 + k)
    p = _param(c, k, h, w, pad, batch)
    p.AssignOutputDim()
    pl = booster.winograd_plan(p)
    assert pl.frequency_points == 36 and pl.tile_outputs == 4 and pl.tiles_x == (p.output_w + 3) // 4 and pl.tiles_y == (p.output_h + 3) // 4
    xt = torch.from_numpy(x).to(cuda)
    wino = ConvLayer(p, torch.from_numpy(wt).to(cuda), torch.from_numpy(b).to(cuda), algo=WINOGRADF63)
    got = wino.Forward(xt).cpu().numpy()
    want = oracle.best().forward(g, x, wt, b)
    assert got.shape == want.shape and nerr(got, want) <= 2e-5, (c, k, h, w, pad, batch, nerr(got, want))
    gemm = ConvLayer(_param(c, k, h, w, pad, batch), torch.from_numpy(wt).to(cuda), torch.from_numpy(b).to(cuda), algo=IM2COL)
    assert nerr(got, gemm.Forward(xt).cpu().numpy()) <= 2e-5
    # garbage in the scratch arena, determinism
    scratch = torch.full((max(wino.buffer_bytes // 4, 1),), float("nan"), device=cuda)
    assert np.array_equal(got, wino.Forward(xt, scratch=scratch).cpu().numpy())


def test_which_planes_take_f43(cuda):
    from feathercnn_amd import _lib, booster
    lib = _lib.load_library()
    for h, w, want in ((7, 7, 36), (8, 8, 36), (7, 8, 36), (6, 6, 64), (5, 5, 64), (9, 9, 64), (14, 14, 64), (8, 12, 64), (4, 7, 36), (56, 56, 64)):
        p = _param(8, 8, h, w, 1, 2)
        p.AssignOutputDim()
        assert booster.winograd_plan(p).frequency_points == want, (h, w)
    # the fused forms are F(6,3)-only
    a, b = _param(8, 8, 8, 8, 1, 2), _param(8, 8, 8, 8, 1, 2)
    a.AssignOutputDim(), b.AssignOutputDim()
    ca, cb = a._c(), b._c()
    assert lib.fhip_conv_can_chain_winograd(ctypes.byref(ca), 4, ctypes.byref(cb), 4, 0) == 0
    assert lib.fhip_conv_can_fuse_maxpool2(ctypes.byref(ca), 4) == 0


def test_f43_stage_api(cuda):
    """U, V, M and the output against numpy, stage by stage (the F(6,3) twin of this test is tests/test_parity_gpu.py::test_winograd_stage_api)."""
    from feathercnn_amd import _lib, booster
    lib = _lib.load_library()
    c_, k_, h, w, n = 12, 20, 7, 8, 6
    g = conv_geom(c_, k_, h, 3, 1, 1, w=w)
    x, wt, b = synth(g, n, seed=5)
    p = _param(c_, k_, h, w, 1, n)
    p.AssignOutputDim()
    pl = booster.winograd_plan(p)
    T, TX, TY, P, Pp = pl.tiles_per_image, pl.tiles_x, pl.tiles_y, pl.columns, pl.columns_padded
    Cp, Kp = pl.in_channels_padded, pl.out_channels_padded
    cp = p._c()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dv = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    xt, wtt, bt = (torch.from_numpy(a).to(cuda) for a in (x, wt, b))
    U = torch.full((36, Cp, Kp), float("nan"), device=cuda)
    V = torch.full((36, c_, Pp), float("nan"), device=cuda)
    M = torch.full((36, k_, Pp), float("nan"), device=cuda)
    out = torch.full((n, k_, p.output_h, p.output_w), float("nan"), device=cuda)
    assert pl.u_bytes == U.numel() * 4 and pl.v_bytes >= V.numel() * 4 and pl.m_bytes >= M.numel() * 4
    assert lib.fhip_winograd_f63_transform_kernel(ctypes.byref(cp), dv(U), dv(wtt), st) == 0
    assert lib.fhip_winograd_f63_input_transform(ctypes.byref(cp), n, dv(V), dv(xt), st) == 0
    assert lib.fhip_winograd_f63_tile_gemm(ctypes.byref(cp), n, dv(M), dv(U), dv(V), st) == 0
    assert lib.fhip_winograd_f63_output_transform(ctypes.byref(cp), n, dv(out), dv(M), dv(bt), st) == 0
    torch.cuda.synchronize()
    Uref = np.einsum("ia,kcab,jb->ijck", _G, wt.astype(np.float64), _G).reshape(36, c_, k_)
    Ug = U.cpu().numpy()
    assert nerr(Ug[:, :c_, :k_], Uref) <= 1e-6 and np.all(Ug[:, c_:, :] == 0) and np.all(Ug[:, :, k_:] == 0)
    xp = np.zeros((n, c_, 4 * TY + 2, 4 * TX + 2))
    xp[:, :, 1:1 + h, 1:1 + w] = x
    patches = np.stack([xp[:, :, 4 * ty:4 * ty + 6, 4 * tx:4 * tx + 6] for ty in range(TY) for tx in range(TX)], axis=2)
    Vref = np.einsum("ia,nctab,jb->ijcnt", _BT, patches, _BT).reshape(36, c_, P)
    assert nerr(V.cpu().numpy()[:, :, :P], Vref) <= 1e-5
    Mref = np.einsum("xck,xcp->xkp", Uref, Vref)
    assert nerr(M.cpu().numpy()[:, :, :P], Mref) <= 1e-5
    Y = np.einsum("ai,ijknt,bj->nktab", _AT, Mref.reshape(6, 6, k_, n, T), _AT)
    full = np.zeros((n, k_, 4 * TY, 4 * TX))
    for t in range(T):
        ty, tx = divmod(t, TX)
        full[:, :, 4 * ty:4 * ty + 4, 4 * tx:4 * tx + 4] = Y[:, :, t]
    yref = np.maximum(full[:, :, :p.output_h, :p.output_w] + b[None, :, None, None], 0)
    assert nerr(out.cpu().numpy(), yref) <= 1e-5
