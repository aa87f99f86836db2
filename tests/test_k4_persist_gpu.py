"""The persistent, prefetching form of the staged Winograd output transform (round 5, wino_output_transform_persist_kernel; replaces reference
winogradOutputTransform, src/booster/avx/winograd_kernels_F63.cpp:1088-1269, where the one-shot grid is more than one round of resident blocks).
Same butterflies on the same values: a large batch (persistent kernel) must give, image for image, the BITS a small batch gives (one-shot kernel:
its grid is resident at once), and both must match the oracle.  Also through the fused 2x2 max pooling."""
import ctypes

import numpy as np
import pytest
import torch

import oracle
from oracle import conv_geom, nerr, synth

pytestmark = pytest.mark.gpu


def _layer(cuda, c, k, h, batch, act=1, bias=True, wb=None):
    from feathercnn_amd import ConvLayer, ConvParam
    from feathercnn_amd.booster import WINOGRADF63
    g = conv_geom(c, k, h, 3, 1, 1, act=act, bias=int(bias))
    x, w, b = synth(g, batch, seed=c + k + h)
    if wb is not None:  # the same filters and bias as another batch size (synth draws the input first, so its weights depend on the batch)
        w, b = wb
    p = ConvParam(output_channels=k, input_channels=c, input_h=h, input_w=h, kernel_h=3, kernel_w=3, stride_h=1, stride_w=1, pad_left=1, pad_right=1,
                  pad_top=1, pad_bottom=1, group=1, bias_term=bias, activation=act, batch=batch)
    return g, x, w, b, p, ConvLayer(p, torch.from_numpy(w).to(cuda), torch.from_numpy(b).to(cuda) if bias else None, algo=WINOGRADF63)


# (C, K, H, big batch, small batch): ResNet-50's 56- and 28-pixel 3x3 layers, a ragged plane (30 = 5 tiles, the last one clipped), no ReLU / no bias
@pytest.mark.parametrize("c,k,h,big,small,act,bias", [(64, 64, 56, 64, 4, 1, True), (128, 128, 28, 64, 8, 1, True), (32, 96, 30, 96, 4, 0, True),
                                                      (32, 64, 42, 64, 4, 1, False)])
def test_persistent_output_transform_equals_one_shot_bit_for_bit(cuda, c, k, h, big, small, act, bias):
    g, x, w, b, p, layer = _layer(cuda, c, k, h, big, act, bias)
    got = layer.Forward(torch.from_numpy(x).to(cuda)).cpu().numpy()
    _, _, _, _, ps, small_layer = _layer(cuda, c, k, h, small, act, bias, wb=(w, b))
    parts = [small_layer.Forward(torch.from_numpy(x[i:i + small]).to(cuda)).cpu().numpy() for i in range(0, big, small)]
    assert np.array_equal(got, np.concatenate(parts))
    n = min(big, 4)
    want = oracle.best().forward(g, x[:n], w, b if bias else None)
    assert nerr(got[:n], want) <= 1e-4


def test_persistent_output_transform_with_fused_pooling(cuda):
    from feathercnn_amd import _lib
    from feathercnn_amd.booster import WINOGRADF63
    lib = _lib.load_library()
    c, k, h, big, small = 32, 64, 56, 64, 4
    g, x, w, b, p, layer = _layer(cuda, c, k, h, big)
    _, _, _, _, ps, small_layer = _layer(cuda, c, k, h, small, wb=(w, b))

    def pooled(lyr, prm, xs):
        n = xs.shape[0]
        cp = prm._c()
        xt = torch.from_numpy(xs).to(cuda)
        out = torch.full((n, k, h // 2, h // 2), float("nan"), device=cuda)
        scratch = torch.empty(max(lyr.buffer_bytes // 4, 1), device=cuda)
        assert lib.fhip_conv_forward_maxpool2(ctypes.byref(cp), WINOGRADF63, n, out.data_ptr(), xt.data_ptr(), lyr.packed.data_ptr(), scratch.data_ptr(),
                                              lyr.bias.data_ptr(), None) == 0
        torch.cuda.synchronize()
        return out.cpu().numpy()

    got = pooled(layer, p, x)
    parts = [pooled(small_layer, ps, x[i:i + small]) for i in range(0, big, small)]
    assert np.array_equal(got, np.concatenate(parts))
    want = oracle.best().forward(g, x[:2], w, b)
    want = want.reshape(2, k, h // 2, 2, h // 2, 2).max(axis=(3, 5))
    assert nerr(got[:2], want) <= 1e-4
