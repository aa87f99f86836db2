"""The fused depthwise 3x3 + pointwise 1x1 route (fhip_conv_forward_dw_pw, MobileNet's dw/pw pairs) through the C-ABI:
against the two layers run one after the other through the same library (same arithmetic: equal to rounding), against the
CPU checker applied twice, and through the Net runtime at fusion level 2 against fusion level 1 (which never fuses the pair)."""
import ctypes

import numpy as np
import pytest

from oracle import Geom, nerr, synth

pytestmark = pytest.mark.gpu
TOL = 1e-4

# C, K, H, W, stride, batch, dw_act, pw_act, dw_bias, pw_bias
# (a pointwise layer with >= 16 k-tiles and < 512 output tiles runs split-K and is not fusable: the C = 256 case uses batch 24)
# and only 64 < K < 160 (stride 1) / 400 (stride 2) is fused: the range where one kernel beats two)
PAIRS = [(32, 96, 112, 112, 1, 2, 1, 1, 1, 1), (64, 128, 112, 112, 2, 2, 1, 1, 1, 1), (128, 128, 56, 56, 1, 3, 1, 1, 1, 1), (128, 256, 56, 56, 2, 2, 1, 1, 1, 1),
         (256, 144, 28, 28, 1, 24, 1, 1, 1, 1), (16, 72, 16, 16, 1, 3, 0, 1, 0, 1), (8, 200, 24, 16, 2, 2, 1, 0, 1, 0), (12, 100, 10, 16, 1, 2, 0, 0, 0, 0),
         (20, 65, 9, 8, 1, 5, 1, 1, 1, 1), (40, 72, 6, 16, 2, 3, 1, 1, 0, 1), (128, 390, 12, 24, 2, 2, 1, 1, 1, 1), (3, 130, 20, 24, 1, 1, 1, 1, 1, 1),
         # round 4: MobileNet's 112- and 56-pixel pair geometries with image heights that are not the benchmark's, no bias / no activation,
         # several blocks of output channels
         # (32 channels on 112-pixel rows behind a stride-1 depthwise layer -- MobileNet-V1's first pair -- take the band-staged, wave-specialised
         # kernel of dwpw_band.h: whole and partial last row groups, one and several blocks of 64 output channels, batches below / above the
         # persistent grid, no bias / no activation)
         (32, 64, 112, 112, 1, 2, 1, 1, 1, 1), (32, 64, 37, 112, 1, 3, 1, 1, 1, 1), (32, 128, 9, 112, 1, 1, 0, 0, 0, 0), (32, 64, 24, 112, 1, 70, 1, 1, 0, 1),
         (32, 128, 10, 112, 1, 2, 0, 1, 0, 1), (64, 128, 100, 112, 2, 2, 1, 1, 1, 1), (64, 256, 18, 112, 2, 1, 1, 0, 1, 0), (128, 128, 50, 56, 1, 2, 1, 1, 1, 1),
         (128, 128, 56, 56, 1, 2, 0, 0, 0, 0), (128, 256, 60, 56, 2, 2, 1, 1, 1, 1), (128, 384, 14, 56, 2, 1, 1, 1, 0, 1)]


def _layers(cuda, c, k, h, w, s, batch, dw_act, pw_act, dw_bias, pw_bias, seed):
    import torch
    from feathercnn_amd import ConvLayer, ConvParam, DEPTHWISE, IM2COL
    rng = np.random.default_rng(seed)
    wd = (rng.uniform(-1, 1, (c, 1, 3, 3)) / 3).astype(np.float32)
    bd = rng.uniform(-0.2, 0.2, c).astype(np.float32)
    wp = (rng.uniform(-1, 1, (k, c, 1, 1)) / np.sqrt(c)).astype(np.float32)
    bp = rng.uniform(-0.1, 0.1, k).astype(np.float32)
    x = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    pd = ConvParam(output_channels=c, input_channels=c, input_h=h, input_w=w, kernel_h=3, kernel_w=3, stride_h=s, stride_w=s, pad_left=1, pad_right=1,
                   pad_top=1, pad_bottom=1, group=c, bias_term=bool(dw_bias), activation=dw_act, batch=batch)
    pd.AssignOutputDim()
    pp = ConvParam(output_channels=k, input_channels=c, input_h=pd.output_h, input_w=pd.output_w, kernel_h=1, kernel_w=1, stride_h=1, stride_w=1,
                   group=1, bias_term=bool(pw_bias), activation=pw_act, batch=batch)
    t = lambda a: torch.from_numpy(a).to(cuda)
    ld = ConvLayer(pd, t(wd), t(bd) if dw_bias else None, algo=DEPTHWISE)
    lp = ConvLayer(pp, t(wp), t(bp) if pw_bias else None, algo=IM2COL)
    return ld, lp, t(x), (x, wd, bd, wp, bp)


@pytest.mark.parametrize("cfg", PAIRS, ids=lambda c: "c%dk%d_%dx%d_s%d_b%d_a%d%d_b%d%d" % c)
def test_fused_pair_equals_the_two_layers(cfg, cuda):
    import torch
    import oracle
    chk = oracle.best()  # the reference itself (oracle/_ref) where it is built, else the restatement
    from feathercnn_amd import _lib
    c, k, h, w, s, batch, dw_act, pw_act, dw_bias, pw_bias = cfg
    ld, lp, xt, (x, wd, bd, wp, bp) = _layers(cuda, *cfg, seed=h * w + c)
    lib = _lib.load_library()
    cd, cp = ld.param._c(), lp.param._c()
    assert lib.fhip_conv_can_fuse_dw_pw(ctypes.byref(cd), ctypes.byref(cp), batch) == 1
    mid = ld.Forward(xt)
    want = lp.Forward(mid)
    out = torch.full_like(want, float("nan"))
    rc = lib.fhip_conv_forward_dw_pw(ctypes.byref(cd), ctypes.byref(cp), batch, out.data_ptr(), xt.data_ptr(), ld.packed.data_ptr(),
                                     ld.bias.data_ptr() if dw_bias else None, lp.packed.data_ptr(), lp.bias.data_ptr() if pw_bias else None, None)
    assert rc == 0, _lib.load_library().fhip_last_error()
    torch.cuda.synchronize()
    assert torch.isfinite(out).all()
    scale = float(want.abs().max())
    assert float((out - want).abs().max()) <= 2e-6 * scale, "fused pair differs from dw -> pw through the same library"
    # and against the CPU checker applied twice
    gd = Geom(c, c, h, w, 3, 3, s, s, 1, 1, 1, 1, c, dw_bias, dw_act)
    m = chk.forward(gd, x, wd, bd if dw_bias else None)
    gp = Geom(c, k, m.shape[2], m.shape[3], 1, 1, 1, 1, 0, 0, 0, 0, 1, pw_bias, pw_act)
    ref = chk.forward(gp, m, wp, bp if pw_bias else None)
    assert nerr(out.cpu().numpy(), ref) <= TOL


def test_pairs_that_do_not_qualify_are_refused(cuda):
    from feathercnn_amd import _lib
    lib = _lib.load_library()
    bad = [(32, 96, 14, 14, 1, 2, 1, 1, 1, 1),    # W % 4 != 0
           (300, 96, 16, 16, 1, 2, 1, 1, 1, 1),   # more channels than the LDS tap table holds
           (32, 64, 16, 16, 1, 2, 1, 1, 1, 1),    # K <= 64: two kernels are as fast
           (32, 256, 16, 16, 1, 2, 1, 1, 1, 1),   # K >= 160 behind a stride-1 depthwise layer: the GEMM dominates, two kernels are as fast
           (32, 192, 9, 112, 1, 1, 1, 1, 1, 1)]   # the band-staged kernel's geometry, but more than two blocks of 64 output channels (round 5 bound)
    for cfg in bad:
        ld, lp, xt, _ = _layers(cuda, *cfg, seed=1)
        cd, cp = ld.param._c(), lp.param._c()
        assert lib.fhip_conv_can_fuse_dw_pw(ctypes.byref(cd), ctypes.byref(cp), cfg[5]) == 0
        import torch
        out = torch.empty((cfg[5], cfg[1], lp.param.output_h, lp.param.output_w), device=cuda)
        rc = lib.fhip_conv_forward_dw_pw(ctypes.byref(cd), ctypes.byref(cp), cfg[5], out.data_ptr(), xt.data_ptr(), ld.packed.data_ptr(),
                                         ld.bias.data_ptr(), lp.packed.data_ptr(), lp.bias.data_ptr(), None)
        assert rc == -1  # FHIP_E_UNSUPPORTED


def test_net_fusion_level_2_fuses_the_pairs_and_matches_level_1(cuda):
    """MobileNet-style stack through the Net runtime: level 2 runs the qualifying pairs as one kernel (fhip_net_layer_fused_pointwise
    reports them), the others (K = 64; a 6-pixel plane) fall back to dw -> pw inside the fused layer; both equal level 1."""
    from feathercnn_amd import model_zoo
    from feathercnn_amd.net import Net
    g = model_zoo.GraphBuilder(3)
    x = g.input("data", 3, 64, 64)
    x = g.conv_bn_relu("conv1", x, 3, 16, 3, 2, 1)
    for i, (c, k, s) in enumerate([(16, 96, 1), (96, 128, 2), (128, 64, 1), (64, 128, 2), (128, 128, 1)]):   # 32, 32->16, 16, 16->8, 8 pixels
        x = g.conv_bn_relu(f"dw{i}", x, c, c, 3, s, 1, group=c)
        x = g.conv_bn_relu(f"pw{i}", x, c, k, 1, 1, 0)
    x = g.conv_bn_relu("dw_odd", x, 128, 128, 3, 1, 0, group=128)   # 8 -> 6 pixels: pad 0, is not absorbed
    x = g.conv_bn_relu("pw_odd", x, 128, 32, 1, 1, 0)
    x = g.conv_bn_relu("dw_seq", x, 32, 32, 3, 1, 1, group=32)      # 6 pixels: absorbed, but W % 4 != 0 -> the two kernels in one layer
    x = g.conv_bn_relu("pw_seq", x, 32, 96, 1, 1, 0)
    p, b = g.finish()
    img = np.random.default_rng(8).uniform(-1, 1, (4, 3, 64, 64)).astype(np.float32)
    outs = {}
    for level in (1, 2):
        net = Net(fusion=level, tuned=True)
        net.LoadParam(p)
        net.LoadWeights(b)
        net.FeedInput("data", img)
        net.Forward()
        outs[level] = net.Extract("pw_seq_relu")
        if level == 2:
            fused = net.fused_pointwise()
            names = [(net.layers()[i][1], one) for i, (_, one) in sorted(fused.items())]
            # 32-, 16- and 8-pixel pairs run as one kernel; dw_odd (pad 0) is not even absorbed
            assert names == [("dw0", True), ("dw1", True), ("dw3", True), ("dw4", True), ("dw_seq", False)], names  # dw2 (K = 64) is not absorbed
            assert [n for t, n, _ in net.layers() if t == "Convolution"] == ["conv1", "pw2", "pw_odd"]  # every other pointwise layer was absorbed
    assert nerr(outs[2], outs[1]) <= 1e-5


def test_absorbed_pointwise_keeps_the_residual_fusion(cuda):
    """MobileNet-V2 style block dw3x3 -> pw(linear) -> Eltwise SUM (-> ReLU) with 64 < K < 160: the depthwise layer absorbs the pointwise
    one at LoadParam time, and the add still moves into the POINTWISE convolution's epilogue (it used to be refused, so the
    block ran conv + a separate add); the pair then runs its two kernels inside one layer.  Levels 2 / 3 equal level 0 and the layer count
    says the Eltwise layers are gone."""
    from feathercnn_amd import model_zoo
    from feathercnn_amd.net import Net
    g = model_zoo.GraphBuilder(11)
    x = g.input("data", 3, 32, 32)
    x = g.conv_bn_relu("stem", x, 3, 96, 3, 1, 1)
    for i in range(2):
        a, b_ = g.split(f"split{i}", x)
        y = g.conv_bn_relu(f"dw{i}", a, 96, 96, 3, 1, 1, group=96)
        y = g.conv_bn_relu(f"pw{i}", y, 96, 96, 1, 1, 0, relu=False)      # linear bottleneck
        x = g.eltwise(f"add{i}", y, b_)
        if i == 1:
            x = g.relu("add1_relu", x)
    p, b = g.finish()
    img = np.random.default_rng(12).uniform(-1, 1, (3, 3, 32, 32)).astype(np.float32)
    outs, counts = {}, {}
    for level in (0, 2, 3):
        net = Net(fusion=level, tuned=True)
        net.LoadParam(p)
        net.LoadWeights(b)
        net.FeedInput("data", img)
        net.Forward()
        outs[level] = net.Extract("add1_relu")
        counts[level] = [t for t, _, _ in net.layers()]
    assert "Eltwise" in counts[0] and "Eltwise" not in counts[2] and "Eltwise" not in counts[3], counts
    assert nerr(outs[2], outs[0]) <= 1e-5 and np.array_equal(outs[3], outs[2])
