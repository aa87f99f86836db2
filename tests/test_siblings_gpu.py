"""Two 1x1 convolutions of the same input as one GEMM (fhip_conv_forward_siblings): each output must equal the layer run on its own --
bit for bit where that layer takes the same LDS-tiled route, to rounding where it splits the reduction or streams -- and the reference."""
import numpy as np
import pytest

from oracle import conv_geom, nerr

pytestmark = pytest.mark.gpu


def _pair(cuda, c, ka, kb, h, stride, batch, bias_a=True, bias_b=True, relu_a=False, relu_b=True, w=None, seed=0):
    import torch

    from feathercnn_amd import ConvParam
    from feathercnn_amd.booster import SiblingConvs
    rng = np.random.default_rng(seed)
    w = h if w is None else w

    def one(k, bias, relu):
        wt = (rng.standard_normal((k, c, 1, 1)) / np.sqrt(c)).astype(np.float32)
        b = rng.uniform(-0.5, 0.5, k).astype(np.float32) if bias else None
        prm = ConvParam(output_channels=k, input_channels=c, input_h=h, input_w=w, kernel_h=1, kernel_w=1, stride_h=stride, stride_w=stride, pad_left=0,
                        pad_right=0, pad_top=0, pad_bottom=0, group=1, bias_term=bias, activation=1 if relu else 0, batch=batch)
        return prm, wt, b
    pa, wa, ba = one(ka, bias_a, relu_a)
    pb, wb, bb = one(kb, bias_b, relu_b)
    t = lambda a: None if a is None else torch.from_numpy(a).to(cuda)
    sib = SiblingConvs(pa, t(wa), t(ba), pb, t(wb), t(bb))
    x = rng.uniform(-1, 1, (batch, c, h, w)).astype(np.float32)
    return sib, x, (pa, wa, ba), (pb, wb, bb)


# (name, C, Ka, Kb, H, stride, batch, bias_a, bias_b, relu_a, relu_b, W)
CASES = [
    ("res3a_like", 64, 128, 32, 28, 2, 12, True, True, False, True, None),
    ("res2a_like_stride1", 16, 128, 16, 28, 1, 6, True, True, False, True, None),
    ("three_row_tiles_ragged_second", 24, 256, 72, 14, 2, 40, True, False, True, True, None),  # Kb = 72: a partial last row tile
    ("no_bias_at_all", 8, 128, 128, 20, 1, 8, False, False, False, False, None),
    ("odd_plane_not_wide", 20, 128, 40, 7, 1, 96, False, True, True, False, None),  # 49 pixels: element-wise stores, columns span images
    ("rectangular_stride2", 12, 128, 24, 18, 2, 30, True, True, False, True, 30),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_sibling_pair_equals_the_two_layers(cuda, checker, case):
    import torch

    from feathercnn_amd import ConvLayer
    from feathercnn_amd.booster import IM2COL
    name, c, ka, kb, h, stride, batch, bias_a, bias_b, relu_a, relu_b, w = case
    sib, x, (pa, wa, ba), (pb, wb, bb) = _pair(cuda, c, ka, kb, h, stride, batch, bias_a, bias_b, relu_a, relu_b, w=w, seed=31)
    assert sib.applicable(batch), name
    xd = torch.from_numpy(x).to(cuda)
    ya, yb = sib.Forward(xd)
    torch.cuda.synchronize()
    for y, (p, wt, b) in ((ya, (pa, wa, ba)), (yb, (pb, wb, bb))):
        lone = ConvLayer(p, torch.from_numpy(wt).to(cuda), None if b is None else torch.from_numpy(b).to(cuda), algo=IM2COL)
        want = lone.Forward(xd)
        assert y.shape == want.shape
        assert nerr(y.cpu().numpy(), want.cpu().numpy()) <= 2e-6, name
        ref = checker.forward(conv_geom(c, p.output_channels, h, 1, stride, 0, bias=1 if b is not None else 0, act=int(p.activation), w=w), x, wt, b)
        assert nerr(y.cpu().numpy(), ref) <= 1e-4, name


def test_refusals(cuda):
    sib, *_ = _pair(cuda, 16, 96, 32, 28, 1, 8)        # Ka is not a multiple of the 128-row tile
    assert not sib.applicable(8)
    sib, *_ = _pair(cuda, 512, 128, 32, 28, 1, 1)      # 26 tiles of 32 k-tiles: the combined grid would run split-K
    assert not sib.applicable(1)
    sib, *_ = _pair(cuda, 16, 128, 32, 2, 1, 4)        # 16 columns: the narrow-N route
    assert not sib.applicable(4)
    import ctypes

    from feathercnn_amd import ConvParam, _lib
    a = ConvParam(output_channels=128, input_channels=16, input_h=28, input_w=28, kernel_h=1, kernel_w=1, stride_h=1, stride_w=1, group=1, batch=8)
    b = ConvParam(output_channels=32, input_channels=16, input_h=28, input_w=28, kernel_h=1, kernel_w=1, stride_h=2, stride_w=2, group=1, batch=8)
    a.AssignOutputDim()
    b.AssignOutputDim()
    ca, cb = a._c(), b._c()
    assert not _lib.load_library().fhip_conv_can_fuse_siblings(ctypes.byref(ca), 1, ctypes.byref(cb), 1, 8)  # different strides
    assert not _lib.load_library().fhip_conv_can_fuse_siblings(ctypes.byref(ca), 1, ctypes.byref(ca), 4, 8)  # not both IM2COL


def test_net_runs_projection_and_first_main_branch_layer_as_one_gemm(cuda):
    """A ResNet-style block: projection shortcut and the first 1x1 layer of the main branch read the same blob through a Split -- at
    fusion level >= 2 the pair is one launch (siblings() says which layer launches), results equal level 1."""
    from feathercnn_amd import model_zoo
    from feathercnn_amd.net import Net
    g = model_zoo.GraphBuilder(21)
    x = g.input("data", 32, 28, 28)
    x = g.relu("r0", g.conv("c0", x, 32, 48, 3, 1, 1))
    s0, s1 = g.split("sp", x)
    short = g.conv("proj", s0, 48, 128, 1, 2, 0)                # projection shortcut: 1x1 / stride 2, no activation
    y = g.relu("r1", g.conv("main_a", s1, 48, 96, 1, 2, 0))     # first layer of the main branch: same input, same stride
    y = g.relu("r2", g.conv("main_b", y, 96, 96, 3, 1, 1))
    y = g.conv("main_c", y, 96, 128, 1, 1, 0)
    out = g.relu("r3", g.eltwise("add", y, short))
    p, b = g.finish()
    img = np.random.default_rng(8).uniform(-1, 1, (6, 32, 28, 28)).astype(np.float32)
    outs = {}
    for level in (1, 2, 3):
        net = Net(fusion=level, tuned=True, concurrency=(level == 3), graph=(level == 3))
        net.LoadParam(p)
        net.LoadWeights(b)
        net.FeedInput("data", img)
        for _ in range(2):
            net.Forward()
        outs[level] = net.Extract("r3")
        names = {net.layers()[k][1]: v for k, v in net.siblings().items()}
        assert names == ({} if level == 1 else {"proj": 1, "main_a": 2}), (level, names)
        if level >= 2:
            assert net.Extract("proj").shape == (6, 128, 14, 14)  # both blobs keep their storage
        net.close()
    assert nerr(outs[2], outs[1]) <= 1e-5 and nerr(outs[3], outs[1]) <= 1e-5
    # a batch at which the stacked grid would need split-K: the pair goes back to two launches (re-planned on Reshape)
    net = Net(fusion=2, tuned=True)
    net.LoadParam(p)
    net.LoadWeights(b)
    net.FeedInput("data", img)
    net.Forward()
    assert net.siblings()
    ref = Net(fusion=1, tuned=True)
    ref.LoadParam(p)
    ref.LoadWeights(b)
    img1 = img[:1, :, :12, :12].copy()
    for n_ in (net, ref):
        n_.FeedInput("data", img1)
        n_.Forward()
    assert nerr(net.Extract("r3"), ref.Extract("r3")) <= 1e-5
