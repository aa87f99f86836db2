"""The three benchmark topologies at resolutions other than 224 x 224: several fast paths are cut for the plane
sizes of a 224-pixel input (flat depthwise for 7 / 14 / 28-pixel planes, band depthwise for 112 / 56, F(4x4,3x3) on 7 / 8-pixel planes, the band-
staged depthwise + pointwise kernel on 112-pixel rows).  At 192 / 256 / 160 pixels the same nets run through the general kernels: the results must
still match the reference feather::Net (looped over the batch) -- bench.py's configuration (fusion 3, MI355X routing, branch stream, graph) --
and the per-image time is printed next to the 224-pixel one, scaled by the pixel count, so a fall-back cliff is visible in the test log
(tools/resolution_bench.py gives the table of DESIGN.md 5)."""
import time

import numpy as np
import pytest

from feathercnn_amd import model_zoo
from oracle import netcheck, nerr

pytestmark = pytest.mark.gpu
TOL = 1e-4
LOGITS = {"mobilenet_v1": "fc7", "resnet50": "fc1000", "vgg16": "fc8"}


def _net(model, x, graph=True):
    from feathercnn_amd.net import Net
    p, b, i, o = model
    net = Net(fusion=3, graph=graph, tuned=True, concurrency=True)
    net.LoadParam(p)
    net.LoadWeights(b)
    net.FeedInput(i, x)
    net.Forward()
    return net


@pytest.mark.parametrize("name,size,batch", [("mobilenet_v1", 192, 4), ("resnet50", 256, 3), ("vgg16", 160, 2), ("mobilenet_v1", 160, 2), ("resnet50", 192, 2)])
def test_other_resolutions_match_the_reference(cuda, name, size, batch):
    model = model_zoo.MODELS[name](size=size)
    p, b, i, o = model
    x = np.random.default_rng(size + batch).uniform(-1, 1, (batch, 3, size, size)).astype(np.float32)
    net = _net(model, x)
    net.Forward()  # a graph replay
    prob, logits = net.Extract(o), net.Extract(LOGITS[name])
    if netcheck.have_ref_net():
        ref = netcheck.RefNet(p, b)
        want, want_logits = ref.run(i, x, o), ref.run(i, x, LOGITS[name])
        ref.close()
    else:
        blobs = netcheck.PortNet(p, b).run(i, x, o, keep=True)
        want, want_logits = blobs[o], blobs[LOGITS[name]]
    assert nerr(logits, want_logits) <= TOL and nerr(prob, want) <= TOL
    assert np.array_equal(prob.reshape(batch, -1).argmax(1), want.reshape(batch, -1).argmax(1))
    net.close()


@pytest.mark.parametrize("name,size,batch", [("mobilenet_v1", 192, 64), ("resnet50", 256, 32)])
def test_fall_back_cost_is_visible(cuda, name, size, batch):
    """Timing smoke (nothing asserted but sanity): images/s at `size` against 224, per pixel.  A ratio well under 1 is a fall-back cliff."""
    import torch
    rates = {}
    for sz in (224, size):
        model = model_zoo.MODELS[name](size=sz)
        x = np.random.default_rng(1).uniform(-1, 1, (batch, 3, sz, sz)).astype(np.float32)
        net = _net(model, x)
        for _ in range(3):
            net.Forward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            net.Forward()
        torch.cuda.synchronize()
        rates[sz] = batch * 10 / (time.perf_counter() - t0)
        net.close()
    per_pixel = rates[size] * size * size / (rates[224] * 224 * 224)
    print(f"\n{name} b{batch}: {rates[224]:.0f} img/s @224, {rates[size]:.0f} img/s @{size} = {per_pixel:.2f} of the 224-pixel rate per pixel")
    assert rates[size] > 0 and per_pixel > 0.3


# every plane size the flat depthwise kernel is instantiated for (depthwise.hip FHIP_DW_FLAT_SIZES), both strides, with a channel count that leaves the
# last chunk ragged and a plane count that is not a multiple of the chunk size: parity against the checker, and against the direct kernel's route on
# a neighbouring size that is NOT in the list (a fall-back must give the same values)
FLAT_SIZES = (5, 6, 7, 8, 9, 10, 12, 14, 16, 18, 20, 24, 28, 32, 36)


@pytest.mark.parametrize("h", FLAT_SIZES + (11, 22))
@pytest.mark.parametrize("stride", (1, 2))
def test_flat_depthwise_plane_sizes(cuda, h, stride):
    import torch

    import oracle
    from feathercnn_amd import DEPTHWISE, ConvLayer, ConvParam
    from oracle import conv_geom, synth
    c, batch = 37, 5  # 185 planes: never a multiple of a chunk
    g = conv_geom(c, c, h, 3, stride, 1, group=c)
    x, w, b = synth(g, batch, seed=h * 10 + stride)
    p = ConvParam(output_channels=c, input_channels=c, input_h=h, input_w=h, kernel_h=3, kernel_w=3, stride_h=stride, stride_w=stride, pad_left=1,
                  pad_right=1, pad_top=1, pad_bottom=1, group=c, bias_term=True, activation=1, batch=batch)
    dev = torch.device("cuda:0")
    layer = ConvLayer(p, torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev), algo=DEPTHWISE)
    y = layer.Forward(torch.from_numpy(x).to(dev)).cpu().numpy()
    want = oracle.best().forward(g, x, w, b)
    assert y.shape == want.shape and nerr(y, want) <= 1e-5


# every plane size the band depthwise kernel is instantiated for (depthwise.hip FHIP_DW_BAND_SIZES), stride 1, plus two neighbours that are not (direct kernel)
@pytest.mark.parametrize("h", (40, 48, 56, 64, 72, 80, 96, 112, 128, 144, 44, 100))
def test_band_depthwise_plane_sizes(cuda, h):
    import torch

    import oracle
    from feathercnn_amd import DEPTHWISE, ConvLayer, ConvParam
    from oracle import conv_geom, synth
    c, batch = 5, 3
    g = conv_geom(c, c, h, 3, 1, 1, group=c)
    x, w, b = synth(g, batch, seed=h)
    p = ConvParam(output_channels=c, input_channels=c, input_h=h, input_w=h, kernel_h=3, kernel_w=3, stride_h=1, stride_w=1, pad_left=1, pad_right=1,
                  pad_top=1, pad_bottom=1, group=c, bias_term=True, activation=1, batch=batch)
    dev = torch.device("cuda:0")
    layer = ConvLayer(p, torch.from_numpy(w).to(dev), torch.from_numpy(b).to(dev), algo=DEPTHWISE)
    y = layer.Forward(torch.from_numpy(x).to(dev)).cpu().numpy()
    want = oracle.best().forward(g, x, w, b)
    assert y.shape == want.shape and nerr(y, want) <= 1e-5
