"""Parity of the layers between the convolutions and of the whole-net runtime on the GPU.

Every result is compared with a CPU checker on the same seeded inputs: the committed fixtures the REAL reference
feather::Net produced (tests/golden/net_golden.npz), the live reference where its prebuilt .so travelled to the box, and
the numpy/C restatement (oracle.netcheck.PortNet) for batches and for configurations the reference crashes on.
Tolerance: normalised max error <= 1e-4; elementwise layers are exact or within 1 ulp."""
import os

import numpy as np
import pytest

from feathercnn_amd import model_zoo
from oracle import nerr, netcheck

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "net_golden.npz")
TOL = 1e-4


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).to(dev)


# ---- layer kernels -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("count", [1, 3, 4, 1000, 4099, 1 << 20])
def test_relu_and_add_exact(cuda, count):
    from feathercnn_amd import net as fnet
    rng = np.random.default_rng(count)
    a, b = rng.uniform(-1, 1, count).astype(np.float32), rng.uniform(-1, 1, count).astype(np.float32)
    ta, tb = _t(a, cuda), _t(b, cuda)
    assert np.array_equal(fnet.relu(ta).cpu().numpy(), np.maximum(a, 0))
    assert np.array_equal(fnet.add(ta, tb).cpu().numpy(), a + b)
    assert np.array_equal(fnet.add(ta, tb, relu=True).cpu().numpy(), np.maximum(a + b, 0))
    if count > 8:  # unaligned views take the scalar path
        assert np.array_equal(fnet.relu(ta[1:]).cpu().numpy(), np.maximum(a[1:], 0))
        assert np.array_equal(fnet.add(ta[1:], tb[1:], relu=True).cpu().numpy(), np.maximum(a[1:] + b[1:], 0))


@pytest.mark.parametrize("shape", [(1, 3, 5, 7), (2, 16, 14, 14), (3, 7, 1, 1), (2, 64, 56, 56)])
def test_affine_matches_reference_formula(cuda, shape):
    from feathercnn_amd import net as fnet
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, shape).astype(np.float32)
    m, a = rng.uniform(0.5, 1.5, shape[1]).astype(np.float32), rng.uniform(-0.1, 0.1, shape[1]).astype(np.float32)
    want = x * m[None, :, None, None] + a[None, :, None, None]
    got = fnet.affine(_t(x, cuda), _t(m, cuda), _t(a, cuda)).cpu().numpy()
    assert np.allclose(got, want, rtol=0, atol=2e-7)
    got = fnet.affine(_t(x, cuda), _t(m, cuda), None, relu=True).cpu().numpy()
    assert np.allclose(got, np.maximum(x * m[None, :, None, None], 0), rtol=0, atol=2e-7)


POOLS = [  # (c, h, w, kernel, stride, (pl, pr, pt, pb), type, global)
    (3, 8, 8, 2, 2, (0, 0, 0, 0), 0, False), (4, 112, 112, 3, 2, (0, 0, 0, 0), 0, False), (5, 13, 11, 3, 2, (1, 1, 1, 1), 0, False),
    (2, 10, 10, 3, 1, (1, 1, 1, 1), 1, False), (3, 9, 7, (3, 2), (1, 2), (0, 0, 1, 0), 1, False), (6, 7, 7, 7, 1, (0, 0, 0, 0), 1, True),
    (2, 5, 9, 1, 1, (0, 0, 0, 0), 0, True), (2, 6, 6, 2, 2, (1, 0, 0, 1), 1, False),
    (3, 12, 12, 3, 2, (0, 0, 0, 0), 0, False), (2, 13, 16, 3, 2, (0, 0, 0, 0), 0, False), (2, 7, 4, 3, 2, (0, 0, 0, 0), 0, False),   # 3x3/s2 fast path
    (2, 9, 20, 3, 2, (0, 0, 0, 0), 0, False), (2, 3, 8, 3, 2, (0, 0, 0, 0), 0, False), (2, 15, 15, 3, 2, (0, 0, 0, 0), 0, False),
    (200, 8, 8, 8, 1, (0, 0, 0, 0), 1, True), (131, 7, 7, 7, 1, (0, 0, 0, 0), 0, True), (3, 14, 14, 14, 1, (0, 0, 0, 0), 1, True), (130, 9, 14, (9, 14), 1, (0, 0, 0, 0), 1, True), (5, 8, 16, (8, 16), 1, (0, 0, 0, 0), 0, True)]  # global pooling: small-plane blocks of 128 (even / odd HW, ragged last block), wave-per-plane form above 128 pixels


@pytest.mark.parametrize("case", POOLS)
def test_pooling_matches_restatement_including_pad_quirk(cuda, case):
    from feathercnn_amd import net as fnet
    c, h, w, k, s, pad, typ, glob = case
    kh, kw = (k, k) if isinstance(k, int) else k
    sh, sw = (s, s) if isinstance(s, int) else s
    x = np.random.default_rng(7).uniform(-1, 1, (2, c, h, w)).astype(np.float32)
    pd = {0: typ, 1: kw, 11: kh, 2: sw, 12: sh, 3: pad[0], 14: pad[1], 13: pad[2], 15: pad[3], 4: int(glob)}
    want = netcheck._pool(x, pd)
    got = fnet.pooling(_t(x, cuda), fnet.pool_param(c, h, w, (kh, kw), (sh, sw), pad, typ, glob)).cpu().numpy()
    assert got.shape == want.shape
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert np.allclose(got[ok], want[ok], rtol=0, atol=5e-7 if typ else 0)


@pytest.mark.parametrize("shape", [(1, 10, 1, 1), (4, 1000, 1, 1), (2, 3, 5, 5), (1, 5000, 1, 1)])
def test_softmax(cuda, shape):
    from feathercnn_amd import net as fnet
    x = np.random.default_rng(2).uniform(-8, 8, shape).astype(np.float32)
    f = x.reshape(shape[0], -1).astype(np.float64)
    e = np.exp(f - f.max(axis=1, keepdims=True))
    want = (e / e.sum(axis=1, keepdims=True)).reshape(shape)
    got = fnet.softmax(_t(x, cuda)).cpu().numpy()
    assert nerr(got, want) <= 1e-6


# ---- whole nets ----------------------------------------------------------------------------------------------------
def _run(model, x, out, fusion=1, graph=False, repeat=1):
    from feathercnn_amd.net import Net
    p, b, i, _ = model
    net = Net(fusion=fusion, graph=graph)
    net.LoadParam(p)
    net.LoadWeights(b)
    net.FeedInput(i, x)
    for _ in range(repeat):
        net.Forward()
    return net, net.Extract(out)


@pytest.mark.parametrize("fusion", [0, 1, 2])
@pytest.mark.parametrize("graph", [False, True])
def test_tiny_net_matches_reference_fixture(cuda, fusion, graph):
    g = np.load(GOLDEN)
    model = (g["tiny/param"].tobytes(), g["tiny/bin"].tobytes(), "data", "prob")
    net, prob = _run(model, g["tiny/x"], "prob", fusion, graph, repeat=3 if graph else 1)
    assert prob.shape == (2, 10, 1, 1)
    assert nerr(prob, g["tiny/blob/prob"]) <= TOL
    survivors = ["pool1", "cat", "drop", "gap", "fc2"] if fusion else [k.split("/")[2] for k in g.files if k.startswith("tiny/blob/")]
    for name in survivors:
        assert nerr(net.Extract(name), g["tiny/blob/" + name]) <= TOL, name
    types = [t for t, _, _ in net.layers()]
    if fusion == 0:
        assert len(types) == 28
    else:
        assert "ReLU" not in types and types.count("Scale") == (1 if fusion == 1 else 0)
        assert types.count("BatchNorm") == (2 if fusion == 1 else 0)
        from feathercnn_amd import FeatherHipError
        with pytest.raises(FeatherHipError, match="fused"):
            net.Extract("conv1")  # its consumer relu1 was absorbed: the conv now writes blob relu1
    algos = {n: a for _, n, a in net.layers()}
    assert algos["conv1"] == "IM2COL" and algos["conv2"] == "WINOGRADF63" and algos["dw"] == "DEPTHWISE" and algos["fc1"] == "IM2COL"


def test_tiny_net_batches_and_reshape(cuda):
    """Batch > 1 (the reference is N = 1: checker loops), then new batch and new image size through the same Net."""
    from feathercnn_amd.net import Net
    p, b, i, o = model_zoo.tiny_allsorts()
    port = netcheck.PortNet(p, b)
    net = Net()
    net.LoadParam(p)
    net.LoadWeights(b)
    for shape in [(5, 3, 20, 20), (1, 3, 20, 20), (2, 3, 32, 26), (7, 3, 20, 20)]:
        x = np.random.default_rng(sum(shape)).uniform(-1, 1, shape).astype(np.float32)
        net.FeedInput(i, x)
        net.Forward()
        got = net.Extract(o)
        want = port.run(i, x, o)
        assert got.shape == want.shape
        assert nerr(got, want) <= TOL, shape
        assert nerr(net.Extract("cat"), port.run(i, x, "cat")) <= TOL, shape
    mem = net.memory()
    assert mem["blob_bytes"] > 0 and mem["weight_bytes"] > 0


def test_squeezenet_matches_reference_fixture(cuda):
    g = np.load(GOLDEN)
    model = model_zoo.squeezenet_v11()
    x = np.random.default_rng(43).uniform(-1, 1, (2, 3, 224, 224)).astype(np.float32)
    net, prob = _run(model, x, "prob")
    assert nerr(prob, g["squeezenet/prob"]) <= TOL
    assert nerr(net.Extract("fire5_concat")[:, :8], g["squeezenet/fire5"]) <= TOL
    assert np.argmax(prob[0]) == np.argmax(g["squeezenet/prob"][0])
    timed = net.forward_timed()
    assert len(timed) == len(net.layers()) and all(ms >= 0 for *_, ms in timed)


@pytest.mark.parametrize("name,batch,size,fusion", [("mobilenet_v1", 3, 224, 1), ("mobilenet_v1", 2, 224, 2), ("resnet50", 2, 224, 1),
                                                     ("resnet50", 1, 224, 2), ("vgg16", 2, 64, 1), ("vgg16", 3, 64, 2)])
def test_benchmark_nets_match_cpu_checker(cuda, name, batch, size, fusion):
    """Full benchmark topologies against the live reference (looped over the batch) when its .so is on the box, else the
    restatement.  Checked before the softmax too: the logits carry the accumulated error of every layer."""
    model = model_zoo.MODELS[name](size=size)
    p, b, i, o = model
    x = np.random.default_rng(11).uniform(-1, 1, (batch, 3, size, size)).astype(np.float32)
    logits = {"mobilenet_v1": "fc7", "resnet50": "fc1000", "vgg16": "fc8"}[name]
    net, prob = _run(model, x, o, fusion)
    got_logits = net.Extract(logits)
    if netcheck.have_ref_net():
        ref = netcheck.RefNet(p, b)
        want, want_logits = ref.run(i, x, o), ref.run(i, x, logits)
        ref.close()
    else:
        blobs = netcheck.PortNet(p, b).run(i, x, o, keep=True)
        want, want_logits = blobs[o], blobs[logits]
    assert nerr(got_logits, want_logits) <= TOL
    assert nerr(prob, want) <= TOL
    assert np.array_equal(prob.reshape(batch, -1).argmax(1), want.reshape(batch, -1).argmax(1))


def test_configurations_the_reference_crashes_on_match_restatement(cuda):
    """Winograd without bias (NULL bias deref), depthwise WITH bias (2.3 #5), InnerProduct without bias
    (inner_product_layer.h:91): the product handles them; the restatement is the checker."""
    g = model_zoo.GraphBuilder(9)
    x = g.input("data", 8, 18, 18)
    x = g.conv("wino_nobias", x, 8, 16, 3, 1, 1, bias=False)
    x = g.relu("r1", x)
    x = g.conv("dw_bias", x, 16, 16, 3, 2, 1, group=16, bias=True)
    x = g.pool("gap", x, 9, 1, avg=True, global_=True)
    x = g.fc("fc_nobias", x, 16, 5, bias=False)
    p, b = g.finish()
    img = np.random.default_rng(3).uniform(-1, 1, (3, 8, 18, 18)).astype(np.float32)
    want = netcheck.PortNet(p, b).run("data", img, "fc_nobias", keep=True)
    net, got = _run((p, b, "data", None), img, "fc_nobias", fusion=0)
    for name in ("wino_nobias", "dw_bias", "fc_nobias"):
        assert nerr(net.Extract(name), want[name]) <= TOL, name


def test_fp16_and_codebook_weight_payloads(cuda):
    """ncnn .bin payload kinds besides raw fp32 (ncnn/modelbin.cpp:78-150): fp16 tag 0x01306B47 and the 256-entry table."""
    import struct
    rng = np.random.default_rng(21)
    w = (rng.uniform(-1, 1, 8 * 4 * 9) * 0.3).astype(np.float32)
    bias = rng.uniform(-0.1, 0.1, 8).astype(np.float32)
    param = b"7767517\n2 2\nInput data 0 1 data 0=12 1=12 2=4\nConvolution c 1 1 data c 0=8 1=3 4=1 5=1 6=288\n"
    img = rng.uniform(-1, 1, (2, 4, 12, 12)).astype(np.float32)
    w16 = w.astype(np.float16)
    table = np.linspace(-0.3, 0.3, 256).astype(np.float32)
    idx = np.abs(w[:, None] - table[None, :]).argmin(1).astype(np.uint8)
    payloads = {"fp16": (struct.pack("<I", 0x01306B47) + w16.tobytes(), w16.astype(np.float32)),
                "table": (bytes([1, 0, 0, 0]) + table.tobytes() + idx.tobytes(), table[idx])}
    for kind, (blob, w_eff) in payloads.items():
        raw = struct.pack("<I", 0) + w_eff.astype("<f4").tobytes() + bias.tobytes()
        want = netcheck.PortNet(param, raw).run("data", img, "c")
        _, got = _run((param, blob + bias.tobytes(), "data", None), img, "c")
        assert nerr(got, want) <= TOL, kind


def test_forward_before_feed_and_unknown_blob_fail_loudly(cuda):
    from feathercnn_amd import FeatherHipError
    from feathercnn_amd.net import Net
    p, b, i, o = model_zoo.tiny_allsorts()
    net = Net()
    net.LoadParam(p)
    with pytest.raises(FeatherHipError, match="weights"):
        net.Forward()
    net.LoadWeights(b)
    with pytest.raises(FeatherHipError, match="not been fed"):
        net.Forward()
    with pytest.raises(FeatherHipError, match="Invalid input blob"):
        net.FeedInput("nope", np.zeros((1, 3, 20, 20), np.float32))
    net.FeedInput(i, np.zeros((1, 3, 20, 20), np.float32))
    net.Forward()
    with pytest.raises(FeatherHipError, match="Cannot find output blob"):
        net.Extract("nope")
    with pytest.raises(FeatherHipError, match="input channels"):
        net.FeedInput(i, np.zeros((1, 4, 20, 20), np.float32))
        net.Forward()


def test_cpp_net_class_forward(cuda, tmp_path):
    """The C++ feather::Net class end to end: LoadParam(FILE*) / LoadWeights / FeedInput / Forward / Extract, checked against
    the reference fixture."""
    import subprocess

    from feathercnn_amd import _lib
    g = np.load(GOLDEN)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    (tmp_path / "m.param").write_bytes(g["tiny/param"].tobytes())
    (tmp_path / "m.bin").write_bytes(g["tiny/bin"].tobytes())
    g["tiny/x"].astype("<f4").tofile(str(tmp_path / "x.f32"))
    exe = str(tmp_path / "net_api_test")
    libdir = os.path.dirname(_lib.lib_path())
    subprocess.run(["g++", "-std=c++11", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "net_api_test.cpp"),
                    "-o", exe, "-L" + libdir, "-lfeather_hip", "-Wl,-rpath," + libdir], check=True, capture_output=True, text=True)
    out = subprocess.run([exe, str(tmp_path / "m.param"), str(tmp_path / "m.bin"), str(tmp_path / "x.f32"), "2", "3", "20", "20", "prob",
                          str(tmp_path / "y.f32")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "net forward ok 2 10 1 1" in out.stdout, out.stdout + out.stderr
    y = np.fromfile(str(tmp_path / "y.f32"), "<f4").reshape(2, 10, 1, 1)
    assert nerr(y, g["tiny/blob/prob"]) <= TOL


def test_reference_style_application_runs(cuda, tmp_path):
    """tests/cpp/reference_style_main.cpp (the reference's own public API, unchanged source) end to end: FeedInput(ncnn::Mat) with a
    padded channel stride (20 x 20 planes are 1600 B: no padding; 19 x 19 are 1444 B -> 1456), Extract(ncnn::Mat), against the checker."""
    import subprocess
    from feathercnn_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    inc = os.path.join(root, "include")
    libdir = os.path.dirname(_lib.lib_path())
    exe = str(tmp_path / "ref_main")
    subprocess.run(["g++", "-std=c++11", "-O1", "-I" + inc, "-I" + os.path.join(inc, "feather"), os.path.join(root, "tests", "cpp", "reference_style_main.cpp"),
                    "-o", exe, "-L" + libdir, "-lfeather_hip", "-Wl,-rpath," + libdir], check=True, capture_output=True, text=True)
    for size in (20, 19):
        p, b, i, o = model_zoo.tiny_allsorts(size=size)
        (tmp_path / "m.param").write_bytes(p)
        (tmp_path / "m.bin").write_bytes(b)
        x = np.random.default_rng(size).uniform(-1, 1, (1, 3, size, size)).astype(np.float32)
        x.astype("<f4").tofile(str(tmp_path / "x.f32"))
        for blob in ("prob", "cat"):
            out = subprocess.run([exe, str(tmp_path / "m.param"), str(tmp_path / "m.bin"), str(tmp_path / "x.f32"), "3", str(size), str(size), blob,
                                  str(tmp_path / "y.f32")], capture_output=True, text=True, timeout=120)
            assert out.returncode == 0 and "reference-style main ok" in out.stdout, out.stdout + out.stderr
            want = (netcheck.RefNet(p, b) if netcheck.have_ref_net() else netcheck.PortNet(p, b)).run(i, x, blob)
            y = np.fromfile(str(tmp_path / "y.f32"), "<f4").reshape(want.shape)
            assert nerr(y, want) <= TOL, (size, blob)


@pytest.mark.parametrize("shape", [(8, 16, 20, 20), (16, 64, 36, 28), (4, 8, 16, 44), (32, 32, 14, 14), (8, 8, 12, 10)])
def test_conv_with_fused_maxpool_equals_conv_then_pool(cuda, shape):
    """fhip_conv_forward_maxpool2 (Winograd output transform writing the pooled tensor) == ConvLayer + PoolingLayer."""
    import ctypes

    import torch
    from feathercnn_amd import ConvLayer, ConvParam, _lib
    from feathercnn_amd import net as fnet
    c, k, h, w = shape
    rng = np.random.default_rng(h * w)
    x = rng.uniform(-1, 1, (3, c, h, w)).astype(np.float32)
    wt = (rng.uniform(-1, 1, (k, c, 3, 3)) / np.sqrt(c * 9)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, k).astype(np.float32)
    for act in (0, 1):
        p = ConvParam.make(c, k, h, 3, 1, 1, bias=True, act=act, w=w, batch=3)
        lyr = ConvLayer(p, _t(wt, cuda), _t(b, cuda))
        full = lyr.Forward(_t(x, cuda))
        want = fnet.pooling(full, fnet.pool_param(k, h, w, 2, 2)).cpu().numpy()
        lib = _lib.load_library()
        cp = p._c()
        assert lib.fhip_conv_can_fuse_maxpool2(ctypes.byref(cp), lyr.booster.algo) == 1
        pooled = torch.empty((3, k, h // 2, w // 2), device=cuda)
        scratch = torch.empty(max(lyr.buffer_bytes // 4, 1), device=cuda)
        rc = lib.fhip_conv_forward_maxpool2(ctypes.byref(cp), lyr.booster.algo, 3, pooled.data_ptr(), _t(x, cuda).data_ptr(), lyr.packed.data_ptr(),
                                            scratch.data_ptr(), lyr.bias.data_ptr(), None)
        assert rc == 0
        torch.cuda.synchronize()
        assert np.array_equal(pooled.cpu().numpy(), want)  # same arithmetic, max is exact
    odd = ConvParam.make(c, k, 15, 3, 1, 1, bias=True, act=1, batch=1)
    assert _lib.load_library().fhip_conv_can_fuse_maxpool2(ctypes.byref(odd._c()), 4) == 0


def test_fused_pool_fallback_on_odd_dims_and_non_winograd(cuda):
    """Fusion level 2 absorbs conv -> (ReLU) -> maxpool 2x2/s2 everywhere; where the fast kernel cannot apply (odd output dims:
    ceil-mode partial windows; IM2COL route) the layer runs conv + pooling itself.  Checked against the restatement."""
    g = model_zoo.GraphBuilder(17)
    x = g.input("data", 3, 15, 15)
    x = g.relu("r0", g.conv("c0", x, 3, 8, 3, 1, 1))        # IM2COL (C = 3), 15x15 -> pool -> 8x8 (ceil)
    x = g.pool("p0", x, 2, 2)
    x = g.relu("r1", g.conv("c1", x, 8, 8, 3, 1, 1))        # 8x8: H <= 8 -> IM2COL, even dims
    x = g.pool("p1", x, 2, 2)
    x = g.conv("c2", x, 8, 12, 3, 1, 4)                     # 4x4 pad 4 -> 10x10: Winograd, even: fast path (no ReLU)
    x = g.pool("p2", x, 2, 2)
    x = g.relu("r3", g.conv("c3", x, 12, 12, 3, 1, 4))      # 5x5 pad 4 -> 11x11: Winograd, odd: fallback
    x = g.pool("p3", x, 2, 2)
    p, b = g.finish()
    img = np.random.default_rng(4).uniform(-1, 1, (3, 3, 15, 15)).astype(np.float32)
    want = netcheck.PortNet(p, b).run("data", img, "p3", keep=True)
    net, got = _run((p, b, "data", None), img, "p3", fusion=2)
    assert [t for t, _, _ in net.layers()] == ["Input", "Convolution", "Convolution", "Convolution", "Convolution"]
    assert got.shape == want["p3"].shape == (3, 12, 6, 6)
    for name in ("p0", "p1", "p2", "p3"):
        assert nerr(net.Extract(name), want[name]) <= TOL, name


@pytest.mark.parametrize("geom", [(64, 256, 14, 1, 1, 0, 2), (256, 64, 7, 1, 1, 0, 3), (1024, 256, 7, 1, 1, 0, 2), (16, 24, 9, 3, 2, 1, 2),
                                  (8, 12, 5, 1, 1, 0, 1)])
def test_conv_with_fused_residual_equals_conv_then_add(cuda, geom):
    """fhip_conv_forward_residual (add in the implicit-GEMM epilogue / in the split-K reduce) == conv, then add (+ReLU)."""
    import ctypes

    import torch
    from feathercnn_amd import ConvLayer, ConvParam, IM2COL, _lib
    c, k, h, ks, s, pad, n = geom
    rng = np.random.default_rng(c + k)
    x = rng.uniform(-1, 1, (n, c, h, h)).astype(np.float32)
    wt = (rng.uniform(-1, 1, (k, c, ks, ks)) / np.sqrt(c * ks * ks)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, k).astype(np.float32)
    lib = _lib.load_library()
    for act in (0, 1):
        plain = ConvParam.make(c, k, h, ks, s, pad, bias=True, act=0, batch=n)
        lyr = ConvLayer(plain, _t(wt, cuda), _t(b, cuda), algo=IM2COL)
        y = lyr.Forward(_t(x, cuda))
        res = _t(rng.uniform(-1, 1, tuple(y.shape)).astype(np.float32), cuda)
        want = y + res
        if act:
            want = torch.clamp_min(want, 0)
        p = ConvParam.make(c, k, h, ks, s, pad, bias=True, act=act, batch=n)
        cp = p._c()
        assert lib.fhip_conv_can_fuse_residual(ctypes.byref(cp), IM2COL) == 1
        out = torch.empty_like(y)
        scratch = torch.empty(max(lyr.buffer_bytes // 4, 1), device=cuda)
        rc = lib.fhip_conv_forward_residual(ctypes.byref(cp), IM2COL, n, out.data_ptr(), _t(x, cuda).data_ptr(), lyr.packed.data_ptr(),
                                            scratch.data_ptr(), lyr.bias.data_ptr(), res.data_ptr(), None)
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(out, want), geom
    assert lib.fhip_conv_can_fuse_residual(ctypes.byref(cp), 4) == 0


def test_residual_fusion_in_the_net_fast_and_fallback_paths(cuda):
    """Fusion level 2 moves conv -> Eltwise(+earlier blob) -> ReLU into the conv: implicit-GEMM epilogue where the route is
    IM2COL, conv + in-place add where it is Winograd; a conv whose ReLU precedes the add is left alone."""
    g = model_zoo.GraphBuilder(23)
    x = g.input("data", 8, 12, 12)
    a, b = g.split("s0", x)
    y = g.conv_bn_relu("c1", a, 8, 8, 1, relu=False)           # IM2COL + BN + Scale, then the add, then ReLU
    x = g.relu("r1", g.eltwise("e1", b, y))
    a, b = g.split("s1", x)
    y = g.conv("c2", a, 8, 8, 3, 1, 1)                          # Winograd: fallback path
    x = g.relu("r2", g.eltwise("e2", y, b))
    a, b = g.split("s2", x)
    y = g.relu("r3a", g.conv("c3", a, 8, 8, 1))                 # ReLU BEFORE the add: must not be fused
    x = g.eltwise("e3", b, y)
    p, w = g.finish()
    img = np.random.default_rng(8).uniform(-1, 1, (3, 8, 12, 12)).astype(np.float32)
    want = netcheck.PortNet(p, w).run("data", img, "e3", keep=True)
    net, got = _run((p, w, "data", None), img, "e3", fusion=2)
    types = [t for t, _, _ in net.layers()]
    assert types.count("Eltwise") == 1 and types.count("ReLU") == 0 and types.count("BatchNorm") == 0
    for name in ("r1", "r2", "e3"):
        assert nerr(net.Extract(name), want[name]) <= TOL, name


def test_tuned_selection_routes_small_3x3_layers_to_winograd_and_keeps_parity(cuda):
    """fhip_conv_select_algo_tuned relaxes the reference's `h,w > 8` Winograd guard; ResNet-50's 7x7 stage then runs F(6,3).
    Logits and probabilities still match the live reference feather::Net (which runs those layers through IM2COL)."""
    from feathercnn_amd.net import Net
    p, b, i, o = model_zoo.resnet50()
    x = np.random.default_rng(12).uniform(-1, 1, (2, 3, 224, 224)).astype(np.float32)
    outs = {}
    for tuned in (False, True):
        net = Net(fusion=2, tuned=tuned)
        net.LoadParam(p)
        net.LoadWeights(b)
        net.FeedInput(i, x)
        net.Forward()
        outs[tuned] = (net.Extract("fc1000"), net.Extract(o), {n: a for _, n, a in net.layers()})
    assert outs[False][2]["res5a_branch2b"] == "IM2COL" and outs[True][2]["res5a_branch2b"] == "WINOGRADF63"
    assert outs[True][2]["res4a_branch2b"] == "WINOGRADF63" and outs[True][2]["conv1"] == "IM2COL"
    if netcheck.have_ref_net():
        ref = netcheck.RefNet(p, b)
        want_logits, want = ref.run(i, x, "fc1000"), ref.run(i, x, o)
        ref.close()
    else:
        blobs = netcheck.PortNet(p, b).run(i, x, o, keep=True)
        want_logits, want = blobs["fc1000"], blobs[o]
    for tuned in (False, True):
        assert nerr(outs[tuned][0], want_logits) <= TOL, tuned
        assert nerr(outs[tuned][1], want) <= TOL, tuned


@pytest.mark.parametrize("name,batch", [("resnet50", 3), ("squeezenet_v1.1", 4)])
@pytest.mark.parametrize("graph", [False, True])
def test_branch_concurrency_keeps_results_bit_identical(cuda, name, batch, graph):
    """fhip_net_set_concurrency moves arena-free convolutions with a distant consumer (projection shortcuts, expand1x1) to a second
    stream.  Same kernels, same inputs: the outputs must equal the single-stream run bit for bit, run after run."""
    from feathercnn_amd.net import Net
    p, b, i, o = model_zoo.MODELS[name]()
    x = np.random.default_rng(31).uniform(-1, 1, (batch, 3, 224, 224)).astype(np.float32)
    outs = []
    for conc in (False, True):
        net = Net(fusion=2, tuned=True, concurrency=conc, graph=graph)
        net.LoadParam(p)
        net.LoadWeights(b)
        net.FeedInput(i, x)
        runs = []
        for _ in range(4 if conc else 1):
            net.Forward()
            runs.append(net.Extract(o).copy())
        assert all(np.array_equal(runs[0], r) for r in runs)
        outs.append(runs[0])
        net.close()
    assert np.array_equal(outs[0], outs[1])
