"""Whole-net path without a GPU: the two CPU checkers agree with each other and with the committed reference fixtures, the
synthetic ncnn models are well-formed, and the product's readers (LoadParam / LoadWeights, host-only) accept them and
reject malformed files with the reference's error codes.  No compute call is made here."""
import hashlib
import os

import numpy as np
import pytest

from feathercnn_amd import model_zoo
from oracle import nerr, netcheck

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "net_golden.npz")


@pytest.fixture(scope="module")
def golden():
    return np.load(GOLDEN)


def test_port_net_matches_reference_fixtures(golden):
    """The numpy/C restatement of every layer against blobs the REAL reference produced (tests/golden/make_net_golden.py)."""
    p, b = golden["tiny/param"].tobytes(), golden["tiny/bin"].tobytes()
    blobs = netcheck.PortNet(p, b).run("data", golden["tiny/x"], "prob", keep=True)
    names = [k.split("/")[2] for k in golden.files if k.startswith("tiny/blob/")]
    assert len(names) >= 14
    for n in names:
        assert blobs[n].shape == golden["tiny/blob/" + n].shape, n
        assert nerr(blobs[n], golden["tiny/blob/" + n]) <= 1e-5, n


def test_tiny_model_regenerates_bit_identically(golden):
    p, b, _, _ = model_zoo.tiny_allsorts()
    assert p == golden["tiny/param"].tobytes()
    assert b == golden["tiny/bin"].tobytes()


def test_port_net_squeezenet_matches_reference_fixture(golden):
    p, b, i, o = model_zoo.squeezenet_v11()
    assert hashlib.sha256(b).digest() == golden["squeezenet/bin_sha256"].tobytes(), "numpy RNG stream changed: regenerate fixtures"
    x = np.random.default_rng(43).uniform(-1, 1, (2, 3, 224, 224)).astype(np.float32)
    blobs = netcheck.PortNet(p, b).run(i, x, o, keep=True)
    assert nerr(blobs[o], golden["squeezenet/prob"]) <= 1e-5
    assert nerr(blobs["fire5_concat"][:, :8], golden["squeezenet/fire5"]) <= 1e-5
    assert abs(float(blobs[o][0].sum()) - 1.0) < 1e-5


@pytest.mark.skipif(not netcheck.have_ref_net(), reason="oracle/_ref/libfeather_net_ref.so not built (needs /root/reference)")
def test_reference_net_live_matches_fixture_and_port(golden):
    p, b = golden["tiny/param"].tobytes(), golden["tiny/bin"].tobytes()
    ref = netcheck.RefNet(p, b)
    y = ref.run("data", golden["tiny/x"], "prob")
    ref.close()
    assert np.array_equal(y, golden["tiny/blob/prob"])
    # pooling quirks: pads shift the window twice, average divides by in-range taps (pooling_layer.h:56-88)
    g = model_zoo.GraphBuilder(3)
    x = g.input("data", 4, 11, 9)
    a, b2 = g.split("s", x)
    g.layer("Pooling", "pmax", [a], ["pmax"], {0: 0, 1: 3, 2: 2, 3: 1})
    g.layer("Pooling", "pavg", [b2], ["pavg"], {0: 1, 1: 2, 11: 3, 2: 2, 12: 1, 3: 0, 13: 1, 15: 0})
    param, weights = g.finish()
    img = np.random.default_rng(5).uniform(-1, 1, (1, 4, 11, 9)).astype(np.float32)
    ref = netcheck.RefNet(param, weights)
    port = netcheck.PortNet(param, weights)
    for name in ("pmax", "pavg"):
        r, q = ref.run("data", img, name), port.run("data", img, name)
        assert r.shape == q.shape and nerr(q, r) <= 1e-6, name
    ref.close()


@pytest.mark.parametrize("name", ["tiny_allsorts", "squeezenet_v1.1", "mobilenet_v1", "resnet50"])
def test_product_readers_accept_zoo_models(name):
    from feathercnn_amd.net import Net
    p, b, _, _ = model_zoo.MODELS[name]()
    layers = netcheck.parse_param(p)
    net = Net()
    net.LoadParam(p)
    net.LoadWeights(b)
    got = net.layers()
    assert [(t, n) for t, n, _ in got] == [(t, n) for t, n, _, _, _ in layers]
    # every weight byte is consumed, in the reference's order: the restatement reads the same stream to its end
    assert netcheck.PortNet(p, b) is not None


def test_product_readers_reject_malformed_models():
    from feathercnn_amd import FeatherHipError
    from feathercnn_amd.net import Net
    cases = [(b"123\n1 1\nInput data 0 1 data\n", "-1"),                                           # utils.cpp:37-41
             (b"7767517\n0 0\n", "-1"),                                                             # net.cpp:79-83
             (b"7767517\n1 1\nFoo a 0 1 a\n", "-200"),                                              # net.cpp:108-112
             (b"7767517\n2 2\nInput data 0 1 data\nReLU r 1 1 nope r\n", "-300"),                   # net.cpp:131-135
             (b"7767517\n2 2\nInput data 0 1 data\nConvolution c 1 1 data c 0=4 1=3 2=2 6=108\n", "-200"),   # dilation
             (b"7767517\n2 2\nInput data 0 1 data\nEltwise e 1 1 data e 0=0\n", "-100")]            # eltwise_layer.h:62-66
    for text, code in cases:
        with pytest.raises(FeatherHipError, match=f"code {code}"):
            Net().LoadParam(text)
    p, b, _, _ = model_zoo.tiny_allsorts()
    net = Net()
    net.LoadParam(p)
    with pytest.raises(FeatherHipError, match="file too short"):
        net.LoadWeights(b[:-8])
    with pytest.raises(FeatherHipError):
        Net().LoadWeights(b)  # net.cpp:193-197: param first
    with pytest.raises(FeatherHipError, match="Cannot open"):
        Net().LoadParam("/nonexistent/model.param")


def test_param_and_bin_files_round_trip(tmp_path):
    from feathercnn_amd.net import Net
    p, b, _, _ = model_zoo.tiny_allsorts()
    (tmp_path / "m.param").write_bytes(p)
    (tmp_path / "m.bin").write_bytes(b)
    net = Net()
    net.LoadParam(str(tmp_path / "m.param"))
    net.LoadWeights(str(tmp_path / "m.bin"))
    assert len(net.layers()) == 28


def test_cpp_net_class_compiles_and_reads_models(tmp_path):
    """include/feather/net.h (the reference's feather::Net API, header-only over the C-ABI) with plain g++, host side only."""
    import subprocess

    from feathercnn_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p, b, _, _ = model_zoo.tiny_allsorts()
    (tmp_path / "m.param").write_bytes(p)
    (tmp_path / "m.bin").write_bytes(b)
    exe = str(tmp_path / "net_api_test")
    libdir = os.path.dirname(_lib.lib_path())
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "net_api_test.cpp"),
                    "-o", exe, "-L" + libdir, "-lfeather_hip", "-Wl,-rpath," + libdir], check=True, capture_output=True, text=True)
    out = subprocess.run([exe, str(tmp_path / "m.param"), str(tmp_path / "m.bin")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "net api ok" in out.stdout, out.stdout + out.stderr


def test_reference_style_application_compiles_unchanged(tmp_path):
    """A main() written against the reference's own public API -- #include <net.h>, feather::Net, FeedInput(const char*, ncnn::Mat&),
    Extract(std::string, ncnn::Mat&) (reference src/net.h:44,50) -- compiles against include/ and links against the product library
    without a source change; the minimal ncnn::Mat behaves like the reference's for the calls such programs make."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from feathercnn_amd import _lib
    libdir = os.path.dirname(_lib.lib_path())
    exe = str(tmp_path / "ref_main")
    inc = os.path.join(root, "include")
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-I" + inc, "-I" + os.path.join(inc, "feather"), os.path.join(root, "tests", "cpp", "reference_style_main.cpp"),
                    "-o", exe, "-L" + libdir, "-lfeather_hip", "-Wl,-rpath," + libdir], check=True, capture_output=True, text=True)
    mat_test = tmp_path / "mat_test.cpp"
    mat_test.write_text(r"""
#include <ncnn/mat.h>
#include <stdio.h>
int main()
{
    ncnn::Mat m(5, 3, 2);                       // w, h, c: plane = 15 floats = 60 B -> channel stride padded to 64 B = 16 floats
    if (m.dims != 3 || m.w != 5 || m.h != 3 || m.c != 2 || m.cstep != 16 || m.elemsize != 4 || m.total() != 32 || m.empty()) return 1;
    m.fill(2.f);
    float* c1 = m.channel(1);
    c1[0] = 7.f;
    if (((float*)m.data)[16] != 7.f || m.channel(1).row(0)[0] != 7.f) return 2;
    ncnn::Mat view = m;                         // shares
    ncnn::Mat deep = m.clone();
    c1[1] = 9.f;
    if (((float*)view.data)[17] != 9.f || ((float*)deep.data)[17] != 2.f) return 3;
    ncnn::Mat ext(4, 4, 3, (void*)c1);          // external data: 4*4*4 = 64 B planes, no padding
    if (ext.cstep != 16 || ext.refcount != 0) return 4;
    m.create(5, 3, 2);                          // same shape: keeps the allocation
    if (((float*)m.data)[16] != 7.f) return 5;
    m.create(8, 8, 1);
    if (m.cstep != 64) return 6;
    printf("mat ok\n");
    return 0;
}
""")
    exe2 = str(tmp_path / "mat_test")
    subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-I" + inc, str(mat_test), "-o", exe2], check=True, capture_output=True, text=True)
    out = subprocess.run([exe2], capture_output=True, text=True)
    assert out.returncode == 0 and "mat ok" in out.stdout, out.stdout + out.stderr


@pytest.mark.skipif(not netcheck.have_ref_net(), reason="needs the compiled reference (oracle/_ref)")
def test_cpu_baseline_helper_sweeps_process_counts(tmp_path):
    """bench.py's cpu_baseline leg (oracle/cpu_bench.py): model loaded once, fork()ed single-thread workers, 1 warm-up + 3 timed
    forwards each, a sweep over process counts bounded by a time budget -- and the extract-only / per-forward-timing shim entries."""
    import json
    import subprocess
    import sys
    p, b, i, o = model_zoo.MODELS["squeezenet_v1.1"]()
    pp, bp = tmp_path / "m.param", tmp_path / "m.bin"
    pp.write_bytes(p)
    bp.write_bytes(b)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "oracle.cpu_bench", "--param", str(pp), "--bin", str(bp), "--input", i, "--output", o,
                          "--procs", "1,2", "--budget", "60"], cwd=root, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-500:]
    r = json.loads(out.stdout.strip().splitlines()[-1])
    assert [s["procs"] for s in r["sweep"]] == [1, 2] and r["reps"] >= 3 and r["warmup"] == 1
    assert all(s["timed_forwards_per_worker"] >= 3 and s["images_per_s"] > 0 for s in r["sweep"])
    assert r["best"]["images_per_s"] == max(s["images_per_s"] for s in r["sweep"])
    # several blobs of ONE reference forward == the blobs of separate runs
    ref = netcheck.RefNet(p, b)
    x = np.random.default_rng(3).uniform(-1, 1, (1, 3, 224, 224)).astype(np.float32)
    prob, pool = ref.run_blobs(i, x[0], (o, "pool10"))
    assert np.array_equal(prob, ref.run(i, x, o)) and np.array_equal(pool, ref.run(i, x, "pool10"))
    assert len(ref.time_each(1, 3)) == 3
    ref.close()
