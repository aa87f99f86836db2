"""Parity of the HIP path (through the C-ABI) against the CPU oracle on the same seeded inputs.

Tolerance (BASELINE.json north_star): normalised max error max|y_gpu - y_ref| / max|y_ref| <= 1e-4
per layer output, in fp32.  The checker is the REAL reference (oracle/_ref, compiled from /root/reference) when its
.so travelled with the snapshot, else the plain-C restatement; both sides are also held against an fp64 direct conv.
"""
import numpy as np
import pytest

import oracle
from oracle import conv_geom, nerr, synth

pytestmark = pytest.mark.gpu

TOL = 1e-4


def run_gpu(g, x, w, b, dev, algo=None):
    import torch
    from feathercnn_amd import ConvLayer, ConvParam
    p = ConvParam(output_channels=g.oc, input_channels=g.ic, input_h=g.ih, input_w=g.iw, kernel_h=g.kh, kernel_w=g.kw,
                  stride_h=g.sh, stride_w=g.sw, pad_left=g.pl, pad_right=g.pr, pad_top=g.pt, pad_bottom=g.pb, group=g.group,
                  bias_term=bool(g.bias), activation=g.act, batch=x.shape[0])
    xt = torch.from_numpy(x).to(dev)
    wt = torch.from_numpy(w).to(dev)
    bt = torch.from_numpy(b).to(dev) if g.bias else None
    layer = ConvLayer(p, wt, bt, algo=algo)
    y = layer.Forward(xt)
    torch.cuda.synchronize()
    return y.cpu().numpy(), layer.booster.algo


def check(g, batch, dev, checker, port, algo=None, seed=1234):
    x, w, b = synth(g, batch, seed)
    y, used = run_gpu(g, x, w, b, dev, algo)
    ref = checker.forward(g, x, w, b, algo=-1 if algo is None else algo)
    f64 = port.direct_f64(g, x, w, b) if algo != oracle.NAIVE else None
    assert y.shape == ref.shape
    assert np.isfinite(y).all()
    e_ref = nerr(y, ref)
    assert e_ref <= TOL, f"{g} algo={used}: normalised error vs oracle {e_ref:.3e}"
    if f64 is not None:
        e64 = nerr(y, f64)
        assert e64 <= TOL, f"{g} algo={used}: normalised error vs fp64 direct conv {e64:.3e}"
    return used


WINO = [
    (conv_geom(16, 16, 9, 3, 1, 1), 2),            # smallest legal Winograd input (h > 8), edge tiles only
    (conv_geom(8, 8, 31, 3, 1, 1, w=17), 3),       # ragged, non-square
    (conv_geom(4, 4, 10, 3, 1, 1), 1),             # C = K = 4: all padding in the GEMM tiles
    (conv_geom(64, 64, 56, 3, 1, 1), 2),           # ResNet 3x3 @56
    (conv_geom(16, 64, 55, 3, 1, 1), 2),           # SqueezeNet fire expand3x3
    (conv_geom(48, 192, 13, 3, 1, 1), 5),          # K = 192 (1.5 row tiles), C = 48 (3 k-tiles)
    (conv_geom(64, 128, 28, 3, 1, 1), 3),
    (conv_geom(128, 256, 14, 3, 1, 1), 4),
    (conv_geom(512, 512, 14, 3, 1, 1), 2),         # VGG conv5
    (conv_geom(64, 64, 224, 3, 1, 1), 1),          # VGG conv1_2 (1444 tiles)
    (conv_geom(24, 36, 12, 3, 1, 0), 2),           # no padding, C % 16 != 0 (not 20: the reference crashes on C % 8 == 4 with ragged tiles)
    (conv_geom(12, 8, 20, 3, 1, 2), 2),            # pad 2 (wider than the kernel needs)
    (conv_geom(128, 128, 14, 3, 1, 1), 10),        # P = 90 columns: the 128 x 96 LDS-DMA tile (one column tile, 6 padding columns)
    (conv_geom(256, 192, 14, 3, 1, 1), 32),        # P = 288 = 3 x 96 (VGG conv5's column count), K = 1.5 row tiles
    (conv_geom(160, 144, 20, 3, 1, 1), 5),         # P = 80, C = 160 (10 k-tiles), K = 144: 96-column tile with ragged rows
]


@pytest.mark.parametrize("g,batch", WINO, ids=lambda v: str(v) if isinstance(v, int) else f"{v.ic}x{v.oc}@{v.ih}x{v.iw}p{v.pl}")
def test_winograd_f63(g, batch, cuda, checker, port):
    assert check(g, batch, cuda, checker, port) == oracle.WINOGRADF63


IM2COL = [
    (conv_geom(256, 64, 56, 1, 1, 0), 2),          # ResNet 1x1 reduce (MODE 2: float4 column loads)
    (conv_geom(64, 256, 56, 1, 1, 0), 2),
    (conv_geom(256, 512, 56, 1, 2, 0), 2),         # strided 1x1 projection (MODE 1)
    (conv_geom(512, 2048, 7, 1, 1, 0), 3),         # HW = 49, not a multiple of 4 (MODE 1)
    (conv_geom(512, 1000, 13, 1, 1, 0), 2),        # SqueezeNet conv10, K = 1000
    (conv_geom(3, 64, 224, 7, 2, 3), 2),           # ResNet conv1
    (conv_geom(3, 64, 224, 3, 2, 0), 1),           # SqueezeNet conv1
    (conv_geom(3, 32, 224, 3, 2, 1), 2),           # MobileNet conv1
    (conv_geom(3, 64, 224, 3, 1, 1), 1),           # VGG conv1_1 (C % 4 != 0 -> IM2COL)
    (conv_geom(512, 512, 7, 3, 1, 1), 2),          # H <= 8 -> IM2COL
    (conv_geom(6, 10, 12, 3, 1, 1), 3),
    (conv_geom(5, 7, 9, 5, 2, 2, w=14), 2),        # 5x5 stride 2 ragged
    (conv_geom(1, 1, 5, 1, 1, 0, group=1), 1),     # degenerate: group == input_channels == 1 -> reference says DEPTHWISE
    # InnerProduct shapes (a 1x1 convolution over a 1x1 image): the weight-streaming kernel (ip_stream.h) at batch <= 32
    (conv_geom(4096, 1000, 1, 1, 1, 0), 32),       # VGG-16 fc8: K = 31.25 m-groups, a full column tile of images
    (conv_geom(4096, 4096, 1, 1, 1, 0), 5),        # fc7 at a ragged batch
    (conv_geom(1028, 300, 1, 1, 1, 0), 7),         # C % 8 != 0 (zero-padded octet), K % 32 != 0
    (conv_geom(1024, 256, 1, 1, 1, 0), 1),         # the thresholds of the route, batch 1 (the reference's GEMV case)
    (conv_geom(25088, 256, 1, 1, 1, 0), 32),       # fc6's reduction length: 3136 octets in pieces of 8 or 9 (loop tails)
    (conv_geom(1024, 256, 1, 1, 1, 0), 33),        # one image too many: stays on the LDS-tiled kernel
    (conv_geom(1016, 256, 1, 1, 1, 0), 8),         # below the threshold: LDS-tiled kernel
]


@pytest.mark.parametrize("g,batch", IM2COL, ids=lambda v: str(v) if isinstance(v, int) else f"{v.ic}x{v.oc}@{v.ih}k{v.kh}s{v.sh}")
def test_im2col_route(g, batch, cuda, checker, port):
    used = check(g, batch, cuda, checker, port)
    assert used == checker.select_algo(g)


DW = [
    (conv_geom(32, 32, 112, 3, 1, 1, group=32), 2),
    (conv_geom(64, 64, 112, 3, 2, 1, group=64), 2),
    (conv_geom(128, 128, 56, 3, 1, 1, group=128), 3),
    (conv_geom(256, 256, 28, 3, 2, 1, group=256), 3),
    (conv_geom(512, 512, 14, 3, 1, 1, group=512), 2),
    (conv_geom(512, 512, 14, 3, 2, 1, group=512), 2),   # OW = 7 -> generic path
    (conv_geom(1024, 1024, 7, 3, 1, 1, group=1024), 2),
    (conv_geom(8, 8, 16, 5, 1, 2, group=8), 2),         # 5x5
    (conv_geom(8, 8, 16, 3, 1, 0, group=8), 2),         # no padding
    (conv_geom(16, 16, 7, 7, 1, 0, group=16), 3),       # global kernel (avx/depthwise.cpp:30-54)
    (conv_geom(24, 24, 20, 3, 1, 1, group=24, w=36), 5),  # chunk tail: planes % planes_per_chunk != 0
]


@pytest.mark.parametrize("g,batch", DW, ids=lambda v: str(v) if isinstance(v, int) else f"dw{v.ic}@{v.ih}k{v.kh}s{v.sh}p{v.pl}")
def test_depthwise(g, batch, cuda, checker, port):
    assert check(g, batch, cuda, checker, port) == oracle.DEPTHWISE


@pytest.mark.parametrize("bias,act", [(0, 0), (1, 0), (0, 1), (1, 1)])
@pytest.mark.parametrize("kind", ["wino", "im2col", "dw"])
def test_epilogue_variants(kind, bias, act, cuda, checker, port):
    g = {"wino": conv_geom(32, 32, 20, 3, 1, 1, bias=bias, act=act),
         "im2col": conv_geom(32, 48, 20, 1, 1, 0, bias=bias, act=act),
         "dw": conv_geom(32, 32, 20, 3, 1, 1, group=32, bias=bias, act=act)}[kind]
    check(g, 2, cuda, checker, port)


def test_force_select_cross_check(cuda, checker, port):
    """ForceSelectAlgo (booster.h:162): the same ConvParam through NAIVE / IM2COL / WINOGRADF63 must agree --
    how 'booster facilitates unit testing' (booster.h:15-16).  NAIVE ignores activation (avx/booster.cpp:41-61)."""
    g = conv_geom(16, 32, 18, 3, 1, 1)
    x, w, b = synth(g, 2)
    outs = {a: run_gpu(g, x, w, b, cuda, a)[0] for a in (oracle.NAIVE, oracle.IM2COL, oracle.WINOGRADF63)}
    assert nerr(outs[oracle.IM2COL], outs[oracle.WINOGRADF63]) <= TOL
    assert (outs[oracle.NAIVE] < 0).any(), "NAIVE must not apply ReLU"
    assert nerr(np.maximum(outs[oracle.NAIVE], 0), outs[oracle.IM2COL]) <= TOL
    check(g, 2, cuda, checker, port, algo=oracle.NAIVE)


def test_unsupported(cuda):
    from feathercnn_amd import ConvBooster, ConvParam, SGECONV, WINOGRADF23, WINOGRADF63FUSED
    p = ConvParam.make(8, 8, 16, 3, 1, 1, group=2)
    b = ConvBooster()
    assert b.SelectAlgo(p) == -1  # partial group, avx/booster.cpp:304-308
    for a in (SGECONV, WINOGRADF23, WINOGRADF63FUSED):
        assert b.ForceSelectAlgo(a) == -1


# ----------------------------------------------------------------------------------------------------------------------
# committed golden vectors (outputs of the real reference, tests/golden/make_golden.py)
from helpers import golden_cases  # noqa: E402

_GOLD = golden_cases()


@pytest.mark.parametrize("case", _GOLD, ids=[c[0] for c in _GOLD])
def test_golden_fixtures(case, cuda):
    name, g, batch, algo, x, w, b, y, sel = case
    got, used = run_gpu(g, x, w, b, cuda, None if algo < 0 else algo)
    assert used == (sel if algo < 0 else algo)
    assert nerr(got, y) <= TOL, name


# ----------------------------------------------------------------------------------------------------------------------
# stage-level entry points (the reference's free functions: transformKernel_F6x6_3x3, input transform, TensorGEMM, output
# transform) against a numpy restatement of the same matrices
_G = np.array([[1, 0, 0], [-2 / 9, -2 / 9, -2 / 9], [-2 / 9, 2 / 9, -2 / 9], [1 / 90, 1 / 45, 2 / 45], [1 / 90, -1 / 45, 2 / 45],
               [1 / 45, 1 / 90, 1 / 180], [1 / 45, -1 / 90, 1 / 180], [0, 0, 1]], np.float64)
_BT = np.array([[1, 0, -5.25, 0, 5.25, 0, -1, 0], [0, 1, 1, -4.25, -4.25, 1, 1, 0], [0, -1, 1, 4.25, -4.25, -1, 1, 0],
                [0, .5, .25, -2.5, -1.25, 2, 1, 0], [0, -.5, .25, 2.5, -1.25, -2, 1, 0], [0, 2, 4, -2.5, -5, .5, 1, 0],
                [0, -2, 4, 2.5, -5, -.5, 1, 0], [0, -1, 0, 5.25, 0, -5.25, 0, 1]], np.float64)
_AT = np.array([[1, 1, 1, 1, 1, 32, 32, 0], [0, 1, -1, 2, -2, 16, -16, 0], [0, 1, 1, 4, 4, 8, 8, 0], [0, 1, -1, 8, -8, 4, -4, 0],
                [0, 1, 1, 16, 16, 2, 2, 0], [0, 1, -1, 32, -32, 1, -1, 1]], np.float64)


@pytest.mark.parametrize("n", [3, 90])  # 36 columns (whole rows of V / M) and 1080 (column blocks of 1024, the second one mostly padding)
def test_winograd_stage_api(cuda, n):
    import ctypes

    import torch

    from feathercnn_amd import ConvParam, _lib, booster
    lib = _lib.load_library()
    g = conv_geom(12, 20, 17, 3, 1, 1, w=23)
    x, w, b = synth(g, n, seed=5)
    p = ConvParam.make(g.ic, g.oc, g.ih, 3, 1, 1, w=g.iw, batch=n)
    pl = booster.winograd_plan(p)
    T, TX, P, Pp = pl.tiles_per_image, pl.tiles_x, pl.columns, pl.columns_padded
    Cp, Kp = pl.in_channels_padded, pl.out_channels_padded
    c = p._c()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    dv = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    xt, wt, bt = (torch.from_numpy(a).to(cuda) for a in (x, w, b))
    U = torch.full((64, Cp, Kp), float("nan"), device=cuda)
    V = torch.full((64, g.ic, Pp), float("nan"), device=cuda)
    M = torch.full((64, g.oc, Pp), float("nan"), device=cuda)
    out = torch.full((n, g.oc, p.output_h, p.output_w), float("nan"), device=cuda)
    assert lib.fhip_winograd_f63_transform_kernel(ctypes.byref(c), dv(U), dv(wt), st) == 0
    assert lib.fhip_winograd_f63_input_transform(ctypes.byref(c), n, dv(V), dv(xt), st) == 0
    assert lib.fhip_winograd_f63_tile_gemm(ctypes.byref(c), n, dv(M), dv(U), dv(V), st) == 0
    assert lib.fhip_winograd_f63_output_transform(ctypes.byref(c), n, dv(out), dv(M), dv(bt), st) == 0
    torch.cuda.synchronize()
    # U = G g G^T, zero padded
    Uref = np.einsum("ia,kcab,jb->ijck", _G, w.astype(np.float64), _G).reshape(64, g.ic, g.oc)
    Ug = U.cpu().numpy()
    assert nerr(Ug[:, :g.ic, :g.oc], Uref) <= 1e-6
    assert np.all(Ug[:, g.ic:, :] == 0) and np.all(Ug[:, :, g.oc:] == 0)
    # V = B^T d B on zero-padded 8x8 patches at stride 6, column p = n*T + ty*TX + tx
    xp = np.zeros((n, g.ic, 6 * pl.tiles_y + 2, 6 * TX + 2))
    xp[:, :, 1:1 + g.ih, 1:1 + g.iw] = x
    patches = np.stack([xp[:, :, 6 * ty:6 * ty + 8, 6 * tx:6 * tx + 8] for ty in range(pl.tiles_y) for tx in range(TX)], axis=2)
    Vref = np.einsum("ia,nctab,jb->ijcnt", _BT, patches, _BT).reshape(64, g.ic, P)
    # (V and M are stored in column blocks -- fhip_winograd_plan.column_block; winograd_rows gives the [64][rows][Pp] view)
    assert nerr(booster.winograd_rows(V, pl, g.ic).cpu().numpy()[:, :, :P], Vref) <= 1e-5
    # M = U V per frequency point
    Mref = np.einsum("xck,xcp->xkp", Uref, Vref)
    assert nerr(booster.winograd_rows(M, pl, g.oc).cpu().numpy()[:, :, :P], Mref) <= 1e-5
    # Y = A^T M A + bias, ReLU, clipped
    Y = np.einsum("ai,ijknt,bj->nktab", _AT, Mref.reshape(8, 8, g.oc, n, T), _AT)
    full = np.zeros((n, g.oc, 6 * pl.tiles_y, 6 * TX))
    for t in range(T):
        ty, tx = divmod(t, TX)
        full[:, :, 6 * ty:6 * ty + 6, 6 * tx:6 * tx + 6] = Y[:, :, t]
    yref = np.maximum(full[:, :, :p.output_h, :p.output_w] + b[None, :, None, None], 0)
    assert nerr(out.cpu().numpy(), yref) <= 1e-5


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE.json's full sizes through size-independent properties (the CPU oracle would take minutes there)
def _layer(dev, ic, oc, h, k, s, p, group, batch, bias=True, act=1, algo=None, seed=0):
    import torch

    from feathercnn_amd import ConvLayer, ConvParam
    prm = ConvParam.make(ic, oc, h, k, s, p, group=group, bias=bias, act=act, batch=batch)
    gen = torch.Generator(device=dev).manual_seed(seed)
    cpg = ic // group
    w = (torch.rand((prm.output_channels, cpg, k, k), device=dev, generator=gen) * 2 - 1) / (cpg * k * k) ** 0.5
    b = (torch.rand((prm.output_channels,), device=dev, generator=gen) * 2 - 1) * 0.1 if bias else None
    return ConvLayer(prm, w, b, algo=algo), w, b


FULL = [("vgg_conv1_2", 64, 64, 224, 3, 1, 1, 1, 32), ("vgg_conv3_2", 256, 256, 56, 3, 1, 1, 1, 32), ("vgg_conv5_1", 512, 512, 14, 3, 1, 1, 1, 32),
        ("r50_1x1", 256, 64, 56, 1, 1, 0, 1, 64), ("r50_proj_s2", 256, 512, 56, 1, 2, 0, 1, 64), ("r50_conv1", 3, 64, 224, 7, 2, 3, 1, 64),
        ("mb_dw_s1", 32, 32, 112, 3, 1, 1, 32, 256), ("mb_dw_s2", 64, 64, 112, 3, 2, 1, 64, 256), ("mb_dw_14", 512, 512, 14, 3, 1, 1, 512, 256)]


@pytest.mark.parametrize("cfg", FULL, ids=[c[0] for c in FULL])
def test_full_size_properties(cfg, cuda):
    import torch
    name, ic, oc, h, k, s, p, group, batch = cfg
    gen = torch.Generator(device=cuda).manual_seed(11)
    x1 = torch.rand((batch, ic, h, h), device=cuda, generator=gen) * 2 - 1
    # (1) determinism: two forwards are bit-identical; (2) the shared scratch arena may hold garbage
    lyr, w, b = _layer(cuda, ic, oc, h, k, s, p, group, batch)
    scratch = torch.full((max(lyr.buffer_bytes // 4, 1),), float("nan"), device=cuda)
    y1 = lyr.Forward(x1, scratch=scratch).clone()
    scratch.fill_(1e30)
    y2 = lyr.Forward(x1, scratch=scratch)
    assert torch.equal(y1, y2)
    assert torch.isfinite(y1).all()
    # (3) batch independence: image i of the batch == the same image run alone (the reference is N=1, conv_layer.h:107)
    one, _, _ = _layer(cuda, ic, oc, h, k, s, p, group, 1)
    for i in (0, batch - 1):
        yi = one.Forward(x1[i:i + 1].contiguous())
        assert float((yi[0] - y1[i]).abs().max()) <= 1e-5 * float(y1[i].abs().max())
    # (4) sampled fp64 direct convolution at 1500 random output positions
    prm = lyr.param
    idx = torch.randint(0, y1.numel(), (1500,), generator=torch.Generator().manual_seed(3))
    n_, k_, oy, ox = np.unravel_index(idx.numpy(), tuple(y1.shape))
    xc, wc = x1.cpu().numpy().astype(np.float64), w.cpu().numpy().astype(np.float64)
    bc = b.cpu().numpy().astype(np.float64)
    cpg = ic // group
    ref = np.empty(1500)
    for j in range(1500):
        c0 = k_[j] if group > 1 else 0
        acc = bc[k_[j]]
        for u in range(k):
            yy = oy[j] * s - p + u
            if yy < 0 or yy >= h:
                continue
            for v in range(k):
                xx = ox[j] * s - p + v
                if 0 <= xx < h:
                    acc += float(np.dot(xc[n_[j], c0:c0 + cpg, yy, xx], wc[k_[j], :, u, v]))
        ref[j] = max(acc, 0.0)
    got = y1.cpu().numpy().reshape(-1)[idx.numpy()]
    assert np.max(np.abs(got - ref)) <= TOL * float(y1.abs().max())
    # (5) linearity without bias / activation
    lin, _, _ = _layer(cuda, ic, oc, h, k, s, p, group, batch, bias=False, act=0)
    x2 = torch.rand((batch, ic, h, h), device=cuda, generator=gen) * 2 - 1
    lhs = lin.Forward((2.0 * x1 - 3.0 * x2).contiguous())
    rhs = 2.0 * lin.Forward(x1).clone() - 3.0 * lin.Forward(x2)
    assert float((lhs - rhs).abs().max()) <= TOL * float(rhs.abs().max())


def test_full_size_winograd_vs_implicit_gemm(cuda):
    """ForceSelectAlgo cross-check at VGG conv4_2 batch 32: two different algorithms, same answer."""
    import torch
    a, w, b = _layer(cuda, 512, 512, 28, 3, 1, 1, 1, 32, algo=oracle.WINOGRADF63)
    c, _, _ = _layer(cuda, 512, 512, 28, 3, 1, 1, 1, 32, algo=oracle.IM2COL)
    x = torch.rand((32, 512, 28, 28), device=cuda, generator=torch.Generator(device=cuda).manual_seed(2)) * 2 - 1
    ya, yc = a.Forward(x), c.Forward(x)
    assert float((ya - yc).abs().max()) <= TOL * float(yc.abs().max())


def test_cpp_host_forward(cuda, tmp_path):
    """The C++ host class (include/booster/booster.h) drives Init + Forward on device buffers exactly like feather::ConvLayer."""
    import os
    import shutil
    import subprocess

    from feathercnn_amd import _lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "host_forward_test")
    libdir = os.path.dirname(_lib.lib_path())
    subprocess.run([hipcc, "-std=c++17", "-O1", "-I" + os.path.join(root, "include"), os.path.join(root, "tests", "cpp", "host_forward_test.cpp"),
                    "-o", exe, "-L" + libdir, "-lfeather_hip", "-Wl,-rpath," + libdir], check=True, capture_output=True, text=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "host forward ok" in out.stdout


def test_forward_is_hipgraph_capturable(cuda, checker):
    """Forward never allocates and keeps no host state, so a whole layer sequence can be captured once and replayed
    (one hipGraph replay per step is what bench.py times)."""
    import torch
    from feathercnn_amd import ConvLayer, ConvParam
    geoms = [conv_geom(16, 32, 20, 3, 1, 1), conv_geom(32, 24, 20, 1, 1, 0), conv_geom(24, 24, 20, 3, 2, 1, group=24)]
    layers, xs, refs = [], [], []
    scratch_bytes = 0
    for g in geoms:
        x, w, b = synth(g, 3, seed=9)
        p = ConvParam(output_channels=g.oc, input_channels=g.ic, input_h=g.ih, input_w=g.iw, kernel_h=g.kh, kernel_w=g.kw,
                      stride_h=g.sh, stride_w=g.sw, pad_left=g.pl, pad_right=g.pr, pad_top=g.pt, pad_bottom=g.pb, group=g.group,
                      bias_term=True, activation=1, batch=3)
        lyr = ConvLayer(p, torch.from_numpy(w).to(cuda), torch.from_numpy(b).to(cuda))
        layers.append(lyr)
        xs.append(torch.from_numpy(x).to(cuda))
        refs.append(checker.forward(g, x, w, b))
        scratch_bytes = max(scratch_bytes, lyr.buffer_bytes)
    scratch = torch.empty(max(scratch_bytes // 4, 1), device=cuda)
    outs = [torch.zeros(l.out_shape(), device=cuda) for l in layers]
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for l, x, o in zip(layers, xs, outs):
            l.Forward(x, out=o, scratch=scratch)  # warm-up on the capture stream
    torch.cuda.synchronize()
    g_ = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g_):
        for l, x, o in zip(layers, xs, outs):
            l.Forward(x, out=o, scratch=scratch)
    for o in outs:
        o.zero_()
    g_.replay()
    g_.replay()
    torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        assert nerr(o.cpu().numpy(), r) <= TOL


@pytest.mark.parametrize("g,batch", [(conv_geom(20, 36, 19, 3, 1, 1), 3), (conv_geom(4, 4, 19, 3, 1, 1), 2), (conv_geom(20, 32, 25, 3, 1, 1), 2)],
                         ids=["20x36@19", "4x4@19", "20x32@25"])
def test_winograd_shapes_the_reference_crashes_on(g, batch, cuda, port):
    """input_channels % 8 == 4 with a ragged tile grid segfaults the compiled reference (tests/test_oracle.py); the HIP path is
    checked against the pinned C restatement and the fp64 direct convolution instead."""
    x, w, b = synth(g, batch, seed=8)
    y, used = run_gpu(g, x, w, b, cuda)
    assert used == oracle.WINOGRADF63
    assert nerr(y, port.forward(g, x, w, b)) <= TOL
    assert nerr(y, port.direct_f64(g, x, w, b)) <= TOL


def test_mfma_calibration_reports_a_plausible_ceiling(cuda):
    """fhip_calibrate_mfma_f32: a pure-MFMA kernel cannot beat the nominal fp32 matrix peak of the part (157.3 TFLOP/s on MI355X) and a
    healthy device sustains well over half of it; bench.py prints this number next to every MFMA roofline."""
    from feathercnn_amd import booster
    tf, mhz = booster.calibrate_mfma_f32()
    assert 60.0 < tf < 165.0, tf
    assert 1000.0 < mhz < 3000.0, mhz
