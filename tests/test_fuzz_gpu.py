"""Seeded random sweep of convolution geometries through every route of the C-ABI against the pinned C restatement (and the
fp64 direct convolution): non-square images, every kernel size 1..7, strides 1..3, pads 0..k, ragged channel counts, depthwise,
all four (bias, activation) epilogues, batches 1..5, both routing rules, and the fused pooling / residual entry points where
they apply.  Deterministic (fixed seed): a failure names its case."""
import ctypes

import numpy as np
import pytest

import oracle
from oracle import Geom, nerr, synth

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _cases(n, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < n:
        kind = rng.choice(["dense", "dense", "dense", "dw", "wino", "wino", "one"])
        if kind == "dw":
            c = int(rng.integers(1, 40))
            ic = oc = group = c
        else:
            ic, oc, group = int(rng.integers(1, 72)), int(rng.integers(1, 80)), 1
        if kind == "wino":
            ic, oc = 4 * int(rng.integers(1, 18)), 4 * int(rng.integers(1, 20))
            kh = kw = 3
            sh = sw = 1
        elif kind == "one":
            kh = kw = 1
            sh = sw = int(rng.integers(1, 3))
        else:
            kh = kw = int(rng.integers(1, 8)) if kind == "dense" else int(rng.choice([3, 3, 3, 5]))
            if rng.random() < 0.2 and kind == "dense":
                kw = int(rng.integers(1, 6))
            sh = sw = int(rng.integers(1, 4))
        ih, iw = int(rng.integers(kh, 40)), int(rng.integers(kw, 44))
        pl = pr = int(rng.integers(0, kw + 1)) if rng.random() < 0.7 else 0
        pt = pb = int(rng.integers(0, kh + 1)) if rng.random() < 0.7 else 0
        if kind == "wino":
            pl = pr = pt = pb = int(rng.integers(0, 3))
        g = Geom(ic, oc, ih, iw, kh, kw, sh, sw, pl, pr, pt, pb, group, int(rng.integers(0, 2)), int(rng.integers(0, 2)))
        oh = (ih + pt + pb - kh) // sh + 1
        ow = (iw + pl + pr - kw) // sw + 1
        if oh < 1 or ow < 1:
            continue
        out.append((g, int(rng.integers(1, 6))))
    return out


CASES = _cases(160, 20260923)


def _param(g, batch):
    from feathercnn_amd import ConvParam
    p = ConvParam(output_channels=g.oc, input_channels=g.ic, input_h=g.ih, input_w=g.iw, kernel_h=g.kh, kernel_w=g.kw, stride_h=g.sh,
                  stride_w=g.sw, pad_left=g.pl, pad_right=g.pr, pad_top=g.pt, pad_bottom=g.pb, group=g.group, bias_term=bool(g.bias),
                  activation=g.act, batch=batch)
    p.AssignOutputDim()
    return p


@pytest.mark.parametrize("idx", range(len(CASES)))
def test_random_geometry(idx, cuda, port):
    import torch
    from feathercnn_amd import ConvLayer, _lib
    g, batch = CASES[idx]
    x, w, b = synth(g, batch, seed=idx)
    want = port.forward(g, x, w, b)
    f64 = port.direct_f64(g, x, w, b if g.bias else None)
    if g.act:
        f64 = np.maximum(f64, 0)
    assert nerr(want, f64) <= TOL, "the checker itself disagrees with fp64"
    xt, wt = torch.from_numpy(x).to(cuda), torch.from_numpy(w).to(cuda)
    bt = torch.from_numpy(b).to(cuda) if g.bias else None
    lib = _lib.load_library()
    for tuned in (False, True):
        lyr = ConvLayer(_param(g, batch), wt, bt, tuned=tuned)
        y = lyr.Forward(xt)
        torch.cuda.synchronize()
        got = y.cpu().numpy()
        assert got.shape == want.shape, (g, batch)
        assert np.isfinite(got).all()
        assert nerr(got, want) <= TOL, (g, batch, tuned, lyr.booster.algo)
        cp = lyr.param._c()
        scratch = torch.empty(max(lyr.buffer_bytes // 4, 1), device=cuda)
        bias_ptr = bt.data_ptr() if bt is not None else None
        if lib.fhip_conv_can_fuse_maxpool2(ctypes.byref(cp), lyr.booster.algo):
            pooled = torch.empty((batch, y.shape[1], y.shape[2] // 2, y.shape[3] // 2), device=cuda)
            rc = lib.fhip_conv_forward_maxpool2(ctypes.byref(cp), lyr.booster.algo, batch, pooled.data_ptr(), xt.data_ptr(), lyr.packed.data_ptr(),
                                                scratch.data_ptr(), bias_ptr, None)
            assert rc == 0
            ref_pool = torch.nn.functional.max_pool2d(y, 2, 2)
            assert torch.equal(pooled, ref_pool), (g, batch, "fused pooling")
        if lib.fhip_conv_can_fuse_residual(ctypes.byref(cp), lyr.booster.algo):
            res = torch.rand(y.shape, device=cuda, generator=torch.Generator(device=cuda).manual_seed(idx)) - 0.5
            plain = ConvLayer(_param(Geom(*[getattr(g, f) for f in ("ic", "oc", "ih", "iw", "kh", "kw", "sh", "sw", "pl", "pr", "pt", "pb", "group", "bias")], 0), batch),
                              wt, bt, algo=lyr.booster.algo)
            expect = plain.Forward(xt) + res
            if g.act:
                expect = expect.clamp_min(0)
            out = torch.empty_like(y)
            rc = lib.fhip_conv_forward_residual(ctypes.byref(cp), lyr.booster.algo, batch, out.data_ptr(), xt.data_ptr(), lyr.packed.data_ptr(),
                                                scratch.data_ptr(), bias_ptr, res.data_ptr(), None)
            assert rc == 0
            torch.cuda.synchronize()
            assert torch.equal(out, expect), (g, batch, "fused residual")


# Asymmetric pads are legal in the C-ABI (ConvParam carries the four pads separately, booster.h:68-71) although feather::ConvLayer
# only ever sets symmetric ones.  ADVICE r01: with pad_right < pad_left the last output column's right tap is a real pixel, not
# padding -- the depthwise kernel used to take it from a lane that belongs to another row.
ASYM = [Geom(c, c, h, w, 3, 3, s, s, pl, pr, pt, pb, c, 1, 1)
        for (c, h, w, s, pl, pr, pt, pb) in [(8, 12, 16, 1, 1, 0, 1, 1), (8, 12, 17, 1, 1, 0, 1, 0), (6, 9, 14, 1, 1, 2, 0, 1), (5, 20, 32, 1, 1, 0, 1, 1),
                                             (4, 16, 16, 2, 1, 0, 1, 0), (7, 15, 18, 2, 1, 2, 1, 1), (3, 28, 28, 1, 1, 0, 0, 0), (16, 14, 14, 2, 1, 0, 0, 1)]] + \
       [Geom(8, 12, 13, 19, 3, 3, 1, 1, 1, 0, 1, 0, 1, 1, 1), Geom(8, 8, 14, 14, 3, 3, 1, 1, 2, 0, 0, 1, 1, 1, 0), Geom(12, 8, 10, 11, 5, 3, 2, 1, 0, 2, 3, 1, 1, 1, 1),
        Geom(16, 16, 9, 9, 1, 1, 1, 1, 1, 0, 0, 1, 1, 0, 1)]


@pytest.mark.parametrize("g", ASYM, ids=lambda g: f"c{g.ic}g{g.group}_{g.ih}x{g.iw}_k{g.kh}x{g.kw}s{g.sh}_p{g.pl}{g.pr}{g.pt}{g.pb}")
def test_asymmetric_pads(g, cuda, port):
    import torch
    from feathercnn_amd import ConvLayer
    batch = 3
    x, w, b = synth(g, batch, seed=99)
    want = port.forward(g, x, w, b)
    f64 = port.direct_f64(g, x, w, b if g.bias else None)
    if g.act:
        f64 = np.maximum(f64, 0)
    assert nerr(want, f64) <= TOL, "the checker itself disagrees with fp64"
    for tuned in (False, True):
        lyr = ConvLayer(_param(g, batch), torch.from_numpy(w).to(cuda), torch.from_numpy(b).to(cuda) if g.bias else None, tuned=tuned)
        y = lyr.Forward(torch.from_numpy(x).to(cuda))
        torch.cuda.synchronize()
        assert nerr(y.cpu().numpy(), want) <= TOL, (g, tuned, lyr.booster.algo)
