"""Seeded random network topologies through the Net runtime at every fusion level and both routing rules, against the numpy/C
restatement of feather::Net (oracle.netcheck.PortNet).  The generator strings together the blocks real models are made of --
conv (+BN / Scale / ReLU in any combination), depthwise, pooling of every flavour, residual blocks with either operand order,
fire-style split/concat blocks, dropout, InnerProduct, softmax -- so the fusion pass meets blobs with two consumers, residuals
whose other operand is produced later, pooling behind non-Winograd convolutions, aliases feeding concats, and so on."""
import numpy as np
import pytest

from feathercnn_amd import model_zoo
from oracle import nerr, netcheck

pytestmark = pytest.mark.gpu
TOL = 1e-4


def random_net(seed, runs=False):
    rng = np.random.default_rng(seed)
    g = model_zoo.GraphBuilder(seed)
    c = int(rng.choice([3, 4, 8]))
    h, w = int(rng.integers(14, 36)), int(rng.integers(14, 36))
    x = g.input("data", c, h, w)
    shape = (c, h, w)
    uid = [0]

    def name(prefix):
        uid[0] += 1
        return f"{prefix}{uid[0]}"

    def conv_out(hw, k, s, p):
        return (hw + 2 * p - k) // s + 1

    def conv(x, c, h, w, cout=None, k=None, s=None, p=None, dw=False):
        k = int(rng.choice([1, 3, 3, 5])) if k is None else k
        s = int(rng.choice([1, 1, 2])) if s is None else s
        p = int(rng.integers(0, k // 2 + 2)) if p is None else p
        if conv_out(h, k, s, p) < 1 or conv_out(w, k, s, p) < 1:
            k, s, p = 1, 1, 0
        cout = c if dw else (int(rng.choice([4, 8, 12, 16, 20, 24])) if cout is None else cout)
        x = g.conv(name("conv"), x, c, cout, k, s, p, group=c if dw else 1, bias=(not dw) and rng.random() < 0.8)
        return x, cout, conv_out(h, k, s, p), conv_out(w, k, s, p)

    def post(x, c):
        if rng.random() < 0.5:
            x = g.bn(name("bn"), x, c)
        if rng.random() < 0.4:
            x = g.scale(name("scale"), x, c, bias=rng.random() < 0.5)
        if rng.random() < 0.6:
            x = g.relu(name("relu"), x)
        return x

    c, h, w = shape
    for _ in range(int(rng.integers(3, 8))):
        kind = rng.choice(["conv", "conv", "dw", "pool", "res", "fire", "drop"] + (["run"] * 4 if runs else []))
        if kind == "run" and min(h, w) >= 6:
            # a VGG-style run: 3x3 / stride-1 / pad-1 layers, some behind a 2x2 / stride-2 max pooling (fusion level 3 chains these)
            for _ in range(int(rng.integers(2, 5))):
                x, c, h, w = conv(x, c, h, w, cout=int(rng.choice([4, 8, 12, 16])), k=3, s=1, p=1)
                x = post(x, c)
                if rng.random() < 0.35 and min(h, w) >= 8:
                    x = g.pool(name("pool"), x, 2, 2, 0)
                    h, w = -(-h // 2), -(-w // 2)
        elif kind == "conv":
            x, c, h, w = conv(x, c, h, w)
            x = post(x, c)
        elif kind == "dw":
            x, c, h, w = conv(x, c, h, w, k=3, p=1, dw=True)
            x = post(x, c)
        elif kind == "pool" and min(h, w) >= 4:
            k = int(rng.choice([2, 3]))
            s = int(rng.choice([1, 2, 2]))
            p = int(rng.choice([0, 0, 1])) if k == 3 else 0  # k <= 2p would make the first window empty (-FLT_MAX / NaN: degenerate)
            avg = bool(rng.random() < 0.4)
            x = g.pool(name("pool"), x, k, s, p, avg=avg)
            import math
            h = int(math.ceil(np.float32(h + 2 * p - k) / np.float32(s))) + 1
            w = int(math.ceil(np.float32(w + 2 * p - k) / np.float32(s))) + 1
        elif kind == "res":
            a, b = g.split(name("split"), x)
            k = int(rng.choice([1, 3]))
            y, _, _, _ = conv(a, c, h, w, cout=c, k=k, s=1, p=k // 2)
            if rng.random() < 0.5:
                y = g.bn(name("bn"), y, c)
            if rng.random() < 0.3:
                y = g.relu(name("relu"), y)          # a ReLU BEFORE the add must block the residual fusion
            x = g.eltwise(name("sum"), y, b) if rng.random() < 0.5 else g.eltwise(name("sum"), b, y)
            if rng.random() < 0.7:
                x = g.relu(name("relu"), x)
        elif kind == "fire":
            a, b = g.split(name("split"), x)
            y1, c1, _, _ = conv(a, c, h, w, k=1, s=1, p=0)
            y1 = post(y1, c1)
            y2, c2, _, _ = conv(b, c, h, w, k=3, s=1, p=1)
            y2 = post(y2, c2)
            x = g.concat(name("cat"), [y1, y2])
            c = c1 + c2
        elif kind == "drop":
            x = g.dropout(name("drop"), x, scale=None if rng.random() < 0.5 else 0.5)
    if rng.random() < 0.6:
        if rng.random() < 0.5:
            x = g.pool(name("gap"), x, 1, 1, avg=True, global_=True)
            h = w = 1
        x = g.fc(name("fc"), x, c * h * w, 10, bias=rng.random() < 0.7)
        if rng.random() < 0.5:
            x = g.softmax(name("prob"), x)
    p, b = g.finish()
    return p, b, shape, x


@pytest.mark.parametrize("seed", range(120))
def test_random_topology(seed, cuda):
    from feathercnn_amd.net import Net
    p, b, shape, out = random_net(1000 + seed)
    batch = 1 + seed % 4
    img = np.random.default_rng(seed).uniform(-1, 1, (batch,) + shape).astype(np.float32)
    blobs = netcheck.PortNet(p, b).run("data", img, out, keep=True)
    want = blobs[out]
    scale = float(np.abs(want).max())
    assert np.isfinite(want).all() or np.isnan(want).any()  # empty average windows are NaN on both sides
    for fusion, tuned, conc in [(0, False, False), (1, False, True), (2, False, False), (2, True, True)]:
        net = Net(fusion=fusion, tuned=tuned, concurrency=conc, graph=conc and seed % 2 == 0)
        net.LoadParam(p)
        net.LoadWeights(b)
        net.FeedInput("data", img)
        for _ in range(3 if conc else 1):
            net.Forward()
        got = net.Extract(out)
        assert got.shape == want.shape, (seed, fusion)
        ok = ~np.isnan(want)
        assert np.array_equal(np.isnan(got), ~ok), (seed, fusion, tuned)
        if scale > 0:
            assert float(np.abs(got[ok] - want[ok]).max()) <= TOL * float(np.abs(want[ok]).max()), (seed, fusion, tuned, p.decode())
        if fusion == 0:
            for name_, ref in blobs.items():
                if name_ == "data":
                    continue
                cur = net.Extract(name_)
                m = ~np.isnan(ref)
                assert cur.shape == ref.shape and np.array_equal(np.isnan(cur), ~m), (seed, name_)
                if m.any() and float(np.abs(ref[m]).max()) > 0:
                    assert float(np.abs(cur[m] - ref[m]).max()) <= TOL * float(np.abs(ref[m]).max()), (seed, name_)
        net.close()


@pytest.mark.parametrize("seed", range(60))
def test_random_topology_with_winograd_runs_at_fusion_level_3(seed, cuda):
    """Topologies rich in consecutive 3x3 layers: level 3 (chained Winograd layers, both routing rules) against the restatement and
    bit for bit against level 2 under the same routing."""
    from feathercnn_amd.net import Net
    p, b, shape, out = random_net(5000 + seed, runs=True)
    batch = 1 + seed % 3
    img = np.random.default_rng(seed).uniform(-1, 1, (batch,) + shape).astype(np.float32)
    want = netcheck.PortNet(p, b).run("data", img, out)
    ok = ~np.isnan(want)
    chained = 0
    for tuned in (False, True):
        outs = {}
        for fusion in (2, 3):
            net = Net(fusion=fusion, tuned=tuned, concurrency=tuned, graph=tuned and seed % 2 == 0)
            net.LoadParam(p)
            net.LoadWeights(b)
            net.FeedInput("data", img)
            for _ in range(3 if tuned else 1):
                net.Forward()
            outs[fusion] = net.Extract(out)
            if fusion == 3:
                chained += len(net.chains())
                first_fused = any(2 in v for v in net.chains(raw=True).values())
            net.close()
        if first_fused:  # the first layer's 27-term sums are added in another order inside the next layer's input transform
            assert np.array_equal(np.isnan(outs[2]), np.isnan(outs[3])), (seed, tuned)
            m = ~np.isnan(outs[2])
            assert not m.any() or float(np.abs(outs[2][m] - outs[3][m]).max()) <= 1e-5 * max(float(np.abs(outs[2][m]).max()), 1e-30), (seed, tuned)
        else:
            assert np.array_equal(outs[2], outs[3], equal_nan=True), (seed, tuned)
        assert np.array_equal(np.isnan(outs[3]), ~ok), (seed, tuned)
        if ok.any() and float(np.abs(want[ok]).max()) > 0:
            assert float(np.abs(outs[3][ok] - want[ok]).max()) <= TOL * float(np.abs(want[ok]).max()), (seed, tuned, p.decode())
    _CHAINED.append(chained)


_CHAINED = []


def test_the_sweep_did_exercise_chains(cuda):
    """Guard against the generator drifting: most of the 60 topologies above must have produced chained layers."""
    if len(_CHAINED) < 60:
        pytest.skip("runs only after the whole sweep")
    assert sum(1 for c in _CHAINED if c > 0) >= 30, _CHAINED
