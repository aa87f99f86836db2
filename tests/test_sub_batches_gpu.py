"""Sub-batch replicas of the Net runtime (fhip_net_set_sub_batches): R complete copies of the net on streams of their own, each
taking a contiguous share of every batch.  Results must equal the single-net ones (images are independent; only the batch-dependent
reduction order of split-K layers may differ), whatever the batch, including batches smaller than R and re-feeds of other sizes."""
import numpy as np
import pytest

from feathercnn_amd import model_zoo
from oracle import nerr, netcheck

pytestmark = pytest.mark.gpu


def _run(p, b, i, img, blobs, **kw):
    from feathercnn_amd.net import Net
    net = Net(**kw)
    net.LoadParam(p)
    net.LoadWeights(b)
    net.FeedInput(i, img)
    for _ in range(3 if kw.get("graph") else 1):
        net.Forward()
    return net, {k: net.Extract(k) for k in blobs}


@pytest.mark.parametrize("replicas,batch", [(2, 8), (2, 5), (3, 7), (4, 2), (2, 1)])
def test_replicas_equal_the_single_net(cuda, replicas, batch):
    p, b, i, o = model_zoo.tiny_allsorts()
    img = np.random.default_rng(replicas * 10 + batch).uniform(-1, 1, (batch, 3, 20, 20)).astype(np.float32)
    _, want = _run(p, b, i, img, [o], fusion=2, tuned=True)
    net, got = _run(p, b, i, img, [o], fusion=2, tuned=True, graph=True, concurrency=True, sub_batches=replicas)
    assert got[o].shape == want[o].shape
    assert nerr(got[o], want[o]) <= 1e-6
    # the reference, image by image
    ref = netcheck.RefNet(p, b) if netcheck.have_ref_net() else netcheck.PortNet(p, b)
    for k in (0, batch - 1):
        assert nerr(got[o][k:k + 1], ref.run(i, img[k:k + 1], o)) <= 1e-4
    # a different batch through the same handle (shares change, a replica may go idle)
    img2 = np.random.default_rng(99).uniform(-1, 1, (max(1, batch - 3), 3, 20, 20)).astype(np.float32)
    net.FeedInput(i, img2)
    net.Forward()
    _, want2 = _run(p, b, i, img2, [o], fusion=2, tuned=True)
    assert nerr(net.Extract(o), want2[o]) <= 1e-6
    assert net.memory()["weight_bytes"] > 0


def test_replicas_on_a_benchmark_net_and_intermediate_blobs(cuda):
    """MobileNet-V1 at 64 x 64 (the net that gains from replicas): logits and an intermediate blob, 2 replicas vs 1."""
    p, b, i, o = model_zoo.mobilenet_v1(size=64, classes=20)
    img = np.random.default_rng(1).uniform(-1, 1, (6, 3, 64, 64)).astype(np.float32)
    net1, want = _run(p, b, i, img, [o, "fc7"], fusion=3, tuned=True, graph=True)
    net2, got = _run(p, b, i, img, [o, "fc7"], fusion=3, tuned=True, graph=True, sub_batches=2)
    for k in want:
        assert got[k].shape == want[k].shape
        assert nerr(got[k], want[k]) <= 1e-6, k
    # device-side feed (ordered on the net's stream) and device-side extract
    import torch
    x = torch.from_numpy(img[::-1].copy()).to(cuda)
    net2.FeedInput(i, x)
    net2.Forward()
    net1.FeedInput(i, x)
    net1.Forward()
    assert nerr(net2.Extract(o), net1.Extract(o)) <= 1e-6


def test_setting_it_late_is_refused(cuda):
    from feathercnn_amd import FeatherHipError
    from feathercnn_amd.net import Net
    import ctypes
    p, b, i, o = model_zoo.tiny_allsorts()
    net = Net()
    net.LoadParam(p)
    assert net._lib.fhip_net_set_sub_batches(net._h, 2) == -2  # FHIP_E_BADARG: after LoadParam
    assert net._lib.fhip_net_set_sub_batches(net._h, 0) == -2
