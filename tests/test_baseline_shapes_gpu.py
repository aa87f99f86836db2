"""Oracle parity AT the BASELINE.json shapes: what bench.py times is what gets checked.

(i)  The three metric nets at their bench configuration -- 224x224, the bench batch (VGG-16 32, ResNet-50 64,
     MobileNet-V1 256), bench.SUB_BATCHES replicas, fusion level 3, MI355X conv routing, branch concurrency, hipGraph replay --
     against the REAL reference feather::Net (oracle/_ref, N = 1): every image of VGG-16's batch, 16+ spread images of the others
     (both sides of the replica split included): logits and probabilities.
(ii) One layer per route at the FULL benchmark batch with the real reference looped over the whole batch (no sampling):
     VGG conv1_2 / conv5_1 b32 (Winograd), ResNet-50 1x1 stride-1 / stride-2 b64 (implicit GEMM), MobileNet depthwise b256.

Tolerance: normalised max error <= 1e-4 (BASELINE.json north_star)."""
import numpy as np
import pytest

import oracle
from feathercnn_amd import model_zoo
from oracle import conv_geom, nerr, netcheck, synth

pytestmark = pytest.mark.gpu
TOL = 1e-4

BENCH_NETS = [("vgg16", 32, "fc8"), ("resnet50", 64, "fc1000"), ("mobilenet_v1", 256, "fc7")]


def bench_picks(batch, replicas, every):
    """Images of the batch that are checked: all of them (VGG-16), or 16 spread over the batch including its first and last image and
    BOTH sides of every replica boundary (the sub-batch replicas deal the batch out in contiguous shares)."""
    if every:
        return list(range(batch))
    from feathercnn_amd.shard import shard_range
    picks = {0, batch - 1}
    for r in range(replicas):
        lo, hi = shard_range(batch, r, replicas)
        picks.update((lo, hi - 1))
    step = max(1, batch // 16)
    picks.update(range(step // 2, batch, step))
    return sorted(picks)


@pytest.mark.parametrize("name,batch,logits", BENCH_NETS, ids=[n for n, _, _ in BENCH_NETS])
def test_bench_configuration_matches_reference_net(cuda, name, batch, logits):
    """EXACTLY the configuration bench.py times (bench.DEFAULT_BATCH, bench.SUB_BATCHES, fusion 3, MI355X routing, branch stream, graph
    replay): every image of VGG-16's batch, >= 16 spread images of the other two, against the real reference feather::Net."""
    import bench
    from feathercnn_amd.net import Net
    assert batch == bench.DEFAULT_BATCH[name]
    replicas = bench.SUB_BATCHES.get(name, 1)
    model = model_zoo.MODELS[name]()  # 224 x 224
    p, b, i, o = model
    x = np.random.default_rng(2024).uniform(-1, 1, (batch, 3, 224, 224)).astype(np.float32)
    net = Net(fusion=3, graph=True, tuned=True, concurrency=True, sub_batches=replicas)  # exactly bench.py's measure_net
    net.LoadParam(p)
    net.LoadWeights(b)
    net.FeedInput(i, x)
    for _ in range(3):  # first forward = Reshape + Init + capture, then two graph replays
        net.Forward()
    prob, got_logits = net.Extract(o), net.Extract(logits)
    assert prob.shape[0] == batch and np.isfinite(prob).all()
    picks = bench_picks(batch, replicas, every=(name == "vgg16"))
    assert len(picks) >= min(batch, 16)
    if netcheck.have_ref_net():
        ref = netcheck.RefNet(p, b)
        want = [ref.run_blobs(i, x[k], (o, logits)) for k in picks]  # one reference forward per image, both blobs
        ref.close()
    else:  # the restatement (slower): one image
        picks = picks[:1]
        blobs = netcheck.PortNet(p, b).run(i, x[:1], o, keep=True)
        want = [(blobs[o], blobs[logits])]
    for k, (wp, wl) in zip(picks, want):
        assert nerr(got_logits[k:k + 1], wl) <= TOL, (name, k)
        assert nerr(prob[k:k + 1], wp) <= TOL, (name, k)
        assert int(prob[k].reshape(-1).argmax()) == int(wp.reshape(-1).argmax())


# name, geometry, full benchmark batch
FULL_BATCH = [
    ("vgg_conv1_2_b32", conv_geom(64, 64, 224, 3, 1, 1), 32),
    ("vgg_conv5_1_b32", conv_geom(512, 512, 14, 3, 1, 1), 32),
    ("r50_res2a_2c_b64", conv_geom(64, 256, 56, 1, 1, 0), 64),
    ("r50_res4a_2a_b64", conv_geom(512, 256, 28, 1, 2, 0), 64),
    ("r50_res5b_2a_b64", conv_geom(2048, 512, 7, 1, 1, 0), 64),
    ("mb_conv2_dw_b256", conv_geom(32, 32, 112, 3, 1, 1, group=32), 256),
    ("mb_conv8_dw_b256", conv_geom(512, 512, 14, 3, 1, 1, group=512), 256),
    ("mb_conv7_dw_s2_b256", conv_geom(256, 256, 28, 3, 2, 1, group=256), 256),
]


@pytest.mark.parametrize("name,g,batch", FULL_BATCH, ids=[c[0] for c in FULL_BATCH])
def test_full_batch_layer_matches_reference(cuda, checker, name, g, batch):
    """The whole benchmark batch through the C-ABI (MI355X routing, as the bench runs it) against the checker looped over
    every image of the batch."""
    import torch

    from feathercnn_amd import ConvLayer, ConvParam
    x, w, b = synth(g, batch, seed=77)
    prm = ConvParam(output_channels=g.oc, input_channels=g.ic, input_h=g.ih, input_w=g.iw, kernel_h=g.kh, kernel_w=g.kw,
                    stride_h=g.sh, stride_w=g.sw, pad_left=g.pl, pad_right=g.pr, pad_top=g.pt, pad_bottom=g.pb, group=g.group,
                    bias_term=True, activation=1, batch=batch)
    layer = ConvLayer(prm, torch.from_numpy(w).to(cuda), torch.from_numpy(b).to(cuda), tuned=True)
    y = layer.Forward(torch.from_numpy(x).to(cuda))
    torch.cuda.synchronize()
    y = y.cpu().numpy()
    ref = checker.forward(g, x, w, b)
    assert y.shape == ref.shape
    scale = float(np.abs(ref).max())
    per_image = np.abs(y - ref).reshape(batch, -1).max(1) / scale
    assert per_image.max() <= TOL, f"{name}: worst image {int(per_image.argmax())} error {per_image.max():.3e}"
