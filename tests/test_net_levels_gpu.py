"""The three benchmark nets at 224 x 224 and odd batches: everything fusion level 3 adds (tuned routes, chained Winograd runs, the first
layer inside the next layer's input transform, sibling 1x1 GEMMs -- planned per batch --, hipGraph replay, the branch stream) against the
plain level-1 net, on the logits (the softmax outputs of randomly initialised nets saturate and would hide a difference)."""
import numpy as np
import pytest

from oracle import nerr

pytestmark = pytest.mark.gpu

LOGITS = {"vgg16": "fc8", "resnet50": "fc1000", "mobilenet_v1": "fc7"}


@pytest.mark.parametrize("name,batch", [("vgg16", 3), ("resnet50", 3), ("resnet50", 17), ("mobilenet_v1", 5)])
def test_level_3_equals_level_1_on_the_logits(cuda, name, batch):
    from feathercnn_amd import model_zoo
    from feathercnn_amd.net import Net
    p, b, i, _ = getattr(model_zoo, name)()
    x = np.random.default_rng(batch).uniform(-1, 1, (batch, 3, 224, 224)).astype(np.float32)
    outs, plans = [], []
    for level, tuned in ((1, False), (3, True)):
        net = Net(fusion=level, tuned=tuned, graph=(level == 3), concurrency=(level == 3))
        net.LoadParam(p)
        net.LoadWeights(b)
        net.FeedInput(i, x)
        for _ in range(2):
            net.Forward()
        outs.append(net.Extract(LOGITS[name]).copy())
        plans.append((len(net.chains()), len(net.siblings())))
        net.close()
    assert plans[0] == (0, 0)
    if name == "vgg16":
        assert plans[1][0] == 12  # the 12 chained Winograd layers; conv1_1 (computed inside conv1_2's input transform) is only in the raw view
    if name == "resnet50":
        assert plans[1][1] == (4 if batch == 17 else 0)  # res4a / res5a pairs at batch 17; at batch 3 the stacked grids would need split-K
    assert np.isfinite(outs[1]).all()
    assert nerr(outs[1], outs[0]) <= 2e-5
