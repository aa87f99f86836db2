"""tools/level_check.py -- the three benchmark nets at odd batches: fusion level 3 (tuned routes, graph, branch stream) against level 1."""
import sys

import numpy as np

sys.path.insert(0, ".")
from feathercnn_amd import model_zoo  # noqa: E402
from feathercnn_amd.net import Net  # noqa: E402
from oracle import nerr  # noqa: E402

worst = 0.0
for name, maker in (("vgg16", model_zoo.vgg16), ("resnet50", model_zoo.resnet50), ("mobilenet_v1", model_zoo.mobilenet_v1)):
    p, b, i, o = maker()
    o = {"vgg16": "fc8", "resnet50": "fc1000", "mobilenet_v1": "fc7"}[name]  # the logits: softmax outputs of random nets saturate
    for batch in (1, 3, 17, 40):
        x = np.random.default_rng(batch).uniform(-1, 1, (batch, 3, 224, 224)).astype(np.float32)
        outs = []
        for level, tuned in ((1, False), (3, True)):
            net = Net(fusion=level, tuned=tuned, graph=(level == 3), concurrency=(level == 3))
            net.LoadParam(p)
            net.LoadWeights(b)
            net.FeedInput(i, x)
            for _ in range(2):
                net.Forward()
            outs.append(net.Extract(o).copy())
            info = (len(net.chains()), len(net.siblings()))
            net.close()
        e = nerr(outs[1], outs[0])
        worst = max(worst, e)
        print(f"{name} batch {batch}: level 3 vs 1 normalised error {e:.2e}  (chained layers {info[0]}, sibling layers {info[1]})")
print("worst", worst)
assert worst <= 1e-4
